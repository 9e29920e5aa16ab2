#!/usr/bin/env python3
"""End-to-end wall clock of the `dsk` binary (file -> .h5) on a synthetic FASTQ, next to the CPU oracle CLI.
usage (GPU box): python tools/e2e.py [workload=ecoli50x] [k=31]"""
import os, subprocess, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from dsk_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "ecoli50x"
k = sys.argv[2] if len(sys.argv) > 2 else "31"
gl, nr, rl = synth.workload(name)
dev = torch.device("cuda:0")
reads = synth.make_reads(synth.make_genome(gl, dev), nr, rl).cpu().numpy().reshape(nr, rl + 1)[:, :rl]
os.makedirs("/tmp/e2e", exist_ok=True)
fq = f"/tmp/e2e/{name}.fastq"
t0 = time.time()
qual = b"I" * rl
with open(fq, "wb") as f:
    buf = []
    for i in range(nr):
        buf.append(b"@r%d\n" % i + reads[i].tobytes() + b"\n+\n" + qual + b"\n")
        if len(buf) == 65536:
            f.write(b"".join(buf)); buf = []
    f.write(b"".join(buf))
print(f"wrote {fq}: {os.path.getsize(fq)/1e6:.0f} MB in {time.time()-t0:.1f} s")
root = os.path.abspath(".")
for attempt in range(2):
    t0 = time.time()
    out = subprocess.run([f"{root}/dsk_amd/host/bin/dsk", "-file", fq, "-kmer-size", k, "-out", "/tmp/e2e/out", "-verbose", "1"],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
    dt = time.time() - t0
    keep = [l.strip() for l in out.splitlines() if any(s in l for s in ("_s ", "EXCEPTION"))]
    print(f"dsk run {attempt}: {dt:.2f} s wall | " + " | ".join(keep))
if len(sys.argv) > 3 and sys.argv[3] == "gz":      # the same reads as ordinary gzip and as BGZF (blocked gzip)
    import gzip, struct, zlib
    data = open(fq, "rb").read()
    t0 = time.time()
    with gzip.open(fq + ".gz", "wb", compresslevel=1) as f:
        f.write(data)
    with open(fq + ".bgzf.gz", "wb") as f:
        for off in list(range(0, len(data), 60000)) + [None]:
            chunk = b"" if off is None else data[off: off + 60000]
            c = zlib.compressobj(1, zlib.DEFLATED, -15)
            body = c.compress(chunk) + c.flush()
            f.write(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1))
            f.write(body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    print(f"compressed copies written in {time.time()-t0:.1f} s: {os.path.getsize(fq + '.gz')/1e6:.0f} MB gzip, {os.path.getsize(fq + '.bgzf.gz')/1e6:.0f} MB BGZF")
    for src in (fq + ".gz", fq + ".bgzf.gz"):
        for attempt in range(2):
            t0 = time.time()
            out = subprocess.run([f"{root}/dsk_amd/host/bin/dsk", "-file", src, "-kmer-size", k, "-out", "/tmp/e2e/outz", "-verbose", "1"],
                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
            dt = time.time() - t0
            keep = [l for l in out.splitlines() if any(s in l for s in ("ingest_s", "total_s", "kmers_nb_valid", "EXCEPTION"))]
            print(f"dsk {os.path.basename(src)} run {attempt}: {dt:.2f} s wall | " + " | ".join(x.strip() for x in keep))
t0 = time.time()
out = subprocess.run([f"{root}/oracle/dsk_oracle_cli", "-file", fq, "-kmer-size", k, "-nb-cores", str(os.cpu_count())],
                     stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
print(f"cpu oracle cli ({os.cpu_count()} threads): {time.time()-t0:.2f} s wall  |  {out.strip()}")
print("h5 size MB:", os.path.getsize("/tmp/e2e/out.h5") / 1e6)
