#!/usr/bin/env python3
"""Corrupted gzip inputs against the host bank (CPU only; run after `make -C tests/host asan`): bit flips, truncations, zeroed and
deleted byte runs in a .fastq.gz, through the checker build of `dsk` (host parser: one zlib stream, the inflate thread, the
parallel inflate with small chunks) and through IBank::streamRaw (the raw-text reader of `dsk -device-parse 1`), all under
AddressSanitizer + UBSan.  Every run must end in a clean `EXCEPTION: ...` (exit code 1) or reproduce the intact file's result --
never a crash, a hang or a silently different count.
   python tools/fuzz_gzip.py [seed=1] [corruptions=100]"""
import gzip, os, re, subprocess, sys, tempfile, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, "tests", "host", "asan", "dsk_cpu_check")
raw = os.path.join(ROOT, "tests", "host", "asan", "test_stream_raw")
tmp = tempfile.mkdtemp()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
recs = b"".join(b"@r%d\n" % i + bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 100)) + b"\n+\n" + b"I" * 100 + b"\n" for i in range(6000))
z = bytearray(gzip.compress(recs, 6))
open(os.path.join(tmp, "x.fastq.gz"), "wb").write(z)
PROGS = ([exe, "-file", "x.fastq.gz", "-kmer-size", "21", "-out", "o", "-verbose", "1", "-nb-cores", "4"], [raw, "x.fastq.gz"])


def sig(prog, p):
    if prog[0] == raw:
        return p.stdout.decode().split()[3:5]
    return re.findall(r"kmers_nb_(?:valid|distinct|solid)\s*:\s*(\d+)", p.stdout.decode())


goods = {}
for prog in PROGS:
    p = subprocess.run(prog, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr
    goods[prog[0]] = sig(prog, p)
n_ok = n_err = n_bad = 0
t0 = time.time()
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 100):
    y = bytearray(z)
    kind = it % 4
    if kind == 0:
        for _ in range(int(rng.integers(1, 4))):
            y[int(rng.integers(0, len(y)))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:
        y = y[: int(rng.integers(20, len(y)))]
    elif kind == 2:
        a = int(rng.integers(0, len(y) - 100)); y[a: a + int(rng.integers(1, 64))] = bytes(int(rng.integers(1, 64)))
    else:
        a = int(rng.integers(0, len(y) - 100)); del y[a: a + int(rng.integers(1, 50))]
    open(os.path.join(tmp, "x.fastq.gz"), "wb").write(y)
    for env in ({}, {"DSK_PGZIP_CHUNK_BYTES": "65536"}):
        for prog in PROGS:
            try:
                p = subprocess.run(prog, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env), timeout=120)
            except subprocess.TimeoutExpired:
                print("HANG", it, kind, env, prog[0]); sys.exit(1)
            if p.returncode == 0 and sig(prog, p) == goods[prog[0]]:
                n_ok += 1
            elif p.returncode == 1 and (b"EXCEPTION" in p.stderr or b"EXCEPTION" in p.stdout):
                n_err += 1
            else:
                n_bad += 1
                print("UNEXPECTED", it, kind, env, os.path.basename(prog[0]), p.returncode, sig(prog, p), p.stderr[-300:])
print(f"{n_ok} runs with the intact result, {n_err} clean errors, {n_bad} unexpected, {time.time() - t0:.0f} s")
sys.exit(1 if n_bad else 0)
