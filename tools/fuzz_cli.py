#!/usr/bin/env python3
"""Differential fuzz of the two `dsk` binaries: dsk_amd/host/bin/dsk (the HIP engine behind the C-ABI) against tests/host/dsk_cpu_check
(the same host layer on the CPU oracle), on random small inputs and random combinations of the options that change results --
1-4 input files (FASTA / FASTQ, plain / gzip, an album), k in 9..128, -abundance-min (numbers and `auto`) / -abundance-max, -histo-max,
-solidity-kind (+ -solidity-custom), -histo2D, -nb-partitions, -out-compress, -nb-gpus, -device-parse.  Compared: the dump of `dsk2ascii` (rows in
the tool's order, then sorted), the histogram dataset, the .histo / .histo2D files.
   python tools/fuzz_cli.py [seed=0] [n=100]          (FUZZ_GROUP=1: every run with -nb-gpus 2 / 4 / 8 and several input files)"""
import gzip
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GPU = os.path.join(ROOT, "dsk_amd", "host", "bin", "dsk")
CPU = os.path.join(ROOT, "tests", "host", "dsk_cpu_check")
D2A = os.path.join(ROOT, "dsk_amd", "host", "bin", "dsk2ascii")


def make_file(rng, tmp, name, genome):
    fq = rng.random() < 0.5
    n = int(rng.choice([0, 1, 2])) if rng.random() < 0.15 else int(rng.integers(1, 400))          # (sometimes a file without reads, or with one)
    out = []
    for i in range(n):
        L = int(rng.integers(0, 260))
        s = int(rng.integers(0, max(1, len(genome) - L)))
        seq = bytearray(genome[s: s + L])
        for _ in range(int(rng.integers(0, 3))):
            if seq:
                seq[int(rng.integers(0, len(seq)))] = int(rng.choice(list(b"ACGTNacgt")))
        seq = bytes(seq)
        if fq:
            out.append(b"@r%d\n" % i + seq + b"\n+\n" + b"I" * len(seq) + b"\n")
        else:
            w = int(rng.choice([60, 1000]))
            out.append(b">s%d\n" % i + b"".join(seq[a: a + w] + b"\n" for a in range(0, len(seq), w)))
    data = b"".join(out)
    path = os.path.join(tmp, name + (".fq" if fq else ".fa"))
    if rng.random() < 0.4:
        path += ".gz"
        open(path, "wb").write(gzip.compress(data, int(rng.integers(1, 10))))
    else:
        open(path, "wb").write(data)
    return path


def run(binary, args, cwd):
    try:
        return subprocess.run([binary] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    except subprocess.TimeoutExpired:
        print(f"HANG (120 s): {os.path.basename(binary)} {args} in {cwd}", flush=True)
        sys.exit(2)


def outputs(cwd, name):
    res = {}
    r = run(D2A, ["-file", name, "-out", name + ".txt", "-verbose", "0"], cwd)
    res["d2a_rc"] = r.returncode
    if r.returncode == 0:
        rows = open(os.path.join(cwd, name + ".txt")).read().splitlines()
        res["rows_sorted"] = sorted(rows)
    h = subprocess.run(["/opt/conda/bin/h5dump", "-y", "-d", "histogram/histogram", name + ".h5"], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    res["histo"] = [ln.strip() for ln in h.stdout.decode().splitlines() if ln.strip() and ln.strip()[0].isdigit()]
    for ext in (".histo", ".histo2D"):
        p = os.path.join(cwd, name + ext)
        res[ext] = open(p).read() if os.path.exists(p) else None
    return res


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    same = errs = 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        tmp = tempfile.mkdtemp(prefix="dskfuzz_")
        genome = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), int(rng.integers(300, 20000))))
        group_focus = os.environ.get("FUZZ_GROUP") == "1"          # every run on a group of ranks, several banks, mostly per-bank modes
        nfiles = int(rng.choice([2, 3, 4])) if group_focus else int(rng.choice([1, 1, 2, 3, 4]))
        files = [make_file(rng, tmp, "in%d" % i, genome) for i in range(nfiles)]
        uri = ",".join(files)
        if nfiles > 1 and rng.random() < 0.25:
            open(os.path.join(tmp, "album.txt"), "w").write("\n".join(files) + "\n")
            uri = os.path.join(tmp, "album.txt")
        k = int(rng.choice([9, 15, 21, 27, 31, 32, 33, 47, 63, 64, 65, 96, 127]))
        args = ["-file", uri, "-kmer-size", str(k), "-verbose", "0"]
        amin = rng.choice(["1", "2", "3", "auto"])
        args += ["-abundance-min", str(amin)]
        if rng.random() < 0.3:
            args += ["-abundance-max", str(int(rng.integers(1, 40)))]
        if rng.random() < 0.3:
            args += ["-histo-max", str(int(rng.choice([5, 50, 10000])))]
        kind = "sum"
        if nfiles > 1 and "album" not in uri and rng.random() < 0.6:
            kind = str(rng.choice(["sum", "min", "max", "one", "all", "custom"]))
            args += ["-solidity-kind", kind]
            if kind == "custom":
                args += ["-solidity-custom", "".join(str(int(rng.integers(0, 2))) for _ in range(nfiles))]
            if rng.random() < 0.4:
                args += ["-histo2D", "1"]
        if rng.random() < 0.3:
            args += ["-histo", "1"]
        if rng.random() < 0.3:
            args += ["-nb-partitions", str(int(rng.choice([1, 3, 16])))]
        if rng.random() < 0.2:
            args += ["-out-compress", str(int(rng.choice([1, 6, 9])))]
        gpu_only = []
        if group_focus or rng.random() < 0.25:
            gpu_only += ["-nb-gpus", str(int(rng.choice([2, 4, 8] if group_focus else [2, 4])))]
        elif rng.random() < 0.4:
            gpu_only += ["-device-parse", "1"]
        if rng.random() < 0.3:
            gpu_only += ["-nb-cores", str(int(rng.choice([1, 3])))]
        rg = run(GPU, args + gpu_only + ["-out", "g"], tmp)
        rc = run(CPU, args + ["-out", "c"], tmp)
        if rg.returncode != rc.returncode:
            print(f"seed {seed}: exit codes differ: gpu {rg.returncode} cpu {rc.returncode}\n  args {args + gpu_only}\n  gpu stderr {rg.stderr[-300:]}\n  cpu stderr {rc.stderr[-300:]}")
            sys.exit(1)
        if rg.returncode != 0:
            errs += 1
            continue
        og, oc = outputs(tmp, "g"), outputs(tmp, "c")
        for key in og:
            if og[key] != oc[key]:
                print(f"seed {seed}: {key} differs\n  args {args + gpu_only}\n  dir {tmp}")
                sys.exit(1)
        same += 1
        subprocess.run(["rm", "-rf", tmp])
    print(f"fuzz ok: {same} runs identical, {errs} with the same error exit on both")


if __name__ == "__main__":
    main()
