#!/usr/bin/env python3
"""Seeded random FASTA / FASTQ TEXT through dskgpu_push_raw (csrc/rawparse.h) against the CPU oracle's count of the records a host
parser would hand on: random record lengths (0 .. a few thousand), line widths, CRLF / LF, lower case and N, quality lines that
look like headers, files with and without a last newline, several files per read set, and -- the point -- random cuts of the text
between pushes (from single bytes to pieces beyond the 32 MB staging chunk).  Also the stream length and the record count against
the plain statement of the device rules in tests/test_gpu_raw_parse.py.
   python tools/stress_raw.py [first_seed=0] [n_seeds=60]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsk_amd import KmerCounter                                                    # noqa: E402
from tests.oracle_py import Oracle                                                 # noqa: E402
from tests.test_gpu_raw_parse import host_records, model_stream, random_cuts      # noqa: E402

ALPHA = np.frombuffer(b"ACGTACGTACGTACGTacgtN", dtype=np.uint8)


def make_file(rng, big):
    fmt = "fq" if rng.random() < 0.5 else "fa"
    eol = b"\r\n" if rng.random() < 0.25 else b"\n"
    n = int(rng.integers(1, 400)) if not big else int(rng.integers(150_000, 300_000))
    lmax = int(rng.choice([40, 151, 300, 5000])) if not big else 200
    width = int(rng.choice([50, 60, 70, 100000]))
    lens = rng.integers(0 if not big else 100, lmax + 1, n)
    pool = rng.choice(ALPHA, int(lens.sum()) + 1)
    out, pos = [], 0
    for i in range(n):
        L = int(lens[i])
        seq = pool[pos: pos + L].tobytes(); pos += L
        if fmt == "fq":
            q = bytearray(rng.integers(33, 74, L, dtype=np.uint8).tobytes()) if not big else bytearray(b"I" * L)
            if L and i % 3 == 0:
                q[0] = b"@+>"[(i // 3) % 3]
            out += [b"@r%d x" % i, eol, seq, eol, b"+", eol, bytes(q), eol]
        else:
            out += [b">s%d >y" % i, eol]
            for a in range(0, L, width):
                out += [seq[a: a + width], eol]
            if i % 11 == 5:
                out += [eol]
    text = b"".join(out)
    if rng.random() < 0.3:
        text = text[: -len(eol)]
    return text, fmt


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    oracle = Oracle(os.path.join(ROOT, "oracle", "libdsk_oracle.so"))
    assert torch.cuda.is_available()
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        big = seed % 10 == 9                                   # every tenth seed: one file beyond the 32 MB staging chunk
        files = [make_file(rng, big)] if big else [make_file(rng, False) for _ in range(int(rng.integers(1, 4)))]
        k = int(rng.choice([15, 21, 31, 32, 33, 47, 63]))
        pieces = int(rng.choice([1, 2, 7, 50, 1000])) if not big else int(rng.choice([1, 3, 9]))
        with KmerCounter(kmer_size=k, abundance_min=1) as kc:
            for text, fmt in files:
                cuts = random_cuts(rng, len(text), pieces)
                started = False
                for a, b in zip(cuts[:-1], cuts[1:]):
                    if b > a or not started:
                        kc.push_raw(text[a:b], kc.RAW_FASTQ if fmt == "fq" else kc.RAW_FASTA, new_file=not started)
                        started = True
            nbytes, recs = kc.raw_finish()
            kc.count()
            rows, ab = kc.rows()
            st, hist = kc.stats(), kc.histogram()
        host = np.frombuffer(b"".join(host_records(t, f) for t, f in files), dtype=np.uint8)
        ref = oracle.count(host, k)
        want_recs = sum(len(host_records(t, f).split(b"\n")) - 1 for t, f in files)
        ok = (st["n_kmers"] == ref.total and st["n_distinct"] == ref.distinct and (rows == ref.words()).all() and (ab == ref.solid(1)[2]).all()
              and (hist == ref.histogram(10000)).all() and recs == want_recs and (big or nbytes == len(model_stream(files))))
        print(f"seed {seed}: files {[(f, len(t)) for t, f in files]} k {k} pieces {pieces} records {recs} stream {nbytes} kmers {st['n_kmers']} {'ok' if ok else 'MISMATCH'}", flush=True)
        if not ok:
            sys.exit(1)
    print("stress ok")


if __name__ == "__main__":
    main()
