#!/usr/bin/env python3
"""Seeded random two-level inputs at two-word key widths (k = 33..64) against the CPU oracle, many seeds: the round-4 two-word paths
(k_count2v3 + its re-count, 2560-key sub-partitions, rowsort2.h) under varying read length, coverage, N rate, read order, repeats.
   python tools/stress_random.py [first_seed=100] [n_seeds=40]
STRESS_KS=15,21,27,31,32 draws k from another list (one-word keys); STRESS_K=k counts the seed's input at another k."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsk_amd import KmerCounter          # noqa: E402
from tests.oracle_py import Oracle       # noqa: E402


def make_input(seed, ks=(33, 34, 41, 47, 55, 62, 63, 64)):
    """-> (stream, k, abundance_min, KmerCounter keywords, description) of one seeded input"""
    rng = np.random.default_rng(seed)
    k = int(rng.choice(list(ks)))
    rl = int(rng.choice([max(k + 5, 80), 150, 251, 1000]))
    n_kmers = int(rng.choice([9_000_000, 12_000_000, 20_000_000]))
    n_reads = n_kmers // (rl - k + 1) + 1
    cov = float(rng.choice([1.5, 8.0, 40.0, 200.0]))
    glen = max(1000, int(n_reads * rl / cov))
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=glen)
    kind = rng.random()
    if kind < 0.3:                                   # a tandem repeat: heavy k-mers
        unit = genome[:37].copy(); genome[glen // 2: glen // 2 + 37 * 400] = np.tile(unit, 400)[: min(37 * 400, glen - glen // 2)]
    elif kind < 0.5:                                 # low complexity: a poly-A stretch with sparse substitutions
        span = min(glen // 3, 200_000)
        pa = np.full(span, ord("A"), np.uint8)
        hit = rng.random(span) < 0.02
        pa[hit] = rng.choice(np.frombuffer(b"CGT", dtype=np.uint8), size=int(hit.sum()))
        genome[:span] = pa
    starts = rng.integers(0, max(1, glen - rl), size=n_reads)
    if rng.random() < 0.5:
        starts.sort()
    idx = starts[:, None] + np.arange(rl)[None, :]
    reads = genome[np.minimum(idx, glen - 1)]
    nrate = float(rng.choice([0.0, 0.001, 0.01]))
    reads = np.where(rng.random(reads.shape) < nrate, np.uint8(ord("N")), reads)
    err = rng.random(reads.shape) < float(rng.choice([0.0, 0.01]))
    sub = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=reads.shape)
    reads = np.where(err & (reads != ord("N")), sub, reads)
    flip = rng.random(n_reads) < 0.5
    comp = np.zeros(256, np.uint8); comp[list(b"ACGTN")] = list(b"TGCAN")
    reads[flip] = comp[reads[flip]][:, ::-1]
    stream = np.concatenate([reads, np.full((n_reads, 1), ord("\n"), np.uint8)], axis=1).reshape(-1)
    amin = int(rng.choice([1, 2, 3]))
    kw = {"max_pass_mkeys": 4} if rng.random() < 0.25 else {}
    return stream, k, amin, kw, f"k {k} rl {rl} cov {cov} N {nrate} kind {kind:.2f}"


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    oracle = Oracle(os.path.join(ROOT, "oracle", "libdsk_oracle.so"))
    dev = torch.device("cuda:0")
    ks = [int(x) for x in os.environ["STRESS_KS"].split(",")] if os.environ.get("STRESS_KS") else (33, 34, 41, 47, 55, 62, 63, 64)
    bad = 0
    for seed in range(first, first + count):
        stream, k, amin, kw, desc = make_input(seed, ks)
        if os.environ.get("STRESS_K"):
            k = int(os.environ["STRESS_K"])              # (the same input counted at another k)
        t = torch.from_numpy(stream).to(dev)
        part = bool(os.environ.get("STRESS_PARTITION"))      # rows in the reference's order (DSKGPU_F_PARTITION_ORDER): ascending inside every output partition
        with KmerCounter(kmer_size=k, abundance_min=amin, partition_order=part, **kw) as kc:
            kc.set_reads_device(t.data_ptr(), t.numel())
            kc.count(); torch.cuda.synchronize()
            rows, ab = kc.rows(); hist = kc.histogram(); st = kc.stats()
            off = kc.partition_offsets().astype(np.int64) if part else None
        ref = oracle.count(stream, k)
        keep = ref.ab >= amin
        part_ok = True
        if part and rows.shape[0] == int(keep.sum()) and rows.shape[0] > 1:
            w = rows.shape[1]
            asc = rows[1:, w - 1] > rows[:-1, w - 1]
            for x in range(w - 2, -1, -1):
                asc = asc | ((rows[1:, x + 1:] == rows[:-1, x + 1:]).all(axis=1) & (rows[1:, x] > rows[:-1, x]))
            st_ = off[1:-1]; st_ = st_[(st_ > 0) & (st_ <= len(asc))]
            asc[st_ - 1] = True
            part_ok = bool(asc.all()) and int(off[-1]) == rows.shape[0] and bool((np.diff(off) >= 0).all())
            order = np.lexsort(rows.T)
            rows, ab = rows[order], ab[order]
        ok = (part_ok and st["n_kmers"] == ref.total and st["n_distinct"] == ref.distinct and rows.shape[0] == int(keep.sum())
              and (rows == ref.words()[keep]).all() and (ab == ref.ab[keep]).all() and (hist == ref.histogram(10000)).all())
        bad += not ok
        print(f"seed {seed}: {desc} (counted at k {k}) passes {st['n_passes']} levels {st['n_levels']} retries {st['n_retries']} "
              f"fallback {st['sort_fallback']} partitions {st['n_partitions']} ext {st['n_ext_regions']} heavy {st['n_heavy']} kmers {ref.total} distinct {ref.distinct} {'ok' if ok else 'MISMATCH'}", flush=True)
    print("stress ok" if not bad else f"stress FAILED: {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
