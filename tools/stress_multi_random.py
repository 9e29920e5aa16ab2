#!/usr/bin/env python3
"""Seeded random inputs through the multi-GPU path on ONE device (emulated ranks: every rank's sender, the exchange as tensor slices,
every rank's receiver), against the CPU oracle: world size, key width, passes per rank and the explicit-key form vary with the seed.
   python tools/stress_multi_random.py [first_seed=1000] [n_seeds=20]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from dsk_amd import KmerCounter          # noqa: E402
from tests.oracle_py import Oracle       # noqa: E402
from stress_random import make_input     # noqa: E402


def run(seed, oracle, dev):
    rng = np.random.default_rng(seed ^ 0xABCDEF)
    ks = (21, 27, 31, 32) if rng.random() < 0.5 else (33, 47, 63, 64)
    stream, k, amin, _, desc = make_input(seed, ks)
    world = int(rng.choice([2, 4, 8]))
    mkeys = int(rng.choice([0, 0, 3]))                     # 3: the receive side in several passes
    lines = np.flatnonzero(stream == 10)
    nreads = len(lines)
    cut = [0] + [int(lines[nreads * (r + 1) // world - 1]) + 1 for r in range(world)]
    ctxs, sends, counts, keep = [], [], [], []
    for r in range(world):
        t = torch.from_numpy(stream[cut[r]:cut[r + 1]].copy()).to(dev)
        kw = {"max_pass_mkeys": mkeys} if mkeys else {}
        kc = KmerCounter(kmer_size=k, abundance_min=amin, world_size=world, rank=r, **kw)
        kc.set_reads_device(t.data_ptr(), t.numel())
        send = torch.zeros(max(1, kc.mg_send_capacity_words()), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        counts.append(kc.mg_scatter(send.data_ptr(), send.numel()))
        ctxs.append(kc); sends.append(send); keep.append(t)
    rows_k, rows_a, hist, tot, retries, passes = [], [], np.zeros(10001, np.uint64), 0, 0, 0
    for d in range(world):
        recv = torch.cat([sends[src][sum(counts[src][:d]): sum(counts[src][:d]) + counts[src][d]] for src in range(world)])
        torch.cuda.synchronize()
        ctxs[d].mg_count(recv.data_ptr() if recv.numel() else 0, recv.numel())
        kk, aa = ctxs[d].rows()
        st = ctxs[d].stats()
        rows_k.append(kk); rows_a.append(aa); hist += ctxs[d].histogram(); tot += st["n_kmers"]; retries += st["n_retries"]; passes = max(passes, st["n_passes"])
    for c in ctxs:
        c.close()
    kk = np.concatenate(rows_k); aa = np.concatenate(rows_a)
    ref = oracle.count(stream, k)
    sel = ref.ab >= amin
    order = np.argsort(kk[:, 0], kind="stable") if k <= 32 else np.lexsort((kk[:, 0], kk[:, 1]))
    ok = (tot == ref.total and kk.shape[0] == int(sel.sum()) and (kk[order] == ref.words()[sel]).all() and (aa[order] == ref.ab[sel]).all()
          and (hist == ref.histogram(10000)).all())
    print(f"seed {seed}: world {world} {desc} passes {passes} retries {retries} kmers {ref.total} {'ok' if ok else 'MISMATCH'}", flush=True)
    return ok


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    oracle = Oracle(os.path.join(ROOT, "oracle", "libdsk_oracle.so"))
    dev = torch.device("cuda:0")
    bad = sum(0 if run(s, oracle, dev) else 1 for s in range(first, first + count))
    print("stress ok" if not bad else f"stress FAILED: {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
