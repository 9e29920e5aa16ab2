#!/usr/bin/env python3
"""Full-size parity: one count of a bench workload against the CPU oracle, row for row (slow: the oracle walks all k-mers).
usage: tools/check_oracle.py [workload] [k]        (env switches DSKGPU_* apply to the engine)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dsk_amd import KmerCounter, synth
from tests.oracle_py import Oracle

wl = sys.argv[1] if len(sys.argv) > 1 else "c2_10Mx150"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
dev = torch.device("cuda:0")
reads, gl, nr, rl = synth.make_workload(wl, dev)
with KmerCounter(kmer_size=k, abundance_min=2, timing=True) as kc:
    kc.set_reads_device(reads.data_ptr(), reads.numel())
    for _ in range(2):
        kc.count()
    torch.cuda.synchronize()
    rows, ab = kc.rows(); hist = kc.histogram(); st = kc.stats(); stages = dict(kc.stage_times())
print("engine:", st, {a: round(b, 3) for a, b in stages.items()})
t0 = time.time()
ref = Oracle(os.path.join(os.path.dirname(__file__), "..", "oracle", "libdsk_oracle.so")).count(reads.cpu().numpy(), k, threads=os.cpu_count())
print("oracle: total", ref.total, "distinct", ref.distinct, "max", int(ref.ab.max()), f"{time.time() - t0:.1f} s")
keep = ref.ab >= 2
ok = st["n_kmers"] == ref.total and st["n_distinct"] == ref.distinct and (hist == ref.histogram(10000)).all() \
    and rows.shape[0] == int(keep.sum()) and (rows == ref.words()[keep]).all() and (ab == ref.ab[keep]).all()
print("PARITY", "OK" if ok else "MISMATCH")
if not ok:
    h = ref.histogram(10000)
    d = np.nonzero(hist != h)[0]
    print("hist diffs at", d[:20], hist[d[:20]], h[d[:20]])
    sys.exit(1)
