#!/usr/bin/env python3
"""Plan the engine picks for a named workload (stats of one count): tools/plan_of.py [workload] [k]"""
import sys, torch
sys.path.insert(0, ".")
from dsk_amd import KmerCounter, synth
wl = sys.argv[1] if len(sys.argv) > 1 else "c2_10Mx150"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
gl, nr, rl = synth.workload(wl)
dev = torch.device("cuda", 0)
reads = synth.make_reads(synth.make_genome(gl, dev), nr, rl)
with KmerCounter(kmer_size=k, abundance_min=2, timing=True) as kc:
    kc.set_reads_device(reads.data_ptr(), reads.numel())
    kc.count(); kc.count()
    print(wl, k, kc.stats(), {n: round(ms, 3) for n, ms in kc.stage_times()})
