#!/usr/bin/env python3
"""PCIe-inclusive rate: the read stream starts in HOST memory (dskgpu_push_reads), never reported as bench `value`."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from dsk_amd import KmerCounter, synth
gl, nr, rl = synth.workload("c2_10Mx150")
dev = torch.device("cuda:0")
host = synth.make_reads(synth.make_genome(gl, dev), nr, rl).cpu().numpy()
for attempt in range(3):
    with KmerCounter(kmer_size=31, abundance_min=2) as kc:
        t0 = time.perf_counter()
        kc.reserve_reads(len(host) + 64)
        step = 256 << 20
        for o in range(0, len(host), step):
            kc.push_reads(host[o:o + step])      # note: a separator is implied between pushes (cuts at most 5 k-mers here)
        t1 = time.perf_counter()
        kc.count()
        t2 = time.perf_counter()
        st = kc.stats()
    print(f"attempt {attempt}: push {1e3*(t1-t0):.1f} ms ({len(host)/1e9/(t1-t0):.1f} GB/s), count {1e3*(t2-t1):.1f} ms, "
          f"end-to-end {st['n_kmers']/(t2-t0)/1e9:.2f} G k-mers/s")
