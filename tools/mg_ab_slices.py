#!/usr/bin/env python3
"""One-piece and sliced step of the emulated rank (tools/mg_stage_times.py) alternating inside one process: stage times per iteration.
   python tools/mg_ab_slices.py [world=8] [k=31] [slices=4] [iters=6]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsk_amd import KmerCounter, synth
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
S = int(sys.argv[3]) if len(sys.argv) > 3 else 4
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 6
dev = torch.device("cuda", 0)
gl, nr, rl = synth.workload("c2_10Mx150")
reads = synth.make_reads(synth.make_genome(gl, dev), nr, rl)
kc = KmerCounter(kmer_size=k, abundance_min=2, world_size=world, rank=0, timing=True, stream=torch.cuda.current_stream().cuda_stream)
kc.set_reads_device(reads.data_ptr(), reads.numel())
send = torch.empty(kc.mg_send_capacity_words() + (1 << 20), dtype=torch.int64, device=dev)
for it in range(iters):
    sliced = it % 2 == 1
    if sliced:
        ns, words, est = kc.mg_slices_prepare(S)
        for sl in range(ns):
            kc.mg_scatter_slice(send.data_ptr(), send.numel(), sl)
        kc.mg_count_sliced(send.data_ptr(), [sum(w) for w in words], sum(est), lambda sl: None)
        kc.mg_slices_finish()
    else:
        counts = kc.mg_scatter(send.data_ptr(), send.numel())
        kc.mg_count(send.data_ptr(), sum(counts), sum(kc.mg_sent_kmers()))
    torch.cuda.synchronize()
    st = dict(kc.stage_times())
    print(("sliced   " if sliced else "one piece"), " ".join(f"{n} {st.get(n, 0):.2f}" for n in ("mg_scatter", "scatter1", "scatter2", "count", "sort")), f"total {sum(st.values()):.2f}")
kc.close()
