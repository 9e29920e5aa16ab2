#!/usr/bin/env python3
"""Per-rank device time of the multi-GPU path, emulated on ONE GPU (no exchange):
one context acts as rank 0 of `world`, scatters the bench shard (10 M x 150 bp) into records for all
owners, then counts ALL of its own records -- the same number of k-mers a rank receives in the
weak-scaling bench.  Prints stage times and the exchange volume per rank.
   python tools/mg_stage_times.py [world=8] [k=31] [explicit=0] [workload=c2_10Mx150] [slices=0] [partition|global]
slices >= 2: the step in that many slices (dskgpu_mg_slices_*: one sender launch and one level-1 launch per slice) -- the device
work of a rank whose exchange is hidden behind it."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsk_amd import KmerCounter, synth
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
explicit = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
wl = sys.argv[4] if len(sys.argv) > 4 else "c2_10Mx150"
slices = int(sys.argv[5]) if len(sys.argv) > 5 else 0
part = len(sys.argv) > 6 and sys.argv[6] == "partition"          # row order of the rank's result (default: global)
dev = torch.device("cuda", 0)
gl, nr, rl = synth.workload(wl)
reads = synth.make_reads(synth.make_genome(gl, dev), nr, rl)
torch.cuda.synchronize()
kc = KmerCounter(kmer_size=k, abundance_min=2, world_size=world, rank=0, timing=True, mg_explicit=explicit, partition_order=part,
                 stream=torch.cuda.current_stream().cuda_stream)
kc.set_reads_device(reads.data_ptr(), reads.numel())
send = None
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if slices >= 2:
        ns, words, est = kc.mg_slices_prepare(slices)
        assert ns == slices, "this input does not take the sampled send layout"
        cap = kc.mg_send_capacity_words()
        if send is None or send.numel() < cap:
            send = torch.empty(cap, dtype=torch.int64, device=dev)
        for sl in range(ns):
            kc.mg_scatter_slice(send.data_ptr(), send.numel(), sl)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        kc.mg_count_sliced(send.data_ptr(), [sum(w) for w in words], sum(est), lambda sl: None)
        assert not kc.mg_slices_finish()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        counts = [sum(words[sl][o] for sl in range(ns)) for o in range(world)]
        if it == 3:
            print(f"  slices {ns}: estimated k-mers {sum(est)} (real {kc.stats()['n_kmers']}: {sum(est) / kc.stats()['n_kmers'] - 1:+.2%})")
        continue
    cap = kc.mg_send_capacity_words()
    if send is None or send.numel() < cap:
        send = torch.empty(cap, dtype=torch.int64, device=dev)
    counts = kc.mg_scatter(send.data_ptr(), send.numel())
    torch.cuda.synchronize(); t1 = time.perf_counter()
    kc.mg_count(send.data_ptr(), sum(counts), sum(kc.mg_sent_kmers()))
    torch.cuda.synchronize(); t2 = time.perf_counter()
st = kc.stats()
print(f"world={world} k={k} explicit={explicit}: scatter side {1e3*(t1-t0):.2f} ms, count side {1e3*(t2-t1):.2f} ms, total {1e3*(t2-t0):.2f} ms")
print(f"  send words per owner: {counts}  ({8*sum(counts)/1e9:.2f} GB total, {8*sum(counts)/st['n_kmers']:.2f} B per k-mer; "
      f"leaves the GPU: {8*sum(counts)*(world-1)/world/1e9:.2f} GB)")
print("  stages:", ", ".join(f"{n} {ms:.2f}" for n, ms in kc.stage_times()))
print("  stats:", st)
kc.close()
