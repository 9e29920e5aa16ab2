import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from dsk_amd import KmerCounter, synth
dev = torch.device("cuda", 0)
reads, gl, nr, rl = synth.make_workload("c2_10Mx150", dev)
torch.cuda.synchronize()
kc = KmerCounter(kmer_size=31, abundance_min=2)
kc.set_reads_device(reads.data_ptr(), reads.numel())
for i in range(3):
    t = time.perf_counter(); kc.count(); print(f"PLACE={os.environ.get('DSKGPU_PLACE','0')} count {i}: {1e3 * (time.perf_counter() - t):.1f} ms")
