#!/usr/bin/env python3
"""Stress of the sliced step under torch.distributed/nccl with ONE rank: the receive buffer is poisoned with valid records of
OTHER reads before every step, so a level-1 launch that runs before its slice arrived changes the counts.
   python tools/check_sliced_nccl.py [iters=10] [k=31] [slices=4]"""
import os, socket, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DSKGPU_SK_MINSLICE", "1")
from dsk_amd import KmerCounter, synth
from dsk_amd.multi import ShardedCounter
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
S = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda", 0)
sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=dev)
reads = synth.make_reads(synth.make_genome(3_000_000, dev), 1_000_000, 150)
other = synth.make_reads(synth.make_genome(3_000_000, dev, seed=77), 1_000_000, 150, seed=78)
kc = KmerCounter(kmer_size=k, abundance_min=2, world_size=1, rank=0, stream=torch.cuda.current_stream().cuda_stream)
# the poison: the records of the other reads, one piece
kc.set_reads_device(other.data_ptr(), other.numel())
poison = torch.empty(kc.mg_send_capacity_words(), dtype=torch.int64, device=dev)
kc.mg_scatter(poison.data_ptr(), poison.numel())
kc.set_reads_device(reads.data_ptr(), reads.numel())
one = ShardedCounter(kc, dev, slices=1); one.count()
want = (kc.stats()["n_kmers"], kc.stats()["n_distinct"], kc.stats()["n_solid"])
print("one piece:", want)
sc = ShardedCounter(kc, dev, slices=S)
bad = 0
for it in range(iters):
    for buf in (sc.recv, sc.send):                   # (the send buffer too: whatever the sender does not write would travel)
        if buf is not None:
            n = min(buf.numel(), poison.numel())
            buf[:n].copy_(poison[:n])
    torch.cuda.synchronize()
    sc.count()
    got = (kc.stats()["n_kmers"], kc.stats()["n_distinct"], kc.stats()["n_solid"])
    ok = got == want and sc.last_step_sliced
    bad += not ok
    print(f"iter {it}: sliced {sc.last_step_sliced} {got} {'ok' if ok else 'MISMATCH'}")
kc.close()
dist.destroy_process_group()
sys.exit(1 if bad else 0)
