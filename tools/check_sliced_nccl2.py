#!/usr/bin/env python3
"""The sliced step under nccl with one rank on DIRTY memory: device memory is filled with valid records of other reads first (torch's
cache and the driver's free pool), then the steps of tests/test_gpu_parity.py::test_exchange_over_rccl_single_rank run."""
import os, socket, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DSKGPU_SK_MINSLICE", "1")
from dsk_amd import KmerCounter, synth
from dsk_amd.multi import ShardedCounter
dev = torch.device("cuda", 0)
sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=dev)
other = synth.make_reads(synth.make_genome(3_000_000, dev, seed=77), 2_000_000, 150, seed=78)
for k in (31, 63):
    with KmerCounter(kmer_size=k, abundance_min=2, world_size=1, rank=0, stream=torch.cuda.current_stream().cuda_stream) as kc:
        kc.set_reads_device(other.data_ptr(), other.numel())
        poison = torch.empty(kc.mg_send_capacity_words(), dtype=torch.int64, device=dev)
        kc.mg_scatter(poison.data_ptr(), poison.numel())
        kc.mg_count(poison.data_ptr(), poison.numel() - 1)
    # dirty torch's cache: many copies of the poison, then freed
    junk = [poison.clone() for _ in range(40)]
    torch.cuda.synchronize()
    del junk
    reads = synth.make_reads(synth.make_genome(1_000_000, dev), 300_000, 150)
    with KmerCounter(kmer_size=k, abundance_min=2, world_size=1, rank=0, stream=torch.cuda.current_stream().cuda_stream) as kc:
        kc.set_reads_device(reads.data_ptr(), reads.numel())
        one = ShardedCounter(kc, dev, slices=1); one.count()
        want = (kc.stats()["n_kmers"], kc.stats()["n_distinct"], kc.stats()["n_solid"])
        del one
        sc = ShardedCounter(kc, dev, slices=4)
        for it in range(4):
            sc.count()
            got = (kc.stats()["n_kmers"], kc.stats()["n_distinct"], kc.stats()["n_solid"])
            print(f"k {k} iter {it}: sliced {sc.last_step_sliced} {got} want {want} {'ok' if got == want else 'MISMATCH'} retries {kc.stats()['n_retries']}")
dist.destroy_process_group()
