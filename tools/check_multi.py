#!/usr/bin/env python3
"""End-to-end check of the sharded count with real processes on ONE GPU (development aid):
   python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/check_multi.py
Ranks share cuda:0 and exchange over gloo (staged through the host); the summed result must equal a
single-context count of the concatenated reads."""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, ".")
from dsk_amd import KmerCounter, synth
from dsk_amd.multi import ShardedCounter, gather_histogram
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
dev = torch.device("cuda:0")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 31
nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000         # >= ~300 000 per rank with DSKGPU_SK_MINSLICE=1: the step runs in slices
genome = synth.make_genome(3 * nreads * world, dev)
shards = [synth.make_reads(genome, nreads, 150, seed=synth.SEED + 1 + r) for r in range(world)]
kc = KmerCounter(kmer_size=k, abundance_min=2, world_size=world, rank=rank)
kc.set_reads_device(shards[rank].data_ptr(), shards[rank].numel())
sc = ShardedCounter(kc, dev)
sc.count(); sc.count()
st = kc.stats()
sliced = sc.last_step_sliced
hist = gather_histogram(torch.from_numpy(kc.histogram().astype(np.int64)))
tot = torch.tensor([st["n_kmers"], st["n_distinct"], st["n_solid"]], dtype=torch.int64)
dist.all_reduce(tot)
if rank == 0:
    allreads = torch.cat(shards)
    # torch builds `allreads` on ITS stream; the context below runs on its own non-blocking stream.  Without this wait the reference
    # count could read the buffer before torch.cat had written it -- the intermittent r03 failure of
    # test_sliced_step_with_real_processes (tools/stress_multi.py with STRESS_RACE=1 reproduces it at will; profiles/r04_stress/).
    # (dskgpu_set_reads_device now waits for the device itself; the explicit wait documents the contract.)
    torch.cuda.synchronize()
    with KmerCounter(kmer_size=k, abundance_min=2) as one:
        one.set_reads_device(allreads.data_ptr(), allreads.numel())
        one.count()
        s1 = one.stats(); h1 = one.histogram().astype(np.int64)
    assert tot.tolist() == [s1["n_kmers"], s1["n_distinct"], s1["n_solid"]], (tot.tolist(), s1)
    assert (hist.numpy() == h1).all()
    print(f"multi ok: world={world} k={k} sliced={sliced} kmers={s1['n_kmers']} distinct={s1['n_distinct']} solid={s1['n_solid']}")
dist.barrier()
dist.destroy_process_group()
