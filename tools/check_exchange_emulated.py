"""Multi-GPU path emulated on one device (several contexts, exchange by hand) for a sweep of (world, k), against the oracle."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsk_amd import KmerCounter
from tests.oracle_py import Oracle
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
oracle = Oracle(os.path.join(root, "oracle", "libdsk_oracle.so"))
s, _ = oracle.load_bank(os.path.join(root, "tests", "golden", "read50x_ref10K_e001.fasta.gz"))
recs = bytes(s).split(b"\n")
dev = torch.device("cuda", 0)
for world, k in [(2, 27), (4, 31), (4, 27), (1, 27), (2, 20), (2, 45), (2, 46), (2, 64), (8, 31)]:
    if world == 1:
        continue
    ctxs, sends, counts = [], [], []
    for r in range(world):
        shard = torch.from_numpy(np.frombuffer(b"\n".join(recs[r::world]) + b"\n", dtype=np.uint8).copy()).to(dev)
        kc = KmerCounter(kmer_size=k, abundance_min=1, world_size=world, rank=r)
        kc.set_reads_device(shard.data_ptr(), shard.numel())
        send = torch.zeros(kc.mg_send_capacity_words(), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()      # the contexts run on their own streams
        c = kc.mg_scatter(send.data_ptr(), send.numel())
        ctxs.append(kc); sends.append(send); counts.append(c)
    tot = 0; hist = np.zeros(10001, np.uint64)
    for d in range(world):
        parts = [sends[src][sum(counts[src][:d]): sum(counts[src][:d]) + counts[src][d]] for src in range(world)]
        recv = torch.cat(parts)
        torch.cuda.synchronize()
        ctxs[d].mg_count(recv.data_ptr(), recv.numel())
        tot += ctxs[d].stats()["n_kmers"]; hist += ctxs[d].histogram()
    ref = oracle.count(s, k)
    print(world, k, "kmers", tot, ref.total, "hist ok", bool((hist == ref.histogram(10000)).all()), "words", [sum(c) for c in counts], flush=True)
    for c in ctxs: c.close()
