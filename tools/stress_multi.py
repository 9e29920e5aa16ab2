#!/usr/bin/env python3
"""Stress of the sharded count with REAL processes sharing ONE GPU (exchange over gloo, every slice staged through the host):
what tests/test_gpu_parity.py::test_sliced_step_with_real_processes runs once, looped, with the evidence kept.

   python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 tools/stress_multi.py [k=31] [reads=400000] [iters=50]

* every rank prints its own exception (rank id + traceback) before it exits, so the first failing rank is in the log;
* send and receive buffers are poisoned with VALID records of other reads before every step (a level-1 launch that runs before
  its slice arrived, or a slice the sender did not fill, changes the counts);
* every sliced step is compared with the one-piece step of the same reads AND with a single-context count of all reads;
* the single-context reference is computed twice on rank 0: once straight after torch.cat (no synchronisation between torch's
  stream, which builds the concatenated reads, and the context's own non-blocking stream) and once after torch.cuda.synchronize().
  STRESS_RACE=1 puts a long matmul chain in front of the torch.cat, which makes that window deterministic.  The r03 intermittent
  failure of test_sliced_step_with_real_processes is exactly a difference between the two references (NOTEBOOK.md section 5).
Exit code 0 = every iteration agreed with the synchronised reference."""
import os
import sys
import traceback

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DSKGPU_SK_MINSLICE", "1")
from dsk_amd import KmerCounter, synth                  # noqa: E402
from dsk_amd.multi import ShardedCounter, gather_histogram   # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])


def totals(kc):
    st = kc.stats()
    hist = gather_histogram(torch.from_numpy(kc.histogram().astype(np.int64)))
    tot = torch.tensor([st["n_kmers"], st["n_distinct"], st["n_solid"]], dtype=torch.int64)
    dist.all_reduce(tot)
    return tot.tolist(), hist.numpy()


def single_context(allreads, k):
    with KmerCounter(kmer_size=k, abundance_min=2) as one:
        one.set_reads_device(allreads.data_ptr(), allreads.numel())
        one.count()
        s = one.stats()
        return [s["n_kmers"], s["n_distinct"], s["n_solid"]], one.histogram().astype(np.int64)


def body():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 31
    nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 400_000
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    dist.init_process_group("gloo")
    dev = torch.device("cuda:0")
    genome = synth.make_genome(3 * nreads * world, dev)
    shards = [synth.make_reads(genome, nreads, 150, seed=synth.SEED + 1 + r) for r in range(world)]
    other = synth.make_reads(synth.make_genome(3 * nreads, dev, seed=77), nreads, 150, seed=78 + rank)
    torch.cuda.synchronize()
    # ---- the reference, twice (rank 0)
    want, want_hist = None, None
    if rank == 0:
        if os.environ.get("STRESS_RACE") == "1":           # a long queue on torch's stream in front of the cat
            x = torch.randn(8192, 8192, device=dev)
            for _ in range(40):
                x = (x @ x) * 1e-4
        allreads = torch.cat(shards)
        racy, racy_hist = single_context(allreads, k)      # the context's own stream does not wait for torch's
        torch.cuda.synchronize()
        want, want_hist = single_context(allreads, k)
        same = racy == want and (racy_hist == want_hist).all()
        print(f"[rank 0] reference straight after torch.cat {racy}, after synchronize {want}: "
              + ("identical" if same else "DIFFERENT -- the unsynchronised reference read the reads before torch.cat had written them"), flush=True)
        del allreads
    # ---- poison: records of other reads, valid for this world size
    kc = KmerCounter(kmer_size=k, abundance_min=2, world_size=world, rank=rank)
    kc.set_reads_device(other.data_ptr(), other.numel())
    poison = torch.empty(kc.mg_send_capacity_words(), dtype=torch.int64, device=dev)
    kc.mg_scatter(poison.data_ptr(), poison.numel())
    kc.set_reads_device(shards[rank].data_ptr(), shards[rank].numel())
    one_piece = ShardedCounter(kc, dev, slices=1)
    one_piece.count()
    base, base_hist = totals(kc)
    del one_piece
    sc = ShardedCounter(kc, dev, slices=4)
    bad = 0
    for it in range(iters):
        for buf in (sc.recv, sc.send):
            if buf is not None:
                n = min(buf.numel(), poison.numel())
                buf[:n].copy_(poison[:n])
        torch.cuda.synchronize()
        sc.count()
        got, hist = totals(kc)
        ok = got == base and (hist == base_hist).all() and sc.last_step_sliced
        if rank == 0:
            ok = ok and got == want and (hist == want_hist).all()
            bad += not ok
            if not ok or it % 10 == 0 or it + 1 == iters:
                print(f"[rank 0] iter {it}: sliced {sc.last_step_sliced} {got} one-piece {base} single-context {want} {'ok' if ok else 'MISMATCH'}", flush=True)
    flag = torch.tensor([bad], dtype=torch.int64)
    dist.broadcast(flag, 0)
    if rank == 0:
        print(f"stress ok: world={world} k={k} iters={iters}" if not bad else f"stress FAILED: {bad} of {iters} iterations", flush=True)
    kc.close()
    dist.barrier()
    dist.destroy_process_group()
    return int(flag.item())


if __name__ == "__main__":
    try:
        rc = body()
    except BaseException:          # every rank says what happened to it before the launcher tears the others down
        sys.stderr.write(f"[rank {rank}] FAILED:\n{traceback.format_exc()}\n")
        sys.stderr.flush()
        sys.exit(1)
    sys.exit(1 if rc else 0)
