"""N>1 path on CPU: world_size-2 (and 4) `gloo` runs of the sharded-count driver
(dsk_amd/multi.py).  The device stages are replaced by a numpy/oracle stand-in
with the same interface and the same owner map as the HIP kernels
(kernels.h key_digit mode 0: owner = (kmix(kmer)[31:12] * G) >> 20); the test
checks that the union of the ranks' results equals the single-process oracle.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def kmix(x):        # dsk_amd/csrc/kmer_device.h::kmix (fold, one multiply, fold)
    x = x.astype(np.uint64).copy()
    x ^= x >> np.uint64(32)
    x *= np.uint64(0xff51afd7ed558ccd)
    x ^= x >> np.uint64(32)
    return x


def kunmix(x):
    x = x.astype(np.uint64).copy()
    x ^= x >> np.uint64(32)
    x *= np.uint64(0x4f74430c22a54005)
    x ^= x >> np.uint64(32)
    return x


def owner_of(h, G):
    return ((((h >> np.uint64(12)) & np.uint64(0xFFFFF)) * np.uint64(G)) >> np.uint64(20)).astype(np.int64)


class CpuStage:
    """Stand-in for KmerCounter's multi-GPU entry points (test only)."""

    def __init__(self, oracle, stream, k, world, can_slice=True):
        self.oracle, self.stream, self.k, self.world = oracle, stream, k, world
        self.rows = None
        self.can_slice = can_slice
        self.gated = []

    # a step in slices (multi.ShardedCounter._count_in_slices): the stand-in cuts its k-mers into S runs by position
    def _owned(self):
        lo, hi, valid = self.oracle.enumerate(self.stream, self.k)
        h = kmix(lo[valid.astype(bool)])
        return h, owner_of(h, self.world)

    def mg_slices_prepare(self, want):
        if not self.can_slice:
            return 0, [], [0] * self.world
        h, own = self._owned()
        cuts = [len(h) * s // want for s in range(want + 1)]
        self.parts = []
        for s in range(want):
            hs, os_ = h[cuts[s]: cuts[s + 1]], own[cuts[s]: cuts[s + 1]]
            order = np.argsort(os_, kind="stable")
            self.parts.append((hs[order], np.bincount(os_, minlength=self.world).tolist()))
        self.sent = np.bincount(own, minlength=self.world).tolist()
        return want, [p[1] for p in self.parts], [int(0.98 * x) for x in self.sent]      # (the k-mer figure is an estimate)

    def mg_scatter_slice(self, ptr, cap, s):
        import ctypes
        off = sum(len(p[0]) for p in self.parts[:s])
        hs = self.parts[s][0]
        if len(hs):
            np.ctypeslib.as_array((ctypes.c_uint64 * (off + len(hs))).from_address(ptr))[off:] = hs

    def mg_slices_finish(self):
        return False

    def mg_count_sliced(self, ptr, slice_words, est, gate):
        import ctypes
        n, got = sum(slice_words), []
        off = 0
        for s, w in enumerate(slice_words):
            gate(s); self.gated.append(s)                # a slice is read only behind its gate
            if w:
                got.append(np.ctypeslib.as_array((ctypes.c_uint64 * (off + w)).from_address(ptr))[off:].copy())
            off += w
        h = np.concatenate(got) if got else np.zeros(0, np.uint64)
        assert abs(est - n) <= 0.05 * max(n, 1)
        keys, cnt = np.unique(kunmix(h), return_counts=True)
        self.rows = (keys, cnt.astype(np.uint32), h)

    def mg_send_capacity_words(self):
        return len(self.stream) + 1

    one_piece = False

    def mg_scatter(self, ptr, cap):
        lo, hi, valid = self.oracle.enumerate(self.stream, self.k)
        h = kmix(lo[valid.astype(bool)])
        own = owner_of(h, self.world)
        order = np.argsort(own, kind="stable")
        h = h[order]
        counts = np.bincount(own, minlength=self.world).tolist()
        import ctypes
        buf = (ctypes.c_uint64 * max(1, len(h))).from_address(ptr)
        np.ctypeslib.as_array(buf)[: len(h)] = h
        self.sent = list(counts)                  # explicit keys: one k-mer per word
        self.one_piece = True
        return counts

    def mg_sent_kmers(self):
        return self.sent

    # minimizer repartition (multi.ShardedCounter.rebalance): the stand-in keeps its own owner map, but takes part in the
    # protocol -- per-rank loads, summed over the ranks, one table derived by every rank from the same sum
    def mg_sample(self):
        loads = np.zeros(4096, dtype=np.uint64)
        loads[: 64] = np.arange(64, dtype=np.uint64) * np.uint64(self.world + 1) + np.uint64(len(self.stream) % 97)
        loads[777] = 10 ** 9                      # one heavy bucket
        return loads

    def mg_set_table(self, table):
        self.table = None if table is None else np.array(table, dtype=np.uint8)

    def mg_count(self, ptr, n, n_kmers=0):
        import ctypes
        assert n_kmers == n                       # the senders' k-mer totals arrived with the word counts
        if n:
            h = np.ctypeslib.as_array((ctypes.c_uint64 * n).from_address(ptr)).copy()
        else:
            h = np.zeros(0, np.uint64)
        keys, cnt = np.unique(kunmix(h), return_counts=True)
        self.rows = (keys, cnt.astype(np.uint32), h)


def _worker(rank, world, port, k, tmpdir, mode="slices"):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dsk_amd.multi import ShardedCounter, gather_histogram
    from tests.oracle_py import Oracle
    oracle = Oracle(os.path.join(ROOT, "oracle", "libdsk_oracle.so"))
    stream, _ = oracle.load_bank(os.path.join(ROOT, "tests", "golden", "read50x_ref10K_e001.fasta.gz"))
    # shard reads by record: rank r takes records r, r+world, ...
    recs = bytes(stream).split(b"\n")
    mine = b"\n".join(recs[rank::world]) + b"\n"
    shard = np.frombuffer(mine, dtype=np.uint8)
    stage = CpuStage(oracle, shard, k, world, can_slice=not (mode == "one_rank_cannot" and rank == 1))
    sc = ShardedCounter(stage, torch.device("cpu"), slices=1 if mode == "one_piece" else 4)
    sc.count()
    # the step ran in slices (every slice read behind its gate, in order) -- or, decided by all ranks together, in one piece
    assert sc.last_step_sliced == (mode == "slices") and stage.one_piece == (mode != "slices")
    assert stage.gated == (list(range(4)) if mode == "slices" else [])
    # the repartition table: same on every rank (built from the all-reduced loads), heavy bucket split, owners inside the world
    tabs = [torch.zeros(4096, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(tabs, torch.from_numpy(stage.table.copy()))
    assert all((t == tabs[0]).all() for t in tabs) and int(tabs[0][777]) == 255 and int(tabs[0][tabs[0] != 255].max()) < world
    keys, cnt, h = stage.rows
    assert (owner_of(h, world) == rank).all()            # every received record is owned by this rank
    assert sum(sc.last_recv_counts) == len(h)
    hist = np.bincount(np.minimum(cnt, 10000), minlength=10001).astype(np.int64)
    total = gather_histogram(torch.from_numpy(hist))
    np.save(os.path.join(tmpdir, f"keys{rank}.npy"), keys)
    np.save(os.path.join(tmpdir, f"cnt{rank}.npy"), cnt)
    if rank == 0:
        np.save(os.path.join(tmpdir, "hist.npy"), total.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,mode", [(2, "slices"), (4, "slices"), (2, "one_piece"), (4, "one_rank_cannot")])
def test_sharded_count_equals_single_process(oracle, golden_dir, tmp_path, world, mode):
    k = 31
    mp.spawn(_worker, args=(world, _free_port(), k, str(tmp_path), mode), nprocs=world, join=True)
    keys = np.concatenate([np.load(tmp_path / f"keys{r}.npy") for r in range(world)])
    cnt = np.concatenate([np.load(tmp_path / f"cnt{r}.npy") for r in range(world)])
    order = np.argsort(keys)
    keys, cnt = keys[order], cnt[order]
    s, _ = oracle.load_bank(os.path.join(golden_dir, "read50x_ref10K_e001.fasta.gz"))
    ref = oracle.count(s, k)
    assert len(np.unique(keys)) == len(keys)              # owners are disjoint
    assert (keys == ref.lo).all() and (cnt == ref.ab).all()
    assert (np.load(tmp_path / "hist.npy") == ref.histogram(10000).astype(np.int64)).all()
    assert (ref.total, ref.distinct) == (350000, 99957)   # BASELINE.json configs[0]


def test_exchange_ragged_and_empty(tmp_path):
    mp.spawn(_exchange_worker, args=(3, _free_port()), nprocs=3, join=True)


def _exchange_worker(rank, world, port):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dsk_amd.multi import exchange
    # rank r sends (r + d) % 3 words to rank d, tagged with (src, dst); rank 2 sends nothing to itself ... etc.
    counts = [(rank + d) % 3 for d in range(world)]
    send = torch.tensor([rank * 100 + d for d in range(world) for _ in range(counts[d])] + [0], dtype=torch.int64)
    out, rc = exchange(send, counts)
    assert rc == [(s + rank) % 3 for s in range(world)]
    want = [s * 100 + rank for s in range(world) for _ in range((s + rank) % 3)]
    assert out.tolist() == want
    dist.barrier()
    dist.destroy_process_group()
