"""Multi-GPU count: one process per GPU, k-mer space sharded by owner(kmer).

    rank r:  reads shard --mg_scatter--> super-k-mer records grouped by owner (owner = f(minimizer);
                                          explicit keys for k < 20), ~2.3 B per k-mer instead of 8
             --RCCL all-to-all (torch.distributed)--> records this rank owns
             --mg_count--> expand + partition + hash-aggregate + histogram + solid rows

The reference has no distributed mode (single process, disk partitions:
doc/paper.tex:60-97); the owner map plays the role of its partition function
across GPUs and the all-to-all replaces the partition files.  torch is
plumbing here: it owns the exchange buffers and the process group.  `stage` is
a `KmerCounter` on a GPU; the CPU tests pass a stand-in with the same two
methods to exercise this driver under gloo.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def exchange(send: torch.Tensor, send_counts: Sequence[int], group=None,
             recv: Optional[torch.Tensor] = None, send_kmers: Optional[Sequence[int]] = None,
             recv_kmers: Optional[List[int]] = None) -> Tuple[torch.Tensor, List[int]]:
    """Variable-size all-to-all of 8-byte words.  send[: sum(send_counts)] is grouped by
    destination rank.  Returns (recv buffer view, per-source counts).  send_kmers (k-mers inside the
    words for every destination) rides along with the counts; the per-source figures are appended to recv_kmers."""
    world = dist.get_world_size(group)
    assert len(send_counts) == world
    if send.is_cuda and dist.get_backend(group) == "gloo":
        # development path (several ranks sharing one GPU, no RCCL): stage the exchange through host memory
        out_cpu, rcounts = exchange(send[: sum(send_counts)].cpu(), send_counts, group, None, send_kmers, recv_kmers)
        n = out_cpu.numel()
        if recv is None or recv.numel() < n:
            recv = torch.empty(max(n, 1), dtype=send.dtype, device=send.device)
        recv[:n].copy_(out_cpu)
        return recv[:n], rcounts
    dev = send.device
    km = list(send_kmers) if send_kmers is not None else [0] * world
    sc = torch.tensor([[int(c), int(k)] for c, k in zip(send_counts, km)], dtype=torch.int64, device=dev)
    rc = torch.empty_like(sc)
    dist.all_to_all_single(rc, sc, group=group)          # one round: (words, k-mers) per peer
    pairs = rc.tolist()
    recv_counts = [int(x[0]) for x in pairs]
    if recv_kmers is not None:
        recv_kmers.extend(int(x[1]) for x in pairs)
    n_recv = sum(recv_counts)
    if recv is None or recv.numel() < n_recv:
        recv = torch.empty(int(n_recv * 1.1) + 1024, dtype=send.dtype, device=dev)   # head-room: sizes wobble step to step
    out = recv[:n_recv]
    dist.all_to_all_single(out, send[: sum(send_counts)], recv_counts, list(send_counts), group=group)
    return out, recv_counts


def scatter_records(stage, send: Optional[torch.Tensor], device: torch.device) -> Tuple[torch.Tensor, List[int]]:
    """mg_send_capacity_words + mg_scatter into `send` (re-allocated when too small).  The capacity normally comes
    from a sampled estimate; if a slice of the send layout overflows, the engine switches to exact counts, which can
    need a larger buffer than estimated -- then the capacity is asked again and the scatter repeated once."""
    for attempt in range(2):
        cap = int(stage.mg_send_capacity_words())
        if send is None or send.numel() < cap:
            send = torch.empty(max(cap, 1), dtype=torch.int64, device=device)
        try:
            return send, stage.mg_scatter(send.data_ptr(), send.numel())
        except RuntimeError as e:
            if attempt == 0 and "send buffer too small" in str(e):
                continue
            raise
    raise AssertionError("unreachable")


class ShardedCounter:
    """Drives one rank of the sharded count.  After `count()`, the stage holds
    this rank's share of the result (its owned k-mers)."""

    def __init__(self, stage, device: torch.device, group=None, balance: bool = True):
        self.stage = stage
        self.device = device
        self.group = group
        self.balance = balance and hasattr(stage, "mg_sample")
        self.table = None            # the repartition table in use (None = the engine's default)
        self.send: Optional[torch.Tensor] = None
        self.recv: Optional[torch.Tensor] = None
        self.last_send_counts: List[int] = []
        self.last_recv_counts: List[int] = []

    def rebalance(self) -> None:
        """Minimizer repartition (gatb-core's RepartitorAlgorithm): every rank samples the k-mer load of its reads per
        minimizer bucket, the loads are summed over the ranks (one 32 KB all-reduce) and every rank derives the same table
        from the sum -- heavy buckets are split by k-mer, the others placed largest first on the least loaded owner."""
        from .engine import make_table
        loads = torch.from_numpy(self.stage.mg_sample().astype("int64")).to(self.device)
        dist.all_reduce(loads, op=dist.ReduceOp.SUM, group=self.group)
        self.table = make_table(loads.cpu().numpy().astype("uint64"), dist.get_world_size(self.group))
        self.stage.mg_set_table(self.table)

    def count(self) -> None:
        if self.balance:                             # part of every count, like the reference's repartition step inside execute()
            self.rebalance()
        self.send, counts = scatter_records(self.stage, self.send, self.device)
        sized = hasattr(self.stage, "mg_sent_kmers")      # the senders counted the k-mers they packed: the receiver need not
        rk: List[int] = []
        out, rcounts = exchange(self.send, counts, self.group, self.recv, self.stage.mg_sent_kmers() if sized else None, rk)
        if self.recv is None or self.recv.numel() < out.numel() or out.data_ptr() != self.recv.data_ptr():
            self.recv = out._base if out._base is not None else out        # keep the (larger) backing buffer for the next step
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()
        self.last_send_counts, self.last_recv_counts = list(counts), rcounts
        if sized:
            self.stage.mg_count(out.data_ptr() if out.numel() else 0, int(out.numel()), sum(rk))
        else:
            self.stage.mg_count(out.data_ptr() if out.numel() else 0, int(out.numel()))


def gather_histogram(hist: torch.Tensor, group=None) -> torch.Tensor:
    """Whole-job histogram = element-wise sum of the ranks' histograms (owners are disjoint)."""
    h = hist.clone()
    dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
    return h
