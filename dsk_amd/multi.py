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

import datetime
import os
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

    def __init__(self, stage, device: torch.device, group=None, balance: bool = True, slices: Optional[int] = None,
                 wait_timeout_s: float = 120.0):
        """wait_timeout_s: longest a rank waits for one slice of the exchange (gloo: the work handle's own timeout; RCCL: the
        process group's watchdog enforces the timeout given to init_process_group -- bench.py passes the same figure there).  A
        wait that fails raises here AND stops the count inside the engine (the gate returns non-zero): no level-1 launch ever
        reads a slice that did not arrive."""
        self.wait_timeout = datetime.timedelta(seconds=wait_timeout_s)
        self.stage = stage
        self.device = device
        self.group = group
        self.balance = balance and hasattr(stage, "mg_sample")
        # a step in slices: the exchange of slice i (async all-to-all on the process group's stream) runs beside the sender of
        # slice i + 1 and the receiver's level 1 of slice i - 1 (the collectives order themselves against torch's current stream:
        # self.stream below, which is also the stage's).  DSK_MG_SLICES = slices per step (default 4; < 2: every step in one piece)
        if slices is None:
            slices = int(os.environ.get("DSK_MG_SLICES", "4"))
        self.slices = slices if hasattr(stage, "mg_slices_prepare") else 0
        self.last_step_sliced = False
        # One stream for the stage's kernels and for what torch.distributed orders its collectives against: a torch side stream
        # handed to the stage.  (A stage created on "torch's current stream" got handle 0 when that was the default stream -- which
        # the C-ABI reads as "a stream of your own", invisible to torch: fine for the one-piece step, whose calls end with host
        # synchronisation, not for slices.)
        self.stream = None
        if device.type == "cuda" and hasattr(stage, "set_stream"):
            self.stream = torch.cuda.Stream(device)
            stage.set_stream(self.stream.cuda_stream)
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
        loads = torch.from_numpy(self.stage.mg_sample().astype("int64"))
        if dist.get_backend(self.group) != "gloo":      # (gloo reduces host tensors anyway; a device tensor would only add a staging hop
            loads = loads.to(self.device)                #  on the step's side stream)
        dist.all_reduce(loads, op=dist.ReduceOp.SUM, group=self.group)
        self.table = make_table(loads.cpu().numpy().astype("uint64"), dist.get_world_size(self.group))
        self.stage.mg_set_table(self.table)

    def _count_in_slices(self) -> bool:
        """One step with the exchange in slices; False when the step has to run in one piece -- some rank's input does not take the
        sampled send layout, or a send slice overflowed (the attempt is then discarded): decided by all ranks together."""
        world, S = dist.get_world_size(self.group), self.slices
        # development path (several ranks sharing one GPU, no RCCL): every slice is staged through host memory -- the same
        # protocol and device work, no overlap
        staged = self.device.type == "cuda" and dist.get_backend(self.group) == "gloo"
        cdev = torch.device("cpu") if staged else self.device
        ns, words, est = self.stage.mg_slices_prepare(S)
        ok = ns == S
        # one host round: per peer, the words of every slice, the estimated k-mers, and whether this rank can run in slices
        sc = torch.tensor([[words[s][p] if ok else 0 for s in range(S)] + [est[p] if ok else 0, 1 if ok else 0] for p in range(world)],
                          dtype=torch.int64, device=cdev)
        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc, group=self.group)
        rows = rc.tolist()
        if not all(int(row[S + 1]) for row in rows):
            return False
        rw = [sum(int(rows[p][s]) for p in range(world)) for s in range(S)]
        sw = [sum(words[s]) for s in range(S)]
        cap = int(self.stage.mg_send_capacity_words())
        if self.send is None or self.send.numel() < cap:
            self.send = torch.empty(max(cap, 1), dtype=torch.int64, device=self.device)
        n_recv = sum(rw)
        if self.recv is None or self.recv.numel() < n_recv:
            self.recv = torch.empty(int(n_recv * 1.1) + 1024, dtype=torch.int64, device=self.device)
        works, landed, so, ro = [], [], 0, 0
        for s in range(S):
            self.stage.mg_scatter_slice(self.send.data_ptr(), self.send.numel(), s)
            src, dst = self.send[so: so + sw[s]], self.recv[ro: ro + rw[s]]
            if staged:
                torch.cuda.current_stream(self.device).synchronize()
                src = src.cpu()
                landed.append((dst, torch.empty(rw[s], dtype=torch.int64)))
                dst = landed[-1][1]
            works.append(dist.all_to_all_single(dst, src, [int(rows[p][s]) for p in range(world)], list(words[s]),
                                                group=self.group, async_op=True))
            so += sw[s]; ro += rw[s]

        gloo = dist.get_backend(self.group) == "gloo"

        def gate(s: int) -> None:                     # the stage's stream waits for slice s (RCCL: an event wait, the host goes on)
            if gloo:
                if works[s].wait(self.wait_timeout) is False:
                    raise RuntimeError(f"slice {s} of the exchange did not arrive within {self.wait_timeout}")
            else:
                works[s].wait()
            if staged:
                landed[s][0].copy_(landed[s][1])
        self.stage.mg_count_sliced(self.recv.data_ptr() if n_recv else 0, rw, sum(int(rows[p][S]) for p in range(world)), gate)
        for w in works:
            w.wait()
        flag = torch.tensor([1 if self.stage.mg_slices_finish() else 0], dtype=torch.int64, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
        if int(flag.item()):
            return False
        self.last_send_counts = [sum(words[s][p] for s in range(S)) for p in range(world)]
        self.last_recv_counts = [sum(int(rows[p][s]) for s in range(S)) for p in range(world)]
        return True

    def count(self) -> None:
        if self.stream is None:
            return self._count()
        self.stream.wait_stream(torch.cuda.current_stream(self.device))       # (the reads, the buffers of the last step)
        with torch.cuda.stream(self.stream):
            self._count()
            self.stream.synchronize()

    def _count(self) -> None:
        if self.balance:                             # part of every count, like the reference's repartition step inside execute()
            self.rebalance()
        self.last_step_sliced = self.slices >= 2 and self._count_in_slices()
        if self.last_step_sliced:
            return
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
