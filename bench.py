#!/usr/bin/env python3
"""bench.py -- headline benchmark of the count path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--kmer-size 31]

One "step" = one pass of the whole hot path (2-bit encode -> canonical k-mers ->
radix partition -> LDS hash aggregate -> histogram + solidity filter -> sorted
rows) over one batch of synthetic reads that is already resident in HBM.
N=1 workload = BASELINE.json configs[1]: 10 M x 150 bp, k=31, 50x of a 30 Mbp
random genome (dsk_amd/synth.py).  N>1 (launched by torch.distributed.run, one
rank per GPU): every rank holds its own 10 M-read shard (weak scaling) of an
N x 30 Mbp genome; k-mers are routed to their owner GPU with one RCCL
all-to-all between the scatter and the count stages.

Prints ONE JSON line on rank 0.  `value` = distinct k-mers counted per second,
whole job.  `roofline` prices the dominant kernel against HBM; `cpu_baseline`
times the CPU oracle (a restatement of DSK's method, NOT GATB/dsk itself, which
cannot be built here: its gatb-core submodule is absent) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2_10Mx150")
    ap.add_argument("--kmer-size", type=int, default=31)
    ap.add_argument("--abundance-min", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-reads", type=int, default=400_000)
    ap.add_argument("--no-sort", action="store_true")
    return ap.parse_args()


def algorithmic_bytes(stage, n_bytes, n_kmers, W=8):
    """Algorithmic HBM bytes of one launch (DESIGN.md 'roofline'; SURVEY.md §8d terms).
    partition kernels: read the bases that produce the k-mers + write W per k-mer;
    key-array passes: read W (+ write W); count: read W per k-mer (table lives in LDS)."""
    enc_words = (n_bytes + 31) // 32
    packed = enc_words * 12                      # 8 B packed + 4 B invalid mask per 32 bases
    return {
        "encode": n_bytes + packed,
        "hist1": packed,
        "scatter1": packed + n_kmers * W,
        "hist2": n_kmers * W,
        "scatter2": 2 * n_kmers * W,
        "count": n_kmers * W,
        "mg_hist": packed,
        "mg_scatter": packed + n_kmers * W,
    }.get(stage)


def cpu_baseline(reads_u8, read_len, n_sample_reads, k, target_s=15.0):
    """Time the CPU oracle (oracle/dsk_oracle.c: partition + sort-count, all host cores) on a
    bounded sample of the SAME stream.  A short probe sizes the sample for ~target_s of CPU work."""
    from tests.oracle_py import Oracle
    so = os.path.join(ROOT, "oracle", "libdsk_oracle.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libdsk_oracle.so"])
    oracle = Oracle(so)
    total_reads = reads_u8.numel() // (read_len + 1)
    cores = os.cpu_count() or 1
    probe_reads = min(n_sample_reads, total_reads)
    sample = reads_u8[: probe_reads * (read_len + 1)].cpu().numpy()
    t0 = time.perf_counter()
    r = oracle.count(sample, k, threads=cores)
    dt = time.perf_counter() - t0
    n_reads = probe_reads
    # grow the sample towards target_s (bounded by the workload and by 16 GiB of k-mer keys)
    want = int(min(total_reads, probe_reads * max(1.0, target_s / max(dt, 1e-3)), 16 * 2**30 // (8 * (read_len - k + 1))))
    if want > probe_reads * 2:
        n_reads = want
        sample = reads_u8[: n_reads * (read_len + 1)].cpu().numpy()
        t0 = time.perf_counter()
        r = oracle.count(sample, k, threads=cores)
        dt = time.perf_counter() - t0
    return {
        "value": r.distinct / dt,
        "unit": "distinct k-mers/s",
        "kmer_occurrences_per_s": r.total / dt,
        "cores": cores,
        "kind": "port",
        "sample": f"first {n_reads} reads of the same synthetic stream ({r.total} k-mer occurrences, "
                  f"{r.distinct} distinct), oracle/dsk_oracle.c partition+sort-count, {cores} threads, {dt:.2f} s",
        "note": "CPU restatement of DSK's method, NOT GATB/dsk (gatb-core submodule absent => reference unbuildable)",
    }


def self_launch(args):
    """`python bench.py --gpus N` started bare (no WORLD_SIZE): become the launcher of the N ranks.  This process never
    touches the GPU (torch is not even imported here): the ranks are children of torch.distributed.run, rank 0's JSON line
    is relayed on stdout and the launcher's exit code is theirs."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: needed by RCCL between processes on this pool
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    line = None
    for ln in proc.stdout.decode(errors="replace").splitlines():
        if ln.startswith('{"metric"'):
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line:
        print(line)
    sys.exit(proc.returncode if proc.returncode else (0 if line else 1))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    import torch
    import torch.distributed as dist
    from dsk_amd import KmerCounter, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the count path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    gl, nr, rl = synth.workload(args.workload)
    # weak scaling: same per-GPU shard size, genome grows with the node so coverage stays 50x
    genome = synth.make_genome(gl * world, dev)
    reads = synth.make_reads(genome, nr, rl, seed=synth.SEED + 1 + rank)
    del genome
    torch.cuda.synchronize()
    n_bytes = reads.numel()

    stream = torch.cuda.current_stream().cuda_stream
    kc = KmerCounter(kmer_size=args.kmer_size, abundance_min=args.abundance_min, device=local_rank, timing=True,
                     sort=not args.no_sort, world_size=world, rank=rank, stream=stream)
    kc.set_reads_device(reads.data_ptr(), n_bytes)

    sharded = None
    if world > 1:
        from dsk_amd.multi import ShardedCounter
        sharded = ShardedCounter(kc, dev)

    stage_acc = {}

    def step():
        if world == 1:
            kc.count()
        else:
            sharded.count()      # mg_scatter -> RCCL all-to-all -> mg_count
        for name, ms in kc.stage_times():
            stage_acc.setdefault(name, []).append(ms)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    stage_acc.clear()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    st = kc.stats()

    # whole-job aggregates (MAX time over ranks, SUM of units)
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    units = torch.tensor([st["n_distinct"], st["n_kmers"], st["n_solid"], n_bytes], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(units, op=dist.ReduceOp.SUM)
        # n_kmers reported by mg_count is what this rank RECEIVED; the sum over ranks is the job total
    dt = float(tt.item())
    n_distinct, n_kmers, n_solid, tot_bytes = [float(x) for x in units.tolist()]
    per_step = dt / args.steps

    if rank == 0:
        stage_ms = {k: sum(v) / len(v) for k, v in stage_acc.items()}
        # dominant kernel = the longest stage that has an algorithmic byte count
        local_kmers = st["n_kmers"]
        cand = {k: v for k, v in stage_ms.items() if algorithmic_bytes(k, n_bytes, local_kmers) is not None}
        dom = max(cand, key=cand.get) if cand else None
        roofline = None
        if dom:
            ab = algorithmic_bytes(dom, n_bytes, local_kmers)
            achieved = ab / (cand[dom] * 1e-3) / 1e9
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
            if os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get(dom, {}).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                        "algorithmic_bytes_per_launch": ab, "avg_launch_ms": round(cand[dom], 4)}
            # the whole step against the same roof: algorithmic bytes of this two-level design (bases read as
            # ASCII + 2-bit, every key written and read once per level, solid rows written) over the wall time of a step
            W = 8 if args.kmer_size <= 32 else 16 if args.kmer_size <= 64 else 32
            step_bytes = n_bytes * 1.375 + local_kmers * (4 * W) + st["n_solid"] * (W + 4)
            roofline["step_algorithmic_bytes"] = int(step_bytes)
            roofline["step_frac"] = round(step_bytes / per_step / 1e9 / HBM_PEAK_GBS, 4)
        out = {
            "metric": "distinct k-mers counted/sec (whole node), k=%d" % args.kmer_size,
            "value": n_distinct / per_step,
            "unit": "distinct k-mers/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": per_step * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {nr} reads x {rl} bp per GPU, k={args.kmer_size}, "
                                   f"abundance-min={args.abundance_min}, genome {gl * world} bp, HBM-resident",
                       "reads_per_gpu": nr, "read_len": rl, "kmer_size": args.kmer_size,
                       "parallelism": f"kmer-space sharded over {world} GPU(s)" + (", RCCL all-to-all" if world > 1 else "")},
            "kmer_occurrences_per_s": n_kmers / per_step,
            "bases_per_s": tot_bytes / per_step,
            "n_distinct": n_distinct, "n_kmers": n_kmers, "n_solid": n_solid,
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(reads, rl, args.cpu_sample_reads, args.kmer_size)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None   # rank 0 at N=1 only
        print(json.dumps(out))
    kc.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
