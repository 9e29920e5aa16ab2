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
import datetime
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

COLLECTIVE_TIMEOUT_S = 180.0   # longest any rank waits inside one collective of the N > 1 bench before it fails loudly
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0   # MI355X_MICROARCH.md: measured float4 copy (79 % of spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None,
                    help="default: c2_10Mx150 (BASELINE.json configs[1]) at --gpus 1; c3_shard_25Mx150 per GPU at --gpus N > 1 "
                         "(N = 8 is exactly configs[2]: 200 M x 150 bp on a 600 Mbp genome; --kmer-size 63 makes it configs[3])")
    ap.add_argument("--kmer-size", type=int, default=31)
    ap.add_argument("--abundance-min", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-reads", type=int, default=1_600_000)
    ap.add_argument("--no-sort", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the file -> .h5 wall-clock block (dsk binary)")
    ap.add_argument("--no-repeat-rich", action="store_true", help="skip the repeat-rich twin of the workload (extra block, headline unchanged)")
    ap.add_argument("--no-k63", action="store_true", help="skip the k = 63 count of the same reads (two-word keys, configs[3]'s key width: extra block, headline unchanged)")
    ap.add_argument("--no-human-standin", action="store_true", help="skip the configs[4] stand-in block (600 M x 150 bp of a repeat-rich 3 Gbp genome on this one GPU, ~20 s)")
    ap.add_argument("--no-place-compare", action="store_true", help="skip the plain-hipMalloc leg that yields ms_per_step_no_place (profiling runs: one set of launches per kernel)")
    ap.add_argument("--check-parity", action="store_true", help="N > 1: before the timed steps, count the first 1/64 of every rank's shard (a) sharded over the N ranks and "
                                                                "(b) on rank 0 alone, and require identical totals and histograms (exit code 3 otherwise)")
    ap.add_argument("--no-self-check", action="store_true", help="N > 1: skip the invariants every rank checks on its own result after the timed steps")
    ap.add_argument("--row-order", choices=("partition", "global"), default="partition",
                    help="order of the solid rows (of every rank) the timed steps produce.  partition (default) = the reference's Partition<Count> contract: "
                         "ascending inside every output partition (DSKGPU_F_PARTITION_ORDER, one pass over the rows); global = ascending over all rows "
                         "(three passes).  The other one is timed beside it (ms_per_step_global_order / _partition_order)")
    ap.add_argument("--no-place", action="store_true", help="plain hipMalloc for the big device buffers instead of the best-placed of 8 candidates (DSKGPU_F_PLACE)")
    return ap.parse_args()


def algorithmic_bytes(stage, n_bytes, n_kmers, W=8, n_solid=None, sent_words=None, recv_words=None):
    """Algorithmic HBM bytes of one launch (DESIGN.md 'roofline'; SURVEY.md §8d terms).
    partition kernels: read the bases that produce the k-mers + write W per k-mer;
    key-array passes: read W (+ write W); count: read W per k-mer (table lives in LDS);
    compact / sort (all their launches together): the solid rows (W + 4 bytes) read once and written once.
    N > 1 (sent_words / recv_words = 8-byte words of super-k-mer records this rank sends / receives per step): the sender reads the
    2-bit stream and writes its records; the receiver's level 1 reads the received records and writes W per k-mer it owns."""
    if stage in ("compact", "sort"):
        return None if n_solid is None else 2 * n_solid * (W + 4)
    enc_words = (n_bytes + 31) // 32
    packed = enc_words * 12                      # 8 B packed + 4 B invalid mask per 32 bases
    multi = sent_words is not None
    return {
        "encode": n_bytes + packed,
        "hist1": packed,
        "scatter1": (recv_words * 8 if multi else packed) + n_kmers * W,
        "hist2": n_kmers * W,
        "scatter2": 2 * n_kmers * W,
        "count": n_kmers * W,
        "mg_hist": packed,
        "mg_scatter": packed + (sent_words * 8 if multi else n_kmers * W),
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
    ncpu = os.cpu_count() or 1
    probe_reads = min(n_sample_reads, total_reads)
    sample = reads_u8[: probe_reads * (read_len + 1)].cpu().numpy()
    # the restatement does not scale to every core of a 256-thread host (its scatter and sort are memory-bound): the thread count
    # with the best rate on the probe is the one used and reported
    cores, dt = ncpu, None
    # (probe sample of 1.6 M reads, best of two runs per thread count: on 0.4 M reads the choice was noise -- r05: 256 threads "won" the
    #  probe and then ran the real sample at half the rate of 32)
    for t in sorted({ncpu, min(ncpu, 128), min(ncpu, 64), min(ncpu, 32)}, reverse=True):
        d = None
        for _ in range(2):
            t0 = time.perf_counter()
            oracle.count_only(sample, k, threads=t)
            d1 = time.perf_counter() - t0
            d = d1 if d is None or d1 < d else d
        if dt is None or d < dt:
            cores, dt = t, d

    class _R:      # (total, distinct) of a count_only call
        def __init__(self, td): self.total, self.distinct = td
    n_reads = probe_reads
    # grow the sample towards target_s (bounded by the workload and by 16 GiB of k-mer keys)
    want = int(min(total_reads, probe_reads * max(1.0, target_s / max(dt, 1e-3)), 16 * 2**30 // (8 * (read_len - k + 1))))
    if want > probe_reads * 2:
        n_reads = want
        sample = reads_u8[: n_reads * (read_len + 1)].cpu().numpy()
    t0 = time.perf_counter()
    r = _R(oracle.count_only(sample, k, threads=cores))
    dt = time.perf_counter() - t0
    # the same code on ONE thread, on a sample sized for a few seconds (DSK v1's published rates are 1-thread)
    n1 = max(1000, min(n_reads, int(n_reads * 6.0 / max(dt, 1e-3) / max(cores, 1)) * 4))
    s1 = reads_u8[: n1 * (read_len + 1)].cpu().numpy()
    t0 = time.perf_counter()
    r1 = _R(oracle.count_only(s1, k, threads=1))
    dt1 = time.perf_counter() - t0
    return {
        "value": r.distinct / dt,
        "unit": "distinct k-mers/s",
        "kmer_occurrences_per_s": r.total / dt,
        "cores": cores,
        "one_thread": {"value": r1.distinct / dt1, "kmer_occurrences_per_s": r1.total / dt1, "cores": 1,
                       "sample": f"first {n1} reads, {dt1:.2f} s"},
        "kind": "port",
        "sample": f"first {n_reads} reads of the same synthetic stream ({r.total} k-mer occurrences, "
                  f"{r.distinct} distinct), oracle/dsk_oracle.c partition+sort-count, {cores} threads, {dt:.2f} s",
        "note": "CPU restatement of DSK's method, NOT GATB/dsk (gatb-core submodule absent => reference unbuildable)",
    }


def write_fastq(reads_u8, n_reads, read_len, path):
    """The reads as a 4-line FASTQ file (id @r<idx>, quality I x read_len), built as one byte
    matrix on the GPU (fixed-width ids) and written in a few large pieces."""
    import torch
    dev = reads_u8.device
    hdr = 10                                            # "@r" + 8 digits
    rec = hdr + 1 + read_len + 1 + 2 + read_len + 1
    step = 1 << 20
    with open(path, "wb") as f:
        for r0 in range(0, n_reads, step):
            r = min(step, n_reads - r0)
            m = torch.empty((r, rec), dtype=torch.uint8, device=dev)
            m[:, 0] = 64; m[:, 1] = 114                  # '@' 'r'
            idx = torch.arange(r0, r0 + r, device=dev, dtype=torch.int64)
            for d in range(8):
                m[:, 2 + d] = ((idx // 10 ** (7 - d)) % 10 + 48).to(torch.uint8)
            m[:, hdr] = 10
            m[:, hdr + 1: hdr + 1 + read_len] = reads_u8[r0 * (read_len + 1): (r0 + r) * (read_len + 1)].view(r, read_len + 1)[:, :read_len]
            m[:, hdr + 1 + read_len] = 10
            m[:, hdr + 2 + read_len] = 43                # '+'
            m[:, hdr + 3 + read_len] = 10
            m[:, hdr + 4 + read_len: hdr + 4 + 2 * read_len] = 73     # 'I'
            m[:, rec - 1] = 10
            f.write(m.cpu().numpy().tobytes())
    return os.path.getsize(path)


def e2e_block(k, amin, budget_s=150.0):
    """File -> .h5 wall clock of the `dsk` binary (HIP engine through the C-ABI): best of 3, page cache warm, on the
    E. coli-like 50x stand-in (plain FASTQ, gzip, BGZF) and on the bench workload itself (plain FASTQ); next to it the
    CPU restatement's CLI on the same E. coli file (all cores, and one thread).  north_star: >= 10x reference-DSK wall
    clock on 50x E. coli -- the reference cannot be built here, so the comparison shown is against the restatement."""
    import gzip
    import shutil
    import struct
    import subprocess
    import tempfile
    import zlib
    import torch
    from dsk_amd import synth
    dsk = os.path.join(ROOT, "dsk_amd", "host", "bin", "dsk")
    cli = os.path.join(ROOT, "oracle", "dsk_oracle_cli")
    if not os.path.exists(dsk):
        return {"error": "dsk binary not built (python -c 'import __graft_entry__ as g; g.build()')"}
    t_start = time.perf_counter()
    tmp = tempfile.mkdtemp(prefix="dsk_e2e_")
    dev = torch.device("cuda", 0)
    out = {"note": "wall clock of the dsk binary, file -> .h5 (HIP runtime start-up, ingest, count, HDF5 write), best of 3 runs 0.4 s apart, page cache warm"}

    def run_dsk(path, extra=()):
        best, info = None, {}
        for _ in range(3):
            for stale in (os.path.join(tmp, "o.h5"),):          # (a fresh output every time: truncating the previous run's 0.7 GB file is not part of a run)
                if os.path.exists(stale):
                    os.remove(stale)
            time.sleep(0.4)          # (the driver takes the previous process's device context apart for up to 0.1 s after its exit, and the next start-up waits for it: runs are timed apart, as single invocations)
            t0 = time.perf_counter()
            p = subprocess.run([dsk, "-file", path, "-kmer-size", str(k), "-abundance-min", str(amin), "-out", os.path.join(tmp, "o"), "-verbose", "1", *extra],
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            dt = time.perf_counter() - t0
            if p.returncode != 0:
                return {"error": p.stdout.decode(errors="replace")[-300:]}
            if best is None or dt < best:
                best = dt
                for ln in p.stdout.decode(errors="replace").splitlines():
                    for key in ("ingest_s", "count_s", "write_s", "total_s", "kmers_nb_valid", "kmers_nb_distinct", "kmers_nb_solid", "banks_parsed_on_device"):
                        if key in ln:
                            info[key] = float(ln.split(":")[-1]) if key.endswith("_s") else int(ln.split(":")[-1])
            if time.perf_counter() - t_start > budget_s:
                break
        info["wall_s"] = round(best, 3)
        if "kmers_nb_distinct" in info:
            info["distinct_kmers_per_s"] = info["kmers_nb_distinct"] / best
            info["kmer_occurrences_per_s"] = info["kmers_nb_valid"] / best
        return info

    try:
        for name in ("ecoli50x", "c2_10Mx150"):
            if time.perf_counter() - t_start > budget_s * 0.6 and name != "ecoli50x":
                out[name] = {"skipped": "time budget"}
                continue
            gl, nr, rl = synth.workload(name)
            reads = synth.make_reads(synth.make_genome(gl, dev), nr, rl)
            fq = os.path.join(tmp, name + ".fastq")
            size = write_fastq(reads, nr, rl, fq)
            del reads
            torch.cuda.empty_cache()
            blk = {"file_bytes": size, "plain": run_dsk(fq)}
            blk["plain_device_parse"] = run_dsk(fq, ("-device-parse", "1"))       # the text goes to the GPU as it is (dskgpu_push_raw): no host parser
            if name == "ecoli50x":
                data = open(fq, "rb").read()
                with gzip.open(fq + ".gz", "wb", compresslevel=1) as f:
                    f.write(data)
                with open(fq + ".bgzf.gz", "wb") as f:          # blocked gzip (bgzip layout): independent <= 64 KB members
                    for off in list(range(0, len(data), 60000)) + [None]:
                        chunk = b"" if off is None else data[off: off + 60000]
                        c = zlib.compressobj(1, zlib.DEFLATED, -15)
                        body = c.compress(chunk) + c.flush()
                        f.write(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1))
                        f.write(body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
                del data
                blk["gzip"] = run_dsk(fq + ".gz")
                blk["gzip_device_parse"] = run_dsk(fq + ".gz", ("-device-parse", "1"))
                blk["bgzf"] = run_dsk(fq + ".bgzf.gz")
                if os.path.exists(cli):                       # the CPU restatement's own CLI on the same file
                    cpu = {}
                    # ("best_threads": up to 32 threads -- the restatement peaks there on the 2 x 64-core host, oracle/dsk_oracle.c; more are slower)
                    for label, threads in (("best_threads", min(os.cpu_count() or 1, 32)), ("one_thread", 1)):
                        if label == "one_thread" and time.perf_counter() - t_start > budget_s * 0.5:
                            cpu[label] = {"skipped": "time budget"}
                            continue
                        t0 = time.perf_counter()
                        p = subprocess.run([cli, "-file", fq, "-kmer-size", str(k), "-abundance-min", str(amin), "-nb-cores", str(threads)],
                                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
                        cpu[label] = {"wall_s": round(time.perf_counter() - t0, 3), "cores": threads, "rc": p.returncode}
                    blk["cpu_restatement_cli"] = cpu
                    if "wall_s" in blk["plain"] and cpu.get("best_threads", {}).get("wall_s"):
                        blk["speedup_vs_cpu_restatement_best_threads"] = round(cpu["best_threads"]["wall_s"] / blk["plain"]["wall_s"], 2)
                    if "wall_s" in blk["plain"] and cpu.get("one_thread", {}).get("wall_s"):
                        blk["speedup_vs_cpu_restatement_one_thread"] = round(cpu["one_thread"]["wall_s"] / blk["plain"]["wall_s"], 2)
            out[name] = blk
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out["reference_context"] = "DSK v1 published: E. coli k=21, one 2012 core, 58.8 s (doc/figure-1/ecoli_log:12); GATB/dsk itself cannot be built here"
    return out


def human_standin_block(k, amin, dev):
    """BASELINE.json configs[4] (30x human short reads, ~90 Gbp, multi-pass) as the stand-in SURVEY.md section 8(d) allows: 600 M x 150 bp
    reads of a repeat-rich 3 Gbp genome (dsk_amd/synth.py c5_human30x: one high-copy family, tandem arrays, 0.2 % poly-A reads), all
    of it on this ONE GPU -- the 8-GPU topology is the driver's to run.  In a process of its own (tools/human_standin.py: 265 GB of
    HBM, plain hipMalloc buffers): the reads encoded once and their bytes released (dskgpu_encode_reads), one count that allocates every
    buffer, one timed, the size-independent invariants checked on the device."""
    import subprocess
    import torch
    free_b, total_b = torch.cuda.mem_get_info()
    if free_b < 275e9:
        return {"skipped": f"needs ~265 GB of free HBM, {free_b / 1e9:.0f} GB are free"}
    env = {x: y for x, y in os.environ.items() if not x.startswith("DSKGPU_")}
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "human_standin.py"), "600", str(k), "1", str(amin)], cwd=ROOT, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
    except subprocess.TimeoutExpired:
        return {"error": "tools/human_standin.py did not finish within 420 s"}
    line = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not line:
        return {"error": p.stderr.decode(errors="replace")[-400:]}
    r = json.loads(line[-1])
    return {"workload": r["workload"] + f", k={k}, abundance-min={amin}, ONE GPU, HBM-resident",
            "reads": 600_000_000, "bases": 90_000_000_000, "count_s": r["count_s"], "first_count_s": r["first_count_s"], "generate_s": r["generate_s"],
            "kmer_occurrences_per_s": r["kmer_occurrences_per_s"], "distinct_kmers_per_s": r["distinct_kmers_per_s"],
            "passes_over_the_key_space": r["n_passes"], "sweeps_over_the_reads": r["n_read_sweeps"],
            **{x: r[x] for x in ("n_kmers", "n_distinct", "n_solid", "n_retries", "sort_fallback", "n_ext_regions", "n_heavy")},
            "stage_ms": r["stage_ms"], "hbm_used_gb": r["hbm_used_gb"], "invariants": r.get("invariants"),
            "ascii_reads_resident_during_count": r.get("ascii_reads_resident_during_count"), "encode_reads_s": r.get("encode_reads_s"),
            "input_handling": "the reads are encoded once (dskgpu_encode_reads: 2 bits per base + the invalid-base mask, 34 GB) and their 90 GB of bytes given back before the count",
            "reference_context": "the reference's own human run: 7 passes over the input, 2.7e9 solid k-mers (doc/human_log:3-4,20-24); README.md:126-130 asks for 'below 10' passes"}


def valid_windows(reads_u8, n_reads, read_len, k):
    """Number of full ACGT windows of a synthetic read stream (<= 1 'N' per read, dsk_amd/synth.py), closed form on the device."""
    import torch
    r = reads_u8.view(n_reads, read_len + 1)[:, :read_len]
    total = 0
    for r0 in range(0, n_reads, 8_000_000):
        bad = r[r0:r0 + 8_000_000] == 78
        has = bad.any(1)
        q = bad.to(torch.uint8).argmax(1).to(torch.int64)
        with_n = torch.clamp(q - k + 1, min=0) + torch.clamp(read_len - q - k, min=0)
        total += int(torch.where(has, with_n, torch.full_like(with_n, read_len - k + 1)).sum())
    return total


def rows_strictly_ascending(kc, dev, words):
    """This rank's sorted result, checked where it lies (HBM): every row's value above its predecessor's -- inside every output
    partition when the rows are in partition order (the partitions' offsets come from the engine; their sizes must add up)
    -> (ok, rows, sum of abundances)."""
    import ctypes
    import numpy as np
    import torch
    kp, ap, n = kc.result_device()
    if n == 0:
        return True, 0, 0
    starts = None
    if kc.num_partitions() > 64:            # partition order (DSKGPU_F_PARTITION_ORDER): a row may be smaller than its predecessor only where a partition starts
        off = kc.partition_offsets().astype(np.int64)
        if int(off[-1]) != n or bool((np.diff(off) < 0).any()):
            return False, int(n), 0
        starts = torch.from_numpy(off[1:-1].copy()).to(dev)
    hip = ctypes.CDLL("libamdhip64.so")
    step = 1 << 26
    kb = torch.empty(step, dtype=torch.int64, device=dev); ab = torch.empty(step, dtype=torch.int32, device=dev)
    ok, last, ab_sum = True, None, 0
    for r0 in range(0, n, step):
        m = min(step, n - r0)
        hip.hipMemcpy(ctypes.c_void_p(ab.data_ptr()), ctypes.c_void_p(ap + r0 * 4), ctypes.c_size_t(m * 4), 3)
        ab_sum += int(ab[:m].to(torch.int64).sum())
        if words == 1:          # (k <= 31: values < 2^62, a signed compare is safe; wider keys: the abundances only -- word 0 alone does not order them)
            hip.hipMemcpy(ctypes.c_void_p(kb.data_ptr()), ctypes.c_void_p(kp + r0 * 8), ctypes.c_size_t(m * 8), 3)
            kk = kb[:m]
            asc = kk[1:] > kk[:-1]
            if starts is not None:
                if m > 1:
                    asc[starts[(starts > r0) & (starts < r0 + m)] - r0 - 1] = True
                ok = ok and bool(asc.all()) and (last is None or int(kk[0]) > last or bool((starts == r0).any()))
            else:
                ok = ok and bool(asc.all()) and (last is None or int(kk[0]) > last)
            last = int(kk[-1])
    return ok, int(n), ab_sum


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
    # watchdog: a rank that dies or hangs inside a collective must not turn into the driver's 30-minute timeout with no message --
    # the children run in their own process group, and past DSK_BENCH_TIMEOUT_S (default 1200 s) all of them are killed
    limit = float(os.environ.get("DSK_BENCH_TIMEOUT_S", "1200"))
    import signal
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, start_new_session=True)
    try:
        stdout, _ = proc.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)
        stdout, _ = proc.communicate()
        print(f"bench.py: the {args.gpus} ranks did not finish within {limit:.0f} s (a rank hung in a collective?): killed", file=sys.stderr)
        sys.stderr.write(stdout.decode(errors="replace")[-4000:])
        sys.exit(124)
    line = None
    for ln in stdout.decode(errors="replace").splitlines():
        if ln.startswith('{"metric"'):
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line:
        print(line)
    sys.exit(proc.returncode if proc.returncode else (0 if line else 1))


def main():
    args = parse()
    if args.workload is None:
        args.workload = "c2_10Mx150" if args.gpus <= 1 else "c3_shard_25Mx150"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (what RCCL between processes needs on this pool): before the HIP runtime starts
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
    # DSK_BENCH_SHARE_GPU=1 (development, never a measurement): all ranks on device 0, the exchange over gloo staged through the
    # host -- runs the N-rank code path of this file on a 1-GPU box
    share_gpu = world > 1 and os.environ.get("DSK_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rdev = torch.device("cpu") if share_gpu else dev          # where the few reduction scalars live
    if world > 1:
        if share_gpu:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=COLLECTIVE_TIMEOUT_S))
        else:
            # (RCCL: the process group's watchdog aborts the communicator and ends the rank when a collective exceeds the timeout)
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=COLLECTIVE_TIMEOUT_S))

    # weak scaling: same per-GPU shard size, genome grows with the node so coverage stays 50x
    reads, gl, nr, rl = synth.make_workload(args.workload, dev, world, rank)
    torch.cuda.synchronize()
    n_bytes = reads.numel()

    stream = torch.cuda.current_stream().cuda_stream
    # the same steps on plain hipMalloc buffers first (3 warm-up + 10 timed, same process, before any context asks for placement --
    # the flag is process-wide): reported next to the headline as ms_per_step_no_place
    no_place_ms = None
    if world == 1 and not args.no_place and not args.no_place_compare:
        with KmerCounter(kmer_size=args.kmer_size, abundance_min=args.abundance_min, device=local_rank, sort=not args.no_sort, stream=stream,
                         partition_order=args.row_order == "partition") as kp:
            kp.set_reads_device(reads.data_ptr(), n_bytes)
            for _ in range(3):
                kp.count()
            torch.cuda.synchronize()
            tp = time.perf_counter()
            for _ in range(10):
                kp.count()
            torch.cuda.synchronize()
            no_place_ms = (time.perf_counter() - tp) / 10 * 1e3
    kc = KmerCounter(kmer_size=args.kmer_size, abundance_min=args.abundance_min, device=local_rank, timing=True,
                     sort=not args.no_sort, world_size=world, rank=rank, stream=stream, place=not args.no_place and not share_gpu,      # (ranks sharing one GPU: no room for placement candidates)
                     partition_order=args.row_order == "partition")      # (every rank of a multi-GPU job orders its own share the same way)
    kc.set_reads_device(reads.data_ptr(), n_bytes)

    sharded = None
    if world > 1:
        from dsk_amd.multi import ShardedCounter
        sharded = ShardedCounter(kc, dev, wait_timeout_s=COLLECTIVE_TIMEOUT_S)

    # ---- N > 1, --check-parity: the sharded count of a 1/64 sample against ONE context counting the same reads (rank 0) -- identical
    # totals and histograms, or the run fails before anything is timed
    parity = None
    if world > 1 and args.check_parity:
        ns = max(1000, nr // 64)
        smp = reads[: ns * (rl + 1)]
        kc.set_reads_device(smp.data_ptr(), smp.numel())
        sharded.count()
        ps = kc.stats()
        ph = torch.from_numpy(kc.histogram().astype("int64")).to(rdev)
        pt = torch.tensor([ps["n_kmers"], ps["n_distinct"], ps["n_solid"]], dtype=torch.int64, device=rdev)
        dist.all_reduce(ph, op=dist.ReduceOp.SUM); dist.all_reduce(pt, op=dist.ReduceOp.SUM)
        gathered = [torch.empty_like(smp) for _ in range(world)] if rank == 0 else None
        if share_gpu:
            host = [torch.empty(smp.numel(), dtype=torch.uint8) for _ in range(world)] if rank == 0 else None
            dist.gather(smp.cpu(), host, dst=0)
            if rank == 0:
                gathered = [h.to(dev) for h in host]
        else:
            dist.gather(smp, gathered, dst=0)
        flag = torch.zeros(1, dtype=torch.int64, device=rdev)
        if rank == 0:
            allr = torch.cat(gathered)
            torch.cuda.synchronize()
            with KmerCounter(kmer_size=args.kmer_size, abundance_min=args.abundance_min, device=local_rank) as k1:
                k1.set_reads_device(allr.data_ptr(), allr.numel())
                k1.count()
                s1, h1 = k1.stats(), k1.histogram().astype("int64")
            same = ([int(x) for x in pt.tolist()] == [s1["n_kmers"], s1["n_distinct"], s1["n_solid"]]) and bool((ph.cpu().numpy() == h1).all())
            parity = {"sample_reads_per_rank": ns, "sharded": [int(x) for x in pt.tolist()], "one_gpu": [s1["n_kmers"], s1["n_distinct"], s1["n_solid"]],
                      "histograms_equal": bool((ph.cpu().numpy() == h1).all()), "ok": bool(same)}
            flag[0] = 0 if same else 1
            del allr, gathered
        dist.broadcast(flag, src=0)
        if int(flag.item()):
            if rank == 0:
                print("bench.py --check-parity FAILED: " + json.dumps(parity), file=sys.stderr)
            dist.destroy_process_group()
            sys.exit(3)
        kc.set_reads_device(reads.data_ptr(), n_bytes)
        torch.cuda.empty_cache()

    stage_acc = {}
    sliced_steps = [0]

    def step():
        if world == 1:
            kc.count()
        else:
            sharded.count()      # records -> RCCL all-to-all -> count; in slices, the exchange beside the sender and the receiver's level 1
            sliced_steps[0] += int(sharded.last_step_sliced)
        for name, ms in kc.stage_times():
            stage_acc.setdefault(name, []).append(ms)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # (the first step also allocates the device buffers -- with DSKGPU_F_PLACE each as the best-placed of 8 candidates: a one-off
    #  cost of a few seconds, reported below, outside the timed region like every first-call allocation)
    torch.cuda.synchronize()
    t_first = time.perf_counter()
    for i in range(max(args.warmup, 1)):
        step()
        if i == 0:
            torch.cuda.synchronize()
            first_step_s = time.perf_counter() - t_first
    stage_acc.clear()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    st = kc.stats()
    # N = 1: the same steps with the OTHER row order (same context, same buffers), outside the timed region
    other_order_ms = None
    if world == 1 and not args.no_sort and not args.no_place_compare:
        kc.set_row_order(args.row_order != "partition")
        for _ in range(2):
            kc.count()
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for _ in range(10):
            kc.count()
        torch.cuda.synchronize()
        other_order_ms = (time.perf_counter() - tp) / 10 * 1e3
        other_stage = dict(kc.stage_times())
        kc.set_row_order(args.row_order == "partition")
        kc.count()                      # (the result the checks below look at is the timed configuration's)
        torch.cuda.synchronize()

    # ---- N > 1: every rank checks its own share of the result, the job-wide sums are checked after the reductions below -- a wrong
    # exchange must not print a `value` (exit code 3)
    self_check = None
    if world > 1 and not args.no_self_check:
        h = kc.histogram().astype("int64")
        asc, rows_n, ab_sum = rows_strictly_ascending(kc, dev, kc.words)
        amin = args.abundance_min
        local_ok = asc and rows_n == st["n_solid"] and int(h.sum()) == st["n_distinct"] and int(h[amin:].sum()) == st["n_solid"]
        if int(h[-1]) == 0:          # nothing saturates the last histogram row: sum(i * hist[i]) is this rank's k-mer count
            local_ok = local_ok and int((h * torch.arange(len(h)).numpy()).sum()) == st["n_kmers"]
        local_ok = local_ok and ab_sum + int((h[:amin] * torch.arange(amin).numpy()).sum()) == st["n_kmers"]
        chk = torch.tensor([0 if local_ok else 1, valid_windows(reads, nr, rl, args.kmer_size), st["n_kmers"]], dtype=torch.int64, device=rdev)
        dist.all_reduce(chk, op=dist.ReduceOp.SUM)
        bad_ranks, windows, counted = [int(x) for x in chk.tolist()]
        self_check = {"ranks_with_a_failed_local_check": bad_ranks, "valid_windows_of_all_shards": windows, "kmers_counted_by_all_ranks": counted,
                      "local_checks": "rows strictly ascending inside every output partition (one-word keys; the partitions' sizes add up to the rows), rows == n_solid == sum(hist[amin:]), sum(hist) == n_distinct, sum(i*hist) == sum of abundances (+ below amin) == n_kmers",
                      "ok": bad_ranks == 0 and windows == counted}
        if not self_check["ok"]:
            if rank == 0:
                print("bench.py self-check FAILED: " + json.dumps(self_check), file=sys.stderr)
            dist.destroy_process_group()
            sys.exit(3)
    # ---- N > 1: the exchange alone, once, outside the timed region: one step in one piece with the all-to-all bracketed by
    # synchronisation (inside a sliced step it overlaps the sender and the receiver's level 1 and cannot be timed by itself)
    exchange_alone = None
    if world > 1:
        from dsk_amd.multi import exchange, scatter_records
        with torch.cuda.stream(sharded.stream) if sharded.stream is not None else torch.cuda.stream(torch.cuda.current_stream()):
            sharded.send, counts = scatter_records(kc, sharded.send, dev)
            torch.cuda.synchronize(); dist.barrier()
            te = time.perf_counter()
            outb, rcounts = exchange(sharded.send, counts, None, sharded.recv)
            torch.cuda.synchronize()
            ex_s = time.perf_counter() - te
        et = torch.tensor([ex_s], dtype=torch.float64, device=rdev)
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
        off_rank = sum(c for p_, c in enumerate(counts) if p_ != rank) * 8
        exchange_alone = {"ms": round(float(et.item()) * 1e3, 3), "bytes_leaving_this_rank": int(off_rank), "bytes_sent_incl_own_share": int(sum(counts)) * 8,
                          "gb_per_s_per_rank_off_gpu": round(off_rank / max(float(et.item()), 1e-9) / 1e9, 2),
                          "note": "one all-to-all-v of a whole step's records (counts round + payload), max over ranks; xGMI egress when every rank has its own GPU"}
        del outb

    # whole-job aggregates (MAX time over ranks, SUM of units)
    tt = torch.tensor([dt], dtype=torch.float64, device=rdev)
    units = torch.tensor([st["n_distinct"], st["n_kmers"], st["n_solid"], n_bytes], dtype=torch.float64, device=rdev)
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
        sent_w = sum(sharded.last_send_counts) if world > 1 else None
        recv_w = sum(sharded.last_recv_counts) if world > 1 else None
        cand = {k: v for k, v in stage_ms.items() if algorithmic_bytes(k, n_bytes, local_kmers, sent_words=sent_w, recv_words=recv_w) is not None}
        dom = max(cand, key=cand.get) if cand else None
        roofline = None
        if dom:
            W = 8 if args.kmer_size <= 32 else 16 if args.kmer_size <= 64 else 32

            def price(stage):
                ab = algorithmic_bytes(stage, n_bytes, local_kmers, W, st["n_solid"], sent_w, recv_w)
                gbs = ab / (stage_ms[stage] * 1e-3) / 1e9
                return {"algorithmic_bytes_per_launch": ab, "avg_launch_ms": round(stage_ms[stage], 4), "achieved": round(gbs, 1),
                        "frac": round(gbs / HBM_PEAK_GBS, 4), "frac_vs_measured_copy": round(gbs / HBM_COPY_GBS, 4)}
            d = price(dom)
            roofline = {"bound": "hbm", "kernel": dom, "achieved": d["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["frac"],
                        "traffic": None,      # PMC counters are not collected inside a timed run (and .git does not travel to the GPU box, so a committed
                                              # profile cannot be matched to this tree's commit): always null; what an EARLIER profiled run of this command
                                              # measured is quoted beside it as traffic_from_profiles, with the commit it was taken at
                        "peak_measured_copy": HBM_COPY_GBS, "frac_vs_measured_copy": d["frac_vs_measured_copy"],
                        "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"], "avg_launch_ms": d["avg_launch_ms"],
                        # every partition + hash kernel of the step, priced the same way (HIP events on the launching stream)
                        "kernels": {st: price(st) for st in ("encode", "scatter1", "scatter2", "count", "compact", "sort", "mg_scatter") if st in stage_ms}}
            pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
            if os.path.exists(pmc):        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an EARLIER run of this command (profiles/)
                try:
                    prof = json.load(open(pmc))
                    if prof.get("_meta", {}).get("workload") == args.workload and prof.get("_meta", {}).get("kmer_size") == args.kmer_size and world == 1:
                        roofline["traffic_from_profiles"] = {"profile": prof["_meta"].get("id"), "commit": prof["_meta"].get("commit"),
                                                             "hbm_bytes_per_launch": {st: prof[st]["hbm_bytes_per_launch"] for st in roofline["kernels"] if st in prof}}
                        if dom in prof:      # the dominant kernel's HBM bytes per launch, from those passes (an earlier run: not collected live)
                            roofline["traffic_from_profiles"]["dominant_kernel_hbm_bytes_per_launch"] = prof[dom]["hbm_bytes_per_launch"]
                            roofline["traffic_from_profiles"]["source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of this command, %s (commit %s)" % (prof["_meta"].get("id"), prof["_meta"].get("commit"))
                except Exception:
                    pass
            # the whole step against the same roof, by the contract's figure: per k-mer occurrence the bases read as
            # ASCII (L/n bytes) + W written by the partition kernels + W read, a (W + 4)-byte slot touched and 4 bytes written by the hash
            # kernel = 33.25 B at k = 31 / 57.7 B at k = 63 on 150 bp reads; the compaction term (distinct (W + 4) / 0.7 read, solid
            # (W + 4) written) is reported beside it.  step_design_bytes = what THIS two-level design moves at least (bases as ASCII + 2-bit,
            # every key written and read once per level, solid rows written) -- until r05 that figure was printed as step_algorithmic_bytes.
            step_bytes = n_bytes + local_kmers * (3 * W + 8)
            k6_bytes = st["n_distinct"] * (W + 4) / 0.7 + st["n_solid"] * (W + 4)
            roofline["step_algorithmic_bytes"] = int(step_bytes)
            roofline["step_frac"] = round(step_bytes / per_step / 1e9 / HBM_PEAK_GBS, 4)
            roofline["step_frac_vs_measured_copy"] = round(step_bytes / per_step / 1e9 / HBM_COPY_GBS, 4)
            roofline["step_algorithmic_bytes_with_compaction_term"] = int(step_bytes + k6_bytes)
            roofline["step_design_bytes"] = int(n_bytes * 1.375 + local_kmers * (4 * W) + st["n_solid"] * (W + 4))
            ph = [x for x in ("scatter1", "scatter2", "count") if x in stage_ms]
            if len(ph) == 3 and world == 1:      # the north_star's "partition + hash kernels" figure: their algorithmic bytes over their summed launch times
                pb = sum(algorithmic_bytes(x, n_bytes, local_kmers, W, st["n_solid"], sent_w, recv_w) for x in ph)
                pm = sum(stage_ms[x] for x in ph)
                roofline["partition_plus_hash"] = {"algorithmic_bytes": int(pb), "ms": round(pm, 4), "frac": round(pb / (pm * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        out = {
            "metric": "distinct k-mers counted/sec (whole node), k=%d" % args.kmer_size,
            "value": n_distinct / per_step,
            "unit": "distinct k-mers/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": max(args.warmup, 1),
            "ms_per_step": per_step * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic" + (" -- DEVELOPMENT RUN: all ranks share one GPU, exchange over gloo (not a measurement)" if share_gpu else ""),
            "config": {"workload": f"{args.workload}: {nr} reads x {rl} bp per GPU, k={args.kmer_size}, "
                                   f"abundance-min={args.abundance_min}, genome {gl * world} bp, HBM-resident",
                       "reads_per_gpu": nr, "read_len": rl, "kmer_size": args.kmer_size,
                       "parallelism": f"kmer-space sharded over {world} GPU(s)" + (
                           f", RCCL all-to-all in {sharded.slices} slices overlapped with the sender and the receiver's level 1"
                           f" ({sliced_steps[0]} of {args.steps + max(args.warmup, 1)} steps ran in slices)" if world > 1 else "")},
            "kmer_occurrences_per_s": n_kmers / per_step,
            "bases_per_s": tot_bytes / per_step,
            "n_distinct": n_distinct, "n_kmers": n_kmers, "n_solid": n_solid,
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "engine_stats": {k: st[k] for k in ("n_passes", "n_retries", "sort_fallback", "n_ext_regions", "n_heavy", "n_final_bins")},
            "buffer_placement": ({"mode": "plain hipMalloc (--no-place)"} if args.no_place else
                                 {"mode": "DSKGPU_F_PLACE: every device buffer >= 256 MB is the best of up to 8 candidate allocations, timed with "
                                          "the level-1 store pattern (where a buffer lies in HBM moves the scatter kernels by up to 18 %)",
                                  "candidates": 8}),
            "first_step_s": round(first_step_s, 3),
            "ms_per_step_no_place": None if no_place_ms is None else round(no_place_ms, 3),
            "row_order": ("partition: ascending inside each of %d output partitions -- the reference's Partition<Count> contract (utils/dsk2ascii.cpp:61,77,85-104), DSKGPU_F_PARTITION_ORDER" % st["n_partitions"])
                         if (args.row_order == "partition" and not args.no_sort) else "global: ascending over all rows",
            ("ms_per_step_global_order" if args.row_order == "partition" else "ms_per_step_partition_order"): None if other_order_ms is None else round(other_order_ms, 3),
            "sort_ms_other_row_order": None if other_order_ms is None else round(other_stage.get("sort", 0.0), 3),
            "roofline": roofline,
        }
        if world > 1:
            # rank 0's view of the exchange (every rank sends and receives about the same: owners are balanced by the repartition table)
            out["exchange_bytes"] = {"sent_per_rank": int(sent_w) * 8, "received_per_rank": int(recv_w) * 8,
                                     "per_kmer_on_the_wire": round(int(sent_w) * 8 / max(1.0, n_kmers / world), 3),
                                     "format": "super-k-mer records (2-3 words per <= 16 k-mers)" if 20 <= args.kmer_size <= 64 else "explicit keys"}
            out["sliced_steps"] = sliced_steps[0]
            out["rccl_ranks"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                                 "devices": "one GPU shared by all ranks (development)" if share_gpu else "one GPU per rank"}
            out["exchange_alone"] = exchange_alone
            out["self_check"] = self_check
            out["check_parity"] = parity
            if roofline:
                roofline["per_rank"] = ("rank 0's kernels: mg_scatter = the sender (2-bit stream -> records), scatter1 = level 1 straight from the received "
                                        "records -- in a sliced step its stage time includes the wait for the slices' arrival")
            which = {31: "configs[2]", 63: "configs[3]"}.get(args.kmer_size)
            if args.workload == "c3_shard_25Mx150" and which:
                out["config"]["baseline_config"] = (f"{which} of BASELINE.json when N = 8 (200 M x 150 bp on a 600 Mbp genome); "
                                                    f"this run: {world} x 25 M reads on a {gl * world} bp genome")
        # the same shape with what real genomes have and the uniform one lacks: a high-copy family, tandem arrays, poly-A reads
        # (dsk_amd/synth.py: make_genome_repeats).  An extra block: the headline above stays the BASELINE.json workload.
        twin = args.workload.replace("_10Mx150", "_repeats_10Mx150")
        if world == 1 and not args.no_repeat_rich and twin in synth.REPEAT_WORKLOADS and twin != args.workload:
            rr, _, _, _ = synth.make_workload(twin, dev)
            torch.cuda.synchronize()
            kc.set_reads_device(rr.data_ptr(), rr.numel())
            for _ in range(args.warmup):
                kc.count()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                kc.count()
            torch.cuda.synchronize()
            rr_ms = (time.perf_counter() - t1) / args.steps * 1e3
            rst = kc.stats()
            out["repeat_rich"] = {"workload": twin + ": 1 % of the genome one 300 bp family (2 % divergence), 4 tandem arrays of 2000 x 37 bp, 0.2 % poly-A reads",
                                  "ms_per_step": round(rr_ms, 3), "vs_uniform": round(rr_ms / (per_step * 1e3), 4),
                                  "n_kmers": rst["n_kmers"], "n_distinct": rst["n_distinct"], "n_solid": rst["n_solid"],
                                  "stage_ms": {k: round(v, 4) for k, v in kc.stage_times()},
                                  **{k: rst[k] for k in ("n_retries", "sort_fallback", "n_ext_regions", "n_heavy")}}
            kc.set_reads_device(reads.data_ptr(), n_bytes)
            del rr
        # the same reads with two-word keys (k = 63: the key width of BASELINE.json's configs[3]).  An extra block, its own context
        # (same placement setting); the headline above stays k = 31.
        if world == 1 and not args.no_k63 and args.kmer_size == 31 and rl >= 63:
            with KmerCounter(kmer_size=63, abundance_min=args.abundance_min, device=local_rank, timing=True, sort=not args.no_sort,
                             stream=stream, place=not args.no_place, partition_order=args.row_order == "partition") as k2:
                k2.set_reads_device(reads.data_ptr(), n_bytes)
                for _ in range(max(2, args.warmup)):
                    k2.count()
                torch.cuda.synchronize()
                acc2 = {}
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    k2.count()
                    for name, ms in k2.stage_times():
                        acc2.setdefault(name, []).append(ms)
                torch.cuda.synchronize()
                ms2 = (time.perf_counter() - t1) / args.steps * 1e3
                st2 = k2.stats()
                stage2 = {name: sum(v) / len(v) for name, v in acc2.items()}
                fr2 = {}
                for name in ("scatter1", "scatter2", "count"):
                    ab = algorithmic_bytes(name, n_bytes, st2["n_kmers"], W=16)
                    if ab and stage2.get(name):
                        fr2[name] = {"algorithmic_bytes_per_launch": ab, "avg_launch_ms": round(stage2[name], 4), "frac": round(ab / (stage2[name] * 1e-3) / 8e12, 4)}
                out["k63"] = {"workload": args.workload + " at k = 63 (two-word keys)", "ms_per_step": round(ms2, 3),
                              "kmer_occurrences_per_s": st2["n_kmers"] / (ms2 * 1e-3), "distinct_kmers_per_s": st2["n_distinct"] / (ms2 * 1e-3),
                              "n_kmers": st2["n_kmers"], "n_distinct": st2["n_distinct"], "n_solid": st2["n_solid"],
                              "stage_ms": {name: round(v, 4) for name, v in stage2.items()}, "roofline_kernels": fr2,
                              **{name: st2[name] for name in ("n_passes", "n_retries", "sort_fallback")}}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(reads, rl, args.cpu_sample_reads, args.kmer_size)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None   # rank 0 at N=1 only
        if world == 1 and not args.no_e2e:
            kc.close()
            del reads
            reads = None
            torch.cuda.empty_cache()
            out["e2e"] = e2e_block(args.kmer_size, args.abundance_min)
        if world == 1 and not args.no_human_standin and args.kmer_size <= 32:
            kc.close()
            reads = None
            torch.cuda.empty_cache()
            out["configs[4]_standin"] = human_standin_block(args.kmer_size, args.abundance_min, dev)
        elif world == 1:
            out["configs[4]_standin"] = {"skipped": "--no-human-standin" if args.no_human_standin else "k > 32"}
        print(json.dumps(out))
    kc.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
