#!/usr/bin/env python3
"""Condense a tools/prof.sh run (gpurun_out/prof_<tag>/) into profiles/:
   profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (our kernels + top others)
   profiles/<tag>_pmc.md             FETCH_SIZE / WRITE_SIZE per kernel, corrected as MI355X_MICROARCH.md §HBM says
   profiles/pmc_summary.json         per-stage HBM bytes per launch (read by bench.py for roofline.traffic)
usage: tools/prof_summary.py <tag>"""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

rows = list(csv.DictReader(open(os.path.join(src, "trace", "t_kernel_stats.csv"))))
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows[:40]:
        w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])


def pmc(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch = pmc(os.path.join(src, "fetch", "f_counter_collection.csv"), "FETCH_SIZE")
write = pmc(os.path.join(src, "write", "w_counter_collection.csv"), "WRITE_SIZE")
avg_ns = {r["Name"]: float(r["AverageNs"]) for r in rows}

# kernel -> bench stage name.  k_hist/k_scatter: <W, SRC, MODE>; SRC 0 = reads (level 1), 1 = key array (level 2)
stage_of = [("k_encode", "encode"), ("k_hist<1, 0,", "sample1"), ("k_bin_moments", "sample1"), ("k_scatter<1, 0,", "scatter1"), ("k_hist<1, 1,", "hist2"),
            ("k_scatter<1, 1,", "scatter2"), ("k_scatter_al<1,", "scatter2"), ("k_count1<", "count"), ("k_count1v3<", "count"), ("k_count_chained", "count"),
            ("k_compact<1>", "compact"),
            # two-word keys (k = 33..64)
            ("k_hist<2, 0,", "sample1"), ("k_scatter<2, 0,", "scatter1"), ("k_scatter_al<2,", "scatter2"), ("k_count2v3<", "count"), ("k_count_mw<2", "count"),
            ("k_count_chained_mw", "count"), ("k_compact<2>", "compact"), ("k_top_key_aos<2>", "sort"), ("k_gather_aos<2>", "sort"), ("k_fix_runs_multi<2>", "sort"),
            ("k_fix_long_runs<2>", "sort"), ("k2_hist", "sort"), ("k2_scatter", "sort"), ("k2_split", "sort"), ("k2_cells", "sort"), ("k2_big", "sort"),
            ("k_part_sort", "sort"), ("k_rs_hist_sp", "sort"), ("k_rs_scatter_sp", "sort"), ("k_rs_hist", "sort"), ("k_rs_scatter", "sort"), ("k_set_rs_scalars", "sort"), ("k_rs_split", "sort"), ("k_rs_cells", "sort"), ("k_rs_big", "sort")]
summary = {}
lines = [f"# PMC summary ({tag})", "",
         "rocprofv3 `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate passes (TCC slots), values are KiB per dispatch,",
         "averaged over dispatches.  Correction per MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE counts a wide",
         "coalesced read at half its bytes (confirmed here: k_encode reads the 1.51 GB ASCII stream with 16 B/lane loads and",
         "reports 0.755 GB), so `read` below = 2 x FETCH_SIZE; WRITE_SIZE is taken as is (k_encode writes 0.566 GB of",
         "packed words and reports 0.566 GB).", "",
         "| kernel | stage | avg ms | FETCH_SIZE KiB | WRITE_SIZE KiB | read GB (2x) | write GB | HBM GB/launch | GB/s |",
         "|---|---|---|---|---|---|---|---|---|"]
for kname in sorted(set(fetch) | set(write)):
    if not any(kname.startswith("void " + p) or kname.startswith(p) for p, _ in stage_of):
        continue
    st = next(s for p, s in stage_of if kname.startswith("void " + p) or kname.startswith(p))
    fk, wk = fetch.get(kname, 0.0), write.get(kname, 0.0)
    rd, wr = 2 * fk * 1024, wk * 1024
    ms = avg_ns.get(kname, 0.0) / 1e6
    tot = rd + wr
    prev = summary.get(st, {"hbm_bytes_per_launch": 0.0, "read_bytes": 0.0, "write_bytes": 0.0, "avg_ms": 0.0})      # a stage of several kernels: sums
    summary[st] = {"hbm_bytes_per_launch": prev["hbm_bytes_per_launch"] + tot, "read_bytes": prev["read_bytes"] + rd,
                   "write_bytes": prev["write_bytes"] + wr, "avg_ms": prev["avg_ms"] + ms}
    lines.append(f"| `{kname[:48]}` | {st} | {ms:.3f} | {fk:.0f} | {wk:.0f} | {rd / 1e9:.2f} | {wr / 1e9:.2f} | {tot / 1e9:.2f} | {tot / 1e9 / (ms / 1e3) if ms else 0:.0f} |")
open(os.path.join(dst, f"{tag}_pmc.md"), "w").write("\n".join(lines) + "\n")
# what the numbers belong to: bench.py quotes them (roofline.traffic_from_profiles) only for the same workload and k
bench_line = [l for l in open(os.path.join(src, "bench_trace.log")) if l.startswith('{"metric"')]
meta = {"id": f"profiles/{tag}_pmc.md", "workload": "c2_10Mx150", "kmer_size": 31}
if bench_line:
    cfg = json.loads(bench_line[-1]).get("config", {})
    meta["workload"] = cfg.get("workload", "").split(":")[0] or meta["workload"]
    meta["kmer_size"] = cfg.get("kmer_size", meta["kmer_size"])
try:
    import subprocess
    meta["commit"] = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"]).decode().strip()
except Exception:
    meta["commit"] = None
summary["_meta"] = meta
# pmc_summary.json is what bench.py reads for the default workload: only the k = 31 bench profile writes it
json.dump(summary, open(os.path.join(dst, "pmc_summary.json" if int(meta["kmer_size"]) == 31 else f"{tag}_pmc_summary.json"), "w"), indent=1)
bench = [l for l in open(os.path.join(src, "bench_trace.log")) if l.startswith('{"metric"')]
if bench:
    open(os.path.join(dst, f"{tag}_bench_under_rocprof.json"), "w").write(bench[-1])
print("\n".join(lines))
