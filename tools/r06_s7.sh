#!/bin/bash
# r06 step 7: the whole GPU suite + smoke + the default bench line (e2e block: the parallel gzip inflate on the GPU box's host) + per-channel
# DRAM credit stalls of the placement probes
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r06_s7; mkdir -p $o
timeout 3000 python -m pytest tests/ -x -q -m gpu --durations=8 2>&1 | tail -30 > $o/suite.log; tail -14 $o/suite.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python3 bench.py ) > $o/bench.log 2>&1; tail -4 $o/bench.log | cut -c1-200
grep '^{"metric"' $o/bench.log | tail -1 > $o/bench.json
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r06_s7/bench.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "ms_per_step_no_place", "ms_per_step_global_order", "first_step_s")})
print("stage_ms", d.get("stage_ms")); print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "frac", "step_frac", "partition_plus_hash")})
print("k63", d.get("k63", {}).get("ms_per_step"), d.get("k63", {}).get("stage_ms"))
print("repeat_rich", d.get("repeat_rich", {}).get("ms_per_step"))
print("e2e", json.dumps(d.get("e2e"))[:3000])
print("standin", json.dumps(d.get("human_standin"))[:1500])
print("cpu", d.get("cpu_baseline"))
PY
# per-channel view of one counter set (json keeps the instances)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
d=$o/chan; mkdir -p $d
DSKGPU_VERBOSE=1 rocprofv3 --pmc TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ --kernel-trace --output-format json -d $d -o p -- python3 bench.py --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --no-k63 --no-repeat-rich --steps 1 --warmup 1 > $d/run.log 2>&1
python3 - <<'PY'
import json, glob, collections
f = glob.glob("gpurun_out/r06_s7/chan/*results.json")[0]
j = json.load(open(f))["rocprofiler-sdk-tool"][0]
names = {}
for c in j.get("counters", []):
    names[c["id"]["handle"]] = (c.get("name"), c.get("dimensions"))
ks = {k["kernel_id"]: k.get("formatted_kernel_name", k.get("kernel_name", "")) for k in j.get("kernel_symbols", [])}
recs = j["callback_records"].get("counter_collection") or j["buffer_records"].get("counter_collection")
print("counters:", list(names.values())[:6])
rows = []
for r in recs:
    di = r["dispatch_data"]["dispatch_info"]
    if "k_place_probe" not in ks.get(di["kernel_id"], ""): continue
    ms = (r["dispatch_data"]["end_timestamp"] - r["dispatch_data"]["start_timestamp"]) / 1e6
    per = collections.defaultdict(list)
    for x in r["records"]:
        per[names.get(x["counter_id"]["handle"], ("?",))[0]].append(x["value"])
    rows.append((di["dispatch_id"], ms, per))
print("probe dispatches", len(rows), "values per counter in one record:", {k: len(v) for k, v in rows[0][2].items()} if rows else None)
big = [r for r in rows if r[1] > 3.0]
big.sort(key=lambda r: r[1])
import statistics
for lab, r in (("best", big[0]), ("worst", big[-1])) if big else ():
    for k, v in r[2].items():
        v = sorted(v)
        print(lab, f"{r[1]:.3f} ms", k, "n", len(v), "sum %.4e" % sum(v), "min %.3e max %.3e" % (v[0], v[-1]), "max/mean %.3f" % (v[-1] / (sum(v) / len(v))), "stdev/mean %.3f" % (statistics.pstdev(v) / (sum(v) / len(v))))
PY
rm -f $d/*.json
