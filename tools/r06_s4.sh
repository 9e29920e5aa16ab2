#!/bin/bash
# r06 step 4: what differs between a well-placed and a badly placed buffer, seen from the memory side: TCC / EA counters
# of every k_place_probe dispatch of one placed bench start-up (8 candidates per big buffer, 3 probes each), joined with the dispatch's
# duration from the kernel trace of the same run.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r06_s4; mkdir -p $o
i=0
for set in "TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_TAG_STALL TCC_TOO_MANY_EA_WRREQS_STALL" "TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_64B TCC_BUSY TCC_CYCLE" "TCC_EA0_WRREQ_LEVEL TCC_WRITEBACK TCC_NORMAL_WRITEBACK TCC_NORMAL_EVICT"; do
  i=$((i+1)); d=$o/set$i; mkdir -p $d
  DSKGPU_VERBOSE=1 rocprofv3 --pmc $set --kernel-trace --output-format csv json -d $d -o p -- python3 bench.py --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --no-k63 --no-repeat-rich --steps 1 --warmup 1 > $d/run.log 2>&1
  grep "placement of" $d/run.log | head -8 > $d/placement.txt
  python3 - $d "$set" <<'PY'
import csv, sys, json, os, collections
d, names = sys.argv[1], sys.argv[2].split()
kt = {r["Dispatch_Id"]: (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(os.path.join(d, "p_kernel_trace.csv")))}
acc = collections.defaultdict(dict)
rows = list(csv.DictReader(open(os.path.join(d, "p_counter_collection.csv"))))
print("csv columns:", list(rows[0].keys()) if rows else None, "rows", len(rows))
for r in rows:
    if "k_place_probe" in r["Kernel_Name"]:
        acc[r["Dispatch_Id"]][r["Counter_Name"]] = acc[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
out = []
for did, c in acc.items():
    out.append((kt.get(did, ("", 0))[1] / 1e6, did, c))
out.sort()
print("k_place_probe dispatches:", len(out))
for ms, did, c in out[:6] + out[-6:]:
    print(f"{ms:8.3f} ms  id {did:>6s}  " + "  ".join(f"{n}={c.get(n, 0):.3e}" for n in names))
# per-instance values if the json keeps them
try:
    j = json.load(open(os.path.join(d, "p_results.json")))
    s = json.dumps(j)[:0]
    recs = j["rocprofiler-sdk-tool"][0]
    print("json keys:", list(recs.keys())[:20])
    cc = recs.get("callback_records", {}).get("counter_collection", []) or recs.get("buffer_records", {}).get("counter_collection", [])
    print("counter_collection records:", len(cc))
    if cc:
        print("first record:", json.dumps(cc[0])[:1500])
except Exception as e:
    print("json:", repr(e)[:300])
PY
  rm -f $d/*.json
done > $o/summary.txt 2>&1
cat $o/summary.txt | cut -c1-400
