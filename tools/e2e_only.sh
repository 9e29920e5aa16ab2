#!/bin/bash
# the bench's file -> .h5 block alone (dsk binary, best of 3): tools/e2e_only.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 - <<PY
import json, sys
sys.path.insert(0, ".")
import bench
r = bench.e2e_block(31, 2)
for name in ("ecoli50x", "c2_10Mx150"):
    b = r.get(name, {})
    print(name, {k: b.get("plain", {}).get(k) for k in ("wall_s", "ingest_s", "count_s", "write_s", "total_s")}, "gzip", b.get("gzip", {}).get("wall_s"), "bgzf", b.get("bgzf", {}).get("wall_s"))
PY
