#!/usr/bin/env python3
"""Thread scaling of the CPU oracle's count (bench.py's cpu_baseline) on this host: tools/oracle_scaling.py [reads=2000000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.oracle_py import Oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
o = Oracle(os.path.join(os.path.dirname(__file__), "..", "oracle", "libdsk_oracle.so"))
rng = np.random.default_rng(1)
g = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n * 3)
st = rng.integers(0, len(g) - 150, size=n)
s = np.concatenate([g[st[:, None] + np.arange(150)[None, :]], np.full((n, 1), 10, np.uint8)], axis=1).reshape(-1)
base = None
for t in (1, 8, 32, 64, 128, os.cpu_count()):
    t0 = time.time(); tot, d = o.count_only(s, 31, threads=t); dt = time.time() - t0
    base = base or tot / dt
    print(f"{t:4d} threads: {dt:6.2f} s  {tot / dt / 1e6:8.1f} M k-mers/s  x{tot / dt / base:.1f}", flush=True)
