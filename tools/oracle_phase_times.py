import os, sys, time, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from tests.oracle_py import Oracle
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
o = Oracle(os.path.join(root, "oracle", "libdsk_oracle.so"))
n = 4_000_000
rng = np.random.default_rng(1)
g = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n * 3)
st = rng.integers(0, len(g) - 150, size=n)
s = np.ascontiguousarray(np.concatenate([g[st[:, None] + np.arange(150)[None, :]], np.full((n, 1), 10, np.uint8)], axis=1).reshape(-1))
for t in (32, 64, 32):
    t0 = time.time(); h = o.lib.dsko_count(s.ctypes.data, len(s), 31, t); t1 = time.time(); o.lib.dsko_free(h); t2 = time.time()
    print(f"{t} threads: count {t1 - t0:.3f} s, free of the result {t2 - t1:.3f} s", flush=True)
