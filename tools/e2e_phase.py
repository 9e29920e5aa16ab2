#!/usr/bin/env python3
"""Wall clock of a `dsk` run split into process start + exit vs main() vs execute() (DSK_PHASE_TIMES=1).
usage (GPU box, after tools/e2e.py wrote /tmp/e2e/<workload>.fastq): tools/e2e_phase.py [workload]"""
import os, re, subprocess, sys, time
wl = sys.argv[1] if len(sys.argv) > 1 else "ecoli50x"
root = os.path.abspath(".")
env = dict(os.environ, DSK_PHASE_TIMES="1")
for i in range(4):
    t0 = time.time()
    p = subprocess.run([f"{root}/dsk_amd/host/bin/dsk", "-file", f"/tmp/e2e/{wl}.fastq", "-kmer-size", "31", "-out", "/tmp/e2e/o2", "-verbose", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
    wall = time.time() - t0
    out = p.stdout.decode()
    def g(pat):
        m = re.search(pat, out)
        return float(m.group(1)) if m else -1.0
    vals = (wall, g(r"main\(\) took ([0-9.]+)"), g(r"total_s\s*:\s*([0-9.]+)"), g(r"teardown ([0-9.]+)"), g(r"engine_startup_s\s*:\s*([0-9.]+)"), g(r"ingest_s\s*:\s*([0-9.]+)"))
    print("wall %.3f  main %.3f  execute %.3f  teardown %.3f  startup_thread %.3f  ingest %.3f" % vals)
t0 = time.time(); subprocess.run(["/bin/true"]); print(f"(spawning /bin/true: {time.time() - t0:.3f} s)")
t0 = time.time(); subprocess.run([f"{root}/dsk_amd/host/bin/dsk", "-help"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL); print(f"(dsk -help: {time.time() - t0:.3f} s)")
