#!/usr/bin/env python3
"""Where a `dsk` run spends its wall clock (DSK_PHASE_TIMES=1): process wall vs main() vs execute() phases, on a synthetic FASTQ
written the fast way (bench.write_fastq).  usage (GPU box): python tools/e2e_phase2.py [workload=c2_10Mx150] [runs=4]"""
import os, re, subprocess, sys, time
sys.path.insert(0, ".")
import torch
from dsk_amd import synth
from bench import write_fastq
wl = sys.argv[1] if len(sys.argv) > 1 else "c2_10Mx150"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
extra = sys.argv[3:]
dev = torch.device("cuda:0")
gl, nr, rl = synth.workload(wl)
reads = synth.make_reads(synth.make_genome(gl, dev), nr, rl)
os.makedirs("/tmp/e2e", exist_ok=True)
fq = f"/tmp/e2e/{wl}.fastq"
write_fastq(reads, nr, rl, fq)
del reads
torch.cuda.empty_cache()
root = os.path.abspath(".")
env = dict(os.environ, DSK_PHASE_TIMES="1")
for i in range(runs):
    t0 = time.time()
    p = subprocess.run([f"{root}/dsk_amd/host/bin/dsk", "-file", fq, "-kmer-size", "31", "-out", "/tmp/e2e/o2", "-verbose", "1", *extra], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
    wall = time.time() - t0
    out = p.stdout.decode()
    def g(pat):
        m = re.search(pat, out)
        return float(m.group(1)) if m else -1.0
    print("wall %.3f | main %.3f | execute-total %.3f: setup %.3f ingest %.3f (engine_startup %.3f reserve_reads %.3f reserve_work %.3f) count %.3f write %.3f | teardown %.3f" % (
        wall, g(r"main\(\) took ([0-9.]+)"), g(r"total_s\s*:\s*([0-9.]+)"), g(r"setup_s\s*:\s*([0-9.]+)"), g(r"ingest_s\s*:\s*([0-9.]+)"), g(r"engine_startup_s\s*:\s*([0-9.]+)"),
        g(r"reserve_reads_s\s*:\s*([0-9.]+)"), g(r"reserve_work_s\s*:\s*([0-9.]+)"), g(r"count_s\s*:\s*([0-9.]+)"), g(r"write_s\s*:\s*([0-9.]+)"), g(r"teardown ([0-9.]+)")))
    for ln in out.splitlines():
        if ln.startswith("[dsk]"):
            print("   ", ln)
t0 = time.time(); subprocess.run([f"{root}/dsk_amd/host/bin/dsk", "-help"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL); print(f"(dsk -help: {time.time() - t0:.3f} s)")
