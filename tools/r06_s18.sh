#!/bin/bash
# r06 step 18: where the wall clock of `dsk reads.fastq.gz` goes outside execute() (BENCH e2e: wall - total_s = 0.16 s for gzip, 0.03 s for the plain file)
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r06_s18; mkdir -p $o
python3 - <<'PY'
import os, subprocess, sys, tempfile, time, gzip, re
import torch
sys.path.insert(0, os.getcwd())
import bench
from dsk_amd import synth
dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp(prefix="dsk_e2e_")
gl, nr, rl = synth.workload("ecoli50x")
reads = synth.make_reads(synth.make_genome(gl, dev), nr, rl)
fq = os.path.join(tmp, "e.fastq")
bench.write_fastq(reads, nr, rl, fq)
del reads
with gzip.open(fq + ".gz", "wb", compresslevel=1) as f: f.write(open(fq, "rb").read())
dsk = os.path.join(os.getcwd(), "dsk_amd", "host", "bin", "dsk")
for name in (fq, fq + ".gz"):
    for rep in range(2):
        if os.path.exists(os.path.join(tmp, "o.h5")): os.remove(os.path.join(tmp, "o.h5"))
        t0 = time.perf_counter(); w0 = time.time()
        import resource
        r0 = resource.getrusage(resource.RUSAGE_CHILDREN)
        p = subprocess.run([dsk, "-file", name, "-kmer-size", "31", "-abundance-min", "2", "-out", os.path.join(tmp, "o"), "-verbose", "1"],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, DSK_PHASE_TIMES="1"))
        dt = time.perf_counter() - t0; w1 = time.time()
        out = p.stdout.decode(); err = p.stderr.decode()
        m = re.search(r"entered at ([\d.]+), leaves at ([\d.]+)", err)
        print("   start-up before main %.3f s, exit after main %.3f s" % (float(m.group(1)) - w0, w1 - float(m.group(2))))
        tot = re.search(r"total_s\s*:\s*([\d.]+)", out).group(1)
        r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
        mainl = [" ".join(l.split())[:60] for l in err.splitlines() if "[dsk]" in l and "RssAnon" in l] + ["user %.2f sys %.2f minflt %d maxrss %d MB" % (r1.ru_utime - r0.ru_utime, r1.ru_stime - r0.ru_stime, r1.ru_minflt - r0.ru_minflt, r1.ru_maxrss // 1024)]
        ing = re.search(r"ingest_s\s*:\s*([\d.]+)", out).group(1); mainl.append("ingest " + ing)
        print(os.path.basename(name), "wall", round(dt, 3), "total_s", tot, " | ".join(x.strip() for x in mainl), flush=True)
PY
