#!/usr/bin/env python3
"""dskgpu_push_raw against dskgpu_push_reads from HOST memory on the bench workload: the 3.15 GB FASTQ text of c2 pushed as it is
(parsed on the device) and the 1.51 GB clean read stream a host parser would have handed on; push time, count time, k-mer totals.
Under `tools/kt_any.sh <tag> tools/raw_push_rate.py` the kernel stats give the k_rp_* times per 32 MB piece."""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from dsk_amd import KmerCounter, synth
gl, nr, rl = synth.workload("c2_10Mx150")
dev = torch.device("cuda:0")
reads = synth.make_reads(synth.make_genome(gl, dev), nr, rl)
tmp = tempfile.mkdtemp(prefix="dsk_raw_")
fq = os.path.join(tmp, "c2.fastq")
bench.write_fastq(reads, nr, rl, fq)
clean = reads.cpu().numpy()
del reads
text = np.fromfile(fq, dtype=np.uint8)
os.remove(fq)
print(f"text {text.size / 1e9:.2f} GB, clean stream {clean.size / 1e9:.2f} GB")
for attempt in range(3):
    for what, data in (("push_reads (clean stream)", clean), ("push_raw (FASTQ text)", text)):
        with KmerCounter(kmer_size=31, abundance_min=2) as kc:
            kc.reserve_reads(len(data) + 4096)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step = 256 << 20
            for o in range(0, len(data), step):
                if data is text:
                    kc.push_raw(data[o:o + step], kc.RAW_FASTQ, new_file=o == 0)
                else:
                    kc.push_reads(data[o:o + step])
            if data is text:
                kc.raw_finish()
            else:
                kc.encode_reads()              # (a synchronisation point for the timing; the count below starts from the encoding)
            t1 = time.perf_counter()
            kc.count()
            t2 = time.perf_counter()
            st = kc.stats()
        print(f"attempt {attempt}: {what}: push {1e3 * (t1 - t0):.1f} ms ({len(data) / 1e9 / (t1 - t0):.1f} GB/s of what is pushed, "
              f"{clean.size / 1e9 / (t1 - t0):.1f} GB/s of bases), count {1e3 * (t2 - t1):.1f} ms, k-mers {st['n_kmers']}", flush=True)
