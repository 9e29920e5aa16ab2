#!/usr/bin/env python3
"""The stand-in for BASELINE.json configs[4] (30x human: 3 Gbp repeat-rich genome, 600 M x 150 bp, k = 31, abundance-min 2) on ONE GPU,
multi-pass, with the size-independent invariants checked on the device (the rows never leave HBM: 3 * 10^9 of them).
   python tools/human_standin.py [reads_millions=600] [k=31] [steps=1] [abundance_min=2] [keep-ascii|-] [partition|global]
reads_millions = 75 is the 1/8 shard of the same genome (what one of 8 GPUs holds)."""
import json, sys, time
import torch
sys.path.insert(0, ".")
from dsk_amd import KmerCounter, synth


from tests.full_size import device_invariants, valid_windows      # noqa: E402


def main():
    mreads = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    amin = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    dev = torch.device("cuda:0")
    nr, rl = int(mreads * 1e6), 150
    t0 = time.perf_counter()
    genome = synth.make_genome_repeats(3_000_000_000, dev)
    reads = synth.make_reads(genome, nr, rl, polya_rate=0.002)
    del genome
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    t_gen = time.perf_counter() - t0
    out = {"workload": f"c5_human30x stand-in: {nr} reads x {rl} bp of a repeat-rich 3 Gbp genome (1 % one 300 bp family, 4 tandem arrays, 0.2 % poly-A reads)",
           "kmer_size": k, "generate_s": round(t_gen, 1)}
    keep_ascii = len(sys.argv) > 5 and sys.argv[5] == "keep-ascii"          # (the round-4 way: the 90 GB of bytes stay in HBM during the count)
    part = not (len(sys.argv) > 6 and sys.argv[6] == "global")          # row order: the reference's contract (ascending inside every output partition) unless "global"
    out["row_order"] = "partition" if part else "global"
    with KmerCounter(kmer_size=k, abundance_min=amin, timing=True, partition_order=part) as kc:
        kc.set_reads_device(reads.data_ptr(), reads.numel())
        n_valid = None
        if not keep_ascii:
            # the reads are turned into their 2-bit form once (dskgpu_encode_reads: 0.375 B per base) and the bytes are given back: what
            # DSK does with its bank -- read once per pass, nothing of it kept (README.md:126-130) -- and what decides how many sweeps
            # over the reads the passes need (90 GB of bytes + 34 GB encoded left room for a third of the records)
            n_valid = valid_windows(reads, nr, rl, k)
            t0 = time.perf_counter(); kc.encode_reads(); out["encode_reads_s"] = round(time.perf_counter() - t0, 3)
            del reads
            reads = None
            torch.cuda.empty_cache()
        out["ascii_reads_resident_during_count"] = bool(keep_ascii)
        times = []
        for _ in range(steps + 1):                 # the first count also allocates every buffer
            t0 = time.perf_counter(); kc.count(); times.append(time.perf_counter() - t0)
        st = kc.stats(); hist = kc.histogram()
        best = min(times[1:]) if steps else times[0]
        out.update({"first_count_s": round(times[0], 3), "count_s": round(best, 3), "kmer_occurrences_per_s": st["n_kmers"] / best,
                    "distinct_kmers_per_s": st["n_distinct"] / best, **{x: st[x] for x in ("n_kmers", "n_distinct", "n_solid", "n_passes", "n_read_sweeps", "n_retries", "sort_fallback", "n_ext_regions", "n_heavy", "n_final_bins")},
                    "stage_ms": {a: round(b, 2) for a, b in kc.stage_times()}})
        free_b, total_b = torch.cuda.mem_get_info()
        out["hbm_used_gb"] = round((total_b - free_b) * 1e-9, 1)
        if k <= 31 and amin == 2:
            out["n_partitions"] = st["n_partitions"]
            out["invariants"] = device_invariants(kc, st, hist, k, reads, nr, rl, dev, n_valid=n_valid, partition_order=part and st["n_partitions"] > 64)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
