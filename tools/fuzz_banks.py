#!/usr/bin/env python3
"""Random multi-bank jobs against one context of the C-ABI: 2-5 banks (some EMPTY, some of one read), pushed as parsed reads or as
raw FASTQ / FASTA text, `dskgpu_next_bank` behind every bank, or the whole stream in HBM with `dskgpu_set_banks`; a random
solidity kind (+ custom mask), abundance window, k (one- and two-word keys), -histo2D.  Rows, histogram, 2-D histogram and k-mer
total against the numpy restatement of the multi-bank semantics over per-bank oracle counts (tests/test_gpu_parity.py
_bank_reference).   python tools/fuzz_banks.py [seed0=0] [n=200]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsk_amd import KmerCounter                                          # noqa: E402
from tests.oracle_py import Oracle                                       # noqa: E402
from tests.test_gpu_parity import _bank_reference                       # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    oracle = Oracle(os.path.join(ROOT, "oracle", "libdsk_oracle.so"))
    dev = torch.device("cuda:0")
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        k = int(rng.choice([15, 21, 27, 31, 33, 47]))
        B = int(rng.integers(2, 6))
        genome = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), int(rng.integers(200, 5000))))
        banks = []
        for b in range(B):
            n = int(rng.choice([0, 1])) if rng.random() < 0.3 else int(rng.integers(2, 200))
            seqs = []
            for _ in range(n):
                L = int(rng.integers(0, 200)); s = int(rng.integers(0, max(1, len(genome) - L)))
                seqs.append(genome[s: s + L])
            banks.append(seqs)
        streams = [np.frombuffer(b"".join(s + b"\n" for s in seqs) or b"", dtype=np.uint8) for seqs in banks]
        kind = str(rng.choice(["sum", "min", "max", "one", "all", "custom"]))
        mask = int(rng.integers(0, 1 << B)) if kind == "custom" else 0
        amin = int(rng.choice([1, 2, 3])); amax = int(rng.choice([2**31 - 1, 5, 30]))
        h2 = bool(rng.random() < 0.5)
        ref_streams = [s if len(s) else np.frombuffer(b"\n", dtype=np.uint8) for s in streams]
        want_k, want_a, want_h, want_h2, want_total = _bank_reference(oracle, ref_streams, k, kind, amin, amax, mask)
        mode = int(rng.integers(0, 3))
        with KmerCounter(kmer_size=k, abundance_min=amin, abundance_max=amax, solidity_kind=kind, solidity_custom=mask, histo2d=h2) as kc:
            if mode == 2:                                     # the whole stream in HBM + end offsets
                whole = np.concatenate([s for s in streams]) if sum(len(s) for s in streams) else np.zeros(0, np.uint8)
                t = torch.from_numpy(whole.copy()).to(dev) if len(whole) else torch.zeros(1, dtype=torch.uint8, device=dev)
                kc.set_reads_device(t.data_ptr(), len(whole))
                kc.set_banks([int(x) for x in np.cumsum([len(s) for s in streams])])
            else:
                for seqs, s in zip(banks, streams):
                    if mode == 1 and seqs and rng.random() < 0.7:                          # raw text of this bank
                        fq = rng.random() < 0.5
                        text = b"".join((b"@r\n" + q + b"\n+\n" + b"I" * len(q) + b"\n") if fq else (b">s\n" + q + b"\n") for q in seqs)
                        kc.push_raw(text, kc.RAW_FASTQ if fq else kc.RAW_FASTA, new_file=True)
                    elif len(s):
                        kc.push_reads(s[:-1])                                              # (the separator behind the push is implied)
                    kc.next_bank()
            kc.count()
            rows, ab = kc.rows()
            st = kc.stats()
            hist = kc.histogram()
            hist2 = kc.histogram2d() if h2 else None
        if k > 32:
            got_k = np.array([(int(h) << 64) | int(l) for l, h in zip(rows[:, 0], rows[:, 1])], dtype=object)
        else:
            got_k = rows[:, 0]
        ok = st["n_kmers"] == want_total and len(got_k) == len(want_k) and (got_k == want_k).all() and (ab == want_a).all() and (hist == want_h).all() and \
            (hist2 is None or (hist2 == want_h2).all())
        if not ok:
            print(f"seed {seed}: k {k} banks {[len(x) for x in banks]} kind {kind} mask {mask:b} window [{amin}, {amax}] mode {mode} histo2D {h2}: "
                  f"engine {st['n_kmers']} k-mers / {len(got_k)} rows, reference {want_total} / {len(want_k)}")
            sys.exit(1)
    print(f"fuzz ok: {count} multi-bank jobs")


if __name__ == "__main__":
    main()
