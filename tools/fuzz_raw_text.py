#!/usr/bin/env python3
"""Differential fuzz of the device-side parser (dskgpu_push_raw, csrc/rawparse.h) against a plain restatement of the HOST parser's
state machine (host/bank.cpp RecordParser: FASTA lines joined, FASTQ qualities read by COUNT, blanks dropped from sequence lines,
records of either format in one file): well-formed FASTA / FASTQ text is damaged -- lines deleted, duplicated, split, joined, bytes
inserted and removed, records of the other format spliced in -- and pushed with random cuts.  For every text the device either
gives it back (DSKGPU_E_FORMAT: the caller parses on the host) or counts EXACTLY what the host parser would have handed on.
   python tools/fuzz_raw_text.py [seed=0] [n=300]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsk_amd import KmerCounter                                          # noqa: E402
from dsk_amd.engine import DskGpuError                                   # noqa: E402
from tests.test_gpu_raw_parse import random_cuts                         # noqa: E402


from tests.raw_text_model import host_parser, base_text, damage       # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    dev = torch.device("cuda:0")
    given_back = same = 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        fmt = "fq" if rng.random() < 0.6 else "fa"
        text = damage(rng, base_text(rng, fmt), fmt)
        if text[:1] not in (b"@", b">") or (fmt == "fq") != (text[:1] == b"@"):
            continue                                   # (the bank looks at the first record before it offers its text: host/bank.cpp record_kind)
        k = int(rng.choice([11, 21, 31]))
        want_stream = np.frombuffer(host_parser(text) + b"\n", dtype=np.uint8).copy()
        t = torch.from_numpy(want_stream).to(dev)
        with KmerCounter(kmer_size=k, abundance_min=1) as kc:
            kc.set_reads_device(t.data_ptr(), t.numel()); kc.count()
            want = (kc.stats()["n_kmers"], kc.rows())
        with KmerCounter(kmer_size=k, abundance_min=1) as kc:
            cuts = random_cuts(rng, len(text), int(rng.choice([1, 3, 20])))
            started = False
            for a, b in zip(cuts[:-1], cuts[1:]):
                if b > a or not started:
                    kc.push_raw(text[a:b], kc.RAW_FASTQ if fmt == "fq" else kc.RAW_FASTA, new_file=not started); started = True
            try:
                kc.raw_finish()
            except DskGpuError as e:
                assert e.code == -6
                given_back += 1
                continue
            kc.count()
            got = (kc.stats()["n_kmers"], kc.rows())
        ok = got[0] == want[0] and got[1][0].shape == want[1][0].shape and (got[1][0] == want[1][0]).all() and (got[1][1] == want[1][1]).all()
        if not ok:
            print(f"seed {seed}: {fmt} k {k}: the device counted {got[0]} k-mers, the host parser's stream holds {want[0]} -- NOT given back\n{text[:1500]!r}")
            sys.exit(1)
        same += 1
    print(f"fuzz ok: {same} texts counted as the host parser would, {given_back} given back")


if __name__ == "__main__":
    main()
