#!/usr/bin/env python3
"""Random CALL SEQUENCES against one context of the C-ABI, with a plain model of the read stream beside it: dskgpu_push_reads,
dskgpu_push_raw (+ dskgpu_raw_finish, or left to the next call that needs the stream), dskgpu_stream_bytes, dskgpu_rewind_reads to an
earlier mark, dskgpu_reserve_reads / dskgpu_reserve_work, dskgpu_encode_reads (the next push starts a new read set), dskgpu_count --
several counts per context, pushes between them.  After every count: the stream's length, the k-mer total, the rows and the
histogram against the CPU oracle on the model's bytes.   python tools/fuzz_abi.py [seed0=0] [n=200]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsk_amd import KmerCounter                                          # noqa: E402
from tests.oracle_py import Oracle                                       # noqa: E402
from tests.test_gpu_raw_parse import make_fasta, make_fastq, model_stream, random_cuts      # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    oracle = Oracle(os.path.join(ROOT, "oracle", "libdsk_oracle.so"))
    assert torch.cuda.is_available()
    alpha = np.frombuffer(b"ACGTACGTACGTacgtN", dtype=np.uint8)
    ncounts = 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        k = int(rng.choice([15, 21, 31, 33, 63]))
        amin = int(rng.choice([1, 2]))
        trace = []
        with KmerCounter(kmer_size=k, abundance_min=amin) as kc:
            model = bytearray()
            marks = []                                   # (stream length, model length)
            encoded = False                              # dskgpu_encode_reads was called: the next push starts a new read set
            raw_open = None                              # a raw session whose finish is still to come: (files so far)
            def start_push():
                nonlocal encoded, model, marks
                if encoded:
                    model = bytearray(); marks = []; encoded = False
            def close_raw():
                nonlocal raw_open
                if raw_open is not None:
                    if model:
                        model.extend(b"\n")
                    model.extend(model_stream(raw_open))          # (its separators between files and the terminator)
                    raw_open = None
            for step in range(int(rng.integers(3, 14))):
                op = int(rng.integers(0, 10))
                if op <= 2:                                                   # push_reads
                    close_raw(); start_push()
                    reads = b"\n".join(bytes(rng.choice(alpha, int(rng.integers(0, 200)))) for _ in range(int(rng.integers(1, 300))))
                    kc.push_reads(reads); model.extend(reads + b"\n"); trace.append(("push_reads", len(reads)))
                elif op <= 4:                                                 # push_raw: one or two files, cut anywhere
                    start_push()
                    files = []
                    for _ in range(int(rng.integers(1, 3))):
                        fq = rng.random() < 0.5
                        text = make_fastq(rng, int(rng.integers(1, 120)), 0, 200, crlf=rng.random() < 0.2, last_newline=rng.random() < 0.7) if fq else \
                            make_fasta(rng, int(rng.integers(1, 60)), 0, 600, crlf=rng.random() < 0.2, last_newline=rng.random() < 0.7)
                        files.append((text, "fq" if fq else "fa"))
                    if raw_open is None:
                        raw_open = []
                    for text, fmt in files:
                        cuts = random_cuts(rng, len(text), int(rng.choice([1, 2, 9])))
                        started = False
                        for a, b in zip(cuts[:-1], cuts[1:]):
                            if b > a or not started:
                                kc.push_raw(text[a:b], kc.RAW_FASTQ if fmt == "fq" else kc.RAW_FASTA, new_file=not started); started = True
                        raw_open.append((text, fmt))
                    trace.append(("push_raw", [(f, len(t)) for t, f in files]))
                    if rng.random() < 0.5:
                        kc.raw_finish(); close_raw(); trace.append(("raw_finish",))
                elif op == 5:                                                 # mark
                    close_raw()
                    n = kc.stream_bytes()
                    assert n == len(model) or (encoded and n == 0), (seed, trace, n, len(model))
                    if not encoded:
                        marks.append((n, len(model))); trace.append(("mark", n))
                elif op == 6 and marks and not encoded:                       # rewind to a mark
                    close_raw()
                    n, m = marks[int(rng.integers(0, len(marks)))]
                    kc.rewind_reads(n); del model[m:]
                    marks = [x for x in marks if x[0] <= n]; trace.append(("rewind", n))
                elif op == 7:
                    close_raw()                                          # (dskgpu_reserve_reads needs the stream's length: it finishes the raw pushes)
                    kc.reserve_reads(int(rng.integers(0, 1 << 22))); trace.append(("reserve_reads",))
                    if rng.random() < 0.5:
                        kc.reserve_work(int(rng.integers(1 << 10, 1 << 24))); trace.append(("reserve_work",))
                elif op == 8 and model and not encoded and raw_open is None:  # encode_reads
                    kc.encode_reads(); encoded = True; trace.append(("encode_reads",))
                else:                                                         # count
                    close_raw()
                    kc.count(); trace.append(("count",))
                    ref = oracle.count(np.frombuffer(bytes(model) if model else b"\n", dtype=np.uint8).copy(), k)
                    rows, ab = kc.rows()
                    st = kc.stats()
                    keep = ref.ab >= amin
                    ok = st["n_kmers"] == ref.total and st["n_distinct"] == ref.distinct and rows.shape[0] == int(keep.sum()) and \
                        (rows == ref.words()[keep]).all() and (ab == ref.ab[keep]).all() and (kc.histogram() == ref.histogram(10000)).all()
                    if not ok:
                        print(f"seed {seed}: count differs from the model (k {k}): engine {st['n_kmers']} k-mers, model {ref.total}\n  {trace}")
                        sys.exit(1)
                    ncounts += 1
    print(f"fuzz ok: {count} call sequences, {ncounts} counts checked")


if __name__ == "__main__":
    main()
