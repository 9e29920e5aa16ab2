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


def host_parser(text):
    """host/bank.cpp RecordParser: -> the read stream it hands on (one sequence per record, '\\n' behind each)"""
    out, st, seq, qleft = [], "HEADER", None, 0
    lines = text.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()                                  # (finish() only handles a non-empty remainder)

    def strip(ln):
        return ln.translate(None, b"\r \t")

    for ln in lines:
        if st == "HEADER":
            if ln[:1] == b">":
                st, seq = "SEQ_FA", b""
            elif ln[:1] == b"@":
                st, seq = "SEQ_FQ", b""
        elif st == "SEQ_FA":
            if ln[:1] == b">":
                out.append(seq); seq = b""
            else:
                seq += strip(ln)
        elif st == "SEQ_FQ":
            if ln[:1] == b"+":
                st, qleft = "QUAL", len(seq)
                if qleft == 0:
                    out.append(seq); seq = None; st = "HEADER"
            else:
                seq += strip(ln)
        else:
            q = len(ln) - ln.count(b"\r")
            if q >= qleft:
                out.append(seq); seq = None; st = "HEADER"
            else:
                qleft -= q
    if seq is not None:
        out.append(seq)
    return b"".join(s + b"\n" for s in out)


def base_text(rng, fmt):
    n = int(rng.integers(3, 60))
    eol = b"\r\n" if rng.random() < 0.2 else b"\n"
    alpha = np.frombuffer(b"ACGTACGTACGTacgtN", dtype=np.uint8)
    out = []
    for i in range(n):
        L = int(rng.integers(0, 120))
        seq = bytes(rng.choice(alpha, L))
        if fmt == "fq":
            q = bytes(rng.integers(33, 74, L, dtype=np.uint8))
            out += [b"@r%d" % i + eol, seq + eol, b"+" + eol, q + eol]
        else:
            out += [b">s%d" % i + eol]
            w = int(rng.choice([30, 60, 1000]))
            out += [seq[a: a + w] + eol for a in range(0, L, w)]
    return out


def damage(rng, lines, fmt):
    lines = list(lines)
    for _ in range(int(rng.integers(0, 4))):
        if not lines:
            break
        i = int(rng.integers(0, len(lines)))
        kind = int(rng.integers(0, 8))
        if kind == 0:
            del lines[i]
        elif kind == 1:
            lines.insert(i, lines[i])
        elif kind == 2 and len(lines[i]) > 3:
            c = int(rng.integers(1, len(lines[i]) - 1)); lines[i: i + 1] = [lines[i][:c] + b"\n", lines[i][c:]]
        elif kind == 3 and i + 1 < len(lines):
            lines[i: i + 2] = [lines[i].rstrip(b"\r\n") + lines[i + 1]]
        elif kind == 4 and len(lines[i]) > 2:
            c = int(rng.integers(0, len(lines[i]) - 1)); lines[i] = lines[i][:c] + bytes([int(rng.choice(list(b"ACGT@>+ \t\r;N")))]) + lines[i][c:]
        elif kind == 5 and len(lines[i]) > 2:
            c = int(rng.integers(0, len(lines[i]) - 1)); lines[i] = lines[i][:c] + lines[i][c + 1:]
        elif kind == 6:
            lines.insert(i, b"\n")
        elif kind == 7:                                # a record of the other format at a record border
            j = next((x for x in range(i, len(lines)) if lines[x][:1] in (b"@", b">")), None)
            if j is not None:
                lines[j:j] = [b">x\n", b"ACGTTGCAACGTTGCAACGTTGCAACGTTGCAAC\n"] if fmt == "fq" else [b"@x\n", b"ACGTTGCAACGTTGCAACGTTGCAACGTTGCAAC\n", b"+\n", b"IIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n"]
    text = b"".join(lines)
    if rng.random() < 0.3:
        text = text.rstrip(b"\r\n")
    return text


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
