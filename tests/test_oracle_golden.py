"""Pins the CPU oracle against the reference's own golden files.

Cases = scripts/simple_test.sh:35-135 (T1..T6) with goldens test/k27.histo,
test/rlong.histo, test/readN.histo, test/short.parse_results (copied as data
into tests/golden/), plus the known answers of SURVEY.md App. A.
"""
import hashlib
import os

import numpy as np
import pytest


def read_histo(path):
    rows = [l.split() for l in open(path) if l.strip()]
    idx = np.array([int(r[0]) for r in rows])
    val = np.array([int(r[1]) for r in rows], dtype=np.uint64)
    assert (idx == np.arange(1, len(rows) + 1)).all()
    return val


def histo_of(oracle, uri, k):
    s, _ = oracle.load_bank(uri)
    return oracle.count(s, k).histogram(10000)[1:]


def test_T1_single_gz(oracle, golden_dir):
    got = histo_of(oracle, os.path.join(golden_dir, "read50x_ref10K_e001.fasta.gz"), 27)
    assert (got == read_histo(os.path.join(golden_dir, "k27.histo"))).all()


def test_T2_multiple_gz(oracle, golden_dir):
    uri = ",".join(os.path.join(golden_dir, f"c{i}.fasta.gz") for i in (1, 2, 3, 4))
    assert (histo_of(oracle, uri, 27) == read_histo(os.path.join(golden_dir, "k27.histo"))).all()


def test_T3_long_reads(oracle, golden_dir):
    got = histo_of(oracle, os.path.join(golden_dir, "longread.fasta"), 27)
    assert (got == read_histo(os.path.join(golden_dir, "rlong.histo"))).all()


def test_T4_k_equals_readlen(oracle, golden_dir):
    s, _ = oracle.load_bank(os.path.join(golden_dir, "shortread.fasta"))
    lines = oracle.ascii_lines(oracle.count(s, 15), amin=1)
    want = open(os.path.join(golden_dir, "short.parse_results")).read().splitlines()
    assert lines == want


def test_T5_k_longer_than_read(oracle, golden_dir):
    s, _ = oracle.load_bank(os.path.join(golden_dir, "shortread.fasta"))
    r = oracle.count(s, 16)
    assert r.total == 0 and r.distinct == 0


def test_T6_reads_with_N(oracle, golden_dir):
    got = histo_of(oracle, os.path.join(golden_dir, "readN.fasta"), 20)
    assert (got == read_histo(os.path.join(golden_dir, "readN.histo"))).all()


def test_iupac_breaks_window(oracle, golden_dir):
    # test/IUPAC.fasta:3 "should be only one kmer, AAAA...AAA's"
    s, _ = oracle.load_bank(os.path.join(golden_dir, "IUPAC.fasta"))
    lines = oracle.ascii_lines(oracle.count(s, 21), amin=1)
    assert lines == ["A" * 21 + " 2"]


@pytest.mark.parametrize("k,total,distinct,solid,maxc,md5", [
    (27, 370000, 93948, 13237, 47, "d8accad74a496705691309c8dd7aa384"),
    (31, 350000, 99957, 13096, 44, "5b4da4c690bb00783eb5fdc49fc19466"),
    (63, 190000, 97702, 10945, 23, "ed2b871b9bbbdd93479ef66330c0b563"),
])
def test_known_answers(oracle, golden_dir, k, total, distinct, solid, maxc, md5):
    s, nreads = oracle.load_bank(os.path.join(golden_dir, "read50x_ref10K_e001.fasta.gz"))
    assert nreads == 5000
    r = oracle.count(s, k, threads=3)
    assert (r.total, r.distinct) == (total, distinct)
    assert int(r.ab.max()) == maxc
    lines = oracle.ascii_lines(r, amin=2)
    assert len(lines) == solid
    assert hashlib.md5(("\n".join(lines) + "\n").encode()).hexdigest() == md5


def test_actg_order_readme(oracle):
    # README.md:111-112: GTA / TAC -> canonical TAC (T < G)
    s = np.frombuffer(b"GTA", dtype=np.uint8)
    r = oracle.count(s, 3)
    assert oracle.ascii_lines(r, amin=1) == ["TAC 1"]


def test_thread_count_invariance(oracle, golden_dir):
    s, _ = oracle.load_bank(os.path.join(golden_dir, "longread.fasta"))
    a = oracle.count(s, 31, threads=1)
    b = oracle.count(s, 31, threads=7)
    assert (a.lo == b.lo).all() and (a.ab == b.ab).all() and a.total == b.total


def test_minimizer_bruteforce_small(oracle):
    s = np.frombuffer(b"ACGTTGCANACGTACGTAGCTAGCTAGCTAGGATC", dtype=np.uint8)
    mm, valid = oracle.minimizers(s, 11, 4)
    code = {65: 0, 67: 1, 84: 2, 71: 3}
    for i in range(len(s)):
        win = s[max(0, i - 10): i + 1]
        ok = len(win) == 11 and all(c in code for c in win)
        assert bool(valid[i]) == ok
        if ok:
            best = None
            for j in range(11 - 4 + 1):
                f = 0
                r = 0
                for t in range(4):
                    c = code[win[j + t]]
                    f = (f << 2) | c
                    r |= (c ^ 2) << (2 * t)
                best = min(f, r) if best is None else min(best, f, r)
            assert mm[i] == best


@pytest.mark.parametrize("k", [65, 80, 96, 97, 127, 128])
def test_oracle_256bit_keys_against_python_integers(oracle, golden_dir, k):
    """k > 64 uses 256-bit keys (C23 _BitInt, clang build).  No reference golden covers these sizes, so the
    width is anchored on an independent pure-Python count with unbounded integers (same rules: A=0 C=1 T=2
    G=3, first base most significant, canonical = min(fwd, revcomp), non-ACGT breaks the window)."""
    from collections import Counter
    assert oracle.max_kmer_size() == 128
    s, _ = oracle.load_bank(os.path.join(golden_dir, "longread.fasta"))
    s = s[:2500]
    code = {65: 0, 67: 1, 84: 2, 71: 3, 97: 0, 99: 1, 116: 2, 103: 3}
    cnt = Counter()
    run = []
    for ch in bytes(s) + b"\n":
        if ch in code:
            run.append(code[ch])
            continue
        for i in range(len(run) - k + 1):
            fw = rc = 0
            for j in range(k):
                fw = (fw << 2) | run[i + j]
                rc |= (run[i + j] ^ 2) << (2 * j)
            cnt[min(fw, rc)] += 1
        run = []
    r = oracle.count(s, k)
    exp = sorted(cnt.items())
    assert r.total == sum(cnt.values()) and r.distinct == len(exp)
    assert [int(v) for v in r.values()] == [e[0] for e in exp]
    assert [int(a) for a in r.ab] == [e[1] for e in exp]
    words, valid = oracle.enumerate_words(s, k)
    assert int(valid.sum()) == r.total
