"""dskgpu_push_raw: FASTA / FASTQ text parsed on the device (dsk_amd/csrc/rawparse.h) against the host-side rules.

What is compared: the length of the read stream the device leaves (dskgpu_raw_finish) with a plain-Python statement of the same
rules, and the COUNT of that stream -- rows, abundances, histogram -- with the oracle's count of the records parsed on the host
(the reference's BankFasta semantics: test/readN.fasta, test/longread.fasta; k-mers never span records, wrapped FASTA lines join).
"""
import gzip
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

E_FORMAT = -6


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (the HIP path has no CPU fallback)")
    return torch.device("cuda:0")


def model_stream(files):
    """The bytes rawparse.h leaves for [(text, 'fa' | 'fq'), ...] pushed as files, terminator included."""
    out = bytearray()
    for text, fmt in files:
        if out:
            out += b"\n"
        lines = text.split(b"\n")
        ends = [True] * (len(lines) - 1) + [False]            # every line but the last is followed by '\n'
        for i, (ln, nl) in enumerate(zip(lines, ends)):
            if fmt == "fq":
                if i % 4 == 1:
                    out += ln.replace(b"\r", b"") + (b"\n" if nl else b"")
            elif ln.startswith(b">"):
                out += b"\n" if nl else b""
            else:
                out += ln.translate(None, b"\r \t")
    out += b"\n"
    return bytes(out)


def host_records(text, fmt):
    """What a host parser hands on: one sequence per record (FASTA lines joined), '\n' behind each."""
    recs, lines = [], text.split(b"\n")
    if fmt == "fq":
        recs = [lines[i].replace(b"\r", b"") for i in range(1, len(lines), 4)]
    else:
        cur = None
        for ln in lines:
            if ln.startswith(b">"):
                if cur is not None:
                    recs.append(cur)
                cur = b""
            elif cur is not None:
                cur += ln.translate(None, b"\r \t")
        if cur is not None:
            recs.append(cur)
    return b"".join(r + b"\n" for r in recs)


def random_cuts(rng, n, pieces):
    c = sorted(set(int(x) for x in rng.integers(0, n + 1, pieces)))
    return [0] + c + [n]


def count_raw(files, k, cuts_rng=None, pieces=1, amin=1, **kw):
    from dsk_amd import KmerCounter
    with KmerCounter(kmer_size=k, abundance_min=amin, **kw) as kc:
        for text, fmt in files:
            f = kc.RAW_FASTQ if fmt == "fq" else kc.RAW_FASTA
            cuts = random_cuts(cuts_rng, len(text), pieces) if cuts_rng is not None else [0, len(text)]
            first = True
            for a, b in zip(cuts[:-1], cuts[1:]):
                if b > a or first:
                    kc.push_raw(text[a:b], f, new_file=first)
                    first = False
        nbytes, lines = kc.raw_finish()
        kc.count()
        rows, ab = kc.rows()
        return nbytes, lines, rows, ab, kc.histogram(), kc.stats()


def check(oracle, files, k, rng=None, pieces=1):
    nbytes, lines, rows, ab, hist, st = count_raw(files, k, rng, pieces)
    model = model_stream(files)
    assert nbytes == len(model), (nbytes, len(model))
    assert lines == sum(len(host_records(t, f).split(b"\n")) - 1 for t, f in files)          # records
    host = np.frombuffer(b"".join(host_records(t, f) for t, f in files), dtype=np.uint8)
    ref = oracle.count(host, k)
    assert oracle.count(np.frombuffer(model, dtype=np.uint8), k).total == ref.total          # the two statements of the rules agree
    lo, hi, rab = ref.solid(1)
    assert st["n_kmers"] == ref.total and st["n_distinct"] == ref.distinct
    assert (rows == ref.words()).all() and (ab == rab).all()
    assert (hist == ref.histogram(10000)).all()


def make_fastq(rng, n, lmin, lmax, crlf=False, nasty_quals=True, last_newline=True):
    eol = b"\r\n" if crlf else b"\n"
    out = []
    for i in range(n):
        L = int(rng.integers(lmin, lmax + 1))
        seq = bytes(rng.choice(np.frombuffer(b"ACGTACGTACGTacgtN", dtype=np.uint8), L))
        q = bytearray(rng.integers(33, 74, L).astype(np.uint8).tobytes())
        if nasty_quals and L:          # quality lines that LOOK like headers: lines are told apart by their number, not their first byte
            q[0] = b"@+>"[i % 3]
        out += [b"@r%d some text" % i, eol, seq, eol, b"+", b"" if i % 2 else b"r%d" % i, eol, bytes(q), eol]
    text = b"".join(out)
    return text if last_newline else text[: -len(eol)]


def make_fasta(rng, n, lmin, lmax, width=60, crlf=False, last_newline=True):
    eol = b"\r\n" if crlf else b"\n"
    out = []
    for i in range(n):
        L = int(rng.integers(lmin, lmax + 1))
        seq = bytes(rng.choice(np.frombuffer(b"ACGTACGTACGTacgtNR", dtype=np.uint8), L))
        out += [b">seq%d >with a bracket" % i, eol]
        w = width if i % 5 else 100000             # some records on one long line
        for a in range(0, L, w):
            out += [seq[a: a + w], eol]
        if i % 7 == 3:
            out += [eol]                           # an empty line inside the file
    text = b"".join(out)
    return text if last_newline else text[: -len(eol)]


@pytest.mark.parametrize("k", [21, 31])
def test_fastq_text_cut_anywhere(oracle, dev, k):
    rng = np.random.default_rng(11 + k)
    text = make_fastq(rng, 3000, 0, 400)
    check(oracle, [(text, "fq")], k)
    check(oracle, [(text, "fq")], k, rng, 40)
    check(oracle, [(make_fastq(rng, 25, 0, 400), "fq")], k, rng, 5000)      # byte-sized pieces
    check(oracle, [(make_fastq(rng, 500, 30, 300, crlf=True), "fq")], k, rng, 9)
    check(oracle, [(make_fastq(rng, 500, 30, 300, last_newline=False), "fq")], k, rng, 3)


@pytest.mark.parametrize("k", [21, 63])
def test_fasta_text_cut_anywhere(oracle, dev, k):
    rng = np.random.default_rng(5 + k)
    text = make_fasta(rng, 400, 0, 5000)
    check(oracle, [(text, "fa")], k)
    check(oracle, [(text, "fa")], k, rng, 60)
    check(oracle, [(text[:4000], "fa")], k, rng, 4000)
    check(oracle, [(make_fasta(rng, 200, 100, 3000, crlf=True), "fa")], k, rng, 7)
    check(oracle, [(make_fasta(rng, 200, 100, 3000, last_newline=False), "fa")], k, rng, 4)
    one = b">chr\n" + bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 300_000))      # one record, one line, no newline at the end
    check(oracle, [(one, "fa")], k, rng, 5)


def test_several_files_and_mixed_pushes(oracle, dev):
    rng = np.random.default_rng(3)
    a = make_fasta(rng, 50, 100, 2000, last_newline=False)       # ends inside a sequence line: the next file's records must not join it
    b = make_fastq(rng, 400, 50, 200)
    c = make_fasta(rng, 30, 100, 900)
    check(oracle, [(a, "fa"), (b, "fq"), (c, "fa")], 25, rng, 6)
    # clean reads first (dskgpu_push_reads), raw text behind them, clean reads again
    from dsk_amd import KmerCounter
    clean = host_records(c, "fa")
    with KmerCounter(kmer_size=25, abundance_min=1) as kc:
        kc.push_reads(clean)
        kc.push_raw(b, new_file=True)
        kc.push_raw(a, kc.RAW_FASTA, new_file=True)
        kc.push_reads(clean)                                  # (finishes the raw pushes by itself)
        kc.count()
        rows, ab = kc.rows()
        st = kc.stats()
    host = np.frombuffer(clean + host_records(b, "fq") + host_records(a, "fa") + clean, dtype=np.uint8)
    ref = oracle.count(host, 25)
    assert st["n_kmers"] == ref.total and (rows == ref.words()).all() and (ab == ref.solid(1)[2]).all()


def test_text_the_device_parser_does_not_handle_is_reported(oracle, dev):
    """A FASTQ file with sequences wrapped over lines, blanks inside a sequence, FASTA text declared FASTQ: DSKGPU_E_FORMAT, the
    stream is what it was before, and the host path takes over on the same context."""
    from dsk_amd import KmerCounter
    from dsk_amd.engine import DskGpuError
    rng = np.random.default_rng(9)
    good = make_fastq(rng, 200, 50, 150)
    wrapped = b"@r1\nACGTACGTAC\nGGGTTTAAAC\n+\nIIIIIIIIII\nIIIIIIIIII\n" * 50
    blanks = b"@r1\nACGT ACGTACGGGTTTAAAC\n+\nIIIIIIIIIIIIIIIIIIIII\n" * 50
    fasta = make_fasta(rng, 20, 100, 500)
    keep = host_records(good, "fq")
    ref = oracle.count(np.frombuffer(keep + keep, dtype=np.uint8), 21)
    junk = b"\n".join(b"x" * 7 for _ in range(40000)) + b"\n"          # every thread of every block sees lines that are no FASTQ
    # four lines per record, but one quality line is shorter than its sequence: a host parser (kseq's rule: as many quality characters
    # as bases) would read on into the next record -- not the same records, so not taken either
    shortq = make_fastq(rng, 20, 50, 150) + b"@s\nACGTACGTACGTACGTACGTACGTACGTAC\n+\nIIIIIIIIII\n" + make_fastq(rng, 50, 50, 150)
    # ... and two records that are off by -1 and +1 do not cancel
    offset = make_fastq(rng, 20, 50, 150) + b"@a\nACGTACGTACGTACGTACGTACGTACGTAC\n+\n" + b"I" * 29 + b"\n@b\nACGTACGTACGTACGTACGTACGTACGTAC\n+\n" + b"I" * 31 + b"\n" + make_fastq(rng, 20, 50, 150)
    noplus = make_fastq(rng, 20, 50, 150, crlf=True) + b"@c\r\nACGTACGTACGTACGTACGTACGTACGTAC\r\n\r\n" + b"I" * 30 + b"\r\n" + make_fastq(rng, 20, 50, 150, crlf=True)      # a '+' line that lost its '+'
    for bad in (wrapped, blanks, fasta, junk, shortq, offset, noplus):
        with KmerCounter(kmer_size=21, abundance_min=1) as kc:
            kc.push_reads(keep)
            kc.push_raw(good, kc.RAW_FASTQ, new_file=True)
            kc.push_raw(bad, kc.RAW_FASTQ, new_file=True)
            with pytest.raises(DskGpuError) as e:
                kc.raw_finish()
            assert e.value.code == E_FORMAT
            kc.push_reads(keep)                    # the raw pushes are gone: the stream is the first push plus this one
            kc.count()
            rows, ab = kc.rows()
            assert kc.stats()["n_kmers"] == ref.total and (rows == ref.words()).all() and (ab == ref.solid(1)[2]).all()


def test_the_reference_fixtures_as_raw_text(oracle, golden_dir, dev):
    for name, k in (("readN.fasta", 27), ("longread.fasta", 31), ("shortread.fasta", 15), ("IUPAC.fasta", 11),
                    ("c1.fasta.gz", 27), ("read50x_ref10K_e001.fasta.gz", 31)):
        path = os.path.join(golden_dir, name)
        text = gzip.open(path).read() if name.endswith(".gz") else open(path, "rb").read()
        stream, _ = oracle.load_bank(path)
        ref = oracle.count(stream, k)
        nbytes, lines, rows, ab, hist, st = count_raw([(text, "fa")], k, np.random.default_rng(1), 5)
        assert st["n_kmers"] == ref.total and (rows == ref.words()).all() and (ab == ref.solid(1)[2]).all(), name
        assert (hist == ref.histogram(10000)).all(), name


def test_raw_pushes_that_outgrow_the_stream_buffer(dev):
    """No reservation, 330 MB of FASTQ text in uneven pieces: the stream's buffer (256 MB at first) is grown while parsed text is
    already in it and kernels of earlier pieces may still be running -- same rows as the clean stream handed over in HBM."""
    from dsk_amd import KmerCounter, synth
    nr, rl = 1_600_000, 100
    reads_t = synth.make_reads(synth.make_genome(2_000_000, dev), nr, rl)
    reads = reads_t.cpu().numpy().reshape(nr, rl + 1)
    rec = np.empty((nr, 2 * rl + 7), dtype=np.uint8)
    rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
    rec[:, 3: 3 + rl] = reads[:, :rl]
    rec[:, 3 + rl: 6 + rl] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 6 + rl: 6 + 2 * rl] = ord("I")
    rec[:, 6 + 2 * rl] = ord("\n")
    text = rec.reshape(-1)
    assert text.size > 330_000_000
    with KmerCounter(kmer_size=31, abundance_min=2) as kc:
        kc.set_reads_device(reads_t.data_ptr(), reads_t.numel())
        kc.count()
        want = (kc.rows(), kc.stats()["n_kmers"], kc.histogram())
    with KmerCounter(kmer_size=31, abundance_min=2) as kc:
        cuts = [0, 100_000_003, 100_000_004, 260_000_000, 260_000_001, text.size]          # (the buffer grows at the third and at the last piece)
        for i in range(len(cuts) - 1):
            kc.push_raw(text[cuts[i]: cuts[i + 1]], kc.RAW_FASTQ, new_file=i == 0)
        nbytes, recs = kc.raw_finish()
        assert nbytes == nr * (rl + 1) + 1 and recs == nr
        kc.count()
        rows, ab = kc.rows()
        assert kc.stats()["n_kmers"] == want[1] and (rows == want[0][0]).all() and (ab == want[0][1]).all() and (kc.histogram() == want[2]).all()


def test_raw_text_beyond_the_staging_chunk(oracle, dev):
    """90 MB of FASTQ text in three pushes (pieces of 32 MB inside a push, 2048 blocks per piece): the count of the clean stream."""
    from dsk_amd import KmerCounter, synth
    nr, rl = 400_000, 100
    reads = synth.make_reads(synth.make_genome(300_000, dev), nr, rl).cpu().numpy().reshape(nr, rl + 1)
    rng = np.random.default_rng(2)
    rec = np.empty((nr, 2 * rl + 7), dtype=np.uint8)
    rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
    rec[:, 3: 3 + rl] = reads[:, :rl]
    rec[:, 3 + rl: 6 + rl] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 6 + rl: 6 + 2 * rl] = rng.integers(33, 74, (nr, rl), dtype=np.uint8)
    rec[:, 6 + rl] = ord("@")
    rec[:, 6 + 2 * rl] = ord("\n")
    text = rec.reshape(-1)
    ref = oracle.count(reads.reshape(-1), 25)
    with KmerCounter(kmer_size=25, abundance_min=2) as kc:
        cuts = [0, 40_000_001, 40_000_002, text.size]
        for i in range(3):
            kc.push_raw(text[cuts[i]: cuts[i + 1]], kc.RAW_FASTQ, new_file=i == 0)
        nbytes, lines = kc.raw_finish()
        assert nbytes == nr * (rl + 1) + 1 and lines == nr
        kc.count()
        rows, ab = kc.rows()
        st = kc.stats()
    keep = ref.ab >= 2
    assert st["n_kmers"] == ref.total and (rows[:, 0] == ref.lo[keep]).all() and (ab == ref.ab[keep]).all()
