"""Host-layer tests (CPU): the six cases of scripts/simple_test.sh:35-135 run
through this repo's `dsk` tool wrapper + `dsk2ascii` + stock h5dump, with the
CPU oracle plugged in as counting backend (tests/host/dsk_cpu_check.cpp -- test
infrastructure).  They pin bank parsing, option handling, the HDF5 layout and
the dsk2ascii text format against the reference's golden files.  The same six
commands run against the real GPU `dsk` binary in tests/test_cli_gpu.py.
"""
import hashlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
H5DUMP = "/opt/conda/bin/h5dump"


@pytest.fixture(scope="module")
def bins():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "dsk_amd", "host"), "bin/dsk2ascii", "bin/libdskhost.a"],
                          stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "host")], stdout=subprocess.DEVNULL)
    if os.environ.get("DSK_TEST_ASAN"):      # the same suite on the AddressSanitizer + UBSan build of the host layer and the oracle (make -C tests/host asan)
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "host"), "asan"], stdout=subprocess.DEVNULL)
        return {"dsk": os.path.join(ROOT, "tests", "host", "asan", "dsk_cpu_check"),
                "dsk2ascii": os.path.join(ROOT, "tests", "host", "asan", "dsk2ascii")}
    return {"dsk": os.path.join(ROOT, "tests", "host", "dsk_cpu_check"),
            "dsk2ascii": os.path.join(ROOT, "dsk_amd", "host", "bin", "dsk2ascii")}


def h5_histo(h5, cwd):
    """scripts/simple_test.sh:37 with stock h5dump standing in for gatb-h5dump."""
    cmd = f"{H5DUMP} -y -d histogram/histogram {h5} | grep '^\\ *[0-9]' | tr -d ' ' | tr -d ',' | paste - -"
    return subprocess.check_output(cmd, shell=True, cwd=cwd).decode()


def make_messy_inputs(tmp, n_reads, genome_len, seed=20251003):
    """What real sequencer output has and the synthetic workloads lack, as three files in `tmp` -- messy.fastq, messy.fastq.gz (8
    concatenated gzip members: what `cat a.gz b.gz` leaves) and messy.fa (every record on two sequence lines): read lengths from 36 to
    251, long headers with blanks, quality lines that begin with '@', '>' or '+', CRLF records among LF ones, lower-case reads, runs of N.
    -> the clean read stream (np.uint8: the sequences separated by one newline)."""
    import gzip
    import numpy as np
    rng = np.random.default_rng(seed)
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=genome_len).tobytes()
    lower = genome.lower()
    qual_pool = bytes(rng.integers(33, 127, size=1 << 16, dtype=np.uint8))
    hdr_pool = [("@SRR%07d.%d %d:N:0:ATCACG+TTAGGC length=%d  " % (rng.integers(1e6), i, i, 150)).encode() + b"x" * int(rng.integers(0, 90)) for i in range(997)]
    starts = rng.integers(0, len(genome) - 260, size=n_reads)
    lens = rng.integers(36, 252, size=n_reads)
    kinds = rng.integers(0, 100, size=n_reads)
    qoff = rng.integers(0, len(qual_pool) - 260, size=n_reads)
    fq, fa, seqs = [], [], []
    for i in range(n_reads):
        s, ln, kd = int(starts[i]), int(lens[i]), int(kinds[i])
        seq = (lower if kd < 7 else genome)[s: s + ln]                     # 7 % lower-case reads
        if kd in (7, 8):                                                    # 2 %: a run of N inside
            seq = seq[: ln // 3] + b"N" * (1 + kd) + seq[ln // 3 + 1 + kd:]
        q = qual_pool[int(qoff[i]): int(qoff[i]) + ln]
        if kd in (9, 10, 11):                                               # 3 %: quality line starting with a record marker
            q = (b"@", b">", b"+")[kd - 9] + q[1:]
        eol = b"\r\n" if kd >= 90 else b"\n"                              # 10 % CRLF records
        fq.append(hdr_pool[i % 997] + eol + seq + eol + b"+" + eol + q + eol)
        fa.append(b">" + hdr_pool[i % 997][1:] + eol + seq[: ln // 2] + eol + seq[ln // 2:] + eol)
        seqs.append(seq)
    clean = np.frombuffer(b"\n".join(seqs) + b"\n", dtype=np.uint8)
    with open(os.path.join(tmp, "messy.fastq"), "wb") as f:
        f.write(b"".join(fq))
    with open(os.path.join(tmp, "messy.fastq.gz"), "wb") as f:
        per = (n_reads + 7) // 8
        for m in range(8):
            f.write(gzip.compress(b"".join(fq[m * per: (m + 1) * per]), compresslevel=1))
    with open(os.path.join(tmp, "messy.fa"), "wb") as f:
        f.write(b"".join(fa))
    return clean


def run_messy_case(dsk, tmp, n_reads, want, extra_env=None):
    """dsk on the three messy files: sequences, k-mer totals and histogram must be `want` = (n_kmers, n_distinct, n_solid, histogram)."""
    import re
    want_histo = "".join(f"{i}\t{int(want[3][i])}\n" for i in range(1, 10001))
    for name in ("messy.fastq", "messy.fastq.gz", "messy.fa"):
        r = subprocess.run([dsk, "-file", name, "-kmer-size", "31", "-abundance-min", "3", "-out", "m", "-verbose", "1"], cwd=tmp,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **(extra_env or {})))
        assert r.returncode == 0, (name, r.stderr.decode()[-1500:])
        info = r.stdout.decode()
        got = tuple(int(re.search(key + r"\s*:\s*(\d+)", info).group(1)) for key in ("nb_sequences", "kmers_nb_valid", "kmers_nb_distinct", "kmers_nb_solid"))
        assert got == (n_reads, want[0], want[1], want[2]), (name, got, want[:3])
        if (extra_env or {}).get("DSK_DEVICE_PARSE"):          # the text went to the GPU as it is and was parsed there (dskgpu_push_raw)
            assert re.search(r"banks_parsed_on_device\s*:\s*1", info), (name, info[-1500:])
        assert h5_histo("m.h5", tmp) == want_histo, name
        os.remove(os.path.join(tmp, "m.h5"))


def test_messy_fastq_fasta_and_multi_member_gzip(bins, tmp_path, oracle):
    """The parser on sequencer-like input (see make_messy_inputs), serial and with the file cut into record-aligned ranges for several
    threads (a quality line that starts with '@' must not be taken for a record start), against the oracle on the clean stream."""
    tmp = str(tmp_path)
    n_reads = 25_000
    clean = make_messy_inputs(tmp, n_reads, 60_000)
    ref = oracle.count(clean, 31)
    want = (ref.total, ref.distinct, int((ref.ab >= 3).sum()), ref.histogram(10000))
    run_messy_case(bins["dsk"], tmp, n_reads, want)
    run_messy_case(bins["dsk"], tmp, n_reads, want, {"DSK_PARSE_MIN_BYTES": "1", "DSK_CHUNK_MB": "1"})


def run_six_cases(dsk, dsk2ascii, tmp):
    def run(*args, **kw):
        return subprocess.run(list(args), cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)
    # T1 single gz
    r = run(dsk, "-file", f"{G}/read50x_ref10K_e001.fasta.gz", "-kmer-size", "27", "-out", "test_dsk27", "-max-memory", "200", "-verbose", "0")
    assert r.returncode == 0, r.stderr
    assert h5_histo("test_dsk27.h5", tmp) == open(f"{G}/k27.histo").read()
    # T2 multiple gz
    files = ",".join(f"{G}/c{i}.fasta.gz" for i in (1, 2, 3, 4))
    r = run(dsk, "-file", files, "-kmer-size", "27", "-out", "test_dsk27", "-max-memory", "200", "-verbose", "0")
    assert r.returncode == 0, r.stderr
    assert h5_histo("test_dsk27.h5", tmp) == open(f"{G}/k27.histo").read()
    # T3 long reads
    r = run(dsk, "-file", f"{G}/longread.fasta", "-kmer-size", "27", "-out", "test_long", "-verbose", "0", "-max-memory", "200")
    assert r.returncode == 0, r.stderr
    assert h5_histo("test_long.h5", tmp) == open(f"{G}/rlong.histo").read()
    # T4 k = readlen; dsk2ascii accepts the name without .h5 (simple_test.sh:89)
    r = run(dsk, "-file", f"{G}/shortread.fasta", "-kmer-size", "15", "-abundance-min", "1", "-out", "test_short", "-verbose", "0", "-max-memory", "200")
    assert r.returncode == 0, r.stderr
    r = run(dsk2ascii, "-file", "test_short", "-out", "test_short.parse_results", "-verbose", "0")
    assert r.returncode == 0, r.stdout
    assert open(os.path.join(tmp, "test_short.parse_results")).read() == open(f"{G}/short.parse_results").read()
    # T5 k = readlen + 1: no k-mer, no hang, empty dump
    r = run(dsk, "-file", f"{G}/shortread.fasta", "-kmer-size", "16", "-out", "test_short16", "-max-memory", "200")
    assert r.returncode == 0, r.stderr
    r = run(dsk2ascii, "-file", "test_short16.h5", "-out", "test_short16.parse_results", "-verbose", "0")
    assert r.returncode == 0 and os.path.getsize(os.path.join(tmp, "test_short16.parse_results")) == 0
    # T6 reads with N
    r = run(dsk, "-file", f"{G}/readN.fasta", "-kmer-size", "20", "-out", "test_N", "-verbose", "0", "-max-memory", "200")
    assert r.returncode == 0, r.stderr
    assert h5_histo("test_N.h5", tmp) == open(f"{G}/readN.histo").read()


def known_answer_dump(dsk, dsk2ascii, tmp, k, md5, nlines):
    r = subprocess.run([dsk, "-file", f"{G}/read50x_ref10K_e001.fasta.gz", "-kmer-size", str(k), "-abundance-min", "2",
                        "-out", f"ka{k}", "-verbose", "0"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([dsk2ascii, "-file", f"ka{k}.h5", "-out", f"ka{k}.txt", "-verbose", "0"], cwd=tmp)
    assert r.returncode == 0
    txt = open(os.path.join(tmp, f"ka{k}.txt"), "rb").read()
    assert txt.count(b"\n") == nlines
    assert hashlib.md5(txt).hexdigest() == md5     # rows globally ascending => same md5 as SURVEY.md App. A


def test_simple_test_sh_cases(bins, tmp_path):
    run_six_cases(bins["dsk"], bins["dsk2ascii"], str(tmp_path))


@pytest.mark.parametrize("k,md5,n", [(31, "5b4da4c690bb00783eb5fdc49fc19466", 13096),
                                     (63, "ed2b871b9bbbdd93479ef66330c0b563", 10945)])
def test_known_answer_dumps(bins, tmp_path, k, md5, n):
    known_answer_dump(bins["dsk"], bins["dsk2ascii"], str(tmp_path), k, md5, n)


@pytest.mark.parametrize("k", [32, 64, 65, 95, 96, 127])
def test_span_borders_and_large_k(bins, tmp_path, oracle, k):
    """Spans 32/64/96/128 (README.md:115-122): a span serves k < span, so k = 32, 64, 96 use the next one up
    (one more, all-zero, word per row); k up to 127 uses four-word values.  Not pinned by any reference golden:
    checked against the oracle, whose 256-bit path is itself checked against Python integers (test_oracle_golden)."""
    tmp = str(tmp_path)
    subprocess.check_call([bins["dsk"], "-file", f"{G}/longread.fasta", "-kmer-size", str(k), "-abundance-min", "1", "-out", "big", "-verbose", "0"], cwd=tmp)
    subprocess.check_call([bins["dsk2ascii"], "-file", "big", "-out", "big.txt", "-verbose", "0"], cwd=tmp)
    s, _ = oracle.load_bank(f"{G}/longread.fasta")
    ref = oracle.count(s, k)
    assert open(os.path.join(tmp, "big.txt")).read().splitlines() == oracle.ascii_lines(ref, amin=1)
    hdr = subprocess.check_output([H5DUMP, "-H", "big.h5"], cwd=tmp).decode()
    words = k // 32 + 1
    assert ('H5T_STD_U64LE "value"' in hdr) if words == 1 else (f"H5T_ARRAY {{ [{words}] H5T_STD_U64LE }}" in hdr)


def test_big_partitions_written_in_place(bins, tmp_path, oracle):
    """Partitions of >= 16 K rows are not handed to H5Dwrite: their dataset is allocated at once (contiguous) and several threads
    write the rows they build straight into the file.  The file must read back -- through this repo's reader (dsk2ascii) AND
    through stock h5dump -- exactly as the rows of the oracle; a compressed output (-out-compress) keeps the library path."""
    import numpy as np
    fa = f"{G}/read50x_ref10K_e001.fasta.gz"
    stream, _ = oracle.load_bank(fa)
    ref = oracle.count(stream, 27)
    want = "".join(f"{''.join('ACTG'[(int(v) >> (2 * (26 - i))) & 3] for i in range(27))} {int(a)}\n" for v, a in zip(ref.lo, ref.ab))
    for extra in ((), ("-out-compress", "3")):
        r = subprocess.run([bins["dsk"], "-file", fa, "-kmer-size", "27", "-abundance-min", "1", "-out", "big", "-verbose", "0", *extra],
                           cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr
        r = subprocess.run([bins["dsk2ascii"], "-file", "big.h5", "-out", "big.txt", "-verbose", "0"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stdout
        got = open(os.path.join(tmp_path, "big.txt")).read()
        assert sorted(got.splitlines()) == sorted(want.splitlines()) and len(got.splitlines()) == 93948
        layout = subprocess.check_output([H5DUMP, "-p", "-H", "-d", "dsk/solid/0", "big.h5"], cwd=tmp_path).decode()
        assert ("CONTIGUOUS" in layout) == (not extra) and "93948" not in layout      # 4 partitions of ~23 K rows
        first = subprocess.check_output(f"{H5DUMP} -d dsk/solid/0 -s 0 -c 1 big.h5 | grep -c abundance", shell=True, cwd=tmp_path).decode()
        assert int(first) >= 1


def test_layout_and_attributes(bins, tmp_path):
    tmp = str(tmp_path)
    subprocess.check_call([bins["dsk"], "-file", f"{G}/longread.fasta", "-kmer-size", "31", "-out", "lay", "-verbose", "0",
                           "-nb-partitions", "3", "-histo", "1"], cwd=tmp)
    hdr = subprocess.check_output([H5DUMP, "-H", "lay.h5"], cwd=tmp).decode()
    for needle in ('GROUP "dsk"', 'ATTRIBUTE "kmer_size"', 'ATTRIBUTE "xml"', 'GROUP "solid"', 'ATTRIBUTE "nb_partitions"',
                   'DATASET "0"', 'DATASET "2"', 'H5T_STD_U64LE "value"', 'H5T_STD_I32LE "abundance"',
                   'GROUP "histogram"', 'DATASET "histogram"', 'H5T_STD_U16LE "index"', 'H5T_STD_U64LE "abundance"', "( 10000 )"):
        assert needle in hdr, needle
    assert 'DATASET "3"' not in hdr
    attr = subprocess.check_output([H5DUMP, "-a", "/dsk/kmer_size", "lay.h5"], cwd=tmp).decode()
    assert '"31"' in attr
    histo = open(os.path.join(tmp, "lay.histo")).read().splitlines()
    assert len(histo) == 10000 and histo[0].split("\t")[0] == "1"
    assert h5_histo("lay.h5", tmp).splitlines() == histo


def test_fastq_multiline_album_and_default_out(bins, tmp_path, oracle):
    import gzip
    tmp = str(tmp_path)
    seqs = ["ACGTACGTACGTTTGACCA", "GGGTTTAAACCCNACGTACGTAGCTAGCTAGCAT", "TTTTTTTTTTTTTTTTTTTTTTTT"]
    with open(os.path.join(tmp, "a.fastq"), "w") as f:
        for i, s in enumerate(seqs):
            f.write(f"@r{i} desc\n{s}\n+\n{'I' * len(s)}\n")
    with gzip.open(os.path.join(tmp, "b.fa.gz"), "wt") as f:     # multi-line FASTA, CRLF, lower case
        f.write(">x\r\nacgtacgtacgt\r\nttgacca\r\n>y\nGGGTTTAAACCC\nNACGTACGTAG\n")
    with open(os.path.join(tmp, "album.txt"), "w") as f:
        f.write("a.fastq\nb.fa.gz\n")
    subprocess.check_call([bins["dsk"], "-file", "album.txt", "-kmer-size", "7", "-abundance-min", "1", "-verbose", "0"], cwd=tmp)
    assert os.path.exists(os.path.join(tmp, "album.h5"))          # default -out = input basename without extension
    subprocess.check_call([bins["dsk2ascii"], "-file", "album.h5", "-out", "album.out", "-verbose", "0"], cwd=tmp)
    got = open(os.path.join(tmp, "album.out")).read().splitlines()
    import numpy as np
    stream = ("\n".join(seqs) + "\n" + "acgtacgtacgtttgacca\nGGGTTTAAACCCNACGTACGTAG\n").encode()
    want = oracle.ascii_lines(oracle.count(np.frombuffer(stream, dtype=np.uint8), 7), amin=1)
    assert got == want
    # comma list of the same files gives the same dump
    subprocess.check_call([bins["dsk"], "-file", "a.fastq,b.fa.gz", "-kmer-size", "7", "-abundance-min", "1", "-out", "cl", "-verbose", "0"], cwd=tmp)
    subprocess.check_call([bins["dsk2ascii"], "-file", "cl", "-out", "cl.out", "-verbose", "0"], cwd=tmp)
    assert open(os.path.join(tmp, "cl.out")).read().splitlines() == want


def test_error_paths(bins, tmp_path):
    tmp = str(tmp_path)
    r = subprocess.run([bins["dsk"], "-file", "/nonexistent.fa"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and r.stderr.startswith(b"EXCEPTION: ")           # src/main.cpp:42-46
    r = subprocess.run([bins["dsk"]], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"-file" in r.stdout and b"mandatory" in r.stdout   # src/main.cpp:37-40
    r = subprocess.run([bins["dsk"], "-file", f"{G}/shortread.fasta", "-bogus", "1"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"Unknown parameter" in r.stdout
    r = subprocess.run([bins["dsk"], "-file", f"{G}/shortread.fasta", "-kmer-size", "128"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"EXCEPTION" in r.stderr and b"k must be < 128" in r.stderr   # KSIZE_LIST 32 64 96 128 (CMakeLists.txt:42)
    r = subprocess.run([bins["dsk2ascii"], "-file", "nope", "-out", "x"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and r.stdout.startswith(b"EXCEPTION: ")           # utils/dsk2ascii.cpp:129-133 (stdout)
    r = subprocess.run([bins["dsk"], "-help"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and b"-kmer-size" in r.stdout
    # a gzip file cut short (an interrupted download) is an error, not a smaller input: small (one zlib stream, serial and with the
    # inflate thread) and large enough for the parallel inflate (its chunks end where the file does)
    import gzip
    import numpy as np
    rng = np.random.default_rng(8)
    recs = b"".join(b">s%d\n" % i + bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 200)) + b"\n" for i in range(30000))
    z = gzip.compress(recs, 6)
    for cut, env in ((len(z) // 2, {}), (len(z) // 2, {"DSK_PGZIP_CHUNK_BYTES": "65536"}), (len(z) - 5, {}), (len(z) - 5, {"DSK_PGZIP_CHUNK_BYTES": "65536"})):
        open(os.path.join(tmp, "cut.fa.gz"), "wb").write(z[:cut])
        for cores in ("1", "4"):
            r = subprocess.run([bins["dsk"], "-file", "cut.fa.gz", "-kmer-size", "21", "-out", "c", "-nb-cores", cores], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               env=dict(os.environ, **env))
            assert r.returncode == 1 and b"EXCEPTION" in r.stderr, (cut, env, cores, r.stderr[-300:])
    open(os.path.join(tmp, "whole.fa.gz"), "wb").write(z)
    r = subprocess.run([bins["dsk"], "-file", "whole.fa.gz", "-kmer-size", "21", "-out", "c", "-verbose", "0"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr


def test_dsk2ascii_variants(bins, tmp_path):
    tmp = str(tmp_path)
    subprocess.check_call([bins["dsk"], "-file", f"{G}/shortread.fasta", "-kmer-size", "15", "-abundance-min", "1", "-out", "s", "-verbose", "0"], cwd=tmp)
    out = subprocess.check_output([bins["dsk2ascii"], "-file", "s", "-out", "unused", "-c", "-verbose", "0"], cwd=tmp).decode()
    assert out == "ACTGTACGTATAAGA 1\n"
    subprocess.check_call([bins["dsk2ascii"], "-file", "s", "-out", "s.fa", "-fasta", "-verbose", "0"], cwd=tmp)
    assert open(os.path.join(tmp, "s.fa")).read() == ">\nACTGTACGTATAAGA\n"
    subprocess.check_call([bins["dsk2ascii"], "-file", "s", "-out", "s.fq", "-fastq", "-verbose", "0"], cwd=tmp)
    assert open(os.path.join(tmp, "s.fq")).read() == "@\nACTGTACGTATAAGA\n+\n" + "-" * 15 + "\n"


def test_abundance_min_auto(bins, tmp_path, oracle):
    run_abundance_min_auto(bins, str(tmp_path), oracle)


def run_abundance_min_auto(bins, tmp, oracle):
    subprocess.check_call([bins["dsk"], "-file", f"{G}/read50x_ref10K_e001.fasta.gz", "-kmer-size", "27", "-abundance-min", "auto",
                           "-out", "auto", "-verbose", "0"], cwd=tmp)
    cutoff = subprocess.check_output([H5DUMP, "-a", "/histogram/cutoff", "auto.h5"], cwd=tmp).decode()
    s, _ = oracle.load_bank(f"{G}/read50x_ref10K_e001.fasta.gz")
    h = oracle.count(s, 27).histogram(10000)
    i = 1
    while h[i + 1] < h[i]:
        i += 1
    assert f'"{i}"' in cutoff
    nbs = subprocess.check_output([H5DUMP, "-a", "/histogram/nbsolids_auto", "auto.h5"], cwd=tmp).decode()
    assert f'"{int(h[i:].sum())}"' in nbs
    subprocess.check_call([bins["dsk2ascii"], "-file", "auto", "-out", "auto.txt", "-verbose", "0"], cwd=tmp)
    rows = open(os.path.join(tmp, "auto.txt")).read().splitlines()
    assert len(rows) == int(h[i:].sum())
    assert rows == oracle.ascii_lines(oracle.count(s, 27), amin=i)      # exactly the k-mers at or above the cutoff


def test_parallel_parser_equals_serial(bins, tmp_path):
    """Large uncompressed files are memory-mapped and parsed by several threads on record-aligned
    ranges (bank.cpp); quality lines starting with '@' or '+', multi-line FASTA and CRLF must not
    confuse the range cutter.  Same dump as the single-thread parse."""
    import hashlib
    import random
    random.seed(5)
    tmp = str(tmp_path)

    def rnd(n):
        return "".join(random.choice("ACGT") for _ in range(n))
    with open(os.path.join(tmp, "big.fastq"), "w") as f:
        for i in range(20000):
            s = rnd(random.randint(30, 160))
            q = "".join(random.choice("@+I#5") for _ in s)
            f.write(f"@r{i}\n{s}\n+\n{q}\n")
    with open(os.path.join(tmp, "big.fa"), "w") as f:
        for i in range(8000):
            s = rnd(random.randint(50, 400))
            f.write(f">s{i} x\r\n")
            for j in range(0, len(s), 70):
                f.write(s[j:j + 70] + "\r\n")
    env = dict(os.environ, DSK_PARSE_MIN_BYTES="1")
    for fn in ("big.fastq", "big.fa"):
        md5 = []
        for cores in ("1", "7"):
            subprocess.check_call([bins["dsk"], "-file", fn, "-kmer-size", "21", "-abundance-min", "1", "-out", f"o{cores}",
                                   "-verbose", "0", "-nb-cores", cores], cwd=tmp, env=env)
            subprocess.check_call([bins["dsk2ascii"], "-file", f"o{cores}", "-out", f"o{cores}.txt", "-verbose", "0"], cwd=tmp)
            md5.append(hashlib.md5(open(os.path.join(tmp, f"o{cores}.txt"), "rb").read()).hexdigest())
        assert md5[0] == md5[1], fn


def test_engine_failure_in_bank_threads_is_reported(bins, tmp_path):
    """An exception thrown by the counting engine while the bank's worker threads feed it (e.g. HBM exhausted by
    dskgpu_push_reads) must surface as `EXCEPTION: <msg>` + exit code 1 (src/main.cpp:42-46), not abort the process:
    memory-mapped multi-thread parse, BGZF slabs, the gzip pipeline and the multi-file thread pool."""
    import gzip
    import random
    random.seed(11)
    tmp = str(tmp_path)
    recs = "".join(f"@r{i}\n{''.join(random.choice('ACGT') for _ in range(100))}\n+\n{'I' * 100}\n" for i in range(30000))
    open(os.path.join(tmp, "f.fastq"), "w").write(recs)
    with gzip.open(os.path.join(tmp, "f.fastq.gz"), "wt") as f:
        f.write(recs)
    with open(os.path.join(tmp, "b.fastq.gz"), "wb") as f:               # BGZF: independent gzip members with a BC extra field
        import struct
        import zlib
        data = recs.encode()
        for off in range(0, len(data), 60000):
            blk = data[off:off + 60000]
            c = zlib.compressobj(6, zlib.DEFLATED, -15); body = c.compress(blk) + c.flush()
            f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(body) + 25) + body + struct.pack("<II", zlib.crc32(blk), len(blk)))
        f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    env = dict(os.environ, DSK_PARSE_MIN_BYTES="1", DSK_TEST_FAIL_AFTER_BYTES="500000")
    for files in ("f.fastq", "f.fastq.gz", "b.fastq.gz", "f.fastq,f.fastq.gz,b.fastq.gz"):
        r = subprocess.run([bins["dsk"], "-file", files, "-kmer-size", "21", "-out", "x", "-verbose", "0", "-nb-cores", "6"], cwd=tmp, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 1, (files, r.returncode, r.stderr[-300:])
        assert b"EXCEPTION: test backend: out of memory" in r.stderr, (files, r.stderr[-300:])
    ok = subprocess.run([bins["dsk"], "-file", "b.fastq.gz", "-kmer-size", "21", "-out", "y", "-verbose", "0"], cwd=tmp, env=dict(os.environ, DSK_PARSE_MIN_BYTES="1"))
    assert ok.returncode == 0


def test_kmer_model_unit(bins):
    """C++ unit test of host/kmer.hpp (Kmer<span>::ModelCanonical, Integer::apply) against the oracle."""
    out = subprocess.run([os.path.join(ROOT, "tests", "host", "test_kmer")], stdout=subprocess.PIPE).stdout.decode()
    assert "ALL OK" in out, out


def test_key_mixers_unit(bins):
    """C++ unit test of the device header's key mixers compiled for the host (tests/host/test_mixer.cpp): kmix / kmixN are bijections,
    and the mixed top word of a two-word key is a hash of the whole key -- the pairs that exposed the two earlier folds are pinned."""
    exe = os.path.join(ROOT, "tests", "host", "test_mixer")
    if not os.path.exists(exe):
        pytest.skip("no hipcc on this machine")
    out = subprocess.run([exe], stdout=subprocess.PIPE).stdout.decode()
    assert "ALL OK" in out, out


def test_boundary_names_compile_and_run(bins, tmp_path):
    """A caller written with exactly the names SURVEY.md section 8(b) lists (Group::getPartition<Count>, Tool::createIterator,
    LOCAL, Integer::apply<Functor,Parameter>, StorageFactory, OptionFailure::displayErrors ...) builds against the host layer
    and round-trips rows through the HDF5 storage (tests/host/test_boundary_names.cpp)."""
    out = subprocess.run([os.path.join(ROOT, "tests", "host", "test_boundary_names"), "probe"], cwd=str(tmp_path), stdout=subprocess.PIPE).stdout.decode()
    assert "ALL OK" in out, out


def run_solidity_cases(dsk, dsk2ascii, tmp, oracle):
    """-solidity-kind / -solidity-custom / -histo2D through the CLI (shared by the CPU and GPU variants)."""
    import numpy as np
    files = [f"{G}/c{i}.fasta.gz" for i in (1, 2, 3)]
    streams = [oracle.load_bank(f)[0] for f in files]
    k, amin = 27, 2
    per = [oracle.count(s, k) for s in streams]
    keys = np.unique(np.concatenate([p.lo for p in per]))
    counts = np.zeros((len(keys), 3), dtype=np.int64)
    for b, p in enumerate(per):
        counts[np.searchsorted(keys, p.lo), b] = p.ab
    tot = counts.sum(1)
    expect = {
        "sum": tot >= amin, "min": counts.min(1) >= amin, "max": counts.max(1) >= amin,
        "one": (counts >= amin).any(1), "all": (counts >= amin).all(1),
        "custom": (counts[:, 0] >= amin) & (counts[:, 2] >= amin) & (counts[:, 1] == 0),
    }
    for kind, solid in expect.items():
        args = [dsk, "-file", ",".join(files), "-kmer-size", str(k), "-abundance-min", str(amin), "-solidity-kind", kind,
                "-out", f"sol_{kind}", "-verbose", "0", "-histo2D", "1"]
        if kind == "custom":
            args += ["-solidity-custom", "101"]
        r = subprocess.run(args, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr
        subprocess.check_call([dsk2ascii, "-file", f"sol_{kind}", "-out", f"sol_{kind}.txt", "-verbose", "0"], cwd=tmp)
        got = open(os.path.join(tmp, f"sol_{kind}.txt")).read().splitlines()
        want = [f"{oracle.kmer_to_string(int(v), 0, k)} {int(c)}" for v, c in zip(keys[solid], tot[solid])]
        assert got == want, kind
        h2 = np.loadtxt(os.path.join(tmp, f"sol_{kind}.histo2D"), dtype=np.int64)
        assert h2.shape == (10001, 12) and (h2[:, 0] == np.arange(10001)).all()
        ref = np.zeros((10001, 11), dtype=np.int64)
        np.add.at(ref, (np.minimum(tot - counts[:, 0], 10000), np.minimum(counts[:, 0], 10)), 1)
        assert (h2[:, 1:] == ref).all(), kind
    r = subprocess.run([dsk, "-file", files[0], "-kmer-size", "27", "-solidity-kind", "bogus", "-out", "x"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"solidity-kind" in r.stderr
    r = subprocess.run([dsk, "-file", files[0], "-kmer-size", "27", "-histo2D", "1", "-out", "x", "-verbose", "0"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"histo2D" in r.stderr          # needs genome + reads


def test_solidity_kinds_and_histo2d_cli(bins, tmp_path, oracle):
    run_solidity_cases(bins["dsk"], bins["dsk2ascii"], str(tmp_path), oracle)


def _fnv1a(data):
    h = 1469598103934665603
    for c in data:
        h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_bank_hands_on_its_text(bins, tmp_path):
    """IBank::streamRaw (what `dsk -device-parse 1` feeds to dskgpu_push_raw): the file's text as it is -- plain, gzip through zlib,
    gzip through the parallel inflate, several members, a comma list -- from its first record on, the first piece of every file
    flagged; and NOTHING for banks that keep the host parser (an album inside a list, BGZF, text that does not start like FASTA / FASTQ)."""
    import gzip
    exe = os.path.join(ROOT, "tests", "host", "test_stream_raw")
    tmp = str(tmp_path)

    def run(uri, env=None):
        return subprocess.run([exe, uri], cwd=tmp, stdout=subprocess.PIPE, env=dict(os.environ, **(env or {}))).stdout.decode().split()

    fa = open(f"{G}/longread.fasta", "rb").read()
    assert run(f"{G}/longread.fasta") == ["RAW", "1", "1", str(len(fa)), "%016x" % _fnv1a(fa), "1"]
    gz = gzip.open(f"{G}/c1.fasta.gz").read()
    assert run(f"{G}/c1.fasta.gz") == ["RAW", "1", "1", str(len(gz)), "%016x" % _fnv1a(gz), "1"]                      # zlib (a small file)
    out = run(f"{G}/c1.fasta.gz,{G}/longread.fasta")
    assert out[:2] == ["RAW", "1"] and int(out[3]) == len(gz) + len(fa) and out[4] == "%016x" % _fnv1a(gz + fa) and out[5] == "1"
    # FASTQ, blank lines in front (skipped), 3 MB: plain, one gzip member through the parallel inflate (small chunks), two members
    import numpy as np
    rng = np.random.default_rng(5)
    recs = b"".join(b"@r%d\n" % i + bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 100)) + b"\n+\n" + b"I" * 100 + b"\n" for i in range(15000))
    open(os.path.join(tmp, "a.fastq"), "wb").write(b"\n\n" + recs)
    want = ["RAW", "2", "1", str(len(recs)), "%016x" % _fnv1a(recs), "1"]
    assert run("a.fastq") == want
    with gzip.open(os.path.join(tmp, "a.fastq.gz"), "wb", compresslevel=6) as f:
        f.write(recs)
    got = run("a.fastq.gz", {"DSK_PGZIP_CHUNK_BYTES": "65536"})
    assert got[:2] == want[:2] and got[3:] == want[3:], got
    with open(os.path.join(tmp, "two.fastq.gz"), "wb") as f:
        f.write(gzip.compress(recs[: len(recs) // 2 // 212 * 212], 1) + gzip.compress(recs[len(recs) // 2 // 212 * 212:], 1))
    got = run("two.fastq.gz", {"DSK_PGZIP_CHUNK_BYTES": "65536"})
    assert got[:2] == want[:2] and got[3:] == want[3:], got
    # banks that do not offer their text
    open(os.path.join(tmp, "album.txt"), "w").write(f"{G}/c1.fasta.gz\n{G}/c2.fasta.gz\n")
    c2 = gzip.open(f"{G}/c2.fasta.gz").read()
    out = run("album.txt")                      # an album's files are banks of their own: each hands on its text
    assert out[:3] == ["RAW", "1", "2"] and int(out[3]) == len(gz) + len(c2) and out[4] == "%016x" % _fnv1a(gz + c2) and out[5] == "1"
    assert run(f"album.txt,{G}/c3.fasta.gz") == ["NO"]          # an album INSIDE a list is one bank of several files: host parser
    write_bgzf(os.path.join(tmp, "b.fastq.gz"), recs)
    assert run("b.fastq.gz") == ["NO"]
    open(os.path.join(tmp, "junk.fa"), "wb").write(b"no header here\nACGT\n")
    assert run("junk.fa") == ["NO"]


def test_damaged_files_are_read_as_one_thread_reads_them(bins, tmp_path, oracle, n_seeds=160, min_reparsed=5):
    """FASTA / FASTQ text with lines deleted, doubled, split, joined, bytes inserted and removed, records of the other format spliced
    in (tests/raw_text_model.py): whatever the thread count -- the parallel parser cuts a file at record starts it recognises by
    their looks -- `dsk` counts what ONE thread reads (the reference's parser is serial: gatb-core BankFasta behind src/DSK.cpp:51), which
    is what the restatement of the parser's state machine reads.  A range that does not end between two records gives the parallel
    parse away: its chunks are dropped from the engine and the file is parsed again by one thread."""
    import re
    import numpy as np
    from tests.raw_text_model import base_text, damage, host_parser
    tmp = str(tmp_path)
    reparsed = 0
    for seed in range(n_seeds):
        rng = np.random.default_rng(seed)
        fmt = "fq" if rng.random() < 0.6 else "fa"
        text = damage(rng, base_text(rng, fmt), fmt)
        if text[:1] not in (b"@", b">"):
            continue
        want = oracle.count(np.frombuffer(host_parser(text) + b"\n", dtype=np.uint8).copy(), 21).total
        open(os.path.join(tmp, "x.txt"), "wb").write(text)
        for cores in ("1", "4"):
            r = subprocess.run([bins["dsk"], "-file", "x.txt", "-kmer-size", "21", "-abundance-min", "1", "-out", "o", "-verbose", "1", "-nb-cores", cores],
                               cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, DSK_PARSE_MIN_BYTES="1"))
            assert r.returncode == 0, (seed, r.stderr[-300:])
            info = r.stdout.decode()
            assert int(re.search(r"kmers_nb_valid\s*:\s*(\d+)", info).group(1)) == want, (seed, fmt, cores)
            reparsed += "banks_parsed_again_by_one_thread" in info
    assert reparsed >= min_reparsed          # (the damage does reach the case this test is about)


def test_parallel_inflate_block_types(bins, tmp_path):
    """host/pgzip.cpp against zlib (tests/host/test_pgzip) on streams that hold what an ordinary FASTQ .gz rarely does: short blocks
    coded with the FIXED Huffman codes that carry back-references (a sync flush, then a short tail: r06 -- the fixed distance code was
    built from 30 lengths, refused as incomplete, and such a block decoded its distances with the previous block's table: a valid
    file was reported as corrupt by the CRC check), empty stored blocks (the flush markers), stored blocks of random bytes."""
    import zlib
    import numpy as np
    exe = os.path.join(ROOT, "tests", "host", "test_pgzip")
    rng = np.random.default_rng(12)
    recs = [b"@r%d\n" % i + bytes(rng.choice(np.frombuffer(b"ACGTacgtN", dtype=np.uint8), 90)) + b"\n+\n" + bytes(rng.integers(33, 74, 90, dtype=np.uint8)) + b"\n" for i in range(4000)]
    body = b"".join(recs)
    for case in range(6):
        c = zlib.compressobj(6, zlib.DEFLATED, 31)
        z = c.compress(body[: 500_000 + 1111 * case]) + c.flush(zlib.Z_SYNC_FLUSH)
        if case % 2:
            z += c.compress(rng.integers(0, 256, 30_000, dtype=np.uint8).tobytes()) + c.flush(zlib.Z_SYNC_FLUSH)      # (incompressible: stored blocks)
        z += c.compress(b"".join(recs[:3])[: 150 + 40 * case]) + c.flush()                                          # a short tail that repeats earlier text: one fixed block with matches
        path = os.path.join(str(tmp_path), "f%d.gz" % case)
        open(path, "wb").write(z)
        for chunk in ("16384", "65536"):
            out = subprocess.run([exe, path, "4", chunk], stdout=subprocess.PIPE).stdout.decode()
            assert out.startswith("OK "), (case, chunk, out)
    # the file on which the bug was found (tools: differential checks of damaged text through gzip, seed 26): 171 KB of text whose
    # deflate stream ends in a fixed block with matches that the parallel path decodes itself
    import gzip
    from tests.raw_text_model import base_text, damage
    rng = np.random.default_rng(26)
    fmt = "fq" if rng.random() < 0.7 else "fa"
    parts = []
    for _ in range(40):
        t = damage(rng, base_text(rng, fmt), fmt) if rng.random() < 0.15 else b"".join(base_text(rng, fmt))
        parts.append(t if t.endswith(b"\n") else t + b"\n")
    path = os.path.join(str(tmp_path), "s26.gz")
    open(path, "wb").write(gzip.compress(b"".join(parts), 6, mtime=0))
    for chunk in ("8192", "16384", "32768"):
        for threads in ("2", "4", "8"):
            out = subprocess.run([exe, path, threads, chunk], stdout=subprocess.PIPE).stdout.decode()
            assert out.startswith("OK 171199 "), (threads, chunk, out)


def run_empty_bank_cases(dsk, dsk2ascii, tmp, oracle, extra=()):
    """An input file without reads is still a bank: with -solidity-kind min no k-mer is solid (its count there is 0), with max / sum
    the other banks decide; the same with the empty file first, and with a bank of ONE read (a group of 4 ranks hands most ranks
    nothing of it: until r06 those ranks counted one bank less than the others and the job hung)."""
    import numpy as np
    rng = np.random.default_rng(21)
    seqs = [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 120)) for _ in range(60)]
    open(os.path.join(tmp, "a.fa"), "wb").write(b"".join(b">s%d\n" % i + s + b"\n" for i, s in enumerate(seqs * 2)))
    open(os.path.join(tmp, "empty.fa"), "wb").write(b"")
    open(os.path.join(tmp, "one.fa"), "wb").write(b">only\n" + seqs[0] + b"\n")
    k = 25
    ra = oracle.count(np.frombuffer(b"\n".join(seqs * 2) + b"\n", dtype=np.uint8).copy(), k)
    r1 = oracle.count(np.frombuffer(seqs[0] + b"\n", dtype=np.uint8).copy(), k)

    def rows(uri, kind):
        r = subprocess.run([dsk, "-file", uri, "-kmer-size", str(k), "-abundance-min", "1", "-solidity-kind", kind, "-out", "e", "-verbose", "0", *extra],
                           cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert r.returncode == 0, (uri, kind, r.stderr[-300:])
        subprocess.check_call([dsk2ascii, "-file", "e", "-out", "e.txt", "-verbose", "0"], cwd=tmp)
        return sorted(open(os.path.join(tmp, "e.txt")).read().splitlines())          # (a group writes its ranks' partitions one after the other: ascending inside each)

    all_a = sorted(f"{oracle.kmer_to_string(int(v), 0, k)} {int(c)}" for v, c in zip(ra.lo, ra.ab))
    for uri in ("a.fa,empty.fa", "empty.fa,a.fa"):
        assert rows(uri, "min") == []
        assert rows(uri, "max") == all_a
        assert rows(uri, "sum") == all_a
    in_one = set(int(v) for v in r1.lo)
    want = sorted(f"{oracle.kmer_to_string(int(v), 0, k)} {int(c) + 1}" for v, c in zip(ra.lo, ra.ab) if int(v) in in_one)
    assert rows("a.fa,one.fa", "min") == want            # (min >= 1 in both banks: the k-mers of the one read; printed with the sum)
    assert rows("one.fa,a.fa", "all") == want


def test_empty_and_tiny_banks(bins, tmp_path, oracle):
    run_empty_bank_cases(bins["dsk"], bins["dsk2ascii"], str(tmp_path), oracle)


def test_parallel_inflate_on_random_valid_files(bins):
    """A short run of tools/fuzz_pgzip.py (valid gzip files of many shapes through the parallel inflate against zlib)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_pgzip.py"), "5", "60"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and b"bad 0" in r.stdout, r.stdout[-600:] + r.stderr[-300:]


def write_bgzf(path, data, block=60000):
    """BGZF writer (htslib's blocked gzip): independent <= 64 KB members with the 'BC' size field + the empty EOF member."""
    import struct, zlib
    with open(path, "wb") as f:
        for off in list(range(0, len(data), block)) + [None]:
            chunk = b"" if off is None else data[off: off + block]
            c = zlib.compressobj(6, zlib.DEFLATED, -15)
            body = c.compress(chunk) + c.flush()
            bsize = 12 + 6 + len(body) + 8 - 1
            f.write(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize))
            f.write(body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))


@pytest.mark.parametrize("fmt", ["fastq", "fasta"])
def test_gzip_and_bgzf_inputs(bins, tmp_path, fmt):
    """gzip input: ordinary .gz goes through the pipelined inflate, BGZF through the parallel block inflate
    (several slabs with a carried-over partial record when DSK_BGZF_SLAB_BYTES is small); same dump as the plain file."""
    import gzip
    import numpy as np
    tmp = str(tmp_path)
    rng = np.random.default_rng(3)
    recs = []
    for i in range(1200):
        n = int(rng.choice([30, 150, 151, 400, 5000])) if i % 60 else 90000   # some reads longer than a BGZF block
        a = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n)
        a[rng.random(n) < 0.002] = ord("N")
        seq = a.tobytes().decode()
        if fmt == "fastq":
            recs.append(f"@r{i}\n{seq}\n+\n{'@' * n}\n")                    # quality lines starting with '@' on purpose
        else:
            recs.append(f">r{i}\n" + "\n".join(seq[j: j + 70] for j in range(0, n, 70)) + "\n")
    data = "".join(recs).encode()
    assert len(data) > (2 << 20)
    open(os.path.join(tmp, "plain." + fmt), "wb").write(data)
    with gzip.open(os.path.join(tmp, "std.gz"), "wb", compresslevel=1) as f:
        f.write(data)
    write_bgzf(os.path.join(tmp, "blocked.gz"), data)
    # one ordinary gzip member at several levels: inflated in parallel from the middle of the deflate stream (host/pgzip.cpp; chunks of
    # 60 KB here so that a 1 MB file is many chunks in several slabs) -- and a file of two members, which that path declines
    for lvl in (6, 9):
        with gzip.open(os.path.join(tmp, f"std{lvl}.gz"), "wb", compresslevel=lvl) as f:
            f.write(data)
    with open(os.path.join(tmp, "two.gz"), "wb") as f:
        f.write(gzip.compress(data[: len(data) // 2], 6) if data[len(data) // 2 - 1: len(data) // 2] == b"\n" else gzip.compress(data, 6))
        f.write(gzip.compress(data[len(data) // 2:], 6) if data[len(data) // 2 - 1: len(data) // 2] == b"\n" else gzip.compress(b"", 6))
    pgz = {"DSK_PGZIP_CHUNK_BYTES": "60000", "DSK_PGZIP_TRACE": "1"}
    dumps = {}
    for name, src, env in (("plain", "plain." + fmt, {"DSK_PARSE_MIN_BYTES": "100000"}), ("std", "std.gz", {}), ("bgzf", "blocked.gz", {}),
                           ("bgzf_slabs", "blocked.gz", {"DSK_BGZF_SLAB_BYTES": "300000"}),
                           ("pgz1", "std.gz", pgz), ("pgz6", "std6.gz", pgz), ("pgz9", "std9.gz", pgz), ("pgz_two_members", "two.gz", pgz),
                           ("no_pgz", "std6.gz", {"DSK_NO_PGZIP": "1"}),
                           ("serial", "plain." + fmt, {"DSK_PARSE_MIN_BYTES": str(1 << 40)})):
        r = subprocess.run([bins["dsk"], "-file", src, "-kmer-size", "25", "-abundance-min", "1", "-out", name, "-verbose", "0"],
                           cwd=tmp, env=dict(os.environ, **env), stderr=subprocess.PIPE)
        assert r.returncode == 0, (name, r.stderr.decode()[-800:])
        if name in ("pgz1", "pgz6", "pgz9"):                 # the parallel path really ran: every slab's chunks all passed
            slabs = [l for l in r.stderr.decode().splitlines() if "[pgzip] slab" in l]
            assert len(slabs) >= 2 and all("0 good" not in l for l in slabs), r.stderr.decode()[-800:]
        subprocess.check_call([bins["dsk2ascii"], "-file", name, "-out", name + ".txt", "-verbose", "0"], cwd=tmp)
        dumps[name] = hashlib.md5(open(os.path.join(tmp, name + ".txt"), "rb").read()).hexdigest()
    assert len(set(dumps.values())) == 1, dumps
