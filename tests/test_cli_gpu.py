"""GPU: the reference's black-box test script (scripts/simple_test.sh:35-135),
case by case, against the real `dsk` binary (HIP engine through the C-ABI) and
`dsk2ascii`, compared with the reference's golden files byte for byte."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bins():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("needs a HIP device")
    b = os.path.join(ROOT, "dsk_amd", "host", "bin")
    if not (os.path.exists(os.path.join(b, "dsk")) and os.path.exists(os.path.join(b, "dsk2ascii"))):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "dsk_amd", "host")])
    return {"dsk": os.path.join(b, "dsk"), "dsk2ascii": os.path.join(b, "dsk2ascii")}


def test_simple_test_sh_cases_on_gpu(bins, tmp_path):
    from tests.test_host_cli import run_six_cases
    run_six_cases(bins["dsk"], bins["dsk2ascii"], str(tmp_path))


@pytest.mark.parametrize("k,md5,n", [(31, "5b4da4c690bb00783eb5fdc49fc19466", 13096), (63, "ed2b871b9bbbdd93479ef66330c0b563", 10945)])
def test_known_answer_dump_on_gpu(bins, tmp_path, k, md5, n):
    from tests.test_host_cli import known_answer_dump
    known_answer_dump(bins["dsk"], bins["dsk2ascii"], str(tmp_path), k, md5, n)


@pytest.mark.parametrize("k", [32, 64, 65, 96, 127])
def test_span_borders_and_large_k_on_gpu(bins, tmp_path, oracle, k):
    from tests.test_host_cli import test_span_borders_and_large_k
    test_span_borders_and_large_k(bins, tmp_path, oracle, k)


def test_engine_is_the_hip_library(bins, tmp_path):
    """-verbose 1 prints the info tree; it must name the gfx950 engine (no CPU path in the product binary)."""
    g = os.path.join(ROOT, "tests", "golden")
    out = subprocess.check_output([bins["dsk"], "-file", f"{g}/longread.fasta", "-kmer-size", "27", "-out", "v"], cwd=str(tmp_path)).decode()
    assert "gfx950" in out and "kmers_nb_valid" in out and "71130" in out


def test_solidity_kinds_and_histo2d_cli_on_gpu(bins, tmp_path, oracle):
    from tests.test_host_cli import run_solidity_cases
    run_solidity_cases(bins["dsk"], bins["dsk2ascii"], str(tmp_path), oracle)


@pytest.mark.parametrize("ngpus", [2, 4])
def test_nb_gpus_writes_one_storage(bins, tmp_path, ngpus):
    """`dsk -nb-gpus N` (ranks share device 0 on the 1-GPU box): ONE .h5 with a flat list of partitions and the summed
    histogram, as after the reference's single execute() (src/DSK.cpp:55-68; utils/dsk2ascii.cpp:61,77) -- reproduces
    test/k27.histo and the k = 31 known-answer md5 (lines put in k-mer order first: the partitions interleave)."""
    import hashlib
    from tests.test_host_cli import G, h5_histo, H5DUMP
    tmp = str(tmp_path)
    out = subprocess.run([bins["dsk"], "-file", f"{G}/read50x_ref10K_e001.fasta.gz", "-kmer-size", "27", "-out", "mg27", "-nb-gpus", str(ngpus), "-verbose", "1"],
                         cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert out.returncode == 0, out.stderr
    assert h5_histo("mg27.h5", tmp) == open(f"{G}/k27.histo").read()
    assert f"nb_gpus" in out.stdout.decode() and "exchange_bytes" in out.stdout.decode()
    files = ",".join(f"{G}/c{i}.fasta.gz" for i in (1, 2, 3, 4))          # the same reads as four files (simple_test.sh:52)
    subprocess.check_call([bins["dsk"], "-file", files, "-kmer-size", "31", "-abundance-min", "2", "-out", "mg31", "-nb-gpus", str(ngpus), "-nb-partitions", "2", "-verbose", "0"], cwd=tmp)
    hdr = subprocess.check_output([H5DUMP, "-n", "mg31.h5"], cwd=tmp).decode()
    for p in range(2 * ngpus):
        assert f"/dsk/solid/{p}\n" in hdr or f"/dsk/solid/{p} " in hdr or hdr.rstrip().endswith(f"/dsk/solid/{p}")
    assert f"/dsk/solid/{2 * ngpus}" not in hdr
    subprocess.check_call([bins["dsk2ascii"], "-file", "mg31.h5", "-out", "mg31.txt", "-verbose", "0"], cwd=tmp)
    lines = open(os.path.join(tmp, "mg31.txt"), "rb").read().splitlines()
    assert len(lines) == 13096
    val = lambda l: int(l.split()[0].translate(bytes.maketrans(b"ACTG", b"0123")), 4)
    lines.sort(key=val)
    assert hashlib.md5(b"\n".join(lines) + b"\n").hexdigest() == "5b4da4c690bb00783eb5fdc49fc19466"


def test_nb_gpus_with_fewer_reads_than_ranks(bins, tmp_path):
    """Ranks that get no reads at all (one short read, four ranks) and a k longer than the read: the group still writes the
    reference's answer (scripts/simple_test.sh T4 / T5 with test/short.parse_results)."""
    from tests.test_host_cli import G
    tmp = str(tmp_path)
    subprocess.check_call([bins["dsk"], "-file", f"{G}/shortread.fasta", "-kmer-size", "15", "-abundance-min", "1", "-out", "s4", "-nb-gpus", "4", "-verbose", "0"], cwd=tmp)
    subprocess.check_call([bins["dsk2ascii"], "-file", "s4", "-out", "s4.txt", "-verbose", "0"], cwd=tmp)
    assert sorted(open(os.path.join(tmp, "s4.txt")).read().splitlines()) == sorted(open(f"{G}/short.parse_results").read().splitlines())
    subprocess.check_call([bins["dsk"], "-file", f"{G}/shortread.fasta", "-kmer-size", "16", "-out", "s16", "-nb-gpus", "2", "-verbose", "0"], cwd=tmp)
    subprocess.check_call([bins["dsk2ascii"], "-file", "s16", "-out", "s16.txt", "-verbose", "0"], cwd=tmp)
    assert os.path.getsize(os.path.join(tmp, "s16.txt")) == 0


@pytest.mark.parametrize("ngpus", [2, 4])
def test_nb_gpus_per_bank_modes(bins, tmp_path, ngpus):
    """-solidity-kind / -solidity-custom / -histo2D over the comma-separated banks (CHANGELOG.md:22, README.md:98-102) with the
    k-mer space sharded over N ranks: every bank is exchanged and counted on its own with ONE repartition table, every rank
    applies the solidity kind to the k-mers it owns.  The rows (put in k-mer order: the partitions interleave), the histogram
    and the 2-D histogram must equal the one-GPU run's."""
    from tests.test_host_cli import G, h5_histo
    tmp = str(tmp_path)
    files = ",".join(f"{G}/c{i}.fasta.gz" for i in (1, 2, 3))
    val = lambda l: int(l.split()[0].translate(bytes.maketrans(b"ACTG", b"0123")), 4)
    for tag, extra in (("min", ["-solidity-kind", "min"]), ("custom", ["-solidity-kind", "custom", "-solidity-custom", "101"]), ("h2d", ["-histo2D", "1", "-histo", "1"])):
        got = {}
        for n in (1, ngpus):
            name = f"{tag}_{n}"
            r = subprocess.run([bins["dsk"], "-file", files, "-kmer-size", "27", "-abundance-min", "2", *extra, "-out", name, "-nb-gpus", str(n), "-verbose", "0"],
                               cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert r.returncode == 0, r.stderr
            subprocess.check_call([bins["dsk2ascii"], "-file", name, "-out", name + ".txt", "-verbose", "0"], cwd=tmp)
            lines = open(os.path.join(tmp, name + ".txt"), "rb").read().splitlines()
            lines.sort(key=val)
            got[n] = (lines, h5_histo(name + ".h5", tmp), open(os.path.join(tmp, name + ".histo2D")).read() if tag == "h2d" else "")
        assert len(got[1][0]) > 10 and got[1] == got[ngpus]


def test_abundance_min_auto_on_gpu(bins, tmp_path, oracle):
    """f2 through the HIP binary: `-abundance-min auto` -> cutoff / nbsolids_auto attributes and the filtered rows."""
    from tests.test_host_cli import run_abundance_min_auto
    run_abundance_min_auto(bins, str(tmp_path), oracle)


def test_nb_gpus_on_a_file_with_more_than_4_gb_per_rank(bins, tmp_path):
    """`dsk -nb-gpus 2` on a file whose halves exceed 2^32 bytes of reads each (60 M x 150 bp: 4.53 GB per
    rank -- what `-nb-gpus 2|4` meets on configs[2]'s 200 M-read file; the reference counts any file in one execute(),
    src/DSK.cpp:55-60) reproduces the 1-GPU run: same histogram, same k-mer totals, same solid rows (a high -abundance-min keeps the
    text dump small: the tail of the 50x coverage peak)."""
    import re
    import torch
    from dsk_amd import synth
    from tests.test_host_cli import h5_histo
    tmp = str(tmp_path)
    dev = torch.device("cuda:0")
    nr, rl = 60_000_000, 150
    reads = synth.make_reads(synth.make_genome(180_000_000, dev), nr, rl).view(nr, rl + 1)
    fa = os.path.join(tmp, "big.fa")
    with open(fa, "wb") as f:                         # ">r\nSEQ\n" records, written in pieces of 4 M reads
        for r0 in range(0, nr, 4_000_000):
            part = reads[r0: r0 + 4_000_000]
            rec = torch.empty((part.shape[0], 3 + rl + 1), dtype=torch.uint8, device=dev)
            rec[:, 0] = 62; rec[:, 1] = 114; rec[:, 2] = 10
            rec[:, 3:] = part
            rec.cpu().numpy().tofile(f)
            del rec
    del reads
    torch.cuda.empty_cache()
    assert os.path.getsize(fa) == nr * (3 + rl + 1)
    val = lambda l: int(l.split()[0].translate(bytes.maketrans(b"ACTG", b"0123")), 4)
    got = {}
    for n in (1, 2):
        name = f"big_{n}"
        r = subprocess.run([bins["dsk"], "-file", fa, "-kmer-size", "31", "-abundance-min", "48", "-out", name, "-nb-gpus", str(n), "-verbose", "1"],
                           cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        info = r.stdout.decode()
        figures = tuple(int(re.search(key + r"\s*:\s*(\d+)", info).group(1)) for key in ("kmers_nb_valid", "kmers_nb_distinct", "kmers_nb_solid"))
        subprocess.check_call([bins["dsk2ascii"], "-file", name, "-out", name + ".txt", "-verbose", "0"], cwd=tmp)
        lines = open(os.path.join(tmp, name + ".txt"), "rb").read().splitlines()
        lines.sort(key=val)
        got[n] = (figures, h5_histo(name + ".h5", tmp), lines)
        os.remove(os.path.join(tmp, name + ".h5"))
    os.remove(fa)
    assert got[1][0][0] > 7_000_000_000 and 1000 < got[1][0][2] == len(got[1][2]) < 20_000_000
    assert got[1] == got[2]


def test_messy_fastq_at_gb_scale(bins, tmp_path):
    """VERDICT r04 'missing' 6: the parser paths no full-size run had touched -- 0.8 GB of FASTQ with what real sequencer output has and
    the synthetic workloads lack (tests/test_host_cli.py::make_messy_inputs: read lengths 36..251, long headers with blanks, quality
    lines that begin with '@', '>' or '+', CRLF records, lower case, runs of N) as a plain file (32 parser threads on record-aligned
    ranges), as multi-member gzip and as two-line FASTA.  Every run of the dsk binary must report the k-mer totals and the
    histogram of the engine counting the clean read stream directly."""
    import torch
    from dsk_amd import KmerCounter
    from tests.test_host_cli import make_messy_inputs, run_messy_case
    tmp = str(tmp_path)
    n_reads = 2_500_000
    clean = make_messy_inputs(tmp, n_reads, 6_000_000)
    assert os.path.getsize(os.path.join(tmp, "messy.fastq")) > 800_000_000
    t = torch.from_numpy(clean.copy()).to(torch.device("cuda:0"))
    with KmerCounter(kmer_size=31, abundance_min=3) as kc:
        kc.set_reads_device(t.data_ptr(), t.numel())
        kc.count()
        st, hist = kc.stats(), kc.histogram()
    del t
    want = (st["n_kmers"], st["n_distinct"], st["n_solid"], hist)
    run_messy_case(bins["dsk"], tmp, n_reads, want)
    run_messy_case(bins["dsk"], tmp, n_reads, want, {"DSK_DEVICE_PARSE": "1"})       # the same files, parsed on the device


def test_device_parse_through_the_binary(bins, tmp_path, oracle, monkeypatch):
    """-device-parse 1 / DSK_DEVICE_PARSE=1: no host parser -- the text of every file goes to the GPU as it is (dskgpu_push_raw).  The
    reference's six black-box cases against its goldens, the messy files (plain, multi-member gzip through the parallel inflate,
    two-line FASTA) against the oracle, and a FASTQ file with wrapped sequences, which the device parser gives back
    (DSKGPU_E_FORMAT) and the host parser then takes -- same totals either way."""
    import re
    import numpy as np
    from tests.test_host_cli import make_messy_inputs, run_messy_case, run_six_cases, run_solidity_cases
    monkeypatch.setenv("DSK_DEVICE_PARSE", "1")
    tmp = str(tmp_path)
    run_six_cases(bins["dsk"], bins["dsk2ascii"], tmp)
    run_solidity_cases(bins["dsk"], bins["dsk2ascii"], tmp, oracle)          # per-bank modes: every bank's text parsed on the device, bank ends where the text ends
    n_reads = 25_000
    clean = make_messy_inputs(tmp, n_reads, 60_000)
    ref = oracle.count(clean, 31)
    want = (ref.total, ref.distinct, int((ref.ab >= 3).sum()), ref.histogram(10000))
    run_messy_case(bins["dsk"], tmp, n_reads, want, {"DSK_DEVICE_PARSE": "1"})
    run_messy_case(bins["dsk"], tmp, n_reads, want, {"DSK_DEVICE_PARSE": "1", "DSK_PGZIP_CHUNK_BYTES": "65536"})       # the gzip file through the parallel inflate
    # a gzip file cut short is an error on this path too (one zlib stream and the parallel inflate)
    import gzip
    z = gzip.compress(open(os.path.join(tmp, "messy.fa"), "rb").read(), 6)
    open(os.path.join(tmp, "cut.fa.gz"), "wb").write(z[: len(z) // 2])
    for env in ({}, {"DSK_PGZIP_CHUNK_BYTES": "65536"}):
        r = subprocess.run([bins["dsk"], "-file", "cut.fa.gz", "-kmer-size", "31", "-out", "c", "-device-parse", "1"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, **env))
        assert r.returncode == 1 and b"EXCEPTION" in r.stderr, r.stderr[-300:]
    # wrapped FASTQ: given back, parsed on the host
    rng = np.random.default_rng(4)
    seqs = [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 120)) for _ in range(2000)]
    with open(os.path.join(tmp, "wrapped.fastq"), "wb") as f:
        for i, s in enumerate(seqs):
            f.write(b"@r%d\n" % i + s[:60] + b"\n" + s[60:] + b"\n+\n" + b"I" * 60 + b"\n" + b"I" * 60 + b"\n")
    ref = oracle.count(np.frombuffer(b"\n".join(seqs) + b"\n", dtype=np.uint8), 31)
    for args in (["-device-parse", "1"], []):
        r = subprocess.run([bins["dsk"], "-file", "wrapped.fastq", "-kmer-size", "31", "-abundance-min", "1", "-out", "w", "-verbose", "1"] + args,
                           cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-1500:]
        info = r.stdout.decode()
        got = tuple(int(re.search(key + r"\s*:\s*(\d+)", info).group(1)) for key in ("nb_sequences", "kmers_nb_valid", "kmers_nb_distinct", "banks_parsed_on_device"))
        assert got == (2000, ref.total, ref.distinct, 0), got


def test_damaged_files_on_gpu_are_read_as_one_thread_reads_them(bins, tmp_path, oracle):
    """The parallel parser's give-away (IBank::stream's `exact`) against the real engine: the chunks of a damaged file are dropped from
    the device's read stream (dskgpu_rewind_reads) and the file is pushed again from one thread -- on one GPU and on a group of two."""
    import re
    import numpy as np
    from tests.test_host_cli import test_damaged_files_are_read_as_one_thread_reads_them
    from tests.raw_text_model import base_text, damage, host_parser
    test_damaged_files_are_read_as_one_thread_reads_them(bins, tmp_path, oracle, n_seeds=40, min_reparsed=1)
    tmp = str(tmp_path)
    done = 0
    for seed in range(400):
        rng = np.random.default_rng(seed)
        fmt = "fq" if rng.random() < 0.6 else "fa"
        text = damage(rng, base_text(rng, fmt), fmt)
        if text[:1] not in (b"@", b">"):
            continue
        open(os.path.join(tmp, "x.txt"), "wb").write(text)
        r = subprocess.run([bins["dsk"], "-file", "x.txt", "-kmer-size", "21", "-abundance-min", "1", "-out", "o", "-verbose", "1", "-nb-cores", "4", "-nb-gpus", "2"],
                           cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, DSK_PARSE_MIN_BYTES="1"))
        assert r.returncode == 0, (seed, r.stderr[-300:])
        info = r.stdout.decode()
        if "banks_parsed_again_by_one_thread" not in info:
            continue
        want = oracle.count(np.frombuffer(host_parser(text) + b"\n", dtype=np.uint8).copy(), 21).total
        assert int(re.search(r"kmers_nb_valid\s*:\s*(\d+)", info).group(1)) == want, (seed, fmt)
        done += 1
        if done == 2:
            break
    assert done == 2


@pytest.mark.parametrize("extra", [(), ("-nb-gpus", "2"), ("-nb-gpus", "4"), ("-device-parse", "1")])
def test_empty_and_tiny_banks_on_gpu(bins, tmp_path, oracle, extra):
    from tests.test_host_cli import run_empty_bank_cases
    run_empty_bank_cases(bins["dsk"], bins["dsk2ascii"], str(tmp_path), oracle, extra)


def test_bench_line_contract():
    """`python bench.py` prints ONE JSON line with the fields the driver reads (metric / value / unit / n_gpus / steps / warmup /
    ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload) plus the `roofline` and `cpu_baseline`
    objects -- checked here on the small workload so that a change to bench.py cannot break the contract unseen.  (The numbers of
    such a small input mean nothing; the default run is the measurement.)"""
    import json
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "small", "--steps", "2", "--warmup", "1", "--no-e2e", "--no-k63",
                        "--no-human-standin", "--no-repeat-rich", "--no-place-compare", "--cpu-sample-reads", "50000"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "u64" and d["data"].startswith("synthetic") and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0 and abs(d["value"] - d["n_distinct"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == d["unit"]
