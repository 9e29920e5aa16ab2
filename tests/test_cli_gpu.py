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
