"""Synthetic read generator of BASELINE.json / SURVEY.md §8(d).

seed 0xD5C0FFEE; genome = i.i.d. uniform ACGT of length G; each read: uniform
start, strand flip p=0.5, fixed length, per-base substitution p=0.01, one 'N'
in 1 read out of 1000.  Output = the read stream the engine consumes: every
read followed by one '\\n' (the FASTQ header / '+' / quality lines carry no
information for the count path and are dropped by the bank front-end).
Runs on whatever torch device it is given (the bench generates directly in HBM).
"""
from __future__ import annotations

import torch

SEED = 0xD5C0FFEE
_ASCII = (65, 67, 71, 84)  # A C G T (alphabetical; complement = 3 - code)


def make_genome(genome_len: int, device, seed: int = SEED) -> torch.Tensor:
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return torch.randint(0, 4, (genome_len,), generator=g, device=device, dtype=torch.uint8)


def make_reads(genome: torch.Tensor, n_reads: int, read_len: int = 150, error_rate: float = 0.01,
               n_rate: float = 0.001, seed: int = SEED + 1, chunk: int = 1 << 20) -> torch.Tensor:
    """-> uint8 tensor of n_reads * (read_len + 1) bytes on genome.device."""
    dev = genome.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    lut = torch.tensor(_ASCII, dtype=torch.uint8, device=dev)
    out = torch.empty((n_reads, read_len + 1), dtype=torch.uint8, device=dev)
    ar = torch.arange(read_len, device=dev, dtype=torch.int64)
    G = genome.numel()
    for r0 in range(0, n_reads, chunk):
        r = min(chunk, n_reads - r0)
        starts = torch.randint(0, G - read_len + 1, (r, 1), generator=g, device=dev, dtype=torch.int64)
        codes = genome[starts + ar]                                   # r x L
        flip = torch.rand((r, 1), generator=g, device=dev) < 0.5
        rc = (3 - codes).flip(1)
        codes = torch.where(flip, rc, codes)
        err = torch.rand((r, read_len), generator=g, device=dev) < error_rate
        delta = torch.randint(1, 4, (r, read_len), generator=g, device=dev, dtype=torch.uint8)
        codes = torch.where(err, (codes + delta) & 3, codes)
        ascii_ = lut[codes.long()]
        has_n = torch.rand((r,), generator=g, device=dev) < n_rate
        npos = torch.randint(0, read_len, (r,), generator=g, device=dev, dtype=torch.int64)
        rows = torch.nonzero(has_n).flatten()
        ascii_[rows, npos[rows]] = 78                                  # 'N'
        out[r0:r0 + r, :read_len] = ascii_
        out[r0:r0 + r, read_len] = 10                                  # '\n'
        del starts, codes, rc, err, delta, ascii_
    return out.view(-1)


def workload(name: str):
    """(genome_len, n_reads, read_len) of the named BASELINE.json configuration."""
    table = {
        "c2_10Mx150": (30_000_000, 10_000_000, 150),      # configs[1]: 50x of a 30 Mbp genome
        "ecoli50x": (4_640_000, 1_550_000, 150),          # E. coli stand-in for the >=10x goal
        "c3_200Mx150": (600_000_000, 200_000_000, 150),   # configs[2] (whole node)
        "c3_shard_25Mx150": (75_000_000, 25_000_000, 150),  # one GPU's share of configs[2] (200 M reads / 8 GPUs)
        "tiny": (20_000, 10_000, 150),
        "small": (1_000_000, 333_334, 150),
    }
    return table[name]
