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


def make_genome_repeats(genome_len: int, device, seed: int = SEED, family_frac: float = 0.01, unit_len: int = 300,
                        divergence: float = 0.02, tandem_unit: int = 37, tandem_copies: int = 2000, tandem_loci: int = 4) -> torch.Tensor:
    """A repeat-rich genome (what the uniform one of SURVEY.md section 8(d) lacks, and every real genome has):
    * a high-copy interspersed family: `family_frac` of the genome is copies of ONE `unit_len`-bp element, every copy with
      `divergence` substitutions per base, dropped at random positions (an Alu-like family: its k-mers occur in hundreds of copies);
    * `tandem_loci` tandem arrays of `tandem_copies` x a `tandem_unit`-bp unit (satellite-like: `tandem_unit` distinct k-mers
      per locus, each `tandem_copies` times in the genome);
    the rest i.i.d. uniform.  Seeded, runs on `device`."""
    g = torch.Generator(device=device)
    g.manual_seed(seed ^ 0x5EED)
    genome = make_genome(genome_len, device, seed)
    n_copies = int(genome_len * family_frac) // unit_len
    if n_copies:
        unit = torch.randint(0, 4, (unit_len,), generator=g, device=device, dtype=torch.uint8)
        copies = unit.repeat(n_copies, 1)
        mut = torch.rand((n_copies, unit_len), generator=g, device=device) < divergence
        delta = torch.randint(1, 4, (n_copies, unit_len), generator=g, device=device, dtype=torch.uint8)
        copies = torch.where(mut, (copies + delta) & 3, copies)
        # the copies sit in distinct slots of unit_len bases (no two overlap: the scatter below is deterministic)
        slots = torch.randperm(genome_len // unit_len, generator=g, device=device)[:n_copies].to(torch.int64).reshape(-1, 1)
        genome[(slots * unit_len + torch.arange(unit_len, device=device)).reshape(-1)] = copies.reshape(-1)
    for locus in range(tandem_loci):
        span = tandem_unit * tandem_copies
        if span * (tandem_loci + 1) >= genome_len:
            break
        unit = torch.randint(0, 4, (tandem_unit,), generator=g, device=device, dtype=torch.uint8)
        at = (locus + 1) * (genome_len // (tandem_loci + 1))
        genome[at: at + span] = unit.repeat(tandem_copies)
    return genome


def make_reads(genome: torch.Tensor, n_reads: int, read_len: int = 150, error_rate: float = 0.01,
               n_rate: float = 0.001, seed: int = SEED + 1, chunk: int = 1 << 20, polya_rate: float = 0.0) -> torch.Tensor:
    """-> uint8 tensor of n_reads * (read_len + 1) bytes on genome.device.
    polya_rate: fraction of the reads that are poly-A (before strand flip and errors): ONE k-mer with millions of occurrences."""
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
        if polya_rate > 0.0:
            pa = torch.rand((r, 1), generator=g, device=dev) < polya_rate
            codes = torch.where(pa, torch.zeros_like(codes), codes)
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


REPEAT_WORKLOADS = {
    # name: (plain workload it mirrors, poly-A read fraction)
    "c2_repeats_10Mx150": ("c2_10Mx150", 0.002),
    "small_repeats": ("small", 0.002),
    "c2_repeats_nopolya_10Mx150": ("c2_10Mx150", 0.0),      # (experiments: the repeat-rich genome without the poly-A reads)
    # stand-in for BASELINE.json configs[4] (30x human short reads, ~90 Gbp; SURVEY.md section 8(d) allows "synthetic G = 3 Gbp,
    # 600 M reads"): the repeat-rich 3 Gbp genome, 600 M x 150 bp, 0.2 % poly-A reads -- ONE k-mer with 1.4 * 10^8 occurrences,
    # family k-mers with ~10^6.  The reference's own human run: doc/human_log:3-4,20-24 (7 passes, 2.7 * 10^9 solid k-mers).
    "c5_human30x": ("c5_600Mx150", 0.002),
    "c5_human30x_shard": ("c5_shard_75Mx150", 0.002),        # one GPU's share of it on 8 GPUs (N x 375 Mbp of genome)
}


def make_workload(name: str, device, world: int = 1, rank: int = 0):
    """The read stream of a named workload on `device` -> (reads, genome_len, n_reads, read_len).  `*_repeats*` workloads use
    make_genome_repeats and a fraction of poly-A reads; the others are the uniform genome of SURVEY.md section 8(d)."""
    if name in REPEAT_WORKLOADS:
        base, polya = REPEAT_WORKLOADS[name]
        gl, nr, rl = workload(base)
        genome = make_genome_repeats(gl * world, device)
        reads = make_reads(genome, nr, rl, seed=SEED + 1 + rank, polya_rate=polya)
    else:
        gl, nr, rl = workload(name)
        genome = make_genome(gl * world, device)
        reads = make_reads(genome, nr, rl, seed=SEED + 1 + rank)
    del genome
    return reads, gl, nr, rl


def workload(name: str):
    """(genome_len, n_reads, read_len) of the named BASELINE.json configuration."""
    table = {
        "c2_10Mx150": (30_000_000, 10_000_000, 150),      # configs[1]: 50x of a 30 Mbp genome
        "ecoli50x": (4_640_000, 1_550_000, 150),          # E. coli stand-in for the >=10x goal
        "c3_200Mx150": (600_000_000, 200_000_000, 150),   # configs[2] (whole node)
        "c3_shard_25Mx150": (75_000_000, 25_000_000, 150),  # one GPU's share of configs[2] (200 M reads / 8 GPUs)
        "c5_600Mx150": (3_000_000_000, 600_000_000, 150),   # configs[4] stand-in (uniform twin; the repeat-rich one: c5_human30x)
        "c5_shard_75Mx150": (375_000_000, 75_000_000, 150), # one GPU's share of it (600 M reads / 8 GPUs)
        "tiny": (20_000, 10_000, 150),
        "small": (1_000_000, 333_334, 150),
    }
    return table[name]
