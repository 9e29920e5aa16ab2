"""Size-independent checks of a finished count whose rows stay in HBM (3 * 10^9 of them on the 30x human stand-in): shared by
tests/test_gpu_parity.py and tools/human_standin.py.  Test infrastructure, not product code."""
import ctypes

import numpy as np
import torch


def valid_windows(reads, nr, rl, k):
    """number of full ACGT windows of a synthetic read stream (<= 1 'N' per read), closed form on the device"""
    r = reads.view(nr, rl + 1)[:, :rl]
    n_valid = 0
    for r0 in range(0, nr, 8_000_000):
        bad = r[r0:r0 + 8_000_000] == 78
        has = bad.any(1)
        q = bad.to(torch.uint8).argmax(1).to(torch.int64)
        full = rl - k + 1
        with_n = torch.clamp(q - k + 1, min=0) + torch.clamp(rl - q - k, min=0)
        n_valid += int(torch.where(has, with_n, torch.full_like(with_n, full)).sum())
    return n_valid


def device_invariants(kc, st, hist, k, reads, nr, rl, dev, amin=2, n_valid=None, partition_order=False):
    """sum(i * hist[i]) == n_kmers (nothing saturates), sum(hist) == n_distinct, sum(hist[amin:]) == n_solid == rows, rows strictly
    ascending (partition_order: inside every output partition, the partitions' sizes adding up to the rows), histogram of the rows'
    abundances == hist tail, n_kmers == number of full ACGT windows (closed form: <= 1 'N' per read)."""
    h = hist.astype(np.int64)
    idx = np.arange(len(h), dtype=np.int64)
    sat = int(h[-1])
    assert int(h.sum()) == st["n_distinct"], "sum(hist) != n_distinct"
    assert int(h[amin:].sum()) == st["n_solid"], "hist tail != n_solid"
    kp, ap, n = kc.result_device()
    assert n == st["n_solid"]
    hip = ctypes.CDLL("libamdhip64.so")
    step = 1 << 27
    bins = torch.zeros(len(h), dtype=torch.int64, device=dev)
    kbuf = torch.empty(step, dtype=torch.int64, device=dev); abuf = torch.empty(step, dtype=torch.int32, device=dev)
    last = None; ab_sum = 0
    starts = None; n_partitions = None
    if partition_order:          # first row of every partition but the first: the only places where a row may be smaller than its predecessor
        sizes = kc.partition_sizes()
        assert int(sizes.sum()) == n, "partition sizes do not add up to the rows"
        n_partitions = int(len(sizes))
        starts = torch.from_numpy(np.cumsum(sizes)[:-1].astype(np.int64)).to(dev)
    for r0 in range(0, n, step):
        m = min(step, n - r0)
        hip.hipMemcpy(ctypes.c_void_p(kbuf.data_ptr()), ctypes.c_void_p(kp + r0 * 8), ctypes.c_size_t(m * 8), 3)
        hip.hipMemcpy(ctypes.c_void_p(abuf.data_ptr()), ctypes.c_void_p(ap + r0 * 4), ctypes.c_size_t(m * 4), 3)
        kk = kbuf[:m]
        asc = kk[1:] > kk[:-1]                                                         # (k <= 31: values < 2^62, signed compare is safe)
        if starts is not None:
            inside = torch.ones(m - 1, dtype=torch.bool, device=dev) if m > 1 else torch.ones(0, dtype=torch.bool, device=dev)
            loc = starts[(starts > r0) & (starts < r0 + m)] - r0 - 1               # asc[i] compares rows i and i + 1: a partition starting at row s excuses asc[s - 1]
            inside[loc] = False
            assert bool(asc[inside].all()), "a partition is not strictly ascending"
            if last is not None and not bool(((starts == r0).any())):
                assert int(kk[0]) > last
        else:
            assert bool(asc.all()), "rows not strictly ascending"
            if last is not None:
                assert int(kk[0]) > last
        last = int(kk[-1])
        a = abuf[:m].to(torch.int64)
        ab_sum += int(a.sum())
        bins += torch.bincount(torch.clamp(a, max=len(h) - 1), minlength=len(h))
    assert (bins.cpu().numpy()[amin:] == h[amin:]).all(), "histogram of the rows != hist tail"
    # k-mer occurrences: rows carry the true abundance even where the histogram saturates at its last row
    assert ab_sum + int((h[:amin] * idx[:amin]).sum()) == st["n_kmers"], "sum of abundances != n_kmers"
    if not sat:
        assert int((h * idx).sum()) == st["n_kmers"]
    if n_valid is None:          # (n_valid given: the reads themselves are gone -- released after dskgpu_encode_reads)
        n_valid = valid_windows(reads, nr, rl, k)
    assert n_valid == st["n_kmers"], (n_valid, st["n_kmers"])
    out = {"rows_checked": int(n), "saturated_histogram_rows": sat}
    if n_partitions is not None:
        out["partitions_checked"] = n_partitions
    return out


