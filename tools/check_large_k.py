"""Enumerate + count on the device for k around the key-width borders (63..128), against the oracle."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsk_amd import KmerCounter
from tests.oracle_py import Oracle
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
oracle = Oracle(os.path.join(root, "oracle", "libdsk_oracle.so"))
s, _ = oracle.load_bank(os.path.join(root, "tests", "golden", "read50x_ref10K_e001.fasta.gz"))
dev = torch.device("cuda", 0)
t = torch.from_numpy(s.copy()).to(dev)
for k in [65, 80, 96, 97, 100, 127, 128, 63, 31]:
    wo = (k + 31) // 32
    # enumerate
    kk = torch.zeros(len(s) * wo, dtype=torch.int64, device=dev); val = torch.zeros(len(s), dtype=torch.uint8, device=dev)
    with KmerCounter(kmer_size=k, abundance_min=1) as kc:
        kc.k_enumerate(t.data_ptr(), len(s), kk.data_ptr(), val.data_ptr())
        torch.cuda.synchronize()
        if k > 64:
            rw, rv = oracle.enumerate_words(s, k)
            got = kk.cpu().numpy().view(np.uint64).reshape(len(s), wo)
            ok_e = bool((val.cpu().numpy() == rv).all() and (got == rw[:, :wo]).all())
        else:
            ok_e = None
        kc.set_reads_device(t.data_ptr(), len(s))
        kc.count()
        rows, ab = kc.rows()
        ref = oracle.count(s, k)
        st = kc.stats()
        rwords = ref.words()
        order = np.lexsort([rows[:, x] for x in range(wo)])
        ok = rows.shape == rwords.shape and bool((rows == rwords).all()) and bool((ab == ref.ab).all())
        okh = bool((kc.histogram() == ref.histogram(10000)).all())
        print(k, "enumerate", ok_e, "rows", ok, "hist", okh, "kmers", st["n_kmers"], ref.total, "distinct", st["n_distinct"], ref.distinct, flush=True)
