"""ctypes wrapper of oracle/libdsk_oracle.so -- TEST INFRASTRUCTURE ONLY.

Nothing under dsk_amd/ imports this; it is the checker for the parity tests,
`__graft_entry__.smoke()` and bench.py's cpu_baseline leg.
"""
import ctypes as C

import numpy as np


class OracleResult:
    def __init__(self, k, total, lo, hi, ab, hist_fn, w2=None, w3=None):
        self.k = k
        self.total = total
        self.lo, self.hi, self.ab = lo, hi, ab
        self.w2, self.w3 = w2, w3          # words 2 and 3 of the value (k > 64)
        self._hist_fn = hist_fn

    def words(self):
        """uint64[n, ceil(k/32)]: the value of every row, least significant word first."""
        cols = [self.lo, self.hi, self.w2, self.w3][: (self.k + 31) // 32]
        return np.stack(cols, axis=1)

    @property
    def distinct(self):
        return len(self.ab)

    def solid(self, amin=2, amax=2147483647):
        m = (self.ab >= amin) & (self.ab <= amax)
        return self.lo[m], self.hi[m], self.ab[m]

    def histogram(self, histo_max=10000):
        return self._hist_fn(histo_max)

    def values(self):
        """k-mer values as Python ints (for k > 32) or uint64 array."""
        if self.k <= 32:
            return self.lo
        if self.k <= 64:
            return np.array([(int(h) << 64) | int(l) for h, l in zip(self.hi, self.lo)], dtype=object)
        return np.array([(int(d) << 192) | (int(c) << 128) | (int(h) << 64) | int(l)
                         for d, c, h, l in zip(self.w3, self.w2, self.hi, self.lo)], dtype=object)


class Oracle:
    def __init__(self, so_path):
        lib = C.CDLL(so_path)
        vp, u64, u32, p8 = C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint8)
        lib.dsko_load_bank.argtypes = [C.c_char_p, C.POINTER(p8), C.POINTER(u64), C.POINTER(u64)]
        lib.dsko_load_bank.restype = C.c_int
        lib.dsko_free_stream.argtypes = [p8]
        lib.dsko_count.argtypes = [vp, u64, C.c_int, C.c_int]
        lib.dsko_count.restype = vp
        lib.dsko_free.argtypes = [vp]
        lib.dsko_total_kmers.argtypes = [vp]
        lib.dsko_total_kmers.restype = u64
        lib.dsko_num_distinct.argtypes = [vp]
        lib.dsko_num_distinct.restype = u64
        lib.dsko_rows.argtypes = [vp, vp, vp, vp]
        lib.dsko_rows4.argtypes = [vp, vp, vp, vp, vp, vp]
        lib.dsko_max_kmer_size.restype = C.c_int
        lib.dsko_enumerate4.argtypes = [vp, u64, C.c_int, vp, vp]
        lib.dsko_enumerate4.restype = C.c_int
        lib.dsko_histogram.argtypes = [vp, vp, u32]
        lib.dsko_num_solid.argtypes = [vp, u32, u32]
        lib.dsko_num_solid.restype = u64
        lib.dsko_kmer_to_string.argtypes = [u64, u64, C.c_int, C.c_char_p]
        lib.dsko_enumerate.argtypes = [vp, u64, C.c_int, vp, vp, vp]
        lib.dsko_minimizers.argtypes = [vp, u64, C.c_int, C.c_int, vp, vp]
        self.lib = lib

    def load_bank(self, uri: str):
        """-> (stream bytes as np.uint8 array, number of reads)"""
        p = C.POINTER(C.c_uint8)()
        n, nr = C.c_uint64(), C.c_uint64()
        rc = self.lib.dsko_load_bank(uri.encode(), C.byref(p), C.byref(n), C.byref(nr))
        if rc != 0:
            raise IOError(f"oracle cannot read {uri}")
        arr = np.ctypeslib.as_array(p, shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint8)
        self.lib.dsko_free_stream(p)
        return arr, nr.value

    def count(self, stream: np.ndarray, k: int, threads: int = 4) -> OracleResult:
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        h = self.lib.dsko_count(stream.ctypes.data, len(stream), k, threads)
        if not h:
            raise ValueError("bad k")
        try:
            d = self.lib.dsko_num_distinct(h)
            total = self.lib.dsko_total_kmers(h)
            lo = np.zeros(d, np.uint64)
            hi = np.zeros(d, np.uint64)
            ab = np.zeros(d, np.uint32)
            w2 = w3 = None
            if k > 64:
                w2 = np.zeros(d, np.uint64); w3 = np.zeros(d, np.uint64)
                self.lib.dsko_rows4(h, lo.ctypes.data, hi.ctypes.data, w2.ctypes.data, w3.ctypes.data, ab.ctypes.data)
            else:
                self.lib.dsko_rows(h, lo.ctypes.data, hi.ctypes.data, ab.ctypes.data)
        finally:
            self.lib.dsko_free(h)

        def hist(histo_max):
            out = np.zeros(histo_max + 1, np.uint64)
            np.add.at(out, np.minimum(ab, histo_max), 1)
            return out

        return OracleResult(k, total, lo, hi, ab, hist, w2, w3)

    def count_only(self, stream: np.ndarray, k: int, threads: int = 4):
        """The count alone -> (total k-mers, distinct k-mers): what bench.py's cpu_baseline times (no copy of the rows to numpy)."""
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        h = self.lib.dsko_count(stream.ctypes.data, len(stream), k, threads)
        if not h:
            raise ValueError("bad k")
        try:
            return int(self.lib.dsko_total_kmers(h)), int(self.lib.dsko_num_distinct(h))
        finally:
            self.lib.dsko_free(h)

    def enumerate(self, stream: np.ndarray, k: int):
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        n = len(stream)
        lo = np.zeros(n, np.uint64)
        hi = np.zeros(n, np.uint64)
        valid = np.zeros(n, np.uint8)
        self.lib.dsko_enumerate(stream.ctypes.data, n, k, lo.ctypes.data, hi.ctypes.data, valid.ctypes.data)
        return lo, hi, valid

    def enumerate_words(self, stream: np.ndarray, k: int):
        """-> (uint64[n, 4] canonical k-mer ending at every byte, valid[n]); any k <= 128."""
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        n = len(stream)
        words = np.zeros((n, 4), np.uint64)
        valid = np.zeros(n, np.uint8)
        if self.lib.dsko_enumerate4(stream.ctypes.data, n, k, words.ctypes.data, valid.ctypes.data) != 0:
            raise RuntimeError("oracle built without 256-bit keys")
        return words, valid

    def max_kmer_size(self) -> int:
        return int(self.lib.dsko_max_kmer_size())

    def minimizers(self, stream: np.ndarray, k: int, m: int):
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        n = len(stream)
        mm = np.zeros(n, np.uint32)
        valid = np.zeros(n, np.uint8)
        self.lib.dsko_minimizers(stream.ctypes.data, n, k, m, mm.ctypes.data, valid.ctypes.data)
        return mm, valid

    def kmer_to_string(self, lo: int, hi: int, k: int) -> str:
        buf = C.create_string_buffer(k + 1)
        self.lib.dsko_kmer_to_string(int(lo), int(hi), k, buf)
        return buf.value.decode()

    def ascii_lines(self, res: OracleResult, amin=2, amax=2147483647):
        if res.k > 64:      # four-word values: letters straight from the integer (A=0 C=1 T=2 G=3, first base most significant)
            keep = (res.ab >= amin) & (res.ab <= amax)
            k = res.k
            return ["".join("ACTG"[(int(v) >> (2 * (k - 1 - i))) & 3] for i in range(k)) + f" {a}"
                    for v, a in zip(res.values()[keep], res.ab[keep])]
        lo, hi, ab = res.solid(amin, amax)
        return [f"{self.kmer_to_string(l, h, res.k)} {a}" for l, h, a in zip(lo, hi, ab)]
