"""ctypes binding of include/dskgpu.h.

`KmerCounter` mirrors the one call the reference makes on this path,
`SortingCountAlgorithm<span>(bank, props).execute()` (src/DSK.cpp:55-60), and
the read-back done by dsk2ascii (utils/dsk2ascii.cpp:61-104): rows of
(kmer value, abundance) per "solid" partition plus the abundance histogram.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class DskGpuError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"dskgpu error {code}: {msg}")
        self.code = code


class _Config(C.Structure):
    _fields_ = [
        ("kmer_size", C.c_uint32),
        ("abundance_min", C.c_uint32),
        ("abundance_max", C.c_uint32),
        ("histo_max", C.c_uint32),
        ("device", C.c_int32),
        ("nb_partitions", C.c_uint32),
        ("minimizer_size", C.c_uint32),
        ("flags", C.c_uint32),
        ("world_size", C.c_uint32),
        ("rank", C.c_uint32),
        ("max_pass_mkeys", C.c_uint32),
        ("solidity_kind", C.c_uint32),
        ("solidity_custom", C.c_uint32),
        ("reserved", C.c_uint32 * 3),
    ]


class _Stats(C.Structure):
    _fields_ = [
        ("n_bytes", C.c_uint64),
        ("n_kmers", C.c_uint64),
        ("n_distinct", C.c_uint64),
        ("n_solid", C.c_uint64),
        ("n_partitions", C.c_uint32),
        ("n_levels", C.c_uint32),
        ("n_final_bins", C.c_uint32),
        ("n_retries", C.c_uint32),
        ("sort_fallback", C.c_uint64),
        ("n_passes", C.c_uint64),
        ("n_ext_regions", C.c_uint64),
        ("n_heavy", C.c_uint64),
        ("n_read_sweeps", C.c_uint64),
    ]


MG_BUCKETS = 4096      # DSKGPU_MG_BUCKETS
MG_SPLIT = 255         # DSKGPU_MG_SPLIT


def make_table(summed_loads: np.ndarray, world_size: int) -> np.ndarray:
    """dskgpu_mg_make_table: the (deterministic) repartition table for loads summed over all ranks."""
    loads = np.ascontiguousarray(summed_loads, dtype=np.uint64)
    assert loads.size == MG_BUCKETS
    table = np.zeros(MG_BUCKETS, dtype=np.uint8)
    load_library().dskgpu_mg_make_table(loads.ctypes.data_as(C.POINTER(C.c_uint64)), world_size, table.ctypes.data_as(C.POINTER(C.c_uint8)))
    return table


F_TIMING = 1
F_NO_SORT = 2
F_HISTO2D = 4
F_MG_EXPLICIT = 8
F_PLACE = 16
F_PARTITION_ORDER = 32
SOLIDITY = {"sum": 0, "min": 1, "max": 2, "one": 3, "all": 4, "custom": 5}

# every symbol include/dskgpu.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "dskgpu_create", "dskgpu_destroy", "dskgpu_last_error", "dskgpu_version", "dskgpu_device_count", "dskgpu_set_stream",
    "dskgpu_push_reads", "dskgpu_push_raw", "dskgpu_raw_finish", "dskgpu_stream_bytes", "dskgpu_rewind_reads", "dskgpu_reserve_reads", "dskgpu_reserve_work", "dskgpu_set_reads_device", "dskgpu_encode_reads", "dskgpu_next_bank", "dskgpu_set_banks", "dskgpu_histogram2d",
    "dskgpu_count", "dskgpu_mg_scatter", "dskgpu_mg_sample", "dskgpu_mg_make_table", "dskgpu_mg_set_table",
    "dskgpu_mg_send_capacity_words", "dskgpu_mg_count", "dskgpu_mg_sent_kmers", "dskgpu_mg_count_sized",
    "dskgpu_mg_slices_prepare", "dskgpu_mg_scatter_slice", "dskgpu_mg_slices_finish", "dskgpu_mg_count_sliced", "dskgpu_get_stats", "dskgpu_histogram",
    "dskgpu_set_row_order", "dskgpu_num_partitions", "dskgpu_partition_size", "dskgpu_partition_offsets", "dskgpu_partition_copy", "dskgpu_result_device",
    "dskgpu_stage_times", "dskgpu_k_encode", "dskgpu_k_enumerate", "dskgpu_k_minimizers",
    "dskgpu_group_create", "dskgpu_group_destroy", "dskgpu_group_last_error", "dskgpu_group_size", "dskgpu_group_ctx",
    "dskgpu_group_transport", "dskgpu_group_count", "dskgpu_group_exchanged_words", "dskgpu_group_sliced_steps", "dskgpu_group_histogram", "dskgpu_group_histogram2d",
    "dskgpu_group_get_stats", "dskgpu_group_num_partitions", "dskgpu_group_partition_size", "dskgpu_group_partition_copy",
]

_lib = None


SLICE_GATE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32)      # dskgpu_slice_gate: 0 = the stream waits for the slice, else the count stops


def library_path() -> str:
    return os.environ.get("DSKGPU_LIB", os.path.join(_HERE, "libdskgpu.so"))


def load_library():
    """Load libdskgpu.so; fails loudly (no fallback) when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C dsk_amd/csrc` (there is no CPU fallback for the count path)")
    lib = C.CDLL(path)
    vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
    lib.dskgpu_create.argtypes = [C.POINTER(_Config), C.POINTER(vp)]
    lib.dskgpu_create.restype = C.c_int
    lib.dskgpu_destroy.argtypes = [vp]
    lib.dskgpu_destroy.restype = None
    lib.dskgpu_last_error.argtypes = [vp]
    lib.dskgpu_last_error.restype = C.c_char_p
    lib.dskgpu_version.argtypes = []
    lib.dskgpu_version.restype = C.c_char_p
    lib.dskgpu_set_stream.argtypes = [vp, vp]
    lib.dskgpu_push_reads.argtypes = [vp, vp, u64]
    lib.dskgpu_push_raw.argtypes = [vp, vp, u64, C.c_int, C.c_int]
    lib.dskgpu_raw_finish.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    lib.dskgpu_stream_bytes.argtypes = [vp, C.POINTER(u64)]
    lib.dskgpu_rewind_reads.argtypes = [vp, u64]
    lib.dskgpu_reserve_reads.argtypes = [vp, u64]
    lib.dskgpu_reserve_work.argtypes = [vp, u64]
    lib.dskgpu_set_reads_device.argtypes = [vp, vp, u64]
    lib.dskgpu_count.argtypes = [vp]
    lib.dskgpu_encode_reads.argtypes = [vp]
    lib.dskgpu_next_bank.argtypes = [vp]
    lib.dskgpu_set_banks.argtypes = [vp, C.POINTER(u64), u32]
    lib.dskgpu_histogram2d.argtypes = [vp, C.POINTER(u64), u32]
    lib.dskgpu_mg_scatter.argtypes = [vp, vp, u64, C.POINTER(u64)]
    lib.dskgpu_mg_sample.argtypes = [vp, C.POINTER(u64)]
    lib.dskgpu_mg_make_table.argtypes = [C.POINTER(u64), u32, C.POINTER(C.c_uint8)]
    lib.dskgpu_mg_make_table.restype = None
    lib.dskgpu_mg_set_table.argtypes = [vp, C.POINTER(C.c_uint8)]
    lib.dskgpu_mg_send_capacity_words.argtypes = [vp]
    lib.dskgpu_mg_send_capacity_words.restype = u64
    lib.dskgpu_mg_count.argtypes = [vp, vp, u64]
    lib.dskgpu_mg_sent_kmers.argtypes = [vp, C.POINTER(u64)]
    lib.dskgpu_mg_count_sized.argtypes = [vp, vp, u64, u64]
    lib.dskgpu_mg_slices_prepare.argtypes = [vp, u32, C.POINTER(u32), C.POINTER(u64), C.POINTER(u64)]
    lib.dskgpu_mg_scatter_slice.argtypes = [vp, vp, u64, u32]
    lib.dskgpu_mg_slices_finish.argtypes = [vp, C.POINTER(C.c_int)]
    lib.dskgpu_mg_count_sliced.argtypes = [vp, vp, u32, C.POINTER(u64), u64, SLICE_GATE, vp]
    lib.dskgpu_get_stats.argtypes = [vp, C.POINTER(_Stats)]
    lib.dskgpu_histogram.argtypes = [vp, C.POINTER(u64), u32]
    lib.dskgpu_num_partitions.argtypes = [vp]
    lib.dskgpu_num_partitions.restype = u32
    lib.dskgpu_set_row_order.argtypes = [vp, C.c_int]
    lib.dskgpu_partition_size.argtypes = [vp, u32]
    lib.dskgpu_partition_offsets.argtypes = [vp, vp]
    lib.dskgpu_partition_size.restype = u64
    lib.dskgpu_partition_copy.argtypes = [vp, u32, vp, vp]
    lib.dskgpu_result_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)]
    lib.dskgpu_stage_times.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]
    lib.dskgpu_k_encode.argtypes = [vp, vp, u64, vp, vp]
    lib.dskgpu_k_enumerate.argtypes = [vp, vp, u64, vp, vp]
    lib.dskgpu_k_minimizers.argtypes = [vp, vp, u64, vp, vp]
    lib.dskgpu_group_create.argtypes = [C.POINTER(_Config), C.POINTER(C.c_int32), u32, C.POINTER(vp)]
    lib.dskgpu_group_destroy.argtypes = [vp]
    lib.dskgpu_group_destroy.restype = None
    lib.dskgpu_group_last_error.argtypes = [vp]
    lib.dskgpu_group_last_error.restype = C.c_char_p
    lib.dskgpu_group_size.argtypes = [vp]
    lib.dskgpu_group_size.restype = u32
    lib.dskgpu_group_ctx.argtypes = [vp, u32]
    lib.dskgpu_group_ctx.restype = vp
    lib.dskgpu_group_transport.argtypes = [vp]
    lib.dskgpu_group_transport.restype = C.c_char_p
    lib.dskgpu_group_count.argtypes = [vp]
    lib.dskgpu_group_exchanged_words.argtypes = [vp]
    lib.dskgpu_group_exchanged_words.restype = u64
    lib.dskgpu_group_sliced_steps.argtypes = [vp]
    lib.dskgpu_group_sliced_steps.restype = u32
    lib.dskgpu_group_histogram.argtypes = [vp, C.POINTER(u64), u32]
    lib.dskgpu_group_histogram2d.argtypes = [vp, C.POINTER(u64), u32]
    lib.dskgpu_group_get_stats.argtypes = [vp, C.POINTER(_Stats)]
    lib.dskgpu_group_num_partitions.argtypes = [vp]
    lib.dskgpu_group_num_partitions.restype = u32
    lib.dskgpu_group_partition_size.argtypes = [vp, u32]
    lib.dskgpu_group_partition_size.restype = u64
    lib.dskgpu_group_partition_copy.argtypes = [vp, u32, vp, vp]
    for name in EXPORTS:          # every declared symbol must resolve (fails loudly on a stale build)
        getattr(lib, name)
    _lib = lib
    return lib


def _make_config(kmer_size, abundance_min, abundance_max, histo_max, device, nb_partitions, timing, sort, world_size, rank,
                 minimizer_size, max_pass_mkeys, solidity_kind, solidity_custom, histo2d, mg_explicit, place=False, partition_order=False) -> "_Config":
    cfg = _Config()
    cfg.kmer_size = kmer_size
    cfg.abundance_min = abundance_min
    cfg.abundance_max = abundance_max
    cfg.histo_max = histo_max
    cfg.device = device
    cfg.nb_partitions = nb_partitions
    cfg.minimizer_size = minimizer_size
    cfg.max_pass_mkeys = max_pass_mkeys
    cfg.flags = (F_TIMING if timing else 0) | (0 if sort else F_NO_SORT) | (F_HISTO2D if histo2d else 0) | (F_MG_EXPLICIT if mg_explicit else 0) | (F_PLACE if place else 0) | (F_PARTITION_ORDER if partition_order else 0)
    cfg.solidity_kind = SOLIDITY[solidity_kind]
    cfg.solidity_custom = solidity_custom
    cfg.world_size = world_size
    cfg.rank = rank
    return cfg


class KmerCounter:
    """One counting context on one GPU (not thread-safe; one per device)."""

    @classmethod
    def _borrowed(cls, handle, kmer_size: int, histo_max: int, world_size: int) -> "KmerCounter":
        """A view of a ctx owned by somebody else (a KmerGroup rank): never destroyed from here."""
        self = cls.__new__(cls)
        self._lib = load_library()
        self._h = C.c_void_p(handle)
        self._owned = False
        self.kmer_size, self.histo_max, self.world_size = kmer_size, histo_max, world_size
        self.words = (kmer_size + 31) // 32
        return self

    def __init__(self, kmer_size: int = 31, abundance_min: int = 2, abundance_max: int = 2147483647,
                 histo_max: int = 10000, device: int = 0, nb_partitions: int = 0, timing: bool = False,
                 sort: bool = True, world_size: int = 1, rank: int = 0, stream: Optional[int] = None,
                 minimizer_size: int = 0, max_pass_mkeys: int = 0, solidity_kind: str = "sum", solidity_custom: int = 0,
                 histo2d: bool = False, mg_explicit: bool = False, place: bool = False, partition_order: bool = False):
        """partition_order: DSKGPU_F_PARTITION_ORDER -- rows ascending inside every output partition only (the reference's Partition<Count>
        contract; thousands of small partitions), one pass over the rows instead of the three of the global order.
        place: DSKGPU_F_PLACE -- every big device buffer becomes the best-placed of 8 candidate allocations (one-off cost of a
        few seconds at the first count, steps ~6 % faster and no longer box- and process-dependent): for contexts that count often."""
        self._lib = load_library()
        self._owned = True
        cfg = _make_config(kmer_size, abundance_min, abundance_max, histo_max, device, nb_partitions, timing, sort, world_size, rank,
                           minimizer_size, max_pass_mkeys, solidity_kind, solidity_custom, histo2d, mg_explicit, place, partition_order)
        self.kmer_size = kmer_size
        self.histo_max = histo_max
        self.words = (kmer_size + 31) // 32          # 64-bit words of a k-mer at the ABI (1..4)
        self.world_size = world_size
        h = C.c_void_p()
        rc = self._lib.dskgpu_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise DskGpuError(rc, self._lib.dskgpu_last_error(None).decode())
        self._h = h
        if stream is not None:
            self._ck(self._lib.dskgpu_set_stream(self._h, C.c_void_p(stream)))

    # -- plumbing
    def _ck(self, rc: int) -> None:
        if rc != 0:
            raise DskGpuError(rc, self._lib.dskgpu_last_error(self._h).decode())

    def close(self) -> None:
        if getattr(self, "_h", None):
            if getattr(self, "_owned", True):
                self._lib.dskgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- input
    def push_reads(self, data) -> None:
        """data: bytes, or a C-contiguous uint8 numpy array (no copy)."""
        if isinstance(data, (bytes, bytearray)):
            buf = (C.c_char * len(data)).from_buffer_copy(data) if isinstance(data, bytes) else (C.c_char * len(data)).from_buffer(data)
            self._ck(self._lib.dskgpu_push_reads(self._h, C.addressof(buf), len(data)))
        else:
            arr = np.ascontiguousarray(data, dtype=np.uint8)
            self._ck(self._lib.dskgpu_push_reads(self._h, arr.ctypes.data, arr.size))

    RAW_FASTA, RAW_FASTQ = 1, 2

    def push_raw(self, text, fmt=None, new_file: bool = False) -> None:
        """FASTA / FASTQ text as it lies in the file (cut anywhere between calls): parsed on the device (dskgpu_push_raw).
        fmt: RAW_FASTA / RAW_FASTQ, or None = from the first byte ('>' / '@') of a text that starts a file."""
        arr = np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray, memoryview)) else np.ascontiguousarray(text, dtype=np.uint8)
        if fmt is None:
            if arr.size == 0 or arr[0] not in (ord(">"), ord("@")):
                raise ValueError("push_raw: cannot tell the format from the first byte; pass fmt")
            fmt = self.RAW_FASTA if arr[0] == ord(">") else self.RAW_FASTQ
        self._ck(self._lib.dskgpu_push_raw(self._h, C.c_void_p(arr.ctypes.data if arr.size else 0), arr.size, int(fmt), int(bool(new_file))))

    def raw_finish(self):
        """-> (bytes of the read stream, records in the raw pushes since the last finish); raises DskGpuError(DSKGPU_E_FORMAT) when the text was not what
        the device parser handles (the raw pushes are dropped then: parse on the host and push_reads)."""
        nb, ln = C.c_uint64(0), C.c_uint64(0)
        self._ck(self._lib.dskgpu_raw_finish(self._h, C.byref(nb), C.byref(ln)))
        return nb.value, ln.value

    def stream_bytes(self) -> int:
        n = C.c_uint64(0)
        self._ck(self._lib.dskgpu_stream_bytes(self._h, C.byref(n)))
        return n.value

    def rewind_reads(self, stream_bytes: int) -> None:
        """Cut the pushed read stream back to its first stream_bytes bytes (dskgpu_rewind_reads)."""
        self._ck(self._lib.dskgpu_rewind_reads(self._h, stream_bytes))

    def reserve_reads(self, nbytes: int) -> None:
        self._ck(self._lib.dskgpu_reserve_reads(self._h, nbytes))

    def reserve_work(self, nbytes: int) -> bool:
        """-> False when the engine declined (DSKGPU_NOT_RESERVED: the request exceeds 60 % of the free HBM; count() sizes its own buffers)."""
        rc = self._lib.dskgpu_reserve_work(self._h, nbytes)
        if rc == 1:
            return False
        self._ck(rc)
        return True

    def set_reads_device(self, ptr: int, nbytes: int) -> None:
        self._ck(self._lib.dskgpu_set_reads_device(self._h, C.c_void_p(ptr), nbytes))

    def encode_reads(self) -> None:
        """Encode the current reads to their 2-bit form now and let go of the bytes: the buffer given to set_reads_device may be freed."""
        self._ck(self._lib.dskgpu_encode_reads(self._h))

    def next_bank(self) -> None:
        self._ck(self._lib.dskgpu_next_bank(self._h))

    def set_banks(self, end_offsets) -> None:
        arr = (C.c_uint64 * len(end_offsets))(*end_offsets)
        self._ck(self._lib.dskgpu_set_banks(self._h, arr, len(end_offsets)))

    def histogram2d(self) -> np.ndarray:
        out = np.zeros((self.histo_max + 1, 11), dtype=np.uint64)
        self._ck(self._lib.dskgpu_histogram2d(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), self.histo_max + 1))
        return out

    def set_stream(self, stream: Optional[int]) -> None:
        """All device work of this context on the given hipStream_t handle.  None or 0 = a stream owned by the context -- NOT the
        legacy default stream (whose handle is 0): to order the context against torch work, hand it a torch.cuda.Stream()."""
        self._ck(self._lib.dskgpu_set_stream(self._h, C.c_void_p(stream) if stream else None))

    # -- hot path
    def count(self) -> None:
        self._ck(self._lib.dskgpu_count(self._h))

    def mg_send_capacity_words(self) -> int:
        n = int(self._lib.dskgpu_mg_send_capacity_words(self._h))
        if n == 0:
            raise DskGpuError(-2, "dskgpu_mg_send_capacity_words: " + (self._lib.dskgpu_last_error(self._h) or b"").decode())
        return n

    def mg_scatter(self, send_ptr: int, capacity_words: int) -> List[int]:
        counts = (C.c_uint64 * self.world_size)()
        self._ck(self._lib.dskgpu_mg_scatter(self._h, C.c_void_p(send_ptr), capacity_words, counts))
        return [int(c) for c in counts]

    def mg_sample(self) -> np.ndarray:
        """Sampled k-mer load of this rank's reads per minimizer bucket (u64[MG_BUCKETS]); sum over ranks, then make_table."""
        loads = np.zeros(MG_BUCKETS, dtype=np.uint64)
        self._ck(self._lib.dskgpu_mg_sample(self._h, loads.ctypes.data_as(C.POINTER(C.c_uint64))))
        return loads

    def mg_set_table(self, table: Optional[np.ndarray]) -> None:
        if table is None:
            self._ck(self._lib.dskgpu_mg_set_table(self._h, None))
        else:
            t = np.ascontiguousarray(table, dtype=np.uint8)
            assert t.size == MG_BUCKETS
            self._ck(self._lib.dskgpu_mg_set_table(self._h, t.ctypes.data_as(C.POINTER(C.c_uint8))))

    def mg_sent_kmers(self) -> List[int]:
        """k-mers inside the records the last mg_scatter wrote for every owner (the receivers' sizing: see mg_count)."""
        k = (C.c_uint64 * self.world_size)()
        self._ck(self._lib.dskgpu_mg_sent_kmers(self._h, k))
        return [int(c) for c in k]

    # -- a step in slices (the exchange of slice i overlaps the sender of slice i + 1 and the receiver's level 1 of slice i - 1)
    def mg_slices_prepare(self, want_slices: int):
        """-> (nslices, send_words[nslices][world], kmers_est[world]); nslices == 0: this input takes the one-piece path."""
        n = C.c_uint32(0)
        words = (C.c_uint64 * (want_slices * self.world_size))()
        est = (C.c_uint64 * self.world_size)()
        self._ck(self._lib.dskgpu_mg_slices_prepare(self._h, want_slices, C.byref(n), words, est))
        ns = int(n.value)
        return ns, [[int(words[s * self.world_size + o]) for o in range(self.world_size)] for s in range(ns)], [int(x) for x in est]

    def mg_scatter_slice(self, send_ptr: int, capacity_words: int, s: int) -> None:
        """Launches the sender of slice s on the context's stream; returns without synchronising."""
        self._ck(self._lib.dskgpu_mg_scatter_slice(self._h, C.c_void_p(send_ptr), capacity_words, s))

    def mg_slices_finish(self) -> bool:
        """True when a slice of the send layout overflowed: every rank then repeats the step in one piece."""
        o = C.c_int(0)
        self._ck(self._lib.dskgpu_mg_slices_finish(self._h, C.byref(o)))
        return bool(o.value)

    def mg_count_sliced(self, recv_ptr: int, slice_words: Sequence[int], n_kmers_est: int, gate) -> None:
        """gate(s) is called right before the first device work that reads slice s is enqueued: make the stream wait for it."""
        arr = (C.c_uint64 * len(slice_words))(*[int(w) for w in slice_words])
        failed: list = []

        def _gate(_user, s):      # ctypes would print and swallow an exception raised in here: keep it, tell the C side, re-raise below
            try:
                gate(int(s))
                return 0
            except BaseException as e:      # noqa: BLE001 -- a failed wait (collective timeout / abort) must stop the count, whatever it is
                failed.append(e)
                return 1
        cb = SLICE_GATE(_gate)
        rc = self._lib.dskgpu_mg_count_sliced(self._h, C.c_void_p(recv_ptr), len(slice_words), arr, n_kmers_est, cb, None)
        if failed:
            raise failed[0]
        self._ck(rc)

    def mg_count(self, recv_ptr: int, recv_words: int, n_kmers: int = 0) -> None:
        """n_kmers = the senders' k-mer total for this rank (sum over sources of mg_sent_kmers()[rank]); 0 = count them here."""
        self._ck(self._lib.dskgpu_mg_count_sized(self._h, C.c_void_p(recv_ptr), recv_words, n_kmers))

    # -- results
    def stats(self) -> dict:
        s = _Stats()
        self._ck(self._lib.dskgpu_get_stats(self._h, C.byref(s)))
        return {k: int(getattr(s, k)) for k, _ in _Stats._fields_ if k != "reserved"}

    def histogram(self) -> np.ndarray:
        out = np.zeros(self.histo_max + 1, dtype=np.uint64)
        self._ck(self._lib.dskgpu_histogram(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), self.histo_max + 1))
        return out

    def num_partitions(self) -> int:
        return int(self._lib.dskgpu_num_partitions(self._h))

    def partition(self, p: int) -> Tuple[np.ndarray, np.ndarray]:
        n = int(self._lib.dskgpu_partition_size(self._h, p))
        kmers = np.zeros((n, self.words), dtype=np.uint64)
        ab = np.zeros(n, dtype=np.uint32)
        self._ck(self._lib.dskgpu_partition_copy(self._h, p, C.c_void_p(kmers.ctypes.data), C.c_void_p(ab.ctypes.data)))
        return kmers, ab

    def set_row_order(self, partition_order: bool) -> None:
        self._ck(self._lib.dskgpu_set_row_order(self._h, 1 if partition_order else 0))

    def partition_offsets(self) -> np.ndarray:
        off = np.zeros(self.num_partitions() + 1, dtype=np.uint64)
        self._ck(self._lib.dskgpu_partition_offsets(self._h, C.c_void_p(off.ctypes.data)))
        return off

    def partition_sizes(self) -> np.ndarray:
        return np.diff(self.partition_offsets().astype(np.int64))

    def rows(self) -> Tuple[np.ndarray, np.ndarray]:
        """All solid rows, partitions concatenated in index order (what dsk2ascii walks)."""
        ks, abs_ = [], []
        for p in range(self.num_partitions()):
            k, a = self.partition(p)
            ks.append(k)
            abs_.append(a)
        if not ks:
            return np.zeros((0, self.words), np.uint64), np.zeros(0, np.uint32)
        return np.concatenate(ks), np.concatenate(abs_)

    def result_device(self) -> Tuple[int, int, int]:
        k, a, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        self._ck(self._lib.dskgpu_result_device(self._h, C.byref(k), C.byref(a), C.byref(n)))
        return int(k.value or 0), int(a.value or 0), int(n.value)

    def stage_times(self) -> List[Tuple[str, float]]:
        cap = 64
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        n = self._lib.dskgpu_stage_times(self._h, names, ms, cap)
        return [(names[i].decode(), float(ms[i])) for i in range(min(n, cap))]

    # -- kernel-level entry points (parity tests)
    def k_encode(self, d_bytes: int, nbytes: int, d_packed: int, d_invalid: int) -> None:
        self._ck(self._lib.dskgpu_k_encode(self._h, C.c_void_p(d_bytes), nbytes, C.c_void_p(d_packed), C.c_void_p(d_invalid)))

    def k_enumerate(self, d_bytes: int, nbytes: int, d_kmers: int, d_valid: int) -> None:
        self._ck(self._lib.dskgpu_k_enumerate(self._h, C.c_void_p(d_bytes), nbytes, C.c_void_p(d_kmers), C.c_void_p(d_valid)))

    def k_minimizers(self, d_bytes: int, nbytes: int, d_minim: int, d_valid: int) -> None:
        self._ck(self._lib.dskgpu_k_minimizers(self._h, C.c_void_p(d_bytes), nbytes, C.c_void_p(d_minim), C.c_void_p(d_valid)))


class KmerGroup:
    """N ranks of one sharded count inside this process (include/dskgpu.h: dskgpu_group_*): what `dsk -nb-gpus N` runs.
    `rank(r)` is a KmerCounter view of rank r for feeding reads and reading that rank's rows."""

    def __init__(self, devices, kmer_size: int = 31, abundance_min: int = 2, abundance_max: int = 2147483647,
                 histo_max: int = 10000, nb_partitions: int = 0, timing: bool = False, sort: bool = True,
                 minimizer_size: int = 0, max_pass_mkeys: int = 0, mg_explicit: bool = False):
        self._lib = load_library()
        devices = list(devices)
        cfg = _make_config(kmer_size, abundance_min, abundance_max, histo_max, 0, nb_partitions, timing, sort, len(devices), 0,
                           minimizer_size, max_pass_mkeys, "sum", 0, False, mg_explicit)
        devs = (C.c_int32 * len(devices))(*devices)
        h = C.c_void_p()
        rc = self._lib.dskgpu_group_create(C.byref(cfg), devs, len(devices), C.byref(h))
        if rc != 0:
            raise DskGpuError(rc, self._lib.dskgpu_group_last_error(None).decode())
        self._h = h
        self.size = len(devices)
        self.kmer_size, self.histo_max = kmer_size, histo_max
        self.words = (kmer_size + 31) // 32
        self._ranks = [KmerCounter._borrowed(self._lib.dskgpu_group_ctx(self._h, r), kmer_size, histo_max, self.size)
                       for r in range(self.size)]

    def _ck(self, rc: int) -> None:
        if rc != 0:
            raise DskGpuError(rc, self._lib.dskgpu_group_last_error(self._h).decode())

    def close(self) -> None:
        if getattr(self, "_h", None):
            for r in self._ranks:
                r._h = None
            self._lib.dskgpu_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def rank(self, r: int) -> KmerCounter:
        return self._ranks[r]

    def transport(self) -> str:
        return self._lib.dskgpu_group_transport(self._h).decode()

    def count(self) -> None:
        self._ck(self._lib.dskgpu_group_count(self._h))

    def exchanged_words(self) -> int:
        return int(self._lib.dskgpu_group_exchanged_words(self._h))

    def sliced_steps(self) -> int:
        """Steps of the last count whose exchange ran in slices (overlapped with the sender and the receiver's level 1)."""
        return int(self._lib.dskgpu_group_sliced_steps(self._h))

    def histogram(self) -> np.ndarray:
        out = np.zeros(self.histo_max + 1, dtype=np.uint64)
        self._ck(self._lib.dskgpu_group_histogram(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), self.histo_max + 1))
        return out

    def stats(self) -> dict:
        s = _Stats()
        self._ck(self._lib.dskgpu_group_get_stats(self._h, C.byref(s)))
        return {k: int(getattr(s, k)) for k, _ in _Stats._fields_ if k != "reserved"}

    def num_partitions(self) -> int:
        return int(self._lib.dskgpu_group_num_partitions(self._h))

    def partition(self, p: int) -> Tuple[np.ndarray, np.ndarray]:
        n = int(self._lib.dskgpu_group_partition_size(self._h, p))
        kmers = np.zeros((n, self.words), dtype=np.uint64)
        ab = np.zeros(n, dtype=np.uint32)
        self._ck(self._lib.dskgpu_group_partition_copy(self._h, p, C.c_void_p(kmers.ctypes.data), C.c_void_p(ab.ctypes.data)))
        return kmers, ab


def kmer_to_string(value: int, k: int) -> str:
    """Kmer<span>::ModelCanonical::toString (utils/dsk2ascii.cpp:104): MSB-first, A C T G = 0 1 2 3."""
    return "".join("ACTG"[(value >> (2 * (k - 1 - i))) & 3] for i in range(k))
