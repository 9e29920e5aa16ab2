"""dsk_amd -- MI355X-native k-mer counting engine (drop-in for the DSK count path).

The product is `libdskgpu.so` (hand-written HIP kernels behind the C-ABI of
`include/dskgpu.h`) plus the C++ host layer in `dsk_amd/host/`.  This Python
package is plumbing only: a ctypes binding used by the tests, `bench.py` and the
multi-GPU launcher (torch owns device memory, streams and the RCCL exchange).
There is no CPU fallback: importing `dsk_amd.engine` without the built library
raises.
"""
from .engine import (  # noqa: F401
    DskGpuError,
    KmerCounter,
    KmerGroup,
    load_library,
    make_table,
    library_path,
)
