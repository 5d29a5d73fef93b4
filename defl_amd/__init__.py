"""MI355X-native inflate engine: Python access to the C-ABI library (ctypes).

The product is `defl_amd/libdefl_amd.so` (HIP kernels for gfx950 behind the C API of
`include/defl/infl.h` and `include/defl/infl_batch.h`); this package only loads it for
the test-suite and the benchmark.  There is no Python or CPU decoder here.
"""
from . import capi  # noqa: F401
