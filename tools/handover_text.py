import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, workloads as wl
from defl_amd import capi
eng = capi.load()
assert eng.lib.infl_gpu_init(0) == 0
fn = eng.lib.infl_debug_par_profile
fn.argtypes = [C.POINTER(C.c_ulonglong)]
sizes = [65536] * 8192
comps = bench.build_pool(0, sizes, 16)
src, so, do = wl.pack(comps, sizes)
b = eng.device_batch(src, so, do)
buf = (C.c_ulonglong * 32)()
assert eng.lib.infl_gpu_set_table_variant(3) == 3
b.run(1, detail=False); fn(buf)
b.run(1, detail=False); fn(buf)
_, st, ol, _ = b.results(want_dst=False)
idx = np.nonzero(st != 0)[0]
print("handed over:", idx.tolist(), [hex(int(x)) for x in st[idx][:10]])
print("tables too large", buf[15], "malformed", buf[29], "serial stretches", buf[28])
