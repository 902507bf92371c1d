"""What hipHostRegister / hipHostUnregister of a 2 GiB numpy buffer cost, against the pageable copy they replace
(decides how fdn_filter_3d moves host volumes)."""
import ctypes, sys, time, numpy as np
sys.path.insert(0, ".")
from flowdenoising_amd import _lib
_lib.load()
import importlib.util, os
hip = ctypes.CDLL(os.path.join(list(importlib.util.find_spec("torch").submodule_search_locations)[0], "lib", "libamdhip64.so"))
a = np.ones(512 * 1024 * 1024, np.float32)
h = _lib.Handle(0)
d = h.malloc(a.nbytes)
for it in range(3):
    t0 = time.perf_counter(); rc = hip.hipHostRegister(ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes), 0); t1 = time.perf_counter()
    h.h2d(d, a); t2 = time.perf_counter()
    rc2 = hip.hipHostUnregister(ctypes.c_void_p(a.ctypes.data)); t3 = time.perf_counter()
    h.h2d(d, a); t4 = time.perf_counter()
    print(f"register {1e3*(t1-t0):.0f} ms (rc {rc}), pinned H2D {1e3*(t2-t1):.0f} ms, unregister {1e3*(t3-t2):.0f} ms (rc {rc2}), pageable H2D {1e3*(t4-t3):.0f} ms", flush=True)
