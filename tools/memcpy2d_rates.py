"""PCIe rates of the strided host <-> device copies the out-of-core mode uses (fdn_memcpy2d_*), by row length:
contiguous, Y-pass rows (cnt * X floats) and X-pass rows (cnt floats) of a 512 x 1024 x 1024 float32 volume."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from flowdenoising_amd import _lib
Z, Y, X = 512, 1024, 1024
vol = np.ones((Z, Y, X), dtype=np.float32)
h = _lib.Handle(0)
assert h.host_register(vol)
d = h.malloc(vol.nbytes // 2)
base = vol.ctypes.data
def t(fn, nbytes, what):
    fn(); h.synchronize()
    t0 = time.perf_counter(); fn(); h.synchronize(); dt = time.perf_counter() - t0
    print(f"{what}: {nbytes / dt / 1e9:.1f} GB/s ({dt * 1e3:.1f} ms for {nbytes >> 20} MiB)", flush=True)
for cnt in (128, 256):
    nb = cnt * Y * X * 4
    t(lambda: h.h2d_2d(d, nb, base, nb, nb, 1), nb, f"H2D contiguous, {cnt} Z slices")
    t(lambda: h.d2h_2d(base, nb, d, nb, nb, 1), nb, f"D2H contiguous, {cnt} Z slices")
    nb = Z * cnt * X * 4
    t(lambda: h.h2d_2d(d, cnt * X * 4, base, Y * X * 4, cnt * X * 4, Z), nb, f"H2D Y-pass chunk of {cnt}: {Z} rows of {cnt * X * 4} B")
    t(lambda: h.d2h_2d(base, Y * X * 4, d, cnt * X * 4, cnt * X * 4, Z), nb, f"D2H Y-pass chunk of {cnt}")
    nb = Z * Y * cnt * 4
    t(lambda: h.h2d_2d(d, cnt * 4, base, X * 4, cnt * 4, Z * Y), nb, f"H2D X-pass chunk of {cnt}: {Z * Y} rows of {cnt * 4} B")
    t(lambda: h.d2h_2d(base, X * 4, d, cnt * 4, cnt * 4, Z * Y), nb, f"D2H X-pass chunk of {cnt}")
h.host_unregister(vol)
