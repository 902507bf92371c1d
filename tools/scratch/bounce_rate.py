"""Rates of the library's host copies: page-locked by the caller, page-locked for the call (>= 8 MB), bounced (a read-only mapping)."""
import mmap, sys, time, numpy as np
sys.path.insert(0, ".")
from flowdenoising_amd import _lib
h = _lib.Handle(0)
n = 1 << 30
a = np.ones(n // 4, np.float32)
d = h.malloc(n)
def rate(f):
    f(); t0 = time.perf_counter(); f(); return n / (time.perf_counter() - t0) / 1e9
print("pageable numpy array (registered for the call): h2d %.1f GB/s, d2h %.1f GB/s" % (rate(lambda: h.h2d(d, a)), rate(lambda: h.d2h(a, d))))
h.host_register(a)
print("caller-registered: h2d %.1f GB/s, d2h %.1f GB/s" % (rate(lambda: h.h2d(d, a)), rate(lambda: h.d2h(a, d))))
h.host_unregister(a)
with open("/dev/shm/fdn_bounce_probe.bin", "wb") as f:
    f.write(a.tobytes())
with open("/dev/shm/fdn_bounce_probe.bin", "rb") as f:
    mm = mmap.mmap(f.fileno(), n, access=mmap.ACCESS_READ)
    ro = np.frombuffer(mm, np.float32)
    print("read-only file mapping: registers:", h.host_register(ro))
    h.host_unregister(ro)
    print("read-only file mapping: h2d %.1f GB/s" % rate(lambda: h.h2d(d, ro)))
    del ro
import os; os.unlink("/dev/shm/fdn_bounce_probe.bin")
small = np.ones((3 << 20) // 4, np.float32)
t0 = time.perf_counter()
for _ in range(50): h.h2d(d, small); h.d2h(small, d)
print("3 MB round trips (bounced): %.2f ms each" % ((time.perf_counter() - t0) / 50 * 1e3))
