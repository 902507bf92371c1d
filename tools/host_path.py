"""End-to-end time of the host-buffer entry point (fdn_filter_3d: H2D + three passes + D2H) on the bench volume."""
import sys, time, numpy as np
sys.path.insert(0, ".")
import flowdenoising_amd as fd
from flowdenoising_amd import _lib
from flowdenoising_amd.synth import make_volume
shape = (512, 1024, 1024)
vol = make_volume(shape, seed=1237, amplitude=100.0)
k = fd.get_gaussian_kernel(2.0)
h = fd.operators.handle(0)
for it in range(3):
    h.enable_timers(True); h.timers(reset=True)
    t0 = time.perf_counter()
    out = fd.OF_filter(vol, [k, k, k], 0, 5)
    dt = time.perf_counter() - t0
    tm = h.timers()
    print(f"run {it}: {dt*1e3:.0f} ms total; timers(ms): " + ", ".join(f"{n}={v[0]:.0f}" for n, v in tm.items() if v[1]), flush=True)
