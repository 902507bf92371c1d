"""Where the out-of-core mode's time goes on the bench volume (chunks of 128 / 256 slices): the host mean, page-locking of the
host arrays, and the passes; against the resident path."""
import sys, time, numpy as np
sys.path.insert(0, ".")
import flowdenoising_amd as fd
from flowdenoising_amd import _lib, streaming
from flowdenoising_amd.synth import make_volume
vol = make_volume((512, 1024, 1024), seed=1237, amplitude=100.0)
k = fd.get_gaussian_kernel(2.0)
t0 = time.perf_counter(); m = _lib.mean_host(vol); print(f"mean_host: {time.perf_counter() - t0:.3f} s", flush=True)
h = _lib.Handle(0)
t0 = time.perf_counter(); ok = h.host_register(vol); t1 = time.perf_counter(); h.host_unregister(vol); print(f"register a touched 2 GiB array: {t1 - t0:.3f} s ({ok})", flush=True)
a = np.empty_like(vol)
t0 = time.perf_counter(); ok = h.host_register(a); t1 = time.perf_counter(); h.host_unregister(a); print(f"register a fresh np.empty of 2 GiB: {t1 - t0:.3f} s ({ok})", flush=True)
del a
for chunk in (128, 256):
    for rep in range(2):
        t0 = time.perf_counter()
        out = streaming.OF_filter_streamed(vol, [k, k, k], 0, 5, chunk, mean=m)
        dt = time.perf_counter() - t0
        print(f"chunk {chunk} (mean given): {dt:.2f} s = {vol.size / dt / 1e6:.0f} Mvox/s", flush=True)
t0 = time.perf_counter(); ref = fd.OF_filter(vol, [k, k, k], 0, 5); print(f"resident OF_filter (numpy in, numpy out): {time.perf_counter() - t0:.2f} s", flush=True)
print("bit-identical:", np.array_equal(out, ref))
