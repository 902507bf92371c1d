"""Out-of-core mode on the bench volume: time with the volume on the host and 128 slices of a pass on the GPU at a time."""
import sys, time, numpy as np
sys.path.insert(0, ".")
import flowdenoising_amd as fd
from flowdenoising_amd import streaming
from flowdenoising_amd.synth import make_volume
vol = make_volume((512, 1024, 1024), seed=1237, amplitude=100.0)
k = fd.get_gaussian_kernel(2.0)
for chunk in (128, 256, None):
    t0 = time.perf_counter()
    out = streaming.OF_filter_streamed(vol, [k, k, k], 0, 5, chunk)
    dt = time.perf_counter() - t0
    print(f"chunk {chunk}: {dt:.2f} s = {vol.size / dt / 1e6:.0f} Mvox/s", flush=True)
ref = fd.OF_filter(vol, [k, k, k], 0, 5)
print("bit-identical to the resident path:", np.array_equal(out, ref))
