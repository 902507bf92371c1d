import sys, numpy as np
sys.path.insert(0, ".")
import flowdenoising_amd as fdn
from flowdenoising_amd.synth import make_volume
from oracle import oracle
oracle.build()
rng = np.random.default_rng(99)
for i in range(400):
    axis = int(rng.integers(0, 3))
    small, a, b = int(rng.integers(3, 8)), int(rng.integers(100, 420)), int(rng.integers(100, 640))
    shape = [0, 0, 0]; shape[axis] = small
    rest = [x for x in range(3) if x != axis]; shape[rest[0]], shape[rest[1]] = a, b
    w = int(rng.choice([3, 4, 5, 5, 6, 7, 8, 9, 11, 15])); l = int(rng.integers(0, 4))
    sigma = float(rng.choice([0.5, 1.0])); border = int(rng.integers(0, 2)); chained = bool(rng.integers(0, 2))
    if shape == [395, 3, 544]:
        break
print(i, shape, axis, l, w, sigma, border, chained)
vol = make_volume(tuple(shape), seed=5000 + i, amplitude=100.0)
k = fdn.get_gaussian_kernel(sigma); mean = vol.mean()
fn = [fdn.OF_filter_along_Z, fdn.OF_filter_along_Y, fdn.OF_filter_along_X][axis]
got = fn(vol, k, l, w, mean, border_mode=border, chained=chained)
import os
for bm, name in ((oracle.BOX_RUNNING, "opencv order"), (2, "kernel order")):
    want = oracle.filter_along_axis(vol, axis, k, l, w, mean, border_mode=border, chained=chained, box_mode=bm, nthreads=16)
    d = np.abs(got - want)
    print(name, "equal", np.array_equal(got, want), "max rel", d.max() / np.abs(want).max(), "n diff", int((d > 0).sum()))
os.environ["FDN_FORCE_STAGED"] = "1"
