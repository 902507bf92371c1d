"""Does the result of fdn_sweep_stack_dev depend on the ALIGNMENT of the caller's device pointers?  (a slab view of a volume with
odd-sized images is only 4-byte aligned)"""
import sys, numpy as np
sys.path.insert(0, ".")
import flowdenoising_amd as fdn
from flowdenoising_amd import _lib
from flowdenoising_amd.operators import _params, handle
from flowdenoising_amd.synth import make_volume
h = handle()
for shape, l, w in (((6 + 8, 131, 97), 2, 5), ((6 + 8, 131, 97), 0, 5), ((6 + 8, 70, 150), 3, 15), ((6 + 8, 131, 97), 2, 15)):
    vol = make_volume(shape, seed=21, amplitude=100.0)
    k = fdn.get_gaussian_kernel(1.0)
    r = k.size // 2
    S, H, W = shape[0] - 2 * r, shape[1], shape[2]
    p = _params(l, w)
    outs = {}
    for off_in, off_out in ((0, 0), (4, 0), (8, 0), (16, 0), (0, 4), (4, 4), (36, 20)):
        d_in = h.malloc(vol.nbytes + 256)
        d_out = h.malloc(S * H * W * 4 + 256)
        h.h2d(d_in + off_in, vol)
        h.sweep_stack_dev(d_in + off_in, d_out + off_out, S, H, W, k, p)
        out = np.empty((S, H, W), np.float32)
        h.d2h(out, d_out + off_out)
        h.free(d_in); h.free(d_out)
        outs[(off_in, off_out)] = out
    ref = outs[(0, 0)]
    print(shape, l, w, {k2: bool(np.array_equal(v, ref)) for k2, v in outs.items()}, flush=True)
