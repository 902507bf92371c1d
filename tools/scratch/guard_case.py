import sys, os, numpy as np
sys.path.insert(0, ".")
import flowdenoising_amd as fdn
from flowdenoising_amd.synth import make_volume
cases = [((7, 70, 150), 3), ((6, 131, 97), 2), ((5, 64, 200), 1), ((6, 131, 97), 2), ((7, 70, 150), 3)]
k = fdn.get_gaussian_kernel(1.0)
ref_path = "/tmp/guard_case_ref.npz"
outs = []
for shape, l in cases:
    vol = make_volume(shape, seed=21, amplitude=100.0)
    outs.append(fdn.OF_filter_along_Z(vol, k, l, 5, vol.mean()))
if sys.argv[1] == "ref":
    np.savez(ref_path, *outs); print("reference saved")
else:
    ref = np.load(ref_path)
    res = []
    for i, o in enumerate(outs):
        r = ref[f"arr_{i}"]
        res.append((cases[i], "%.3g" % float(np.abs(o - r).max() / np.abs(r).max()), sorted(set(np.argwhere(o != r)[:, 0].tolist()))))
    print(sys.argv[1], res, flush=True)
