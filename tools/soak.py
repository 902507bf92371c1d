"""Determinism soak: the same sweep run repeatedly (and under each occupancy build) must give identical bits --
a race in the LDS hand-over or the window refill would show up here."""
import os, sys, subprocess, numpy as np
sys.path.insert(0, ".")
code = r'''
import sys, numpy as np
sys.path.insert(0, ".")
import flowdenoising_amd as fd
from flowdenoising_amd.synth import make_volume
vol = make_volume((48, 1024, 1024), seed=77, amplitude=100.0)
k = fd.get_gaussian_kernel(2.0)
ref = None
for it in range(6):
    out = fd.OF_filter_along_Z(vol, k, int(sys.argv[1]), int(sys.argv[3]), vol.mean())
    if ref is None: ref = out
    assert np.array_equal(ref, out), ("run", it)
np.save(sys.argv[2], ref)
print("ok", sys.argv[1:], flush=True)
'''
outs = []
for l in (0, 3):
    for occ in ("", "3", "4", "5", "8"):      # "": the launcher's own choice; 8: two bands per workgroup
        fn = f"/tmp/soak_{l}_{occ or 'auto'}.npy"
        env = dict(os.environ, FDN_FUSED_OCC=occ) if occ else {k: v for k, v in os.environ.items() if k != "FDN_FUSED_OCC"}
        subprocess.run([sys.executable, "-c", code, str(l), fn, "5"], env=env, check=True)
        outs.append((l, fn))
# the one-iteration kernel (winsize 15): its own runs, and against the per-stage kernels
for l in (0, 3):
    fns = []
    for path in ("0", "1"):
        fn = f"/tmp/soak_w15_{l}_{path}.npy"
        subprocess.run([sys.executable, "-c", code, str(l), fn, "15"], env=dict(os.environ, FDN_PATH=path), check=True)
        fns.append(fn)
    assert np.array_equal(np.load(fns[0]), np.load(fns[1])), ("w15", l)
for l in (0, 3):
    a = [np.load(fn) for ll, fn in outs if ll == l]
    assert all(np.array_equal(a[0], b) for b in a[1:]), l
print("all runs and builds bit-identical")
