"""How far can FMA contraction inside cv2's GaussianBlur / resize move a result?  (The oracle is unpinned against
cv2; for levels > 0 the blur taps are not powers of two, and stock x86 wheels run those filters through SIMD code
that fuses multiply-adds on FMA3 machines.)  Runs the oracle with separate and with fused multiply-adds on samples of
BASELINE configs[4] (-l 3 -w 15, uint16-range data) and of par's default (-l 3 -w 5) and prints the output change.
CPU only; results are quoted in DESIGN.md 5."""
import sys, numpy as np
sys.path.insert(0, ".")
from oracle import oracle as O
from flowdenoising_amd.synth import make_volume
O.build()
for name, shape, sig, l, w, scale16 in [("configs[4] Z sample", (17, 1024, 1024), 2.0, 3, 15, True), ("par default", (17, 512, 512), 2.0, 3, 5, False),
                                        ("levels 0 (control)", (17, 256, 256), 2.0, 0, 5, False)]:
    vol = make_volume(shape, seed=1239, amplitude=100.0)
    if scale16:
        vol = np.round((vol - vol.min()) / (vol.max() - vol.min()) * 4095).astype(np.float32)
    k = O.get_gaussian_kernel(sig)
    outs = []
    for fma in (0, 1):
        O.set_fma(fma)
        outs.append(O.filter_axis_range(vol, 0, k, l, w, vol.mean(), 8, 9, nthreads=1)[8])
    O.set_fma(0)
    d = np.abs(outs[0].astype(np.float64) - outs[1])
    print(f"{name}: max |diff| / max |out| = {d.max() / np.abs(outs[0]).max():.3g}, voxels changed {np.count_nonzero(d) / d.size:.3%}, "
          f"99.9th percentile {np.quantile(d, 0.999) / np.abs(outs[0]).max():.3g}", flush=True)
