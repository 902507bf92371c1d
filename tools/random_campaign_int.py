"""Randomised parity campaign for integer volumes: int16 / uint16 / uint8 volumes through the three-pass
filter with seq's semantics (float64 padded volume) and par's (integer images), random shapes, sigmas, levels and windows,
GPU against the oracle's restatements, bit for bit.
usage: random_campaign_int.py [n] [seed]"""
import sys, numpy as np
sys.path.insert(0, ".")
import flowdenoising_amd as fdn
from flowdenoising_amd.synth import make_volume
from oracle import oracle
oracle.build()
import os        # the product reads these at handle creation (opencv_fma, remap_model): the oracle follows the same switches
oracle.set_fma(int(os.environ.get("FDN_OPENCV_FMA", 0)), int(os.environ.get("FDN_OPENCV_FMA_LANES", 8)))
oracle.set_remap_model(int(os.environ.get("FDN_REMAP_MODEL", 0)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 31)
bad = 0
for i in range(n):
    shape = (int(rng.integers(5, 14)), int(rng.integers(33, 150)), int(rng.integers(33, 150)))
    dtype = [np.int16, np.uint16, np.uint8][int(rng.integers(3))]       # uint8 (round 4): par's fixed-point remap, seq's float64 padding
    span = 255 if dtype == np.uint8 else int(rng.choice([255, 4095, 30000]))
    v = make_volume(shape, seed=int(rng.integers(1 << 30)), amplitude=100.0)
    lo, hi = float(v.min()), float(v.max())
    vi = np.round((v - lo) / (hi - lo) * span)
    if dtype == np.int16:
        vi -= span // 2
    vi = vi.astype(dtype)
    ks = [fdn.get_gaussian_kernel(float(s)) if rng.random() > 0.15 else None for s in rng.choice([0.5, 1.0, 1.5, 2.0], 3)]
    l, w = int(rng.integers(0, 4)), int(rng.choice([3, 5, 5, 7, 9, 11, 15]))
    if rng.random() < 0.5:
        got = fdn.OF_filter(vi, ks, l, w)
        want = oracle.OF_filter_integer_input(vi, ks, l, w, nthreads=8)
        what = "seq"
    else:
        chained = bool(rng.random() < 0.7)
        x = vi.copy()
        fdn.FlowDenoising(1, x, l, w, fdn.get_flow_with_prev_flow if chained else fdn.get_flow_without_prev_flow).filter(ks)
        got = x
        want = oracle.filter_par_integer_input(vi, ks, l, w, nthreads=8, chained=chained).astype(dtype)
        what = "par"
    if not np.array_equal(got, want):
        bad += 1
        print("MISMATCH", what, shape, np.dtype(dtype).name, span, l, w, [None if k is None else k.size for k in ks],
              float(np.abs(got.astype(np.float64) - want).max()), flush=True)
print(f"{n} integer-volume cases: {n - bad} bit-identical, {bad} mismatches")
