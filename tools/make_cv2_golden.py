#!/usr/bin/env python3
"""Dump real-cv2 fixtures into tests/golden/ (run on any box where `import cv2` works; none in this
pipeline so far -- see DESIGN.md 5).  cv2 is called directly by this harness (tests/test_cv2_pin.py's
helpers); the reference's files are not involved.  Outputs are data only: seeded inputs + cv2's results.

  cv2_pair_<H>x<W>_l<l>_w<w>.npz   target, reference, init flow, cv2 flow, cv2-warped reference
  cv2_config0.npz                  128x128x64 volume (BASELINE configs[0]), sigma 2, l 0, w 5 -> seq-shaped result
  cv2_int16_seq.npz / _par.npz     an int16 volume through seq- / par-shaped sweeps with numpy's and cv2's own dtypes

After committing them, change DESIGN.md 5 to say "pinned by cv2 fixtures" (tests/test_cv2_pin.py checks it)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def which_options_match(cv2, T, O):
    """Which of the product's / the oracle's switches for the two cv2 unknowns reproduce THIS cv2 bit for bit: `remap_model`
    (0 = the 1/32-pixel table, 1 = unquantised float32 bilinear) on a warp, `opencv_fma` x `opencv_fma_lanes` on pyramid flows
    (levels 3, window 5 and 15, images whose rows are not multiples of 16).  The answer goes into cv2_identity.json next to the
    version and build lines: on that box `flowdenoising.py --opencv_fma M --opencv_fma_lanes L --remap_model R`, or
    fdn_set_option / fdo_set_fma / fdo_set_remap_model, select the matching arithmetic -- no kernel has to be written."""
    res = {"remap_model": None, "opencv_fma": None, "tried": []}
    a, b, f0 = T.make_pair((70, 150), 171)
    flow = (f0 * 3).astype(np.float32)
    want = T.cv2_warp(cv2, b, flow)
    for model in (0, 1):
        O.set_remap_model(model)
        try:
            if np.array_equal(O.warp_slice(b, flow), want):
                res["remap_model"] = model
        finally:
            O.set_remap_model(0)
    wants = {(l, w): T.cv2_flow(cv2, a, b, l, w, f0) for (l, w) in ((3, 5), (3, 15))}
    for mode, lanes in ((0, 8), (1, 8), (2, 4), (2, 8), (2, 16)):
        O.set_fma(mode, lanes)
        try:
            errs = {f"l{l}w{w}": float(np.abs(O.get_flow(b, a, l, w, f0.copy()) - wants[(l, w)]).max()) for (l, w) in wants}
        finally:
            O.set_fma(0)
        res["tried"].append({"opencv_fma": mode, "lanes": lanes, "max_abs_flow_diff": errs})
        if all(v == 0.0 for v in errs.values()) and res["opencv_fma"] is None:
            res["opencv_fma"] = {"mode": mode, "lanes": lanes}
    return res


def main():
    import cv2
    import test_cv2_pin as T
    from flowdenoising_amd.synth import make_volume
    ident = T.cv2_identity(cv2)
    print("cv2", ident)
    import json
    with open(os.path.join(T.GOLD, "cv2_identity.json"), "w") as f:      # which build produced the fixtures (version, SIMD / IPP lines)
        json.dump(ident, f, indent=1)
    # which remap model does this build follow?  (OpenCV >= 4.11 may not use the 1/32-pixel table for float maps)
    from oracle import oracle as O0
    a0, b0, f00 = T.make_pair((64, 64), 164)
    fl = (f00 * 3).astype(np.float32)
    print("remap model:", T.classify_remap(b0, fl, T.cv2_warp(cv2, b0, fl), O0.warp_slice(b0, fl).astype(np.float64)))
    ident["matched"] = which_options_match(cv2, T, O0)
    print("options that reproduce this cv2 bit for bit:", ident["matched"])
    with open(os.path.join(T.GOLD, "cv2_identity.json"), "w") as f:      # ... now with the matching option values
        json.dump(ident, f, indent=1)
    for shape in T.PAIR_SHAPES[:2]:
        for l, w in T.PAIR_PARAMS:
            a, b, f0 = T.make_pair(shape, 100 + shape[0])
            flow = T.cv2_flow(cv2, a, b, l, w, f0)
            np.savez_compressed(os.path.join(T.GOLD, f"cv2_pair_{shape[0]}x{shape[1]}_l{l}_w{w}.npz"), target=a, reference=b,
                                init=f0, flow=flow, warped=T.cv2_warp(cv2, b, flow), l=l, w=w, cv2_version=cv2.__version__)
    vol = make_volume((64, 128, 128), seed=1234 + 1, amplitude=100.0)
    from oracle import oracle as O
    k = O.get_gaussian_kernel(2.0)
    out = T.cv2_of_filter(cv2, vol, [k, k, k], 0, 5)
    np.savez_compressed(os.path.join(T.GOLD, "cv2_config0.npz"), vol=vol, out=out, sigma=2.0, l=0, w=5, cv2_version=cv2.__version__)
    # integer volumes (seq: float64 padded volume; par: integer images), through sweeps that let numpy and cv2 pick the dtypes
    vi = T._int16_volume((7, 34, 38), 21)
    ks = [O.get_gaussian_kernel(1.0), O.get_gaussian_kernel(0.5), O.get_gaussian_kernel(0.5)]
    np.savez_compressed(os.path.join(T.GOLD, "cv2_int16_seq.npz"), vol=vi, out=T.numpy_seq_sweep(cv2, vi, ks, 0, 5), sigmas=[1.0, 0.5, 0.5],
                        l=0, w=5, semantics="seq", cv2_version=cv2.__version__)
    np.savez_compressed(os.path.join(T.GOLD, "cv2_int16_par.npz"), vol=vi, out=T.numpy_par_sweep(cv2, vi, ks, 0, 5), sigmas=[1.0, 0.5, 0.5],
                        l=0, w=5, semantics="par", cv2_version=cv2.__version__)
    print("written to", T.GOLD)


if __name__ == "__main__":
    main()
