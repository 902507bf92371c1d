#!/usr/bin/env python3
"""Generate the reference-derived golden vectors in tests/golden/.

Runs ONLY in the build container (needs /root/reference).  It imports the reference's
src/flowdenoising_sequential.py ("seq") and src/flowdenoising.py ("par") and runs THEIR functions; nothing of
the reference travels: the outputs are numeric arrays (inputs and expected outputs) in .npz files.

The reference's top-level imports of cv2 / mrcfile / skimage / tifffile / imageio (absent from this image; an
ordinary ModuleNotFoundError, not a refusal) are satisfied by placeholder modules.

Part 1 -- reference arithmetic only (EMPTY placeholders: three integer constants, no functionality):
  ref_kernels.npz   sigma list + kernels from seq.get_gaussian_kernel (seq:30-41)
  ref_no_of.npz     seeded input volume, sigmas, seq.no_OF_filter output (seq:426-431 -> 171-192, 290-311, 396-417)
  Every number there was computed by the reference's own Python + numpy + scipy.

Part 2 -- the reference's own CONTROL FLOW around the two OpenCV calls:
  ref_sweep_*.npz   seq.OF_filter / seq.OF_filter_along_Z (seq:78-130, 235-288, 313-364, 419-424) and par's
                    FlowDenoising(...).filter(kernels) (par:285-290, 299-373) run as they stand -- padding, tap order,
                    chain reset, in-place flow aliasing, numpy's dtype propagation, the thread pool, the wrap-around
                    indexing, the truncating store into an integer volume -- on a placeholder `cv2` whose
                    calcOpticalFlowFarneback and remap FORWARD TO THE ORACLE (oracle/oracle.py: the CPU restatement of
                    OpenCV's two routines).  Every file says so in its `cv2_calls` field.  These fixtures therefore pin
                    everything of rows a-5 ... a-8 and a-10 EXCEPT the two cv2 calls (seq:56, seq:62), which stay
                    unpinned ("parity partial") until a box has real cv2.
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/src/flowdenoising_sequential.py"
REF_PAR = "/root/reference/src/flowdenoising.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_seq():
    cv2 = types.ModuleType("cv2")
    cv2.BORDER_REPLICATE = 1
    cv2.INTER_LINEAR = 1
    cv2.OPTFLOW_USE_INITIAL_FLOW = 4
    placeholders = {"cv2": cv2}
    for name in ("mrcfile", "skimage", "skimage.io", "tifffile", "imageio"):
        placeholders[name] = types.ModuleType(name)
    placeholders["skimage"].io = placeholders["skimage.io"]
    saved = {k: sys.modules.get(k) for k in placeholders}
    sys.modules.update(placeholders)
    try:
        spec = importlib.util.spec_from_file_location("fd_seq_reference", REF)
        seq = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(seq)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return seq


class OracleCv2(types.ModuleType):
    """Placeholder `cv2` for part 2: the two calls the reference makes, forwarded to the oracle (which dispatches on the
    image depth like cv2: convertTo(CV_32F) inside Farneback; remap per depth, returning the image's type)."""

    def __init__(self):
        super().__init__("cv2")
        self.BORDER_REPLICATE, self.INTER_LINEAR, self.OPTFLOW_USE_INITIAL_FLOW = 1, 1, 4
        sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
        from oracle import oracle
        oracle.build()
        self._o = oracle

    def calcOpticalFlowFarneback(self, prev, next, flow, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags):
        return self._o.calcOpticalFlowFarneback(np.asarray(prev, np.float32), np.asarray(next, np.float32), flow, pyr_scale,
                                                levels, winsize, iterations, poly_n, poly_sigma, flags)

    def remap(self, src, map1, map2, interpolation, borderMode):
        assert map2 is None and interpolation == self.INTER_LINEAR and borderMode == self.BORDER_REPLICATE
        return self._o.remap_any(src, map1)


def load_reference(path, name, cv2):
    placeholders = {"cv2": cv2}
    for n in ("mrcfile", "skimage", "skimage.io", "tifffile", "imageio"):
        placeholders[n] = types.ModuleType(n)
    placeholders["skimage"].io = placeholders["skimage.io"]
    saved = {k: sys.modules.get(k) for k in placeholders}
    sys.modules.update(placeholders)
    try:
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def volume(shape, seed, dtype=np.float32):
    """A structured test volume (drifting blobs + noise; the package's own generator) in the dtype asked for."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from flowdenoising_amd.synth import make_volume
    v = make_volume(shape, seed=seed, amplitude=100.0)
    if dtype == np.uint8:
        lo, hi = float(v.min()), float(v.max())
        return np.round((v - lo) / (hi - lo) * 255).astype(np.uint8)
    if np.issubdtype(dtype, np.integer):
        lo, hi = float(v.min()), float(v.max())
        return (np.round((v - lo) / (hi - lo) * 4095) - 1000).astype(dtype)
    return v.astype(dtype)


CV2_NOTE = ("cv2.calcOpticalFlowFarneback and cv2.remap were the ORACLE's (oracle/oracle.py) behind a placeholder module; everything else "
            "-- control flow, numpy, scipy -- is the reference's own code run as it stands")


def sweeps():
    cv2 = OracleCv2()
    seq = load_reference(REF, "fd_seq_reference_cv", cv2)
    par = load_reference(REF_PAR, "fd_par_reference_cv", cv2)
    sig = [1.0, 1.0, 0.5]
    ks = [seq.get_gaussian_kernel(s) for s in sig]

    def save(name, **kw):
        np.savez(os.path.join(HERE, name), cv2_calls=np.array(CV2_NOTE), **kw)
        print(name, {k: (v.shape, str(v.dtype)) for k, v in kw.items() if hasattr(v, "shape") and v.ndim})

    # --- seq.OF_filter: float32 volumes, the default window and a wider one with a pyramid level asked for -------------
    vol = volume((12, 40, 48), 101)
    for l, w in ((0, 5), (1, 7)):
        out = seq.OF_filter(vol.copy(), ks, l, w)
        assert out.dtype == np.float32
        save(f"ref_sweep_seq_f32_l{l}_w{w}.npz", vol=vol, sigmas=np.array(sig), l=np.array(l), w=np.array(w), out=out)
    # --- seq.OF_filter on an int16 volume: float64 mean -> float64 padded volume in all three passes (seq:88-89, 420) ----
    vi = volume((12, 40, 48), 102, np.int16)
    out = seq.OF_filter(vi.copy(), ks, 0, 5)
    assert out.dtype == np.float32
    save("ref_sweep_seq_i16_l0_w5.npz", vol=vi, sigmas=np.array(sig), l=np.array(0), w=np.array(5), out=out)
    # --- seq.OF_filter_along_Z with a real pyramid level (images of 64 x 80: level 1 exists, cv2's >= 32 rule) ----------
    vz = volume((6, 64, 80), 103)
    mean = vz.mean()
    out = seq.OF_filter_along_Z(vz.copy(), ks[0], 1, 7, mean)
    save("ref_sweep_seq_alongZ_f32_l1_w7.npz", vol=vz, sigma=np.array(sig[0]), l=np.array(1), w=np.array(7), mean=np.float32(mean), out=out)

    # --- par: FlowDenoising(P, vol, l, w, get_flow, warp_slice).filter(kernels) (par:285-290, 299-373) ------------------
    # module globals its methods read: l, w (par:313), get_flow (par:366), vol (par:268), args.input (par:127)
    def run_par(v, l, w, get_flow):
        par.args = types.SimpleNamespace(input="fixture")
        par.l, par.w, par.get_flow, par.vol = l, w, get_flow, v
        fd = par.FlowDenoising(3, v, l, w, get_flow, par.warp_slice)
        assert fd.filter(ks) is None                     # par:285-290 returns nothing
        # what par's main keeps is `vol` (par:520): the Z and Y passes; the X pass stays behind in filtered_vol
        return fd.vol.copy(), fd.filtered_vol.copy()

    vp = volume((12, 40, 48), 104)
    zy, zyx = run_par(vp.copy(), 0, 5, par.get_flow_with_prev_flow)
    save("ref_sweep_par_f32_l0_w5.npz", vol=vp, sigmas=np.array(sig), l=np.array(0), w=np.array(5), chained=np.array(1), out_zy=zy, out_zyx=zyx)
    zy, zyx = run_par(vp.copy(), 0, 5, par.get_flow_without_prev_flow)
    save("ref_sweep_par_f32_l0_w5_recompute.npz", vol=vp, sigmas=np.array(sig), l=np.array(0), w=np.array(5), chained=np.array(0), out_zy=zy, out_zyx=zyx)
    vpi = volume((12, 40, 48), 105, np.int16)
    zy, zyx = run_par(vpi.copy(), 0, 5, par.get_flow_with_prev_flow)
    assert zy.dtype == np.int16 and zyx.dtype == np.int16      # par:131: results live in arrays of the input's dtype
    save("ref_sweep_par_i16_l0_w5.npz", vol=vpi, sigmas=np.array(sig), l=np.array(0), w=np.array(5), chained=np.array(1), out_zy=zy, out_zyx=zyx)
    # a uint8 volume: cv2.remap interpolates 8-bit images in fixed point (the oracle's remap_any restates FixedPtCast<int, uchar, 15>)
    vpu = volume((12, 40, 48), 106, np.uint8)
    zy, zyx = run_par(vpu.copy(), 0, 5, par.get_flow_with_prev_flow)
    assert zy.dtype == np.uint8 and zyx.dtype == np.uint8
    save("ref_sweep_par_u8_l0_w5.npz", vol=vpu, sigmas=np.array(sig), l=np.array(0), w=np.array(5), chained=np.array(1), out_zy=zy, out_zyx=zyx)


def sweeps_with_levels():
    """Round 5: real pyramid levels OFF the Z axis under the reference's control flow.  cv2 keeps level
    k only while both image sides times 0.5^k stay >= 32, so every image of a pass needs sides >= 64: a 64 x 64 x 72 volume
    gives the Y pass (Z x X = 64 x 72) and the X pass (Z x Y = 64 x 64) one coarser level, like the Z pass.
      ref_sweep_seq_lv_f32_l1_w7.npz   seq.OF_filter, float32, -l 1 -w 7
      ref_sweep_seq_lv_i16_l1_w7.npz   seq.OF_filter on an int16 volume (float64 padded volume), -l 1 -w 7
      ref_sweep_par_lv_f32_l3_w5.npz   par's FlowDenoising(...).filter(kernels) at par's own default l = 3 (par:48), w = 5"""
    cv2 = OracleCv2()
    seq = load_reference(REF, "fd_seq_reference_cv", cv2)
    par = load_reference(REF_PAR, "fd_par_reference_cv", cv2)
    sig = [1.0, 0.5, 1.0]
    ks = [seq.get_gaussian_kernel(s) for s in sig]

    def save(name, **kw):
        np.savez_compressed(os.path.join(HERE, name), cv2_calls=np.array(CV2_NOTE), **kw)
        print(name, {k: (v.shape, str(v.dtype)) for k, v in kw.items() if hasattr(v, "shape") and v.ndim})

    shape = (64, 64, 72)
    vol = volume(shape, 111)
    out = seq.OF_filter(vol.copy(), ks, 1, 7)
    assert out.dtype == np.float32
    save("ref_sweep_seq_lv_f32_l1_w7.npz", vol=vol, sigmas=np.array(sig), l=np.array(1), w=np.array(7), out=out)
    vi = volume(shape, 112, np.int16)
    out = seq.OF_filter(vi.copy(), ks, 1, 7)
    assert out.dtype == np.float32
    save("ref_sweep_seq_lv_i16_l1_w7.npz", vol=vi, sigmas=np.array(sig), l=np.array(1), w=np.array(7), out=out)

    par.args = types.SimpleNamespace(input="fixture")
    vp = volume(shape, 113)
    v = vp.copy()
    par.l, par.w, par.get_flow, par.vol = 3, 5, par.get_flow_with_prev_flow, v
    fd = par.FlowDenoising(3, v, 3, 5, par.get_flow_with_prev_flow, par.warp_slice)
    assert fd.filter(ks) is None
    save("ref_sweep_par_lv_f32_l3_w5.npz", vol=vp, sigmas=np.array(sig), l=np.array(3), w=np.array(5), chained=np.array(1),
         out_zy=fd.vol.copy(), out_zyx=fd.filtered_vol.copy())


def main():
    seq = load_seq()
    sigmas = [0.1, 0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 4.0]
    kernels = {f"k{i}": seq.get_gaussian_kernel(s) for i, s in enumerate(sigmas)}
    np.savez(os.path.join(HERE, "ref_kernels.npz"), sigmas=np.array(sigmas), **kernels)

    rng = np.random.default_rng(20261003)
    vol = (rng.standard_normal((12, 10, 14)) * 50 + 100).astype(np.float32)
    sig = [1.0, 1.5, 0.5]
    ks = [seq.get_gaussian_kernel(s) for s in sig]
    seq.l, seq.w = 0, 5  # module globals the no_OF log lines read (seq:173)
    out = seq.no_OF_filter(vol, ks)
    assert out.dtype == np.float32
    np.savez(os.path.join(HERE, "ref_no_of.npz"), vol=vol, sigmas=np.array(sig), out=out,
             mean=np.float32(vol.mean()))
    print("kernels:", [len(k) for k in kernels.values()], " no_OF out:", out.shape, out.dtype)


if __name__ == "__main__":
    if "--levels-only" in sys.argv:
        sweeps_with_levels()
    else:
        if "--sweeps-only" not in sys.argv:
            main()
        sweeps()
        sweeps_with_levels()
