"""The sweeps under the reference's OWN control flow.

tests/golden/ref_sweep_*.npz were written by tests/golden/make_golden.py in the build container: it imported the reference's
flowdenoising_sequential.py ("seq") and flowdenoising.py ("par") and ran seq.OF_filter / seq.OF_filter_along_Z and par's
FlowDenoising(...).filter(kernels) as they stand, on a placeholder `cv2` whose calcOpticalFlowFarneback and remap forward to
the oracle.  So the padding, tap order, chain reset, in-place flow aliasing, numpy's dtype propagation, par's wrap-around
indexing, thread pool and truncating stores are the reference's; only the two cv2 calls (seq:56, seq:62) are the oracle's
-- and those stay unpinned (no cv2 exists here), which every fixture states in its `cv2_calls` field.

CPU tests: the oracle's own sweeps (fdn_oracle.c restates those loops in C) reproduce the fixtures bit for bit.
GPU tests (-m gpu): the HIP path, through the C ABI, does too."""
import glob
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def load(name):
    g = np.load(os.path.join(GOLD, name))
    assert "ORACLE" in str(g["cv2_calls"])           # the fixture says whose cv2 it was
    return g


def test_fixture_set_is_complete():
    names = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLD, "ref_sweep_*.npz")))
    assert names == ["ref_sweep_par_f32_l0_w5.npz", "ref_sweep_par_f32_l0_w5_recompute.npz", "ref_sweep_par_i16_l0_w5.npz",
                     "ref_sweep_par_lv_f32_l3_w5.npz", "ref_sweep_par_u8_l0_w5.npz",
                     "ref_sweep_seq_alongZ_f32_l1_w7.npz", "ref_sweep_seq_f32_l0_w5.npz", "ref_sweep_seq_f32_l1_w7.npz",
                     "ref_sweep_seq_i16_l0_w5.npz", "ref_sweep_seq_lv_f32_l1_w7.npz", "ref_sweep_seq_lv_i16_l1_w7.npz"]
    for n in names:                                   # the outputs are not trivial copies of the inputs
        g = load(n)
        out = g["out"] if "out" in g else g["out_zyx"]
        assert out.shape == g["vol"].shape and np.abs(out.astype(np.float64) - g["vol"]).max() > 1.0


# ---- CPU: the oracle's C restatement of the sweeps against the reference's Python control flow --------------------------
@pytest.mark.parametrize("name", ["ref_sweep_seq_f32_l0_w5.npz", "ref_sweep_seq_f32_l1_w7.npz"])
def test_oracle_OF_filter_equals_seq_control_flow(oracle, name):
    g = load(name)
    ks = [oracle.get_gaussian_kernel(float(s)) for s in g["sigmas"]]
    assert np.array_equal(oracle.OF_filter(g["vol"], ks, int(g["l"]), int(g["w"])), g["out"])


def test_oracle_integer_volume_equals_seq_control_flow(oracle):
    g = load("ref_sweep_seq_i16_l0_w5.npz")
    ks = [oracle.get_gaussian_kernel(float(s)) for s in g["sigmas"]]
    assert np.array_equal(oracle.OF_filter_integer_input(g["vol"], ks, 0, 5), g["out"])


def test_oracle_along_Z_with_a_pyramid_level_equals_seq_control_flow(oracle):
    g = load("ref_sweep_seq_alongZ_f32_l1_w7.npz")
    k = oracle.get_gaussian_kernel(float(g["sigma"]))
    assert np.float32(g["vol"].mean()) == g["mean"]
    assert np.array_equal(oracle.OF_filter_along_Z(g["vol"], k, 1, 7, g["mean"]), g["out"])


@pytest.mark.parametrize("name", ["ref_sweep_par_f32_l0_w5.npz", "ref_sweep_par_f32_l0_w5_recompute.npz"])
def test_oracle_wrap_sweeps_equal_par_control_flow(oracle, name):
    g = load(name)
    ks = [oracle.get_gaussian_kernel(float(s)) for s in g["sigmas"]]
    chained = bool(g["chained"])
    assert np.array_equal(oracle.OF_filter(g["vol"], ks, 0, 5, border_mode=1, chained=chained), g["out_zyx"])
    # what par's main writes is `vol` after the Z and Y passes (par:290 + par:520): the X pass stays in filtered_vol
    assert np.array_equal(oracle.OF_filter(g["vol"], [ks[0], ks[1], None], 0, 5, border_mode=1, chained=chained), g["out_zy"])


@pytest.mark.parametrize("name,dtype", [("ref_sweep_par_i16_l0_w5.npz", np.int16), ("ref_sweep_par_u8_l0_w5.npz", np.uint8)])
def test_oracle_integer_wrap_sweeps_equal_par_control_flow(oracle, name, dtype):
    """int16: cv2.remap in float, rounded and saturated; uint8: cv2.remap's 8-bit fixed point (oracle.remap_any)."""
    g = load(name)
    ks = [oracle.get_gaussian_kernel(float(s)) for s in g["sigmas"]]
    assert g["out_zyx"].dtype == dtype
    assert np.array_equal(oracle.filter_par_integer_input(g["vol"], ks, 0, 5), g["out_zyx"].astype(np.float32))
    assert np.array_equal(oracle.filter_par_integer_input(g["vol"], [ks[0], ks[1], None], 0, 5), g["out_zy"].astype(np.float32))


# ---- round 5: real pyramid levels off the Z axis (64 x 64 x 72 volumes: every pass's images keep a coarser level) ----------
def test_lv_fixtures_really_have_a_level_on_every_axis(oracle):
    """cv2 keeps level k while both image sides x 0.5^k stay >= 32: on these volumes the result of `-l 1`
    differs from `-l 0` in every single pass -- the Y and X passes run a pyramid under the reference's control flow."""
    g = load("ref_sweep_seq_lv_f32_l1_w7.npz")
    vol = g["vol"][:, :, :]
    for axis, sigma in enumerate(g["sigmas"]):
        k = oracle.get_gaussian_kernel(float(sigma))
        sl = [slice(None)] * 3
        sl[axis] = slice(20, 20 + k.size + 1)                         # a few targets are enough (and quick)
        sub = np.ascontiguousarray(vol[tuple(sl)])
        a = oracle.filter_along_axis(sub, axis, k, 1, 7, vol.mean(), nthreads=8)
        b = oracle.filter_along_axis(sub, axis, k, 0, 7, vol.mean(), nthreads=8)
        assert not np.array_equal(a, b), axis


def test_oracle_equals_seq_control_flow_with_levels_on_every_axis(oracle):
    g = load("ref_sweep_seq_lv_f32_l1_w7.npz")
    ks = [oracle.get_gaussian_kernel(float(s)) for s in g["sigmas"]]
    assert np.array_equal(oracle.OF_filter(g["vol"], ks, 1, 7, nthreads=8), g["out"])
    g = load("ref_sweep_seq_lv_i16_l1_w7.npz")
    assert np.array_equal(oracle.OF_filter_integer_input(g["vol"], ks, 1, 7, nthreads=8), g["out"])


def test_oracle_equals_par_control_flow_at_pars_default_levels(oracle):
    g = load("ref_sweep_par_lv_f32_l3_w5.npz")
    ks = [oracle.get_gaussian_kernel(float(s)) for s in g["sigmas"]]
    assert int(g["l"]) == 3
    assert np.array_equal(oracle.OF_filter(g["vol"], ks, 3, 5, border_mode=1, nthreads=8), g["out_zyx"])
    assert np.array_equal(oracle.OF_filter(g["vol"], [ks[0], ks[1], None], 3, 5, border_mode=1, nthreads=8), g["out_zy"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ref_sweep_seq_lv_f32_l1_w7.npz", "ref_sweep_seq_lv_i16_l1_w7.npz"])
def test_hip_equals_seq_control_flow_with_levels_on_every_axis(fdn, name):
    g = load(name)
    ks = [fdn.get_gaussian_kernel(float(s)) for s in g["sigmas"]]
    got = fdn.OF_filter(g["vol"], ks, 1, 7)
    assert got.dtype == np.float32 and np.array_equal(got, g["out"])


@pytest.mark.gpu
def test_hip_FlowDenoising_at_pars_default_levels(fdn):
    g = load("ref_sweep_par_lv_f32_l3_w5.npz")
    ks = [fdn.get_gaussian_kernel(float(s)) for s in g["sigmas"]]
    for kernels, want in ((ks, g["out_zyx"]), ([ks[0], ks[1], None], g["out_zy"])):
        vol = g["vol"].copy()
        fd = fdn.FlowDenoising(3, vol, 3, 5, fdn.get_flow_with_prev_flow, fdn.warp_slice)
        assert fd.filter(kernels) is None
        assert np.array_equal(vol, want)


# ---- GPU: the HIP path against the same fixtures ---------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ref_sweep_seq_f32_l0_w5.npz", "ref_sweep_seq_f32_l1_w7.npz", "ref_sweep_seq_i16_l0_w5.npz"])
def test_hip_OF_filter_equals_seq_control_flow(fdn, name):
    g = load(name)
    ks = [fdn.get_gaussian_kernel(float(s)) for s in g["sigmas"]]
    got = fdn.OF_filter(g["vol"], ks, int(g["l"]), int(g["w"]))          # an int16 array keeps its dtype's semantics (seq:513)
    assert got.dtype == np.float32 and np.array_equal(got, g["out"])


@pytest.mark.gpu
def test_hip_along_Z_with_a_pyramid_level_equals_seq_control_flow(fdn):
    g = load("ref_sweep_seq_alongZ_f32_l1_w7.npz")
    k = fdn.get_gaussian_kernel(float(g["sigma"]))
    assert np.array_equal(fdn.OF_filter_along_Z(g["vol"], k, 1, 7, g["mean"]), g["out"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ref_sweep_par_f32_l0_w5.npz", "ref_sweep_par_f32_l0_w5_recompute.npz", "ref_sweep_par_i16_l0_w5.npz", "ref_sweep_par_u8_l0_w5.npz"])
def test_hip_FlowDenoising_equals_par_control_flow(fdn, name):
    """The class of flowdenoising.py, same constructor and filter(kernels): with all three kernels `vol` holds the full
    Z -> Y -> X result (= par's filtered_vol; the documented deviation from par's lost X pass), with [kz, ky, None] what
    par's main actually keeps (par:520)."""
    g = load(name)
    ks = [fdn.get_gaussian_kernel(float(s)) for s in g["sigmas"]]
    get_flow = fdn.get_flow_with_prev_flow if bool(g["chained"]) else fdn.get_flow_without_prev_flow
    for kernels, want in ((ks, g["out_zyx"]), ([ks[0], ks[1], None], g["out_zy"])):
        vol = g["vol"].copy()
        fd = fdn.FlowDenoising(3, vol, 0, 5, get_flow, fdn.warp_slice)
        assert fd.filter(kernels) is None
        assert vol.dtype == want.dtype and np.array_equal(vol, want)
        assert np.array_equal(fd.filtered_vol, want)
