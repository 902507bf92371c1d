"""The oracle against (a) the goldens generated from the reference itself (tests/golden/
make_golden.py: get_gaussian_kernel and the whole no-OF path, computed by the reference's own
Python) and (b) analytic known-answer tests for the OpenCV restatement, since no
cv2 output is available to pin it (parity unpinned, see oracle/fdn_oracle.c)."""
import os

import numpy as np
import pytest

from conftest import assert_kernel

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _img(rng, H, W, amp=200.0, smooth=2.5):
    import scipy.ndimage
    a = scipy.ndimage.gaussian_filter(rng.standard_normal((H, W)), smooth)
    return (a / np.abs(a).max() * amp).astype(np.float32)


# ---- reference goldens ---------------------------------------------------------------------
def test_kernel_matches_reference(oracle):
    g = np.load(os.path.join(GOLD, "ref_kernels.npz"))
    for i, s in enumerate(g["sigmas"]):
        k = oracle.get_gaussian_kernel(float(s))
        ref = g[f"k{i}"]
        assert k.shape == ref.shape                      # K = 2*int(4 sigma + .5) + 1
        assert_kernel(k, ref, float(s))
        assert abs(k.sum() - 1) < 1e-15


def test_kernel_closed_form_values(oracle):
    k = oracle.get_gaussian_kernel(2.0)
    assert k.size == 17
    assert abs(k[8] - 0.19947464786) < 1e-10 and abs(k[0] - 6.6916289573e-05) < 1e-14


def test_no_of_filter_matches_reference_bit_for_bit(oracle):
    n = np.load(os.path.join(GOLD, "ref_no_of.npz"))
    ks = [oracle.get_gaussian_kernel(float(s)) for s in n["sigmas"]]
    out = oracle.no_OF_filter(n["vol"], ks)
    assert out.dtype == np.float32 and np.array_equal(out, n["out"])


def test_no_of_equals_scipy_correlate(oracle):
    """KAT 2: mean-padded separable sweep == correlate1d(mode='constant', cval=mean) per axis."""
    import scipy.ndimage
    rng = np.random.default_rng(5)
    vol = (rng.standard_normal((9, 11, 13)) * 30 + 50).astype(np.float32)
    ks = [oracle.get_gaussian_kernel(s) for s in (1.0, 0.5, 1.5)]
    want = vol.astype(np.float64)
    for ax, k in enumerate(ks):
        want = scipy.ndimage.correlate1d(want, k, axis=ax, mode="constant", cval=float(vol.mean()))
    got = oracle.no_OF_filter(vol, ks)
    np.testing.assert_allclose(got, want, rtol=2e-6)


# ---- known-answer tests of the OpenCV restatement -----------------------------------------------
def test_polyexp_recovers_quadratic(oracle):
    """KAT 3: I = c + a x + b y + d x^2 + e y^2 + f xy -> channels [dI/dy, dI/dx, e, d, f] away from borders."""
    H, W = 40, 48
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    c0, a, b, d, e, f = 0.7, 0.3, -0.2, 0.05, -0.04, 0.03
    R = oracle.poly_exp((c0 + a * x + b * y + d * x * x + e * y * y + f * x * y).astype(np.float32))
    sl = (slice(6, H - 6), slice(6, W - 6))
    np.testing.assert_allclose(R[sl + (0,)], (b + 2 * e * y + f * x)[sl], atol=2e-4)
    np.testing.assert_allclose(R[sl + (1,)], (a + 2 * d * x + f * y)[sl], atol=2e-4)
    np.testing.assert_allclose(R[sl + (2,)], e, atol=2e-5)
    np.testing.assert_allclose(R[sl + (3,)], d, atol=2e-5)
    np.testing.assert_allclose(R[sl + (4,)], f, atol=2e-5)


def test_polyexp_constants(oracle):
    g, xg, xxg, ig = oracle.polyexp_consts(5, 1.2)
    assert g.size == 11 and abs(float(g.sum()) - 1) < 1e-6
    assert np.allclose(xg, np.arange(-5, 6) * g) and np.allclose(xxg, np.arange(-5, 6) ** 2 * g)
    m2 = float((np.arange(-5, 6) ** 2 * g.astype(np.float64)).sum())
    assert abs(ig[0] - 1 / m2) < 1e-6 * ig[0]            # ig11 = 1 / second moment
    assert abs(ig[3] - 1 / m2 ** 2) < 1e-5 * ig[3]       # ig55 = 1 / m2^2


def test_identical_images_give_zero_flow_in_the_interior(oracle):
    """KAT 4 (h == 0 wherever the bilinear sample is inside the image)."""
    img = _img(np.random.default_rng(0), 96, 80)
    fl = oracle.get_flow(img, img, 0, 5, np.zeros((96, 80, 2), np.float32))
    assert np.all(fl[:96 - 8, :80 - 8] == 0)


def test_translation_is_recovered(oracle):
    """KAT 7: target(p) ~ reference(p + d) (seq:59-67 passes prev=target, next=reference)."""
    import scipy.ndimage
    rng = np.random.default_rng(1)
    img = _img(rng, 128, 128, amp=800.0, smooth=3.0)
    ref = scipy.ndimage.shift(img.astype(np.float64), (0.4, -0.7), order=3, mode="nearest").astype(np.float32)
    for l, w in ((0, 5), (3, 15)):
        fl = oracle.get_flow(ref, img, l, w, np.zeros((128, 128, 2), np.float32))
        c = fl[24:-24, 24:-24]
        assert abs(np.median(c[..., 0]) + 0.7) < 0.05 and abs(np.median(c[..., 1]) - 0.4) < 0.05
        warped = oracle.warp_slice(ref, fl)
        assert np.abs(warped - img)[24:-24, 24:-24].mean() < 0.25 * np.abs(ref - img)[24:-24, 24:-24].mean()


def test_warp_quantisation(oracle):
    """KAT 6: zero flow = identity, integer flow = clamped shift, 1/64 px rounds half-to-even."""
    img = _img(np.random.default_rng(2), 20, 24)
    z = np.zeros((20, 24, 2), np.float32)
    assert np.array_equal(oracle.warp_slice(img, z), img)
    f = z.copy(); f[..., 0] = 3
    assert np.array_equal(oracle.warp_slice(img, f), img[:, np.clip(np.arange(24) + 3, 0, 23)])
    f = z.copy(); f[..., 0] = 1 / 64   # sx = x*32 + 0.5 -> ties to even: stays x*32
    assert np.array_equal(oracle.warp_slice(img, f), img)
    f = z.copy(); f[..., 0] = 3 / 64   # 1.5/32 -> rounds to 2/32
    g = z.copy(); g[..., 0] = 2 / 32
    assert np.array_equal(oracle.warp_slice(img, f), oracle.warp_slice(img, g))
    # weights: a = 16/32 in x gives the mean of two neighbours
    f = z.copy(); f[..., 0] = 0.5
    want = (img * np.float32(0.5) + img[:, np.clip(np.arange(24) + 1, 0, 23)] * np.float32(0.5))
    np.testing.assert_array_equal(oracle.warp_slice(img, f), want)


def test_box_filter_of_constant_matrices(oracle):
    """KAT 8: constant M -> the box filter returns M (replicate borders, 1/w^2 scale), so the flow is
    the closed-form 2x2 solve everywhere; even w uses a (w+1)-wide window with the same scale."""
    H, W = 24, 30
    M = np.empty((H, W, 5), np.float32)
    M[...] = (4.0, 1.0, 3.0, 0.5, -0.25)
    R = np.zeros((H, W, 5), np.float32)
    for w, gain in ((5, 1.0), (7, 1.0), (4, (5 / 4) ** 2)):
        g11, g12, g22, h1, h2 = (gain * v for v in (4.0, 1.0, 3.0, 0.5, -0.25))
        idet = 1 / (g11 * g22 - g12 * g12 + 1e-3)
        fl, _ = oracle.update_flow_blur(R, R, np.zeros((H, W, 2), np.float32), M, w, False)
        np.testing.assert_allclose(fl[..., 0], (g11 * h2 - g12 * h1) * idet, rtol=1e-6)
        np.testing.assert_allclose(fl[..., 1], (g22 * h1 - g12 * h2) * idet, rtol=1e-6)


def test_running_vs_direct_box_sums(oracle):
    """The f32-fed running sum (OpenCV) and an exact window sum agree to ~1e-5 px on ordinary data but
    not to 1e-7: this is why the HIP kernels carry the running sum instead of a tile-local box filter."""
    import scipy.ndimage
    rng = np.random.default_rng(3)
    a = _img(rng, 128, 96)
    b = scipy.ndimage.shift(a.astype(np.float64), (0.5, 0.3), order=3, mode="nearest").astype(np.float32)
    z = np.zeros((128, 96, 2), np.float32)
    f0 = oracle.get_flow(b, a, 0, 5, z.copy(), box_mode=oracle.BOX_RUNNING)
    f1 = oracle.get_flow(b, a, 0, 5, z.copy(), box_mode=oracle.BOX_DIRECT)
    d = np.abs(f0 - f1).max()
    assert 0 < d < 1e-2


@pytest.mark.parametrize("w", [3, 5, 7, 10, 11, 15, 21, 31, 49])
def test_horizontal_summation_orders_agree(oracle, w):
    """The oracle's horizontal-window orders -- OpenCV's running sum (0), left to right (2), by doubling (3), blocks of
    floor(sqrt(w)) columns (4, the one-iteration HIP kernel's order) -- are the same sum: bit-identical flows when the
    f64 sums cannot round (small-integer matrices), and equal to f64 rounding on ordinary data."""
    rng = np.random.default_rng(100 + w)
    H, W = 40, 150
    R = rng.standard_normal((H, W, 5)).astype(np.float32)
    z = np.zeros((H, W, 2), np.float32)
    Mi = rng.integers(-50, 50, (H, W, 5)).astype(np.float32)
    Mi[..., 0] += 200; Mi[..., 2] += 200           # keep the 2 x 2 systems well conditioned
    Mr = Mi + rng.standard_normal((H, W, 5)).astype(np.float32)
    modes = (oracle.BOX_RUNNING, oracle.BOX_VRUN_HDIRECT, oracle.BOX_VRUN_HDOUBLING, oracle.BOX_VRUN_HBLOCKS)
    exact = [oracle.update_flow_blur(R, R, z, Mi, w, False, box_mode=m)[0] for m in modes]
    for f in exact[1:]:
        assert np.array_equal(f, exact[0])
    close = [oracle.update_flow_blur(R, R, z, Mr, w, False, box_mode=m)[0] for m in modes]
    for f in close[1:]:
        np.testing.assert_allclose(f, close[0], rtol=0, atol=1e-6 * np.abs(close[0]).max())


def test_constant_volume(oracle):
    """KAT 5: constant volume v (pad = v) -> v * sum(w) per pass."""
    vol = np.full((5, 34, 36), 12.5, np.float32)
    ks = [oracle.get_gaussian_kernel(0.5)] * 3
    np.testing.assert_allclose(oracle.OF_filter(vol, ks, 0, 5), 12.5, rtol=1e-6)


def test_gaussian_blur_and_resize(oracle):
    img = _img(np.random.default_rng(4), 33, 40)
    # sigma 0, size 3 -> [.25 .5 .25] separably with reflect-101 borders
    k = np.array([0.25, 0.5, 0.25])
    p = np.pad(img.astype(np.float64), 1, mode="reflect")
    want = sum(k[i] * k[j] * p[i:i + 33, j:j + 40] for i in range(3) for j in range(3))
    np.testing.assert_allclose(oracle.gaussian_blur(img, 3, 0.0), want, rtol=1e-5, atol=1e-4)
    # constant images survive any blur / resize
    c = np.full((37, 45), 3.25, np.float32)
    np.testing.assert_allclose(oracle.gaussian_blur(c, 9, 1.5), 3.25, rtol=1e-6)
    np.testing.assert_allclose(oracle.resize(c, 18, 22, oracle.INTER_LINEAR), 3.25, rtol=1e-6)
    np.testing.assert_allclose(oracle.resize(c, 9, 11, oracle.INTER_AREA), 3.25, rtol=1e-5)
    # exact 2x shrink = 2x2 block mean (also what INTER_LINEAR is promoted to)
    e = img[:32, :40]
    bm = e.reshape(16, 2, 20, 2).mean(axis=(1, 3))
    np.testing.assert_allclose(oracle.resize(e, 16, 20, oracle.INTER_AREA), bm, rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(oracle.resize(e, 16, 20, oracle.INTER_LINEAR), bm, rtol=1e-5, atol=1e-4)


def test_wrap_mode_is_roll_equivariant(oracle):
    """par:312 wrap-around: rolling the volume along the sweep axis rolls the result."""
    from flowdenoising_amd.synth import make_volume
    vol = make_volume((7, 34, 36), seed=2, amplitude=100.0)
    k = oracle.get_gaussian_kernel(0.5)
    a = oracle.filter_along_axis(vol, 0, k, 0, 5, 0.0, border_mode=1)
    b = oracle.filter_along_axis(np.roll(vol, 3, axis=0), 0, k, 0, 5, 0.0, border_mode=1)
    assert np.array_equal(np.roll(a, 3, axis=0), b)


def test_threads_do_not_change_results(oracle):
    from flowdenoising_amd.synth import make_volume
    vol = make_volume((6, 34, 36), seed=4, amplitude=100.0)
    k = oracle.get_gaussian_kernel(0.5)
    a = oracle.filter_along_axis(vol, 1, k, 0, 5, vol.mean(), nthreads=1)
    b = oracle.filter_along_axis(vol, 1, k, 0, 5, vol.mean(), nthreads=4)
    assert np.array_equal(a, b)


def _int_volume(shape, seed):
    from flowdenoising_amd.synth import make_volume
    v = make_volume(shape, seed=seed, amplitude=100.0)
    lo, hi = float(v.min()), float(v.max())
    return np.round((v - lo) / (hi - lo) * 4095).astype(np.int16)


def test_integer_volume_semantics_are_not_the_float32_ones(oracle):
    """seq on an integer MRC pads and remaps in float64 (seq:420, seq:88-89): against the same numbers converted to
    float32 first, single results move by 1e-4 of the range -- a last-bit difference of one pass is amplified by the
    next pass's flows -- which is why the HIP library implements the integer semantics instead of converting
    (tests/test_gpu_integer.py).  Without optical flow the two differ by float rounding only."""
    vi = _int_volume((10, 48, 56), 11)
    ks = [oracle.get_gaussian_kernel(s) for s in (1.0, 1.0, 0.5)]
    a = oracle.OF_filter(vi.astype(np.float32), ks, 0, 5, nthreads=8)
    b = oracle.OF_filter_integer_input(vi, ks, 0, 5, nthreads=8)
    d = np.abs(a - b).max() / np.abs(b).max()
    assert 1e-6 < d < 1e-2
    a = oracle.no_OF_filter(vi.astype(np.float32), ks)
    b = oracle.OF_filter_integer_input(vi, ks, 0, 5, use_of=False)
    assert np.abs(a - b).max() / np.abs(b).max() < 5e-7


def test_integer_volume_known_answers(oracle):
    """Constant integer volume: seq gives c * sum(taps) per pass (the float64 pad is c too); par's wrap-around passes
    leave every voxel at c exactly (remap of a constant integer image is that integer, the taps sum to 1 +- 1e-16 and
    the truncation must not drop it to c - 1 ... which it does when the float32 sum lands just below c: the reference
    has that bias, and the restatement keeps it)."""
    vol = np.full((5, 34, 36), 1000, np.int16)
    ks = [oracle.get_gaussian_kernel(0.5)] * 3
    out = oracle.OF_filter_integer_input(vol, ks, 0, 5)
    np.testing.assert_allclose(out, 1000.0, rtol=1e-6)
    par = oracle.filter_par_integer_input(vol, ks, 0, 5)
    assert np.array_equal(par, np.trunc(par)) and np.all((par == 1000) | (par == 999))


def test_fma_modes_of_the_pyramid(oracle):
    """fdo_set_fma (the product's "opencv_fma" option): mode 2 is mode 0 when a row holds less than one vector, mode 1 when
    every element is in the vector body; on rows that are not a multiple of the lane count it is neither; at levels = 0
    (blur taps 1/4, 1/2, 1/4) no mode changes anything."""
    rng = np.random.default_rng(3)
    a = (rng.standard_normal((70, 150)) * 50).astype(np.float32)      # levels: 35 x 75, then 18 x 38 (both with row tails at 8 lanes)
    b = (np.roll(a, 1, axis=1) + rng.standard_normal(a.shape).astype(np.float32)).astype(np.float32)
    f0 = np.zeros(a.shape + (2,), np.float32)
    got = {}
    try:
        for key, (mode, lanes) in {"plain": (0, 8), "fused": (1, 8), "body8": (2, 8), "body1": (2, 1), "body4096": (2, 4096)}.items():
            oracle.set_fma(mode, lanes)
            got[key] = oracle.get_flow(b, a, 2, 5, f0.copy())
            got[key + "_l0"] = oracle.get_flow(b, a, 0, 5, f0.copy())
    finally:
        oracle.set_fma(0)
    assert np.array_equal(got["body1"], got["fused"]) and np.array_equal(got["body4096"], got["plain"])
    assert not np.array_equal(got["fused"], got["plain"])
    assert not np.array_equal(got["body8"], got["plain"]) and not np.array_equal(got["body8"], got["fused"])
    assert all(np.array_equal(got[k + "_l0"], got["plain_l0"]) for k in ("fused", "body8", "body1"))
    assert np.abs(got["fused"] - got["plain"]).max() < 1e-2            # a last-bit effect amplified by the solve, not another algorithm


def test_remap_model_switch(oracle):
    """fdo_set_remap_model (the product's "remap_model" option): model 1 is plain float32 bilinear interpolation at the map
    position -- two lerps, no 1/32-pixel table -- on float32, float64 and 16-bit images; 8-bit images keep their fixed-point table."""
    rng = np.random.default_rng(4)
    H, W = 37, 53
    src = (rng.standard_normal((H, W)) * 100).astype(np.float32)
    flow = (rng.standard_normal((H, W, 2)) * 1.7).astype(np.float32)
    flow[0, 0] = (-5.0, -7.0)
    flow[-1, -1] = (9.0, 3.0)                                           # clamped taps (BORDER_REPLICATE)
    mx = (flow[..., 0].astype(np.float64) + np.arange(W)[None, :]).astype(np.float32)
    my = (flow[..., 1].astype(np.float64) + np.arange(H)[:, None]).astype(np.float32)
    m = np.stack([mx, my], axis=-1)
    x0, y0 = np.floor(mx), np.floor(my)
    fx, fy = (mx - x0).astype(np.float32), (my - y0).astype(np.float32)
    xa, xb = np.clip(x0.astype(int), 0, W - 1), np.clip(x0.astype(int) + 1, 0, W - 1)
    ya, yb = np.clip(y0.astype(int), 0, H - 1), np.clip(y0.astype(int) + 1, 0, H - 1)
    one = np.float32(1)

    def lerp(s, dt):
        s = s.astype(dt)
        wx0, wx1, wy0, wy1 = (one - fx).astype(dt), fx.astype(dt), (one - fy).astype(dt), fy.astype(dt)
        return (s[ya, xa] * wx0 + s[ya, xb] * wx1) * wy0 + (s[yb, xa] * wx0 + s[yb, xb] * wx1) * wy1
    classic = oracle.warp_slice(src, flow)
    try:
        oracle.set_remap_model(1)
        assert np.array_equal(oracle.warp_slice(src, flow), lerp(src, np.float32))
        assert np.array_equal(oracle.remap_any(src.astype(np.float64), m), lerp(src, np.float64))
        i16 = np.round(src).astype(np.int16)
        assert np.array_equal(oracle.remap_any(i16, m), np.clip(np.rint(lerp(i16, np.float32)), -32768, 32767).astype(np.int16))
        u8 = np.clip(np.round(src + 128), 0, 255).astype(np.uint8)
        unq_u8 = oracle.remap_any(u8, m)
    finally:
        oracle.set_remap_model(0)
    assert np.array_equal(unq_u8, oracle.remap_any(u8, m))              # fixed point either way
    assert not np.array_equal(classic, lerp(src, np.float32)) and np.abs(classic - lerp(src, np.float32)).max() < 0.05 * np.abs(src).max()
