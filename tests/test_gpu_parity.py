"""GPU parity tests: the HIP path (through the C ABI, via flowdenoising_amd) against the CPU
oracle on the same seeded inputs, and against the reference-generated goldens.

Tolerance (float32 path).  BASELINE.json asks for "within 1e-4 relative of the sequential
reference"; the kernels reproduce the oracle's arithmetic operation by operation (including
OpenCV's f32-fed vertical running sums), so the tests hold them to a much tighter bar:
  TIGHT_TOL = 2e-6  max|gpu - oracle| / max|oracle| (flows: / max(|flow|, 1 px)).
The only deliberate differences are f64 summation orders (horizontal box sum, mean), ~1e-16.
"""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

TIGHT_TOL = 2e-6


def _img(rng, H, W, amp=200.0, smooth=2.0):
    import scipy.ndimage
    a = scipy.ndimage.gaussian_filter(rng.standard_normal((H, W)), smooth)
    return (a / np.abs(a).max() * amp).astype(np.float32)


from conftest import assert_kernel as _assert_kernel


def test_gaussian_kernel_matches_reference_golden(fdn):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_kernels.npz"))
    for i, s in enumerate(g["sigmas"]):
        k = fdn.get_gaussian_kernel(float(s))
        assert k.shape == g[f"k{i}"].shape
        _assert_kernel(k, g[f"k{i}"], float(s))


def test_warp_bit_exact(fdn, oracle):
    rng = np.random.default_rng(1)
    H, W = 37, 53
    img = _img(rng, H, W)
    flow = (rng.standard_normal((H, W, 2)) * 3).astype(np.float32)
    flow[0, 0] = (1 / 64, 0)        # exactly half a quantisation step: round-half-even
    flow[0, 1] = (3 / 64, -1 / 64)
    flow[1, 0] = (-100, 100)        # far outside: clamped taps
    flow[2, 2] = (0, 0)
    got = fdn.warp_slice(img, flow)
    want = oracle.warp_slice(img, flow)
    assert np.array_equal(got, want)


def test_warp_zero_and_integer_flow(fdn):
    rng = np.random.default_rng(2)
    img = _img(rng, 24, 40)
    assert np.array_equal(fdn.warp_slice(img, np.zeros((24, 40, 2), np.float32)), img)
    flow = np.zeros((24, 40, 2), np.float32)
    flow[..., 0] = 2
    flow[..., 1] = -1
    got = fdn.warp_slice(img, flow)
    yy = np.clip(np.arange(24) - 1, 0, 23)
    xx = np.clip(np.arange(40) + 2, 0, 39)
    assert np.array_equal(got, img[yy][:, xx])


@pytest.mark.parametrize("shape", [(64, 96), (33, 47), (130, 70)])
@pytest.mark.parametrize("w", [5, 7, 4])
def test_farneback_pair(fdn, oracle, shape, w):
    rng = np.random.default_rng(3)
    H, W = shape
    import scipy.ndimage
    a = _img(rng, H, W)
    b = scipy.ndimage.shift(a.astype(np.float64), (0.6, -0.4), order=3, mode="nearest").astype(np.float32)
    b += (rng.standard_normal((H, W)) * 2).astype(np.float32)
    for init in ("zero", "random"):
        f0 = np.zeros((H, W, 2), np.float32) if init == "zero" else (rng.standard_normal((H, W, 2)) * 0.5).astype(np.float32)
        got = fdn.get_flow(b, a, 0, w, f0.copy())
        want = oracle.get_flow(b, a, 0, w, f0.copy())
        scale = max(np.abs(want).max(), 1.0)
        assert np.abs(got - want).max() / scale < TIGHT_TOL, (init, np.abs(got - want).max())
    # without initial flow (par:89-114)
    got = fdn.get_flow_without_prev_flow(b, a, 0, w)
    want = oracle.calcOpticalFlowFarneback(a, b, None, 0.5, 0, w, 3, 5, 1.2, 0)
    assert np.abs(got - want).max() < TIGHT_TOL * max(np.abs(want).max(), 1.0)


@pytest.mark.parametrize("shape,l", [((128, 160), 1), ((130, 151), 2), ((256, 300), 3), ((97, 300), 3)])
@pytest.mark.parametrize("w", [5, 15])
def test_farneback_pyramid(fdn, oracle, shape, l, w):
    """levels > 0 (par's default is 3; BASELINE configs[4] uses -l 3 -w 15): Gaussian pyramid with
    INTER_LINEAR images, INTER_AREA initial flow (non-integer ratios for odd sizes), level cropping
    at 32 pixels."""
    rng = np.random.default_rng(6)
    H, W = shape
    import scipy.ndimage
    a = _img(rng, H, W, smooth=4.0)
    b = scipy.ndimage.shift(a.astype(np.float64), (2.3, -1.6), order=3, mode="nearest").astype(np.float32)
    for init in ("zero", "random"):
        f0 = np.zeros((H, W, 2), np.float32) if init == "zero" else (rng.standard_normal((H, W, 2)) * 0.7).astype(np.float32)
        got = fdn.get_flow(b, a, l, w, f0.copy())
        want = oracle.get_flow(b, a, l, w, f0.copy())
        assert np.abs(got - want).max() / max(np.abs(want).max(), 1.0) < TIGHT_TOL, (init, np.abs(got - want).max())
    got = fdn.get_flow_without_prev_flow(b, a, l, w)
    want = oracle.calcOpticalFlowFarneback(a, b, None, 0.5, l, w, 3, 5, 1.2, 0)
    assert np.abs(got - want).max() / max(np.abs(want).max(), 1.0) < TIGHT_TOL


def test_of_filter_with_pyramid(fdn, oracle):
    vol = _vol((6, 70, 150), seed=13)
    k = fdn.get_gaussian_kernel(1.0)
    got = fdn.OF_filter_along_Z(vol, k, 3, 15, vol.mean())
    want = oracle.filter_along_axis(vol, 0, k, 3, 15, vol.mean(), nthreads=8)
    assert rel_err(got, want) < TIGHT_TOL
    got = fdn.OF_filter(vol, [None, None, k], 2, 5)      # X pass: images 6 x 70 -> no level survives the 32-px crop
    want = oracle.OF_filter(vol, [None, None, k], 2, 5, nthreads=8)
    assert rel_err(got, want) < TIGHT_TOL


@pytest.mark.parametrize("shape,l", [((7, 70, 150), 3), ((6, 131, 97), 2), ((5, 64, 200), 1)])
def test_of_filter_pyramid_on_the_fused_kernel(fdn, oracle, shape, l):
    """par's default (-l 3 -w 5): every pyramid level of every chain step is one launch of the fused
    kernel (flow only on the coarser levels), with the INTER_AREA shrink of the previous step's flow
    and the INTER_LINEAR upsampling between them; odd sizes give non-integer area ratios."""
    vol = _vol(shape, seed=21)
    k = fdn.get_gaussian_kernel(1.0)      # K = 9: chains of four steps either side
    got = fdn.OF_filter_along_Z(vol, k, l, 5, vol.mean())
    want = oracle.filter_along_axis(vol, 0, k, l, 5, vol.mean(), nthreads=8)
    assert rel_err(got, want) < TIGHT_TOL


def test_pair_entry_points_take_views_and_device_pointers(fdn, oracle):
    """The reference hands cv2 VIEWS of the padded volume: padded_vol[:, y + i, :] (seq:255) and
    padded_vol[:, :, x + i] (seq:333).  fdn_farneback_strided / fdn_warp_strided take them as they are
    (pointer + element strides, gathered into the handle's pinned staging); fdn_farneback_dev / fdn_warp_dev do
    the same on a volume that already lives in HBM, flow included, without any host round trip."""
    torch = pytest.importorskip("torch")
    from flowdenoising_amd.operators import handle
    vol = _vol((40, 44, 52), seed=77)
    h = handle()
    d_vol = torch.from_numpy(vol).cuda()
    for name, tgt, ref, dt, dr in [
            ("Y", vol[:, 20, :], vol[:, 21, :], d_vol[:, 20, :], d_vol[:, 21, :]),      # row stride Y*X, column stride 1
            ("X", vol[:, :, 30], vol[:, :, 31], d_vol[:, :, 30], d_vol[:, :, 31])]:    # row stride Y*X, column stride X
        assert not tgt.flags["C_CONTIGUOUS"]
        H, W = tgt.shape
        f0 = (np.random.default_rng(5).standard_normal((H, W, 2)) * 0.3).astype(np.float32)
        want = oracle.get_flow(np.ascontiguousarray(ref), np.ascontiguousarray(tgt), 0, 5, f0.copy())
        got = fdn.get_flow(ref, tgt, 0, 5, f0.copy())                                   # views in, no copy on this side
        assert np.array_equal(got, want), name
        assert np.array_equal(fdn.warp_slice(ref, got), oracle.warp_slice(np.ascontiguousarray(ref), want)), name
        d_flow = torch.from_numpy(f0).cuda()
        torch.cuda.synchronize()
        h.farneback_dev(dt.data_ptr(), dt.stride(), dr.data_ptr(), dr.stride(), d_flow.data_ptr(), H, W, 0, 5, 3, 5, 1.2, 4)
        d_out = torch.empty((H, W), dtype=torch.float32, device="cuda")
        h.warp_dev(dr.data_ptr(), dr.stride(), d_flow.data_ptr(), d_out.data_ptr(), H, W)
        h.synchronize()
        assert np.array_equal(d_flow.cpu().numpy(), want), name
        assert np.array_equal(d_out.cpu().numpy(), oracle.warp_slice(np.ascontiguousarray(ref), want)), name


def test_get_flow_updates_prev_flow_in_place(fdn):
    rng = np.random.default_rng(4)
    a, b = _img(rng, 40, 40), _img(rng, 40, 40)
    f = np.zeros((40, 40, 2), np.float32)
    out = fdn.get_flow(b, a, 0, 5, f)
    assert out is f and np.abs(f).max() > 0


def test_no_of_filter_matches_reference_golden(fdn):
    import os
    n = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_no_of.npz"))
    ks = [fdn.get_gaussian_kernel(float(s)) for s in n["sigmas"]]
    got = fdn.no_OF_filter(n["vol"], ks)
    assert got.dtype == np.float32
    assert np.array_equal(got, n["out"])


def _vol(shape, seed=7):
    from flowdenoising_amd.synth import make_volume
    return make_volume(shape, seed=seed, amplitude=100.0)


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_of_filter_single_axis(fdn, oracle, axis):
    vol = _vol((18, 40, 44))
    k = fdn.get_gaussian_kernel(1.0)
    mean = vol.mean()
    fn = [fdn.OF_filter_along_Z, fdn.OF_filter_along_Y, fdn.OF_filter_along_X][axis]
    got = fn(vol, k, 0, 5, mean)
    want = oracle.filter_along_axis(vol, axis, k, 0, 5, mean)
    assert rel_err(got, want) < TIGHT_TOL


def test_of_filter_3d(fdn, oracle):
    vol = _vol((20, 36, 40), seed=11)
    ks = [fdn.get_gaussian_kernel(s) for s in (1.0, 1.5, 0.5)]
    got = fdn.OF_filter(vol, ks, 0, 5)
    want = oracle.OF_filter(vol, ks, 0, 5)
    assert got.dtype == np.float32 and got.shape == vol.shape
    assert rel_err(got, want) < TIGHT_TOL
    # and it is not the plain Gaussian
    assert rel_err(got, oracle.no_OF_filter(vol, ks)) > 1e-3


def test_of_filter_wrap_and_recompute(fdn, oracle):
    """par's variant: wrap-around neighbours (par:312) and --recompute_flow (par:89-114)."""
    vol = _vol((10, 34, 38), seed=5)
    ks = [fdn.get_gaussian_kernel(1.0), None, None]
    from flowdenoising_amd import _lib
    got = fdn.OF_filter(vol, ks, 0, 5, border_mode=_lib.BORDER_WRAP, chained=False)
    want = oracle.OF_filter(vol, ks, 0, 5, border_mode=1, chained=False)
    assert rel_err(got, want) < TIGHT_TOL


def test_constant_volume(fdn):
    """SURVEY 8c KAT 5: constant volume v -> v * sum(w) per pass."""
    vol = np.full((6, 34, 36), 37.5, np.float32)
    ks = [fdn.get_gaussian_kernel(0.5)] * 3
    got = fdn.OF_filter(vol, ks, 0, 5)
    np.testing.assert_allclose(got, 37.5, rtol=1e-6)


@pytest.mark.parametrize("shape,axis,border", [((1, 40, 44), 0, 0), ((1, 40, 44), 0, 1), ((2, 40, 44), 0, 0), ((9, 2, 70), 2, 0), ((9, 70, 2), 1, 1)])
@pytest.mark.parametrize("l,w", [(0, 5), (1, 15)])
def test_degenerate_extents(fdn, oracle, shape, axis, border, l, w):
    """Edge cases of the sweep: an axis of length 1 or 2 under a kernel much longer than it (every neighbour of a target is a
    mean-pad slice, or the target itself once wrapped), and images two pixels wide or high -- on both Farneback kernels.
    Bit-equal to the oracle."""
    vol = _vol(shape, seed=77)
    k = fdn.get_gaussian_kernel(2.0)          # 17 taps
    fn = [fdn.OF_filter_along_Z, fdn.OF_filter_along_Y, fdn.OF_filter_along_X][axis]
    got = fn(vol, k, l, w, vol.mean(), border_mode=border)
    want = oracle.filter_along_axis(vol, axis, k, l, w, vol.mean(), border_mode=border, nthreads=4)
    assert np.array_equal(got, want), rel_err(got, want)


def test_kernel_size_one_is_identity(fdn):
    vol = _vol((4, 34, 36))
    got = fdn.OF_filter(vol, [np.array([1.0]), None, None], 0, 5)
    assert np.array_equal(got, vol)


def test_permute_roundtrip(fdn):
    from flowdenoising_amd.operators import handle
    h = handle()
    rng = np.random.default_rng(0)
    Z, Y, X = 5, 37, 70
    a = rng.standard_normal((Z, Y, X)).astype(np.float32)
    d_in = h.malloc(a.nbytes)
    d_out = h.malloc(a.nbytes)
    h.h2d(d_in, a)
    try:
        # out[y][z][x]
        h.permute_dev(d_in, d_out, (Y, Z, X), (X, Y * X, 1))
        out = np.empty((Y, Z, X), np.float32); h.d2h(out, d_out)
        assert np.array_equal(out, a.transpose(1, 0, 2))
        # out[x][z][y]
        h.permute_dev(d_in, d_out, (X, Z, Y), (1, Y * X, X))
        out = np.empty((X, Z, Y), np.float32); h.d2h(out, d_out)
        assert np.array_equal(out, a.transpose(2, 0, 1))
        # from [x][z][y] back to [z][y][x]
        h.h2d(d_in, np.ascontiguousarray(a.transpose(2, 0, 1)))
        h.permute_dev(d_in, d_out, (Z, Y, X), (Y, 1, Z * Y))
        out = np.empty((Z, Y, X), np.float32); h.d2h(out, d_out)
        assert np.array_equal(out, a)
    finally:
        h.free(d_in); h.free(d_out)


def test_mean_dev(fdn):
    from flowdenoising_amd.operators import handle
    h = handle()
    a = _vol((7, 33, 35))
    d = h.malloc(a.nbytes)
    h.h2d(d, a)
    try:
        m = h.mean_dev(d, a.size)
    finally:
        h.free(d)
    assert abs(float(m) - float(a.astype(np.float64).mean())) <= 1e-6 * abs(float(a.mean()))


def test_errors_are_loud(fdn):
    from flowdenoising_amd._lib import FlowdnError
    vol = _vol((4, 34, 36))
    with pytest.raises(FlowdnError):
        fdn.OF_filter(vol, [np.array([0.5, 0.5]), None, None], 0, 5)  # even kernel (seq:93 assert)
    k = fdn.get_gaussian_kernel(1.0)
    with pytest.raises(FlowdnError, match="at least 2x2"):
        fdn.OF_filter_along_Z(_vol((4, 1, 36)), k, 0, 5, np.float32(0))     # optical flow between one-pixel-high images
    with pytest.raises((FlowdnError, ValueError)):
        fdn.OF_filter_along_Z(np.zeros((0, 34, 36), np.float32), k, 0, 5, np.float32(0))     # an empty volume


def _random_cases(n, seed):
    """Reproducible random sweep configurations: odd shapes (images from 2 pixels up, not multiples of
    the kernels' 52-column bands), every window half-width the kernels distinguish, pyramid levels,
    both volume-end conventions, chained and recomputed flow."""
    rng = np.random.default_rng(seed)
    cases = []
    for i in range(n):
        axis = int(rng.integers(0, 3))
        shape = [int(rng.integers(2, 14)), int(rng.integers(2, 90)), int(rng.integers(2, 140))]
        if i % 5 == 0:                    # some larger images so that pyramids have levels to keep
            shape[1], shape[2] = int(rng.integers(64, 150)), int(rng.integers(64, 230))
        img = [s for a, s in enumerate(shape) if a != axis]
        if min(img) < 2:
            continue
        w = int(rng.choice([3, 4, 5, 5, 5, 6, 7, 9, 15]))
        l = int(rng.integers(0, 4))
        sigma = float(rng.choice([0.5, 1.0, 1.5, 2.0]))
        cases.append((tuple(shape), axis, l, w, sigma, int(rng.integers(0, 2)), bool(rng.integers(0, 2)), 1000 + i))
    return cases


@pytest.mark.parametrize("shape,axis,l,w,sigma,border,chained,seed", _random_cases(40, 20261003))
def test_randomised_sweeps(fdn, oracle, shape, axis, l, w, sigma, border, chained, seed):
    """Bit-level agreement with the (OpenCV-order) oracle over random shapes and parameters."""
    vol = _vol(shape, seed=seed)
    k = fdn.get_gaussian_kernel(sigma)
    mean = vol.mean()
    fn = [fdn.OF_filter_along_Z, fdn.OF_filter_along_Y, fdn.OF_filter_along_X][axis]
    got = fn(vol, k, l, w, mean, border_mode=border, chained=chained)
    want = oracle.filter_along_axis(vol, axis, k, l, w, mean, border_mode=border, chained=chained, nthreads=8)
    assert rel_err(got, want) < TIGHT_TOL, (shape, axis, l, w, sigma, border, chained)


@pytest.mark.parametrize("w", [2, 3, 4, 6, 7, 8, 9])
@pytest.mark.parametrize("l", [0, 2])
def test_fused_kernel_window_sizes(fdn, oracle, w, l):
    """winsize 2-9 (window half-widths 1 to 4) run on the fused kernel's builds: multi-band images with
    interior and edge bands, chains of four steps, with and without pyramid."""
    vol = _vol((7, 130, 300), seed=31 + w)
    k = fdn.get_gaussian_kernel(1.0)
    got = fdn.OF_filter(vol, [k, None, k], l, w)
    want = oracle.OF_filter(vol, [k, None, k], l, w, nthreads=8)
    assert np.array_equal(got, want), (w, l, rel_err(got, want))


@pytest.mark.parametrize("n", [1, 7, 130, 8191, 8192, 8193, 3 * 8192 + 77, 40 * 8192, 64 * 128 * 131])
def test_device_mean_is_numpys_float32_mean(fdn, n):
    """vol.mean() (seq:420) on the GPU: numpy's pairwise sums per 8192-element chunk, then its left-to-right
    float32 accumulation -- the padded volume ends are sensitive to the last bit of this value."""
    from flowdenoising_amd import _lib
    from flowdenoising_amd.operators import handle
    rng = np.random.default_rng(n)
    a = (rng.standard_normal(n) * 100 + 37).astype(np.float32)
    h = handle(0)
    d = h.malloc(a.nbytes)
    try:
        h.h2d(d, a)
        assert h.mean_dev(d, n) == a.mean()
        assert np.array_equal(h.np_chunk_sums_dev(d, n), [a[s:s + 8192].sum(dtype=np.float32) for s in range(0, n, 8192)])
    finally:
        h.free(d)


def test_flower_protocol(fdn, oracle):
    """src/flowdenoising_GPU.py's flower objects (gpu:92-177): set_target once, then a chain of get_flow(reference, prev_flow)
    calls -- the target stays on the device; flows equal cv2-order Farneback (the oracle) call by call, with and without
    OPTFLOW_USE_INITIAL_FLOW."""
    vol = _vol((6, 70, 90), seed=51)
    fl = fdn.GPU_flower(1, 5, 3, 5, 1.2)
    fl.set_target(vol[2])
    flow = np.zeros((70, 90, 2), np.float32)
    want = flow.copy()
    for i in (3, 4, 5):
        flow = fl.get_flow(vol[i], flow)
        want = oracle.get_flow(vol[i], vol[2], 1, 5, want)
        assert np.array_equal(flow, want)
    cold = fdn.CPU_flower(l=0, w=7, flags=0)
    cold.set_target(vol[0])
    assert np.array_equal(cold.get_flow(vol[1]), oracle.calcOpticalFlowFarneback(vol[0], vol[1], None, 0.5, 0, 7, 3, 5, 1.2, 0))
    with pytest.raises(RuntimeError):
        fdn.CPU_flower().get_flow(vol[0], flow)


@pytest.mark.parametrize("shape,l,w", [((14, 131, 97), 2, 5), ((14, 131, 97), 0, 5), ((14, 70, 150), 3, 15), ((12, 33, 61), 0, 7)])
def test_results_do_not_depend_on_the_alignment_of_device_pointers(fdn, shape, l, w):
    """fdn_sweep_stack_dev on a caller's device pointers that are only 4-byte aligned -- a slab view `volume[z0:z1]` of a volume
    with odd-sized images is -- gives the bits of the 256-byte aligned call: no kernel assumes more than the element's alignment
    (the polynomial expansion's 8- and 16-byte loads address library-owned buffers)."""
    from flowdenoising_amd.operators import _params, handle
    vol = _vol(shape, seed=21)
    k = fdn.get_gaussian_kernel(1.0)
    r = k.size // 2
    S, H, W = shape[0] - 2 * r, shape[1], shape[2]
    h = handle()
    outs = []
    for off_in, off_out in ((0, 0), (4, 0), (8, 4), (36, 20)):
        d_in, d_out = h.malloc(vol.nbytes + 256), h.malloc(S * H * W * 4 + 256)
        try:
            h.h2d(d_in + off_in, vol)
            h.sweep_stack_dev(d_in + off_in, d_out + off_out, S, H, W, k, _params(l, w))
            out = np.empty((S, H, W), np.float32)
            h.d2h(out, d_out + off_out)
        finally:
            h.free(d_in)
            h.free(d_out)
        outs.append(out)
    assert all(np.array_equal(o, outs[0]) for o in outs[1:])
