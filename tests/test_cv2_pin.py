"""The only pin this pipeline can ever get for the OpenCV part of the oracle: real cv2 output.

The reference delegates its arithmetic to `cv2.calcOpticalFlowFarneback` (src/flowdenoising_sequential.py:62)
and `cv2.remap` (seq:56).  cv2 is absent from the build container and from the GPU image (probe recorded in
DESIGN.md 5), so these tests SKIP there; on any box that has opencv-python they run and compare
  * the CPU oracle (oracle/fdn_oracle.c)           -- `-m "not gpu"`
  * the HIP path through the C ABI (libflowdn.so)  -- `-m gpu`
with cv2 called directly from this harness (own code: the reference's files stay where they are), on seeded
pairs (64^2, 130x70, 256x300; levels 0 and 3; winsize 5 and 15; zero and random initial flow) and on
BASELINE configs[0] (128x128x64, sigma 2) through a seq-shaped sweep written here.  If committed cv2 fixtures
exist (tests/golden/cv2_*.npz, written by tools/make_cv2_golden.py on a box with cv2), they are checked on
every box, cv2 or not.

Tolerance: north_star's 1e-4 relative.  Bit equality is reported (and expected at levels = 0, where the
pyramid blur taps are powers of two); at levels > 0 stock x86 wheels run GaussianBlur / resize through
FMA-contracting SIMD code, which the oracle's separate multiply-add may differ from in the last bit."""
import glob
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-4

PAIR_SHAPES = [(64, 64), (130, 70), (256, 300)]
PAIR_PARAMS = [(0, 5), (0, 15), (3, 5), (3, 15)]


def make_pair(shape, seed):
    """Seeded (target, reference, random initial flow): smooth structure, sub-pixel shift, noise."""
    import scipy.ndimage
    rng = np.random.default_rng(seed)
    H, W = shape
    a = scipy.ndimage.gaussian_filter(rng.standard_normal((H, W)), 3.0)
    a = (a / np.abs(a).max() * 200).astype(np.float32)
    b = scipy.ndimage.shift(a.astype(np.float64), (0.7, -0.45), order=3, mode="nearest").astype(np.float32)
    b += (rng.standard_normal((H, W)) * 2).astype(np.float32)
    f0 = (rng.standard_normal((H, W, 2)) * 0.5).astype(np.float32)
    return a, b, f0


def cv2_flow(cv2, target, reference, l, w, f0):
    """seq:62 as the reference calls it: prev = target, next = reference, flow in/out."""
    flags = cv2.OPTFLOW_USE_INITIAL_FLOW if f0 is not None else 0
    return cv2.calcOpticalFlowFarneback(prev=target, next=reference, flow=None if f0 is None else f0.copy(), pyr_scale=0.5,
                                        levels=l, winsize=w, iterations=3, poly_n=5, poly_sigma=1.2, flags=flags)


def cv2_warp(cv2, reference, flow):
    """seq:51-57: map = float32(flow + grid); remap INTER_LINEAR, BORDER_REPLICATE."""
    H, W = flow.shape[:2]
    m = np.empty((H, W, 2), np.float32)
    m[..., 0] = (flow[..., 0].astype(np.float64) + np.arange(W)[None, :]).astype(np.float32)
    m[..., 1] = (flow[..., 1].astype(np.float64) + np.arange(H)[:, None]).astype(np.float32)
    return cv2.remap(reference, m, None, interpolation=cv2.INTER_LINEAR, borderMode=cv2.BORDER_REPLICATE)


def cv2_of_filter(cv2, vol, kernels, l, w):
    """A seq-shaped sweep on cv2 (own harness code): mean padding, back chain near -> far, centre tap, forward chain;
    f64 product / f32 store of the accumulate (numpy >= 2 semantics of seq:107)."""
    mean = vol.mean()
    cur = vol
    for axis, k in enumerate(kernels):
        if k is None:
            continue
        K, r = k.size, k.size // 2
        n = cur.shape[axis]
        moved = np.moveaxis(cur, axis, 0)
        out = np.empty_like(moved)
        pad = np.full((r,) + moved.shape[1:], mean, dtype=np.float32)
        padded = np.concatenate([pad, moved, pad])
        for s in range(n):
            target = np.ascontiguousarray(moved[s])
            acc = np.zeros_like(target)
            for side in (0, 1):
                if side == 1:
                    acc = (acc.astype(np.float64) + target.astype(np.float64) * k[r]).astype(np.float32)
                flow = np.zeros(target.shape + (2,), np.float32)
                for step in range(r):
                    i = r - 1 - step if side == 0 else r + 1 + step
                    ref = np.ascontiguousarray(padded[s + i])
                    flow = cv2_flow(cv2, target, ref, l, w, flow)
                    acc = (acc.astype(np.float64) + cv2_warp(cv2, ref, flow).astype(np.float64) * k[i]).astype(np.float32)
            out[s] = acc
        cur = np.ascontiguousarray(np.moveaxis(out, 0, axis))
    return cur


def numpy_seq_sweep(cv2, vol, kernels, l, w):
    """seq's sweep with NO casts of this harness's own: the dtypes are whatever numpy and cv2 make of the volume's dtype,
    as in the reference (seq:86-89 np.zeros_like(...).astype(float32), np.full(fill_value=mean), in-place += of seq:107,
    seq:420 mean once for all passes).  For a float32 volume this is cv2_of_filter; for an integer volume the padded
    volume is float64 and cv2.remap runs on CV_64F."""
    mean = vol.mean()
    cur = vol
    for axis, k in enumerate(kernels):
        if k is None:
            continue
        K = k.size
        moved = np.moveaxis(cur, axis, 0)
        n = moved.shape[0]
        filtered = np.zeros_like(moved).astype(np.float32)
        padded = np.full(shape=(n + K,) + moved.shape[1:], fill_value=mean)
        padded[K // 2:n + K // 2] = moved
        for s in range(n):
            tmp = np.zeros_like(moved[s]).astype(np.float32)
            prev_flow = np.zeros(moved.shape[1:] + (2,), dtype=np.float32)
            for i in range(K // 2 - 1, -1, -1):
                prev_flow = cv2_flow(cv2, np.ascontiguousarray(moved[s]), np.ascontiguousarray(padded[s + i]), l, w, prev_flow)
                tmp += cv2_warp(cv2, np.ascontiguousarray(padded[s + i]), prev_flow) * k[i]
            tmp += moved[s] * k[K // 2]
            prev_flow = np.zeros(moved.shape[1:] + (2,), dtype=np.float32)
            for i in range(K // 2 + 1, K):
                prev_flow = cv2_flow(cv2, np.ascontiguousarray(moved[s]), np.ascontiguousarray(padded[s + i]), l, w, prev_flow)
                tmp += cv2_warp(cv2, np.ascontiguousarray(padded[s + i]), prev_flow) * k[i]
            filtered[s] = tmp
        cur = np.ascontiguousarray(np.moveaxis(filtered, 0, axis))
    return cur


def numpy_par_sweep(cv2, vol, kernels, l, w):
    """par's passes (par:306-373, 285-290) with numpy's own dtype propagation: wrap-around neighbours taken from the volume
    in ITS dtype, filtered_vol = np.zeros_like(vol), vol[...] = filtered_vol after each pass (all three kept, unlike par)."""
    vol = vol.copy()
    for axis, k in enumerate(kernels):
        if k is None:
            continue
        K, ks2 = k.size, k.size // 2
        filtered_vol = np.zeros_like(vol)
        moved, fmoved = np.moveaxis(vol, axis, 0), np.moveaxis(filtered_vol, axis, 0)
        n = moved.shape[0]
        for s in range(n):
            tmp = np.zeros_like(moved[s]).astype(np.float32)
            for side in (range(ks2 - 1, -1, -1), range(ks2 + 1, K)):
                if side.start > ks2:
                    tmp += moved[s] * k[ks2]
                prev_flow = np.zeros(moved.shape[1:] + (2,), dtype=np.float32)
                for i in side:
                    ref = np.ascontiguousarray(moved[(s + i - ks2) % n])
                    prev_flow = cv2_flow(cv2, np.ascontiguousarray(moved[s]), ref, l, w, prev_flow)
                    tmp += cv2_warp(cv2, ref, prev_flow) * k[i]
            fmoved[s] = tmp
        vol[...] = filtered_vol
    return vol


def cv2_identity(cv2):
    """Version and the build lines that decide the arithmetic (SIMD baseline / dispatch, IPP, FMA): recorded with every
    fixture and printed with every mismatch, so that a difference can be attributed to a build, not guessed at."""
    info = {"version": getattr(cv2, "__version__", "stand-in")}
    try:
        lines = cv2.getBuildInformation().splitlines()
        keep = [ln.strip() for ln in lines if any(k in ln for k in ("CPU/HW features", "Baseline:", "Dispatched code", "requested:", "Intel IPP", "Version control", "FP16", "AVX"))]
        info["build"] = keep[:16]
    except Exception:
        info["build"] = []
    return info


def remap_unquantised(src, flow):
    """The OTHER reading of cv2.remap(INTER_LINEAR, BORDER_REPLICATE) on a float map: plain float32 bilinear
    interpolation at the map position, no 1/32-pixel coordinate table.  The oracle and the HIP kernels model the classic
    path (INTER_BITS = 5: coordinates rounded to 1/32 px, weights from the 32 x 32 table).  OpenCV 4.11+
    is reported to ship new linear remap kernels for float maps that may not quantise; `opencv-python` is unpinned in
    the reference (src/requirements.txt:2), so a fresh install could be either.  Diagnostic model only."""
    H, W = flow.shape[:2]
    mx = (flow[..., 0].astype(np.float64) + np.arange(W)[None, :]).astype(np.float32)
    my = (flow[..., 1].astype(np.float64) + np.arange(H)[:, None]).astype(np.float32)
    x0 = np.floor(mx)
    y0 = np.floor(my)
    fx = (mx - x0).astype(np.float32)
    fy = (my - y0).astype(np.float32)
    x0 = x0.astype(np.int64)
    y0 = y0.astype(np.int64)
    xa, xb = np.clip(x0, 0, W - 1), np.clip(x0 + 1, 0, W - 1)
    ya, yb = np.clip(y0, 0, H - 1), np.clip(y0 + 1, 0, H - 1)
    s = np.asarray(src, dtype=np.float32)
    one = np.float32(1)
    top = s[ya, xa] * (one - fx) + s[ya, xb] * fx
    bot = s[yb, xa] * (one - fx) + s[yb, xb] * fx
    return (top * (one - fy) + bot * fy).astype(np.float32)


def classify_remap(src, flow, observed, quantised):
    """Which remap model does `observed` (cv2's output) follow?  Returns (verdict, err_quantised, err_unquantised)."""
    eq = float(np.abs(observed.astype(np.float64) - quantised).max())
    eu = float(np.abs(observed.astype(np.float64) - remap_unquantised(src, flow)).max())
    scale = max(float(np.abs(np.asarray(src, dtype=np.float64)).max()), 1e-30)
    if eq == 0:
        verdict = "classic 1/32-pixel table (INTER_BITS = 5): the modelled path, bit for bit"
    elif eu <= 4e-7 * scale and eu < eq:
        verdict = "UNQUANTISED float bilinear: this cv2 build does not use the 1/32-pixel table for float maps"
    elif eq < eu:
        verdict = "closer to the classic 1/32-pixel table, but not bit-equal (weights / accumulation order differ)"
    else:
        verdict = "closer to unquantised float bilinear than to the 1/32-pixel table"
    return verdict, eq, eu


def assert_warp_matches(cv2, src, flow, got, label):
    """`got` (oracle or HIP remap) against cv2.remap; on a mismatch say WHICH model this cv2 follows, with its version."""
    want = cv2_warp(cv2, src, flow)
    if np.array_equal(got, want):
        return
    verdict, eq, eu = classify_remap(src, flow, want, np.asarray(got, dtype=np.float64))
    raise AssertionError(f"{label}: cv2.remap differs from the modelled remap (max |diff| {eq:.3g}; against unquantised float bilinear "
                         f"{eu:.3g}) -> {verdict}.  cv2 {cv2_identity(cv2)}")


def _flow_err(got, want):
    return float(np.abs(got - want).max() / max(np.abs(want).max(), 1.0))


def _report(name, got, want):
    exact = np.array_equal(got, want)
    print(f"[cv2 pin] {name}: {'bit-equal' if exact else 'max err %.3g' % np.abs(got - want).max()}")
    return exact


# ---- oracle vs cv2 (CPU) ------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", PAIR_SHAPES)
@pytest.mark.parametrize("l,w", PAIR_PARAMS)
def test_oracle_farneback_and_remap_against_cv2(oracle, shape, l, w):
    cv2 = pytest.importorskip("cv2")
    a, b, f0 = make_pair(shape, 100 + shape[0])
    for init in (np.zeros_like(f0), f0):
        want = cv2_flow(cv2, a, b, l, w, init)
        got = oracle.get_flow(b, a, l, w, init.copy())
        exact = _report(f"oracle flow {shape} l={l} w={w}", got, want)
        assert _flow_err(got, want) < TOL
        if l == 0:
            assert exact
        assert_warp_matches(cv2, b, want, oracle.warp_slice(b, want), f"oracle remap {shape}")
    want = cv2_flow(cv2, a, b, l, w, None)                        # par:89-114: flags = 0
    assert _flow_err(oracle.calcOpticalFlowFarneback(a, b, None, 0.5, l, w, 3, 5, 1.2, 0), want) < TOL


def test_oracle_config0_against_cv2(oracle):
    cv2 = pytest.importorskip("cv2")
    from flowdenoising_amd.synth import make_volume
    vol = make_volume((64, 128, 128), seed=1234 + 1, amplitude=100.0)
    k = oracle.get_gaussian_kernel(2.0)
    want = cv2_of_filter(cv2, vol, [k, k, k], 0, 5)
    got = oracle.OF_filter(vol, [k, k, k], 0, 5, nthreads=8)
    _report("oracle configs[0]", got, want)
    assert float(np.abs(got - want).max() / np.abs(want).max()) < TOL


# ---- HIP path vs cv2 (GPU box with cv2) -----------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("shape", PAIR_SHAPES)
@pytest.mark.parametrize("l,w", PAIR_PARAMS)
def test_hip_farneback_and_remap_against_cv2(fdn, shape, l, w):
    cv2 = pytest.importorskip("cv2")
    a, b, f0 = make_pair(shape, 100 + shape[0])
    for init in (np.zeros_like(f0), f0):
        want = cv2_flow(cv2, a, b, l, w, init)
        got = fdn.get_flow(b, a, l, w, init.copy())
        _report(f"hip flow {shape} l={l} w={w}", got, want)
        assert _flow_err(got, want) < TOL
        assert_warp_matches(cv2, b, want, fdn.warp_slice(b, want), f"hip remap {shape}")


@pytest.mark.gpu
def test_hip_config0_against_cv2(fdn):
    cv2 = pytest.importorskip("cv2")
    from flowdenoising_amd.synth import make_volume
    vol = make_volume((64, 128, 128), seed=1234 + 1, amplitude=100.0)
    k = fdn.get_gaussian_kernel(2.0)
    want = cv2_of_filter(cv2, vol, [k, k, k], 0, 5)
    got = fdn.OF_filter(vol, [k, k, k], 0, 5)
    _report("hip configs[0]", got, want)
    assert float(np.abs(got - want).max() / np.abs(want).max()) < TOL


# ---- the harness itself (runs everywhere) -----------------------------------------------------------------------
def test_remap_model_diagnostic_tells_the_two_models_apart(oracle):
    """The day cv2 exists, a remap mismatch must be diagnosed in one run, not read as an oracle bug: a stand-in cv2
    whose remap is the classic table model passes; one whose remap is plain float bilinear (what OpenCV >= 4.11 may do)
    fails with a message that names the unquantised model and the cv2 version."""
    a, b, f0 = make_pair((64, 64), 164)
    flow = (f0 * 3).astype(np.float32)

    class Classic:
        __version__ = "stand-in classic"
        INTER_LINEAR, BORDER_REPLICATE = 1, 1

        @staticmethod
        def remap(src, m, _, interpolation, borderMode):
            return oracle.remap(src, m)

    class Unquantised(Classic):
        __version__ = "stand-in 4.11-like"

        @staticmethod
        def remap(src, m, _, interpolation, borderMode):
            H, W = m.shape[:2]
            f = np.empty_like(m)
            f[..., 0] = m[..., 0] - np.arange(W, dtype=np.float32)[None, :]
            f[..., 1] = m[..., 1] - np.arange(H, dtype=np.float32)[:, None]
            return remap_unquantised(src, f)

    modelled = oracle.warp_slice(b, flow)
    assert_warp_matches(Classic, b, flow, modelled, "classic stand-in")
    # integer-valued flows: both models agree exactly (no fractional coordinate to quantise)
    fi = np.round(flow)
    assert np.array_equal(oracle.warp_slice(b, fi), remap_unquantised(b, fi))
    with pytest.raises(AssertionError, match="UNQUANTISED float bilinear.*stand-in 4.11-like"):
        assert_warp_matches(Unquantised, b, flow, modelled, "4.11-like stand-in")
    assert classify_remap(b, flow, modelled.astype(np.float32), modelled.astype(np.float64))[0].startswith("classic")
    assert cv2_identity(Classic)["version"] == "stand-in classic"



@pytest.mark.parametrize("fma,lanes,model", [(0, 8, 0), (1, 8, 1), (2, 8, 1), (2, 4, 0)])
def test_option_matcher_finds_the_settings_of_a_stand_in(oracle, fma, lanes, model):
    """tools/make_cv2_golden.py records, next to a real cv2's version and build lines, which values of the product's switches
    for the two cv2 unknowns reproduce it bit for bit (`opencv_fma` / `opencv_fma_lanes`, `remap_model`).  Here a stand-in cv2
    that IS the oracle in one of those settings must be identified as exactly that setting."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools"))
    import make_cv2_golden as M

    class StandIn:
        __version__ = f"stand-in fma {fma}/{lanes} remap {model}"
        OPTFLOW_USE_INITIAL_FLOW, INTER_LINEAR, BORDER_REPLICATE = 4, 1, 1

        @staticmethod
        def calcOpticalFlowFarneback(prev, next, flow, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags):
            oracle.set_fma(fma, lanes)
            try:
                return oracle.calcOpticalFlowFarneback(prev, next, flow, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags)
            finally:
                oracle.set_fma(0)

        @staticmethod
        def remap(src, m, _, interpolation, borderMode):
            oracle.set_remap_model(model)
            try:
                return oracle.remap_any(src, m)
            finally:
                oracle.set_remap_model(0)

    import types
    T = types.SimpleNamespace(make_pair=make_pair, cv2_warp=cv2_warp, cv2_flow=cv2_flow)
    res = M.which_options_match(StandIn, T, oracle)
    assert res["remap_model"] == model
    assert res["opencv_fma"] == {"mode": fma, "lanes": 8 if fma < 2 else lanes}, res


def test_harness_sweep_is_seq_shaped(oracle):
    """cv2_of_filter / cv2_flow / cv2_warp above are this file's own code; so that they are known to be right on
    the day cv2 appears, they are run here with a stand-in object whose two functions forward to the oracle:
    the result must then be the oracle's OF_filter, bit for bit (padding, tap order, chain reset, accumulate)."""
    from flowdenoising_amd.synth import make_volume

    class StandIn:
        OPTFLOW_USE_INITIAL_FLOW, INTER_LINEAR, BORDER_REPLICATE = 4, 1, 1

        @staticmethod
        def calcOpticalFlowFarneback(prev, next, flow, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags):
            return oracle.calcOpticalFlowFarneback(prev, next, flow, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags)

        @staticmethod
        def remap(src, m, _, interpolation, borderMode):
            return oracle.remap(src, m)

    vol = make_volume((7, 34, 38), seed=3, amplitude=100.0)
    ks = [oracle.get_gaussian_kernel(1.0), oracle.get_gaussian_kernel(0.5), None]
    assert np.array_equal(cv2_of_filter(StandIn, vol, ks, 0, 5), oracle.OF_filter(vol, ks, 0, 5))


class OracleCv2:
    """cv2 stand-in on the oracle, dispatching on the image depth like cv2 does (convertTo(CV_32F) inside Farneback;
    remap per depth)."""
    OPTFLOW_USE_INITIAL_FLOW, INTER_LINEAR, BORDER_REPLICATE = 4, 1, 1

    @staticmethod
    def calcOpticalFlowFarneback(prev, next, flow, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags):
        from oracle import oracle
        return oracle.calcOpticalFlowFarneback(np.asarray(prev, np.float32), np.asarray(next, np.float32), flow, pyr_scale,
                                               levels, winsize, iterations, poly_n, poly_sigma, flags)

    @staticmethod
    def remap(src, m, _, interpolation, borderMode):
        from oracle import oracle
        return oracle.remap_any(src, m)


def _int16_volume(shape, seed):
    from flowdenoising_amd.synth import make_volume
    v = make_volume(shape, seed=seed, amplitude=100.0)
    lo, hi = float(v.min()), float(v.max())
    return (np.round((v - lo) / (hi - lo) * 4095) - 1000).astype(np.int16)


def test_numpy_dtype_propagation_of_integer_volumes(oracle):
    """The NUMPY half of the integer-volume semantics, checked with real numpy: sweeps written like the reference's, with
    no casts of their own, on the cv2 stand-in must give the oracle's integer restatements bit for bit -- np.full's dtype
    (float64 padded volume), the in-place += into float32, the truncating store into an integer volume.  What stays
    unpinned is cv2's side (remap on CV_64F / CV_16S), restated in oracle.remap_any."""
    vi = _int16_volume((7, 34, 38), 21)
    ks = [oracle.get_gaussian_kernel(1.0), oracle.get_gaussian_kernel(0.5), oracle.get_gaussian_kernel(0.5)]
    assert np.array_equal(numpy_seq_sweep(OracleCv2, vi, ks, 0, 5), oracle.OF_filter_integer_input(vi, ks, 0, 5))
    vf = vi.astype(np.float32)
    assert np.array_equal(numpy_seq_sweep(OracleCv2, vf, ks, 0, 5), oracle.OF_filter(vf, ks, 0, 5))
    got = numpy_par_sweep(OracleCv2, vi, ks, 0, 5)
    assert got.dtype == np.int16 and np.array_equal(got, oracle.filter_par_integer_input(vi, ks, 0, 5).astype(np.int16))
    assert np.array_equal(numpy_par_sweep(OracleCv2, vf, ks, 0, 5), oracle.OF_filter(vf, ks, 0, 5, border_mode=1))


def test_integer_volumes_against_cv2(oracle):
    """With real cv2: the same sweeps pin the integer semantics (remap on CV_64F and CV_16S, Farneback fed mixed depths)."""
    cv2 = pytest.importorskip("cv2")
    vi = _int16_volume((7, 34, 38), 21)
    ks = [oracle.get_gaussian_kernel(1.0), oracle.get_gaussian_kernel(0.5), oracle.get_gaussian_kernel(0.5)]
    want = numpy_seq_sweep(cv2, vi, ks, 0, 5)
    got = oracle.OF_filter_integer_input(vi, ks, 0, 5)
    _report("oracle int16 seq", got, want)
    assert float(np.abs(got - want).max() / np.abs(want).max()) < TOL
    want = numpy_par_sweep(cv2, vi, ks, 0, 5)
    got = oracle.filter_par_integer_input(vi, ks, 0, 5).astype(np.int16)
    _report("oracle int16 par", got, want)
    assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1


@pytest.mark.gpu
def test_hip_integer_volumes_against_cv2(fdn):
    """The HIP path on an int16 volume against the same reference-shaped sweeps on real cv2 (seq and par semantics),
    and the typed pair operators against cv2 called on the same arrays."""
    cv2 = pytest.importorskip("cv2")
    vi = _int16_volume((7, 34, 38), 21)
    ks = [fdn.get_gaussian_kernel(1.0), fdn.get_gaussian_kernel(0.5), fdn.get_gaussian_kernel(0.5)]
    want = numpy_seq_sweep(cv2, vi, ks, 0, 5)
    got = fdn.OF_filter(vi, ks, 0, 5)
    _report("hip int16 seq", got, want)
    assert float(np.abs(got - want).max() / np.abs(want).max()) < TOL
    want = numpy_par_sweep(cv2, vi, ks, 0, 5)
    v = vi.copy()
    fdn.FlowDenoising(1, v, 0, 5).filter(ks)
    _report("hip int16 par", v, want)
    assert np.abs(v.astype(np.int32) - want.astype(np.int32)).max() <= 1
    padded = np.full(shape=(9, 34, 38), fill_value=vi.mean())
    padded[1:8] = vi
    flow = cv2_flow(cv2, vi[3], padded[5], 0, 5, np.zeros((34, 38, 2), np.float32))
    assert _flow_err(fdn.get_flow(padded[5], vi[3], 0, 5, np.zeros((34, 38, 2), np.float32)), flow) < TOL
    for ref in (padded[5], vi[4], vi[4].astype(np.uint16)):
        assert np.array_equal(fdn.warp_slice(ref, flow), cv2_warp(cv2, ref, flow))


# ---- committed cv2 fixtures (none until a box with cv2 runs tools/make_cv2_golden.py) ---------------------------
def _fixtures():
    return sorted(glob.glob(os.path.join(GOLD, "cv2_*.npz")))


def test_cv2_probe_is_recorded():
    """DESIGN.md 5 must say whether parity is pinned: 'parity unpinned' while no cv2 fixture is committed."""
    text = open(os.path.join(os.path.dirname(GOLD), "..", "DESIGN.md")).read()
    if _fixtures():
        assert "pinned by cv2 fixtures" in text
    else:
        assert "parity unpinned" in text and "import cv2" in text


@pytest.mark.parametrize("path", _fixtures() or [None])
def test_oracle_against_committed_cv2_fixtures(oracle, path):
    if path is None:
        pytest.skip("no cv2 fixtures committed (cv2 has never been importable in this pipeline): parity unpinned")
    g = np.load(path)
    if "semantics" in g:
        ks = [oracle.get_gaussian_kernel(float(v)) for v in g["sigmas"]]
        if str(g["semantics"]) == "seq":
            got = oracle.OF_filter_integer_input(g["vol"], ks, int(g["l"]), int(g["w"]))
            assert float(np.abs(got - g["out"]).max() / np.abs(g["out"]).max()) < TOL
        else:
            got = oracle.filter_par_integer_input(g["vol"], ks, int(g["l"]), int(g["w"])).astype(g["out"].dtype)
            assert np.abs(got.astype(np.int64) - g["out"].astype(np.int64)).max() <= 1
        return
    if "flow" in g:
        l, w = int(g["l"]), int(g["w"])
        got = oracle.get_flow(g["reference"], g["target"], l, w, g["init"].copy())
        assert _flow_err(got, g["flow"]) < TOL
        got_w = oracle.warp_slice(g["reference"], g["flow"])
        if not np.array_equal(got_w, g["warped"]):
            verdict, eq, eu = classify_remap(g["reference"], g["flow"], g["warped"], got_w.astype(np.float64))
            raise AssertionError(f"{os.path.basename(path)} (cv2 {g['cv2_version'] if 'cv2_version' in g else '?'}): remap differs from the "
                                 f"modelled one by {eq:.3g} (unquantised model: {eu:.3g}) -> {verdict}")
    else:
        k = oracle.get_gaussian_kernel(float(g["sigma"]))
        got = oracle.OF_filter(g["vol"], [k, k, k], int(g["l"]), int(g["w"]), nthreads=8)
        assert float(np.abs(got - g["out"]).max() / np.abs(g["out"]).max()) < TOL
