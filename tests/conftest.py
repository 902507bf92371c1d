import os
import sys

import pytest

# the HIP runtime reports its own errors on stderr (level 1 = errors only): a failure inside the runtime then says what it was
os.environ.setdefault("AMD_LOG_LEVEL", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_subprocess: a GPU test that starts other GPU processes (runs after the in-process GPU tests)")


def _starts_gpu_processes(item):
    """A GPU test that runs other GPU processes (the CLI, bench.py, rank processes of the transports, run_in_fresh_process):
    marked `@pytest.mark.gpu_subprocess` by hand (tests/test_abi_and_host.py checks that no test that starts one is unmarked)."""
    return item.get_closest_marker("gpu") is not None and item.get_closest_marker("gpu_subprocess") is not None


def pytest_collection_modifyitems(config, items):
    """GPU tests that start other GPU processes run LAST, in their own order, and GPU work that is multi-threaded runs in
    processes of its own (run_in_fresh_process).  Round 5: in 6 of 19 full `-m gpu` sessions the HIP runtime aborted the
    interpreter inside an ordinary in-process call, each time one to three tests after work of that kind -- the out-of-core
    mode's worker threads, rank threads, rank processes -- and never in a short session or one without them; what the runtime
    objects to is not known (profiles/history/NOTES_r05.md, section 5).  The long-lived test process now only ever does
    single-threaded GPU work, and all of it before the first other GPU process starts."""
    if os.environ.get("FDN_TEST_INPROCESS") == "1":      # diagnostic sessions (run_in_fresh_process below): round 5's original order
        return
    tail = [it for it in items if _starts_gpu_processes(it)]
    if tail:
        ids = {id(it) for it in tail}
        items[:] = [it for it in items if id(it) not in ids] + tail


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure only)."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def fdn():
    """The product package; its HIP library must exist (no fallback)."""
    import flowdenoising_amd
    flowdenoising_amd._lib.load()
    return flowdenoising_amd


def run_in_fresh_process(code, arrays, tmp_path, timeout=600, env=None):
    """Run `code` (Python source) in a process of its own with the arrays of `arrays` (a dict) loaded as variables of the same
    names; the script puts what it wants to hand back into a dict called `out` (name -> array), which is returned.
    For GPU work that is multi-threaded (the out-of-core mode's workers, rank threads): round 5's long sessions showed the
    HIP runtime aborting the interpreter some tests AFTER such work had run in the pytest process itself
    (profiles/history/NOTES_r05.md, section 5); the long-lived test process therefore stays single-threaded on the GPU."""
    import subprocess
    import numpy as np
    if os.environ.get("FDN_TEST_INPROCESS") == "1":
        # Diagnostic only (never the default): run the code in THIS process and keep collection order -- the configuration in
        # which round 5's sessions aborted -- to get the runtime's own message: `--capture=sys` leaves fd 2 to the runtime
        # (AMD_LOG_LEVEL), tools/abort_trace.c the native call chain.
        import contextlib
        import io
        scope = {k: np.asarray(v) for k, v in arrays.items()}
        scope["np"] = np
        scope["out"] = {}
        for p in (ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        buf = io.StringIO()
        old = dict(os.environ)
        os.environ.update(env or {})
        try:
            with contextlib.redirect_stdout(buf):
                exec(compile(code, "<run_in_fresh_process>", "exec"), scope)
        finally:
            os.environ.clear()
            os.environ.update(old)
        return {k: np.asarray(v) for k, v in scope["out"].items()}, buf.getvalue()
    np.savez(tmp_path / "fresh_in.npz", **{k: np.asarray(v) for k, v in arrays.items()})
    script = ("import sys, numpy as np\nsys.path.insert(0, %r)\nsys.path.insert(0, %r)\n"
              "_in = np.load(%r, allow_pickle=False)\n"
              "globals().update({k: _in[k] for k in _in.files})\nout = {}\n" % (ROOT, os.path.join(ROOT, "tests"), str(tmp_path / "fresh_in.npz"))
              + code + "\nnp.savez(%r, **out)\n" % str(tmp_path / "fresh_out.npz"))
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=timeout, env={**os.environ, **(env or {})})
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(tmp_path / "fresh_out.npz", allow_pickle=False)
    return {k: got[k] for k in got.files}, r.stdout


def to_host(t):
    """A torch device tensor's values as a numpy array, copied by the library (fdn_memcpy_d2h: page-locked for the call or
    bounced) instead of `tensor.cpu()`: large pageable copies are page-locked by the HIP runtime on the fly -- the path the GPU
    memory fault of profiles/history/NOTES_r06.md section 2 was caught in; the long-lived test process stays off it."""
    import numpy as np
    import torch
    from flowdenoising_amd.operators import handle
    t = t.contiguous()
    torch.cuda.synchronize()
    arr = np.empty(tuple(t.shape), dtype=np.float32)
    handle().d2h(arr, t.data_ptr())
    return arr


def rel_err(a, b):
    """max |a-b| / max |b|: the volume-range-relative error the 1e-4 bar is stated in."""
    import numpy as np
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def assert_kernel(k, ref, sigma):
    """get_gaussian_kernel (seq:30-41) against the reference-generated golden.  The taps are
    exp(-j^2 / 2 sigma^2) / sum: the sum follows numpy's pairwise order (bit-equal), but the golden's
    np.exp ran numpy's AVX512F float64 kernel on the build container's CPU, which is not correctly
    rounded; libm's exp differs from it by one ulp in two taps at sigma = 2 and sigma = 4 (so the
    reference's own taps depend on the CPU it runs on).  Every other sigma in the golden is bit-equal."""
    import numpy as np
    assert k.shape == ref.shape
    if sigma in (2.0, 4.0):
        assert np.abs(k - ref).max() <= np.spacing(ref.max()) and np.count_nonzero(k != ref) <= 2
    else:
        assert np.array_equal(k, ref)


@pytest.fixture(autouse=True)
def _resource_trace(request):
    """FDN_TEST_TRACE=<file>: one line per test with the process's open file descriptors, threads and resident memory
    (a diagnostic for failures that depend on what earlier tests left behind; off by default)."""
    path = os.environ.get("FDN_TEST_TRACE")
    if not path:
        yield
        return
    import resource
    import threading

    def snap(tag):
        try:
            nfd = len(os.listdir("/proc/self/fd"))
        except OSError:
            nfd = -1
        rss = 0
        try:
            with open("/proc/self/statm") as f:
                rss = int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") >> 20
        except (OSError, ValueError):
            pass
        gpu = ""
        try:                                    # device memory: free / total, and what the process-wide handle holds
            from flowdenoising_amd import operators
            if operators._handles:
                h = next(iter(operators._handles.values()))
                fre, tot = h.mem_info()
                gpu = f" gpu_free={fre >> 20}MiB/{tot >> 20}MiB handle={h.workspace_bytes() >> 20}MiB"
        except Exception as e:                  # noqa: BLE001 -- a diagnostic must not fail a test
            gpu = f" gpu=? ({type(e).__name__})"
        with open(path, "a") as f:
            f.write(f"{tag} {request.node.nodeid} fds={nfd}/{resource.getrlimit(resource.RLIMIT_NOFILE)[0]} threads={threading.active_count()} rss={rss}MiB{gpu}\n")
    snap("begin")
    yield
    snap("end")
