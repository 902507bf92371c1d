"""GPU tests beyond the small parity cases: the CLI end to end, the slab engine on the HIP backend,
and BASELINE.json's full-size images checked through (a) spot parity against the oracle on
sub-volumes and (b) size-independent properties."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import rel_err, to_host

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TIGHT_TOL = 2e-6


def _vol(shape, seed=7):
    from flowdenoising_amd.synth import make_volume
    return make_volume(shape, seed=seed, amplitude=100.0)


@pytest.mark.gpu_subprocess
def test_cli_mrc_end_to_end_config0(fdn, oracle, tmp_path):
    """BASELINE configs[0] in miniature: float32 MRC in, float32 MRC out, defaults of the oracle script."""
    from flowdenoising_amd import io as fio
    vol = _vol((10, 40, 44), seed=3)
    fio.write_mrc(str(tmp_path / "in.mrc"), vol)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "flowdenoising.py"), "-i", str(tmp_path / "in.mrc"),
                        "-o", str(tmp_path / "out.mrc"), "-s", "1.0", "0.5", "1.0", "-v", "1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = fio.read_mrc(str(tmp_path / "out.mrc"))
    want = oracle.OF_filter(vol, [oracle.get_gaussian_kernel(s) for s in (1.0, 0.5, 1.0)], 0, 5)
    assert got.dtype == np.float32 and rel_err(got, want) < TIGHT_TOL


@pytest.mark.gpu_subprocess
def test_cli_tiff_uint16_no_of_and_par_compat(fdn, oracle, tmp_path):
    from flowdenoising_amd import io as fio
    vol = (np.clip(_vol((6, 34, 36), seed=4), 0, None) * 4).astype(np.uint16)
    fio.write_tiff(str(tmp_path / "in.tif"), vol)
    exe = [sys.executable, os.path.join(ROOT, "flowdenoising.py"), "-i", str(tmp_path / "in.tif"), "-s", "0.5", "0.5", "0.5"]
    r = subprocess.run(exe + ["-o", str(tmp_path / "a.tif"), "-n"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    ks = [oracle.get_gaussian_kernel(0.5)] * 3
    want = oracle.no_OF_filter(vol.astype(np.float32), ks)
    got = fio.read_tiff(str(tmp_path / "a.tif"))
    assert got.dtype == np.uint16 and np.array_equal(got, want.astype(np.uint16))     # seq:566-571
    r = subprocess.run(exe + ["-o", str(tmp_path / "b.tif"), "--compat", "par", "-l", "0", "--recompute_flow"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = fio.read_tiff(str(tmp_path / "b.tif"))
    want = oracle.OF_filter(vol.astype(np.float32), ks, 0, 5, border_mode=1, chained=False)
    assert got.dtype == np.float32 and rel_err(got, want) < TIGHT_TOL                 # par:548, par:312


@pytest.mark.gpu_subprocess
def test_cli_gpus_2_shards_reads_and_gathers(fdn, tmp_path):
    """flowdenoising.py --gpus 2: re-launches itself under torch.distributed.run, every rank reads its own Z-slab of
    the file, the mean is assembled from chunk sums, rank 0 gathers and writes -- the output equals the single-GPU
    run's bit for bit.  (On a one-GPU box the two ranks share GPU 0 and exchange through the host.)"""
    from flowdenoising_amd import io as fio
    vol = _vol((11, 66, 130), seed=41)
    fio.write_mrc(str(tmp_path / "in.mrc"), vol)
    exe = [sys.executable, os.path.join(ROOT, "flowdenoising.py"), "-i", str(tmp_path / "in.mrc"), "-s", "1.0", "0.5", "1.0"]
    for out, extra in (("one.mrc", []), ("two.mrc", ["--gpus", "2"])):
        r = subprocess.run(exe + ["-o", str(tmp_path / out)] + extra, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
    assert np.array_equal(fio.read_mrc(str(tmp_path / "two.mrc")), fio.read_mrc(str(tmp_path / "one.mrc")))


@pytest.mark.gpu_subprocess
def test_bench_line_contract(fdn):
    """bench.py on a small volume: ONE JSON line with the contract's keys, a roofline fraction that is a fraction, the
    post-run oracle check green and a CPU baseline beside it."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--shape", "24,96,160", "--steps", "1", "--warmup", "1",
                        "--cpu-targets", "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "checked"):
        assert key in d, key
    assert d["unit"] == "Mvoxels/s" and d["dtype"] == "f32" and d["vs_baseline"] is None and "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["kernel"] == "k_farneback_fused" and 0 < rf["frac"] <= 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["traffic"] is None            # the committed PMC pass is for the 512 x 1024 x 1024 workload, not this one
    assert d["checked"]["ok"] and d["checked"]["bit_equal"] and d["checked"]["timed_output_equals_pass_by_pass_rerun"]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert 0 < d["sweep"]["frac"] <= 1


def test_flowdenoising_class_mirrors_par(fdn, oracle):
    vol = _vol((8, 34, 36), seed=6)
    ks = [fdn.get_gaussian_kernel(0.5)] * 3
    v = vol.copy()
    fd = fdn.FlowDenoising(4, v, 0, 5, fdn.get_flow_with_prev_flow, fdn.warp_slice)
    assert fd.filter(ks) is None                                                      # par:285-290 returns None
    want = oracle.OF_filter(vol, ks, 0, 5, border_mode=1)
    assert rel_err(fd.filtered_vol, want) < TIGHT_TOL and np.array_equal(fd.vol, fd.filtered_vol)


@pytest.mark.gpu_subprocess
def test_pair_operators_on_one_shared_handle(fdn, tmp_path):
    """FDN_PAIR_HANDLES=1: every pool thread's get_flow / warp_slice goes through the ONE process-wide handle; the C ABI's
    lock per handle serialises them: eight threads = one thread, bit for bit."""
    from conftest import run_in_fresh_process
    vol = _vol((17, 40, 83), seed=24)
    ks = [fdn.get_gaussian_kernel(1.0), None, fdn.get_gaussian_kernel(0.5)]
    code = (
        "import flowdenoising_amd as fdn, _par_pool\n"
        "from flowdenoising_amd import operators\n"
        "ks = [k0, None, k2]\n"
        "out['pool'] = _par_pool.par_sweep(fdn.get_flow_with_prev_flow, fdn.warp_slice, vol, ks, 0, 5, 8)\n"
        "out['single'] = _par_pool.par_sweep(fdn.get_flow_with_prev_flow, fdn.warp_slice, vol, ks, 0, 5, 1)\n"
        "out['made'] = np.array([operators._pair_pool(0).made])\n")
    got, _ = run_in_fresh_process(code, {"vol": vol, "k0": ks[0], "k2": ks[2]}, tmp_path, env={"FDN_PAIR_HANDLES": "1"})
    assert got["made"][0] == 1
    assert np.array_equal(got["pool"], got["single"])


@pytest.mark.gpu_subprocess
def test_pair_handle_pool_follows_the_options_of_the_process_wide_handle(fdn, oracle, tmp_path):
    """What a caller sets on the process-wide handle (here remap_model = 1 and opencv_fma = 1 with a pyramid) holds for the
    pair operators whichever pool handle serves a thread -- also when it is changed after the pool's handles exist."""
    from conftest import run_in_fresh_process
    import _par_pool
    vol = _vol((9, 70, 90), seed=25)
    ks = [fdn.get_gaussian_kernel(0.5), None, None]
    code = (
        "import flowdenoising_amd as fdn, _par_pool\n"
        "from flowdenoising_amd import operators\n"
        "ks = [k0, None, None]\n"
        "out['default_pool'] = _par_pool.par_sweep(fdn.get_flow_with_prev_flow, fdn.warp_slice, vol, ks, 2, 5, 4)\n"
        "out['made'] = np.array([operators._pair_pool(0).made])\n"
        "operators.handle().set_option('remap_model', 1)\n"
        "operators.handle().set_option('opencv_fma', 1)\n"
        "out['pool'] = _par_pool.par_sweep(fdn.get_flow_with_prev_flow, fdn.warp_slice, vol, ks, 2, 5, 4)\n"
        "out['single'] = _par_pool.par_sweep(fdn.get_flow_with_prev_flow, fdn.warp_slice, vol, ks, 2, 5, 1)\n"
        "operators.release_pair_handles()\n")
    got, _ = run_in_fresh_process(code, {"vol": vol, "k0": ks[0]}, tmp_path)
    assert got["made"][0] >= 2, "the pool never made a second handle: the test did not test it"
    assert np.array_equal(got["pool"], got["single"])
    assert not np.array_equal(got["pool"], got["default_pool"])

    def o_warp(reference, flow):
        H, W = flow.shape[:2]
        m = np.empty((H, W, 2), np.float32)
        m[..., 0] = (flow[..., 0].astype(np.float64) + np.arange(W)[None, :]).astype(np.float32)
        m[..., 1] = (flow[..., 1].astype(np.float64) + np.arange(H)[:, None]).astype(np.float32)
        return oracle.remap_any(np.ascontiguousarray(reference), m)

    def o_flow(reference, target, l, w, prev_flow):
        return oracle.get_flow(np.asarray(reference, np.float32), np.asarray(target, np.float32), l, w, prev_flow)

    oracle.set_remap_model(1)
    oracle.set_fma(1)
    try:
        want = _par_pool.par_sweep(o_flow, o_warp, vol, ks, 2, 5, 1)
    finally:
        oracle.set_remap_model(0)
        oracle.set_fma(0)
    assert np.array_equal(got["pool"], want)


@pytest.mark.gpu_subprocess
@pytest.mark.parametrize("dtype", [np.float32, np.int16])
def test_pair_operators_under_pars_thread_pool(fdn, oracle, tmp_path, dtype):
    """The seam par injects into (`FlowDenoising(P, vol, l, w, get_flow, warp_slice)`, par:506) as par itself uses it:
    P = 8 pool threads call get_flow / warp_slice at once on views of one shared volume (par:187-193, 299-327;
    tests/_par_pool.py restates the scheduler and the slice loops).  Concurrent callers of
    `flowdenoising_amd.get_flow_with_prev_flow` / `warp_slice` each take a handle from the operators' small pool (the
    process-wide handle first; FDN_PAIR_HANDLES = 1 puts them all on that one, which the C ABI serialises with its lock
    per handle -- test_pair_operators_on_one_shared_handle); the eight-thread result must be the single-thread one bit
    for bit -- and both the oracle's for the same loops.
    Axis lengths 17 / 66 / 83 leave a remainder round on every pass.  Runs in a process of its own (conftest)."""
    from conftest import run_in_fresh_process
    import _par_pool
    if dtype == np.float32:
        vol = _vol((17, 66, 83), seed=21)
    else:
        v = _vol((17, 66, 83), seed=22)
        vol = (np.round((v - v.min()) / (v.max() - v.min()) * 4095) - 1500).astype(np.int16)
    ks = [fdn.get_gaussian_kernel(1.0), fdn.get_gaussian_kernel(0.5), fdn.get_gaussian_kernel(1.0)]
    code = (
        "import flowdenoising_amd as fdn, _par_pool, threading\n"
        "ks = [k0, k1, k2]\n"
        "seen = set()\n"
        "def gf(reference, target, l, w, prev_flow):\n"
        "    seen.add(threading.get_ident())\n"
        "    return fdn.get_flow_with_prev_flow(reference, target, l, w, prev_flow)\n"
        "out['pool'] = _par_pool.par_sweep(gf, fdn.warp_slice, vol, ks, 0, 5, 8)\n"
        "out['threads'] = np.array([len(seen)])\n"
        "out['single'] = _par_pool.par_sweep(fdn.get_flow_with_prev_flow, fdn.warp_slice, vol, ks, 0, 5, 1)\n"
        "v = vol.copy()\n"
        "fdn.FlowDenoising(8, v, 0, 5, fdn.get_flow_with_prev_flow, fdn.warp_slice).filter(ks)\n"
        "out['batched'] = v\n")
    got, _ = run_in_fresh_process(code, {"vol": vol, "k0": ks[0], "k1": ks[1], "k2": ks[2]}, tmp_path)
    assert got["threads"][0] >= 4, "the pool's threads did not all reach the library"
    assert got["pool"].dtype == vol.dtype and np.array_equal(got["pool"], got["single"])

    def o_flow(reference, target, l, w, prev_flow):
        return oracle.get_flow(np.asarray(reference, np.float32), np.asarray(target, np.float32), l, w, prev_flow)

    def o_warp(reference, flow):
        H, W = flow.shape[:2]
        m = np.empty((H, W, 2), np.float32)
        m[..., 0] = (flow[..., 0].astype(np.float64) + np.arange(W)[None, :]).astype(np.float32)     # seq:53-55
        m[..., 1] = (flow[..., 1].astype(np.float64) + np.arange(H)[:, None]).astype(np.float32)
        return oracle.remap_any(np.ascontiguousarray(reference), m)

    want = _par_pool.par_sweep(o_flow, o_warp, vol, ks, 0, 5, 1)
    if dtype == np.float32:
        assert rel_err(got["pool"], want) < TIGHT_TOL
    else:
        assert np.array_equal(got["pool"], want)
    # and the batched sweep the package runs for the same class (one launch per chain step of every target)
    assert rel_err(got["batched"], want) < TIGHT_TOL if dtype == np.float32 else np.array_equal(got["batched"], want)


def test_slab_engine_on_hip_backend_world1(fdn, oracle):
    torch = pytest.importorskip("torch")
    from flowdenoising_amd import _lib
    from flowdenoising_amd.distributed import SlabEngine, SlabPlan
    from flowdenoising_amd.operators import handle
    vol = _vol((9, 36, 40), seed=8)
    h = handle()
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        eng = SlabEngine(SlabPlan(vol.shape, 1, 0), h, None)
        ks = [fdn.get_gaussian_kernel(s) for s in (1.0, 0.5, 0.5)]
        params = _lib.SweepParams(0, 5, 3, 5, 1.2, 0, 1, 1)
        t = torch.from_numpy(vol).cuda()
        assert abs(float(eng.global_mean(t)) - float(vol.mean())) <= 1.2e-7 * abs(float(vol.mean()))
        out = eng.filter_3d(t, ks, params, mean=vol.mean()).cpu().numpy()   # seq:420's own value
    finally:
        h.reset_stream()
    want = oracle.OF_filter(vol, ks, 0, 5)
    assert rel_err(out, want) < TIGHT_TOL


@pytest.mark.gpu_subprocess
@pytest.mark.parametrize("world,shape,sig,border,l", [(2, (12, 70, 150), "1.0,0.5,1.0", 0, 0), (3, (13, 64, 128), "1.0,-,0.5", 1, 1),
                                                      (4, (10, 66, 140), "1.5,0.5,1.0", 0, 0)])    # 4 ranks + this process: under the box's limit of 6
def test_slab_engine_on_hip_backend_multi_rank(fdn, tmp_path, world, shape, sig, border, l):
    """N > 1 on the HIP backend: `world` processes, each with its own fdn handle, run the slab engine (persistent
    buffers, fdn_permute_dev packing, one exchange per pass, the chunk-sum mean) and must reproduce the single-GPU
    OF_filter of the whole volume bit for bit.  On a node with >= `world` GPUs (the driver's scaling box) every rank
    has its own GPU and the exchange is RCCL; on the one-GPU box the ranks share GPU 0 and the exchange is staged
    through the host over gloo (tests/_dist_hip_worker.py)."""
    import socket
    from flowdenoising_amd import _lib
    vol = _vol(shape, seed=33)
    np.save(tmp_path / "v.npy", vol)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(world):
        env = {**os.environ, "RANK": str(r), "WORLD_SIZE": str(world), "LOCAL_RANK": str(r), "MASTER_ADDR": "127.0.0.1",
               "MASTER_PORT": str(port)}
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_hip_worker.py"), str(tmp_path / "v.npy"),
                                       str(tmp_path / "o"), sig, str(border), str(l), "5"], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    print(outs[0][0].strip())
    got = np.concatenate([np.load(f"{tmp_path}/o.{r}.npy") for r in range(world)])
    ks = [None if s == "-" else fdn.get_gaussian_kernel(float(s)) for s in sig.split(",")]
    want = fdn.OF_filter(vol, ks, l, 5, border_mode=border)
    assert np.array_equal(got, want)
    assert np.load(f"{tmp_path}/o.mean.npy") == vol.mean()
    # the C-level entry point fdn_filter_3d_sharded (schedule and mean in libflowdn.so, transport in two callbacks): same bits
    got_c = np.concatenate([np.load(f"{tmp_path}/o.c.{r}.npy") for r in range(world)])
    assert np.array_equal(got_c, want)


@pytest.mark.gpu_subprocess
def test_rccl_world_size_1_carries_the_slab_engine(fdn, tmp_path):
    """One real RCCL communicator on the one GPU every box has: a world-size-1 `nccl` process group carries the slab
    engine in loopback mode -- the blocks of every exchange travel as send-to-self messages inside the batched group
    (ncclSend / ncclRecv under ncclGroupStart / End, exactly the calls of an N > 1 run), the mean goes through
    all_gather -- and the result equals the single-GPU OF_filter bit for bit."""
    import socket
    vol = _vol((12, 70, 150), seed=35)
    np.save(tmp_path / "v.npy", vol)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {**os.environ, "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)}
    sig = "1.0,0.5,1.0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_dist_hip_worker.py"), str(tmp_path / "v.npy"), str(tmp_path / "o"),
                        sig, "0", "0", "5", "loopback"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    print(r.stdout.strip())
    assert "backend nccl world 1" in r.stdout
    ks = [fdn.get_gaussian_kernel(float(s)) for s in sig.split(",")]
    want = fdn.OF_filter(vol, ks, 0, 5)
    assert np.array_equal(np.load(f"{tmp_path}/o.0.npy"), want)
    assert np.array_equal(np.load(f"{tmp_path}/o.c.0.npy"), want)             # fdn_filter_3d_sharded, one rank
    assert np.array_equal(np.load(f"{tmp_path}/o.gathered.npy"), want)
    assert np.load(f"{tmp_path}/o.mean.npy") == vol.mean()


@pytest.mark.gpu_subprocess
def test_bench_gpus_2_from_a_bare_shell(fdn):
    """`python3 bench.py --gpus 2 ...` the way the driver calls it (no torch.distributed.run around it): bench.py starts
    its ranks as a child process, relays rank 0's line and exits with the child's code.  On a one-GPU box the two ranks
    share GPU 0 (a rehearsal, labelled so); on a node with two GPUs the same command runs over RCCL."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shape", "24,96,160", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["value"] > 0
    assert d["backend"].split()[0] in ("rccl", "shm"), d["backend"]         # the native transport: no torch.distributed
    assert len(d["phase_ms_per_step_per_rank"]["compute"]) == 2 and "exchange" in d["phase_ms_per_step_per_rank"]
    # an N > 1 line carries its own correctness check: the gathered sharded output equals a single-GPU rerun bit for bit,
    # and the oracle recomputed one target slice per pass
    c = d["checked"]
    assert c["ok"] and c["sharded_output_equals_single_gpu_rerun"] and c["bit_equal"] and len(c["slices"]) == 3


@pytest.mark.gpu_subprocess
def test_bench_gpus_2_under_torch_distributed_run(fdn):
    """The round driver's launch for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...`.  torch.distributed.run only starts the ranks: they find each other through
    the rendezvous directory flowdenoising_amd.launch derives from their common parent and the port (no process group is
    initialised), exchange through the native transport, and rank 0 prints the one line -- with its `checked` block."""
    import json
    import socket
    pytest.importorskip("torch")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "FDN_RANK", "FDN_WORLD", "FDN_RDV")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shape", "24,96,160", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0 and d["backend"].split()[0] in ("rccl", "shm")
    assert d["checked"]["ok"] and d["checked"]["sharded_output_equals_single_gpu_rerun"]
    import glob
    import tempfile
    assert not glob.glob(os.path.join("/dev/shm", f"fdn_rdv_{os.getuid()}_{port}_*")) and not glob.glob(os.path.join(tempfile.gettempdir(), f"fdn_rdv_{os.getuid()}_{port}_*"))


@pytest.mark.gpu_subprocess
def test_bench_gpus_2_python_engine(fdn):
    """--engine python: the torch.distributed slab engine (gloo rehearsal on a one-GPU box, RCCL with two GPUs), started by
    bench.py itself under torch.distributed.run."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--engine", "python", "--shape", "24,96,160", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["backend"] in ("nccl", "gloo") and d["value"] > 0
    assert len(d["phase_ms_per_step_per_rank"]["compute"]) == 2


@pytest.mark.parametrize("l,w", [(0, 5), (1, 5), (0, 15)])
def test_workspace_limit_bounds_every_buffer(fdn, oracle, l, w):
    """fdn_set_workspace_limit caps everything the handle owns (stack, re-oriented pass output, intermediate volumes,
    expansions, flows): passes are cut into chunks of target slices, results do not change by a bit, and a cap too
    small for one target slice is an error."""
    from flowdenoising_amd._lib import FlowdnError
    from flowdenoising_amd.operators import handle
    vol = _vol((14, 40, 70), seed=9)
    ks = [fdn.get_gaussian_kernel(1.0), fdn.get_gaussian_kernel(0.5), fdn.get_gaussian_kernel(1.0)]
    want = fdn.OF_filter(vol, ks, l, w)
    h = handle()
    limit = 8 * vol.nbytes + (6 << 20 if l else 0)        # two intermediate volumes + chunks of a few target slices
    h.set_workspace_limit(limit)
    try:
        assert h.workspace_bytes() == 0
        got = fdn.OF_filter(vol, ks, l, w)
        assert h.workspace_bytes() <= limit
        h.set_workspace_limit(vol.nbytes // 4)
        with pytest.raises(FlowdnError, match="too small"):
            fdn.OF_filter(vol, ks, l, w)
    finally:
        h.set_workspace_limit(0)
    assert np.array_equal(got, want)
    assert rel_err(got, oracle.OF_filter(vol, ks, l, w, nthreads=8)) < TIGHT_TOL


@pytest.mark.parametrize("shape,sig,border,use_of", [((3, 40, 70), 1.0, 1, True), ((5, 36, 40), 1.5, 0, False), ((4, 34, 36), 1.0, 1, False)])
def test_workspace_limit_with_wrapped_ends_and_short_axes(fdn, oracle, shape, sig, border, use_of):
    """Chunked passes where the halo (K//2 = 4 or 6 slices) is longer than the axis and the ends wrap around (par:312)
    or the plain Gaussian runs (-n): same bits as the unchunked pass and as the oracle."""
    from flowdenoising_amd import _lib
    from flowdenoising_amd.operators import _params, handle
    vol = _vol(shape, seed=19)
    ks = [fdn.get_gaussian_kernel(sig)] * 3
    p = _params(0, 5, use_of, border, True)
    h = handle()
    want = h.filter_3d(vol, ks, vol.mean(), p)
    from flowdenoising_amd._lib import FlowdnError
    done = 0
    try:
        for limit in (4 << 20, 2 << 20, 1200 << 10, 1000 << 10, 900 << 10, 800 << 10, 600 << 10, 400 << 10):    # ever fewer slices per chunk
            h.set_workspace_limit(limit)
            try:
                got = h.filter_3d(vol, ks, vol.mean(), p)
            except FlowdnError as e:
                assert "too small" in str(e)
                break
            assert h.workspace_bytes() <= limit and np.array_equal(got, want), limit
            done += 1
    finally:
        h.set_workspace_limit(0)
    assert done >= 2
    assert np.array_equal(got, want)
    ref = oracle.OF_filter(vol, ks, 0, 5, border_mode=border, nthreads=8) if use_of else None
    if use_of:
        assert rel_err(got, ref) < TIGHT_TOL


def test_config2_under_a_16_gib_workspace_limit(fdn):
    """configs[2] (2 GiB volume, 26 GiB of scratch when unlimited) completes bit-equal when the handle may own 16 GiB."""
    import torch
    from flowdenoising_amd import _lib, synth
    shape = (512, 1024, 1024)
    vol = synth.make_volume(shape, seed=1234 + 3, amplitude=100.0, xp=torch, device=torch.device("cuda", 0))
    k = _lib.gaussian_kernel(2.0)
    params = _lib.SweepParams(0, 5, 3, 5, 1.2, _lib.BORDER_MEAN_PAD, 1, 1)
    h = _lib.Handle(0)
    try:
        h.set_stream(torch.cuda.current_stream().cuda_stream)
        mean = h.mean_dev(vol.data_ptr(), vol.numel())
        a, b = torch.empty_like(vol), torch.empty_like(vol)
        h.filter_3d_dev(vol.data_ptr(), a.data_ptr(), shape, [k, k, k], mean, params)
        torch.cuda.synchronize()
        unlimited = h.workspace_bytes()
        h.set_workspace_limit(16 << 30)
        h.filter_3d_dev(vol.data_ptr(), b.data_ptr(), shape, [k, k, k], mean, params)
        torch.cuda.synchronize()
        assert h.workspace_bytes() <= (16 << 30) < unlimited
        assert torch.equal(a, b)
    finally:
        h.close()
        del vol
        torch.cuda.empty_cache()


def test_wide_window_small_image(fdn, oracle):
    """winsize 15 (BASELINE configs[4]) on an image narrower than one band of the one-iteration kernel."""
    vol = _vol((6, 48, 52), seed=10)
    k = fdn.get_gaussian_kernel(0.5)
    got = fdn.OF_filter_along_Z(vol, k, 0, 15, vol.mean())
    assert rel_err(got, oracle.filter_along_axis(vol, 0, k, 0, 15, vol.mean())) < TIGHT_TOL


@pytest.mark.parametrize("w", [10, 11, 13, 15, 17, 21, 31])
@pytest.mark.parametrize("l", [0, 2])
def test_iter_kernel_window_sizes(fdn, oracle, w, l):
    """winsize >= 10 runs on k_farneback_iter (one Farneback iteration per launch, matrices in an LDS ring):
    compile-time windows (11, 15) and the runtime-width build, multi-band images with interior and edge
    bands, chains of four steps, with and without pyramid; bit-equal to the oracle."""
    vol = _vol((7, 130, 300), seed=61 + w)
    k = fdn.get_gaussian_kernel(1.0)
    got = fdn.OF_filter(vol, [k, None, k], l, w)
    want = oracle.OF_filter(vol, [k, None, k], l, w, nthreads=8)
    assert np.array_equal(got, want), (w, l, rel_err(got, want))


def test_paths_can_be_switched_on_a_live_handle(fdn, oracle):
    """fdn_set_option("path"): the same sweep on the 3-iteration fused kernel, the staged kernels and the
    one-iteration kernels of one handle gives the same bits; unknown options are errors."""
    from flowdenoising_amd._lib import FlowdnError
    from flowdenoising_amd.operators import handle
    vol = _vol((8, 70, 200), seed=14)
    k = fdn.get_gaussian_kernel(1.0)
    h = handle()
    outs = []
    try:
        for path in (0, 1, 2):
            h.set_option("path", path)
            outs.append(fdn.OF_filter_along_Z(vol, k, 1, 5, vol.mean()))
        with pytest.raises(FlowdnError):
            h.set_option("no_such_option", 1)
    finally:
        h.set_option("path", 0)
    want = oracle.filter_along_axis(vol, 0, k, 1, 5, vol.mean(), nthreads=8)
    for o in outs:
        assert np.array_equal(o, want)


def test_strict_order_refuses_rows_it_cannot_hold(fdn):
    """Strict mode never falls back silently: a row too wide for the LDS of its serial kernel (the row's vertical sums, 40 B
    per column, must fit next to a segment of window sums: about 4 000 columns) is an error."""
    from flowdenoising_amd._lib import FlowdnError
    from flowdenoising_amd.operators import handle
    vol = np.zeros((3, 4, 4200), np.float32)
    h = handle()
    h.set_option("strict_order", 1)
    try:
        with pytest.raises(FlowdnError, match="strict"):
            fdn.OF_filter_along_Z(vol, fdn.get_gaussian_kernel(0.5), 0, 5, 0.0)
    finally:
        h.set_option("strict_order", 0)


@pytest.mark.gpu_subprocess
@pytest.mark.parametrize("env", [{}, {"FDN_FUSED_OCC": "3"}, {"FDN_FUSED_OCC": "4"}, {"FDN_FUSED_OCC": "5"}, {"FDN_FUSED_OCC": "8"}, {"FDN_FORCE_STAGED": "1"}, {"FDN_PATH": "2"},
                                 {"FDN_SUB_BATCHES": "1"}, {"FDN_SUB_BATCHES": "2"}, {"FDN_SUB_BATCHES": "1", "FDN_PATH": "2"}, {"FDN_SUB_BATCHES": "2", "FDN_PATH": "2"}])
def test_kernel_variants_agree_bit_for_bit(fdn, oracle, tmp_path, env):
    """Every implementation of the chain step (the fused stage-pipelined kernel in its builds for 3, 4
    and 5 workgroups per CU -- different LDS windows and unrolls --, its two-bands-per-workgroup build, the staged per-stage
    kernels and the one-iteration kernels; the batch's targets on one stream or as two sub-batches on two) must give the
    oracle's bits on a multi-band image with interior and edge bands."""
    vol = _vol((10, 70, 300), seed=12)
    np.save(tmp_path / "v.npy", vol)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import flowdenoising_amd as fd; v = np.load(%r); "
            "k = fd.get_gaussian_kernel(1.0); np.save(%r, fd.OF_filter(v, [k, None, k], 0, 5))"
            % (ROOT, str(tmp_path / "v.npy"), str(tmp_path / "o.npy")))
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    k = oracle.get_gaussian_kernel(1.0)
    want = oracle.OF_filter(vol, [k, None, k], 0, 5, nthreads=8)
    assert np.array_equal(np.load(tmp_path / "o.npy"), want)


# ---- full-size images (BASELINE.json configs[1] and configs[2]) ---------------------------------
@pytest.mark.parametrize("axis,shape,levels,w", [(0, (40, 1024, 1024), 0, 5), (1, (512, 40, 1024), 0, 5), (2, (512, 1024, 40), 0, 5),
                                                 (0, (24, 1024, 1024), 3, 5), (1, (512, 20, 1000), 3, 5),
                                                 (0, (24, 1024, 1024), 0, 7), (2, (512, 1000, 20), 1, 9)])
def test_full_size_images_spot_parity(fdn, oracle, axis, shape, levels, w):
    """sigma=2 (K=17) sweeps over full-size images (1024x1024 for Z; 512x1024 for Y and X), without and
    with a 3-level pyramid (par's default): two target slices are compared with the oracle run on the
    17-slice sub-volume that feeds them; winsize 7 and 9 exercise the fused kernel's wider-window builds."""
    from flowdenoising_amd.synth import make_volume
    vol = make_volume(shape, seed=1234 + 3, amplitude=100.0)
    k = fdn.get_gaussian_kernel(2.0)
    mean = vol.mean()
    fn = [fdn.OF_filter_along_Z, fdn.OF_filter_along_Y, fdn.OF_filter_along_X][axis]
    got = fn(vol, k, levels, w, mean)
    n = shape[axis]
    for t in (3, n // 2):            # one target whose window reaches the mean padding, one interior
        lo, hi = max(0, t - 8), min(n, t + 9)
        sub = np.take(vol, range(lo, hi), axis=axis)
        want = oracle.filter_axis_range(sub, axis, k, levels, w, mean, t - lo, t - lo + 1, nthreads=1)
        assert rel_err(np.take(got, [t], axis=axis), np.take(want, [t - lo], axis=axis)) < TIGHT_TOL


def test_full_size_roll_equivariance_wrap(fdn):
    """Size-independent property at configs[1] size (256x512x512, Z only): with wrap-around ends
    (par:312) rolling the volume along Z rolls the result, bit for bit."""
    from flowdenoising_amd import _lib
    from flowdenoising_amd.synth import make_volume
    vol = make_volume((256, 512, 512), seed=1234 + 2, amplitude=100.0)
    k = fdn.get_gaussian_kernel(2.0)
    a = fdn.OF_filter_along_Z(vol, k, 0, 5, 0.0, border_mode=_lib.BORDER_WRAP)
    b = fdn.OF_filter_along_Z(np.roll(vol, 37, axis=0), k, 0, 5, 0.0, border_mode=_lib.BORDER_WRAP)
    assert np.array_equal(np.roll(a, 37, axis=0), b)


def test_full_size_identical_slices(fdn):
    """Every slice equal (1024x1024 images): away from the padded ends and the image border the flow
    is exactly zero, so the output is the f32 fold of v * w_i in the reference's tap order."""
    rng = np.random.default_rng(0)
    import scipy.ndimage
    img = (scipy.ndimage.gaussian_filter(rng.standard_normal((1024, 1024)), 3) * 2000).astype(np.float32)
    vol = np.broadcast_to(img, (24, 1024, 1024)).copy()
    k = fdn.get_gaussian_kernel(2.0)
    got = fdn.OF_filter_along_Z(vol, k, 0, 5, vol.mean())
    acc = np.zeros_like(img)
    order = list(range(7, -1, -1)) + [8] + list(range(9, 17))     # seq:95, 108, 110
    for i in order:
        acc = (acc.astype(np.float64) + img.astype(np.float64) * k[i]).astype(np.float32)
    assert np.array_equal(got[12, :900, :900], acc[:900, :900])


def test_wide_kernel_on_a_gib_volume_spot_parity(fdn, oracle):
    """BASELINE.json configs[3] on one GPU: 1024 x 1024 x 1024 float32, sigma = 4 (K = 33, chains of 16
    steps either side), volume resident in HBM; the Z pass is spot-checked against the oracle run on
    the 33-slice sub-volume that feeds one target slice."""
    import torch
    from flowdenoising_amd import _lib, synth
    shape = (1024, 1024, 1024)
    dev = torch.device("cuda", 0)
    vol = synth.make_volume(shape, seed=1234 + 4, amplitude=100.0, xp=torch, device=dev)
    out = torch.empty_like(vol)
    h = _lib.Handle(0)
    try:
        h.set_stream(torch.cuda.current_stream().cuda_stream)
        k = _lib.gaussian_kernel(4.0)
        params = _lib.SweepParams(0, 5, 3, 5, 1.2, _lib.BORDER_MEAN_PAD, 1, 1)
        mean = h.mean_dev(vol.data_ptr(), vol.numel())
        h.filter_3d_dev(vol.data_ptr(), out.data_ptr(), shape, [k, None, None], mean, params)
        torch.cuda.synchronize()
        t = 500
        sub = to_host(vol[t - 16:t + 17])
        got = to_host(out[t])
    finally:
        h.close()
        del vol, out
        torch.cuda.empty_cache()
    want = oracle.filter_axis_range(sub, 0, k, 0, 5, mean, 16, 17, nthreads=16)
    assert np.array_equal(got, want[16])


@pytest.mark.parametrize("axis,shape", [(1, (1024, 40, 1024)), (2, (1024, 1024, 40))])
def test_wide_kernel_y_and_x_passes_spot_parity(fdn, oracle, axis, shape):
    """BASELINE.json configs[3] has sigma = 4 on all three axes (K = 33, 33, 33): the Y and X passes on its
    1024 x 1024 images (Z x X and Z x Y planes of the 1024^3 volume, seq:255 / seq:333), on stacks thin along the
    pass's own axis; two target slices -- one whose chain reaches the mean padding, one with all 32 neighbours
    inside -- against the oracle run on the sub-volume that feeds them."""
    from flowdenoising_amd.synth import make_volume
    vol = make_volume(shape, seed=1234 + 4, amplitude=100.0)
    k = fdn.get_gaussian_kernel(4.0)
    assert k.size == 33
    mean = vol.mean()
    fn = [fdn.OF_filter_along_Z, fdn.OF_filter_along_Y, fdn.OF_filter_along_X][axis]
    got = fn(vol, k, 0, 5, mean)
    n = shape[axis]
    for t in (5, n // 2):
        lo, hi = max(0, t - 16), min(n, t + 17)
        sub = np.take(vol, range(lo, hi), axis=axis)
        want = oracle.filter_axis_range(sub, axis, k, 0, 5, mean, t - lo, t - lo + 1, nthreads=16)
        assert np.array_equal(np.take(got, [t], axis=axis), np.take(want, [t - lo], axis=axis))


def test_refused_host_registration_does_not_poison_the_next_call(fdn, oracle):
    """hipHostRegister refuses some ranges (here: one that is not mapped at all); fdn_host_register then reports an error,
    Handle.host_register returns False and the callers fall back to pageable copies.  The runtime's sticky last-error of
    that refusal must not surface in the next kernel-launch check (advisor finding, round 2): a filter run right after it
    succeeds and is right."""
    import ctypes
    from flowdenoising_amd.operators import handle
    h = handle()
    vol = _vol((7, 40, 70), seed=31)
    ks = [fdn.get_gaussian_kernel(0.5), fdn.get_gaussian_kernel(1.0), None]
    want = oracle.OF_filter(vol, ks, 0, 5)
    rc = h._lib.fdn_host_register(h._h, ctypes.c_void_p(1 << 20), ctypes.c_size_t(1 << 20))      # an unmapped range
    assert rc != 0 and b"hipHostRegister" in h._lib.fdn_last_error()
    assert h._lib.fdn_host_unregister(h._h, ctypes.c_void_p(1 << 20)) != 0                          # never registered
    assert np.array_equal(fdn.OF_filter(vol, ks, 0, 5), want)


def test_reserve_allocates_what_the_filter_uses(fdn):
    """fdn_reserve_3d (the CLI runs it while it reads the file): after it, the real call allocates nothing more."""
    from flowdenoising_amd import _lib
    h = _lib.Handle(0)
    try:
        shape = (20, 96, 160)
        ks = [fdn.get_gaussian_kernel(1.0), fdn.get_gaussian_kernel(1.0), fdn.get_gaussian_kernel(0.5)]
        for l, w in ((0, 5), (2, 5), (1, 15)):
            params = _lib.SweepParams(l, w, 3, 5, 1.2, _lib.BORDER_MEAN_PAD, 1, 1)
            h.reserve_3d(shape, [k.size for k in ks], params)
            held = h.workspace_bytes()
            assert held > 0
            vol = _vol(shape, seed=5)
            n = vol.nbytes
            d_in, d_out = h.malloc(n), h.malloc(n)
            try:
                h.h2d(d_in, vol)
                h.filter_3d_dev(d_in, d_out, shape, ks, vol.mean(), params)
                h.synchronize()
                assert h.workspace_bytes() == held, (l, w, held, h.workspace_bytes())
            finally:
                h.free(d_in)
                h.free(d_out)
    finally:
        h.close()


def test_device_statistics_and_casts(fdn):
    """The CLI's host-side numpy passes moved to the GPU: fdn_stats_dev (the statistics seq:529-532 / 547-550 log and
    mrcfile writes into the output header), fdn_convert_dev (seq:517's astype(np.float32) of an integer stack) and
    fdn_truncate_dev (seq:566-571's astype(np.uint8 / np.uint16) of the result, numpy's truncating, wrapping cast)."""
    from flowdenoising_amd.operators import handle
    rng = np.random.default_rng(5)
    h = handle()
    v = (rng.standard_normal(1_000_003) * 300 + 40).astype(np.float32)
    v[17], v[900_000] = -1.75, 70000.5              # a negative voxel wraps, one above 65535 wraps too
    d = h.malloc(v.nbytes)
    d2 = h.malloc(v.nbytes)
    try:
        h.h2d(d, v)
        st = h.stats_dev(d, v.size)
        v64 = v.astype(np.float64)
        assert st["min"] == v.min() and st["max"] == v.max()
        assert abs(st["mean"] - v64.mean()) <= 1e-12 * abs(v64.mean()) + 1e-12 and abs(st["std"] - v64.std()) <= 1e-10 * v64.std()
        for dt in (np.uint16, np.uint8):
            h.truncate_dev(d, dt, d2, v.size)
            got = np.empty(v.size, dtype=dt)
            h.d2h(got, d2)
            with np.errstate(invalid="ignore"):
                assert np.array_equal(got, v.astype(np.int32).astype(dt))     # truncate toward zero, keep the low bits
        for dt in (np.uint16, np.int16, np.uint8, np.int8):
            info = np.iinfo(dt)
            raw = rng.integers(info.min, info.max + 1, size=100_001).astype(dt)
            h.h2d(d, raw)
            h.convert_dev(d, dt, d2, raw.size)
            got = np.empty(raw.size, dtype=np.float32)
            h.d2h(got, d2)
            assert np.array_equal(got, raw.astype(np.float32))
    finally:
        h.free(d)
        h.free(d2)


def test_operator_with_device_side_casts_equals_host_side_casts(fdn):
    """filter_3d_own_mean on a uint16 stack with float32 semantics (a TIFF, seq:517) and the TIFF down-cast of
    seq:566-571 on the device: the same voxels as astype(np.float32) before and astype(np.uint16) after on the host."""
    from flowdenoising_amd.operators import _params, filter_3d_own_mean
    vol = (np.clip(_vol((9, 40, 70), seed=23), 0, None) * 40).astype(np.uint16)
    ks = [fdn.get_gaussian_kernel(1.0), fdn.get_gaussian_kernel(0.5), fdn.get_gaussian_kernel(0.5)]
    stats = {}
    got = filter_3d_own_mean(vol, ks, _params(0, 5), stats=stats, float32_semantics=True, tiff_downcast=True)
    want = fdn.OF_filter(vol.astype(np.float32), ks, 0, 5)
    assert got.dtype == np.uint16 and np.array_equal(got, want.astype(np.uint16))
    assert stats["out"]["max"] == want.max() and stats["in"]["min"] == vol.min()
    small = (vol // 300).astype(np.uint16)          # maximum below 256: the reference writes uint8 (seq:566)
    got8 = filter_3d_own_mean(small, ks, _params(0, 5), float32_semantics=True, tiff_downcast=True)
    assert got8.dtype == np.uint8 and np.array_equal(got8, fdn.OF_filter(small.astype(np.float32), ks, 0, 5).astype(np.uint8))


def test_non_finite_voxels_do_not_derail_the_kernels(fdn):
    """A NaN and an Inf voxel (dead detector pixels happen): every gather / remap index is clamped, so the
    sweep completes, and the damage stays in the columns the running sums carry it down (OpenCV's box
    filter does the same): columns far from both voxels stay finite."""
    vol = _vol((12, 64, 160), seed=5)
    vol[5, 20, 30] = np.nan
    vol[6, 40, 100] = np.inf
    k = fdn.get_gaussian_kernel(1.0)
    for l in (0, 1):
        out = fdn.OF_filter_along_Z(vol, k, l, 5, np.float32(100.0))
        assert out.shape == vol.shape and out.dtype == np.float32
        assert np.isfinite(out[:, :, 56:76]).all()
        assert not np.isfinite(out[5, 20:, 28:33]).all()


@pytest.mark.gpu_subprocess
@pytest.mark.parametrize("l,border,chunk", [(0, 0, 3), (1, 0, 4), (0, 1, 5), (0, 0, None)])
def test_streamed_filter_equals_resident_filter(fdn, tmp_path, l, border, chunk):
    """Out-of-core mode (volume on the host, chunks of a pass's slices on the GPU): bit-identical to the
    resident OF_filter, with mean-padded and wrap-around volume ends, with a pyramid, and for no_OF.
    (The streamed runs -- worker threads with handles of their own -- happen in a fresh process: conftest.run_in_fresh_process.)"""
    from conftest import run_in_fresh_process
    vol = _vol((11, 70, 90), seed=17)
    ks = [fdn.get_gaussian_kernel(1.0), fdn.get_gaussian_kernel(0.5), fdn.get_gaussian_kernel(1.5)]
    want = fdn.OF_filter(vol, ks, l, 5, border_mode=border)
    code = ("from flowdenoising_amd import streaming\n"
            "ks = [k0, k1, k2]\nchunk = None if int(chunk) < 0 else int(chunk)\n"
            "out['of'] = streaming.OF_filter_streamed(vol, ks, int(l), 5, chunk, border_mode=int(border))\n"
            "out['no_of'] = streaming.no_OF_filter_streamed(vol, ks, chunk)\n")
    got, _ = run_in_fresh_process(code, dict(vol=vol, k0=ks[0], k1=ks[1], k2=ks[2], l=l, border=border, chunk=-1 if chunk is None else chunk), tmp_path)
    assert np.array_equal(got["of"], want)
    assert np.array_equal(got["no_of"], fdn.no_OF_filter(vol, ks))


@pytest.mark.gpu_subprocess
def test_streamed_filter_survives_a_failing_chunk(fdn, tmp_path):
    """A chunk that fails ends the call with its error only after the chunks in flight have ended (they copy to and from
    page-locked arrays that the call then releases), chunks not yet started never run -- and the kept workers serve the next
    call, bit-identical to the resident filter."""
    from conftest import run_in_fresh_process
    vol = _vol((14, 70, 90), seed=23)
    ks = [fdn.get_gaussian_kernel(1.0), fdn.get_gaussian_kernel(0.5), fdn.get_gaussian_kernel(1.0)]
    want = fdn.OF_filter(vol, ks, 0, 5)
    code = ("import threading\n"
            "from flowdenoising_amd import streaming, _lib\n"
            "ks = [k0, k1, k2]\n"
            "real = _lib.Handle.sweep_stack_dev\n"
            "calls = []; lock = threading.Lock()\n"
            "def flaky(self, *a, **kw):\n"
            "    with lock:\n"
            "        calls.append(1); n = len(calls)\n"
            "    if n == 3:\n"
            "        raise _lib.FlowdnError('injected: chunk 3 fails')\n"
            "    return real(self, *a, **kw)\n"
            "_lib.Handle.sweep_stack_dev = flaky\n"
            "try:\n"
            "    streaming.OF_filter_streamed(vol, ks, 0, 5, 2)\n"
            "    out['raised'] = np.zeros(1)\n"
            "except _lib.FlowdnError as e:\n"
            "    out['raised'] = np.ones(1) * ('injected' in str(e))\n"
            "out['calls_first'] = np.array([len(calls)])\n"
            "_lib.Handle.sweep_stack_dev = real\n"
            "out['of'] = streaming.OF_filter_streamed(vol, ks, 0, 5, 2)\n"
            "streaming.release_workers()\n")
    got, _ = run_in_fresh_process(code, dict(vol=vol, k0=ks[0], k1=ks[1], k2=ks[2]), tmp_path)
    assert got["raised"][0] == 1
    assert 3 <= int(got["calls_first"][0]) <= 5               # the failing chunk and at most the two in flight beside it (three workers): the Z pass has 7, Y and X never start
    assert np.array_equal(got["of"], want)


@pytest.mark.gpu_subprocess
def test_strict_order_mode_reproduces_opencvs_horizontal_running_sum(oracle, tmp_path):
    """The fast kernels sum the box filter's horizontal window directly; OpenCV runs a serial f64 chain along
    the row.  In one of 7 080 random configurations that 1e-16 difference flipped an f32 rounding (5.8e-5 in
    the output).  FDN_STRICT_ORDER=1 runs OpenCV's chain: that very configuration then equals the OpenCV-order
    oracle bit for bit -- the summation order is the only difference there is."""
    from flowdenoising_amd.synth import make_volume
    shape, axis, l, w, sigma, seed = (395, 3, 544), 1, 3, 3, 1.0, 5000 + 268
    vol = make_volume(shape, seed=seed, amplitude=100.0)
    np.save(tmp_path / "v.npy", vol)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import flowdenoising_amd as fd; v = np.load(%r); "
            "k = fd.get_gaussian_kernel(%r); np.save(%r, fd.OF_filter_along_Y(v, k, %d, %d, v.mean(), border_mode=1))"
            % (ROOT, str(tmp_path / "v.npy"), sigma, str(tmp_path / "o.npy"), l, w))
    k = oracle.get_gaussian_kernel(sigma)
    want = oracle.filter_along_axis(vol, axis, k, l, w, vol.mean(), border_mode=1, nthreads=16)
    outs = {}
    for strict in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], env={**os.environ, "FDN_STRICT_ORDER": strict}, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[strict] = np.load(tmp_path / "o.npy")
    assert np.array_equal(outs["1"], want)
    assert not np.array_equal(outs["0"], want) and rel_err(outs["0"], want) < 1e-4


@pytest.mark.gpu_subprocess
@pytest.mark.parametrize("shape,l,w", [((5, 24, 2048), 0, 5), ((4, 70, 2048), 3, 15), ((3, 20, 3000), 0, 7), ((4, 9, 63), 0, 15), ((3, 5, 3), 0, 5)])
def test_strict_order_on_rows_of_2048_pixels_and_more(fdn, oracle, tmp_path, shape, l, w):
    """(The last two shapes: rows narrower than a segment's minimum are one segment.)
    configs[4]'s images have 2048-pixel rows: the strict mode's serial chain walks such a row in segments (its running
    value stays in a register, one segment of window sums in LDS at a time), so OpenCV's own f64 order can be verified on
    them too -- until round 4 the mode refused rows wider than about 2040.  Z pass, every slice compared with the
    OpenCV-order oracle bit for bit.
    Runs in a process of its own, like the strict test above (round 5: the HIP runtime aborted the interpreter inside this call in
    3 of 13 long sessions -- profiles/history/NOTES_r05.md, section 5 -- and a fresh process keeps such an abort from taking the
    session down with it; the comparison is as strict as before)."""
    from flowdenoising_amd.synth import make_volume
    vol = make_volume(shape, seed=77, amplitude=100.0)
    np.save(tmp_path / "v.npy", vol)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import flowdenoising_amd as fd; v = np.load(%r); "
            "k = fd.get_gaussian_kernel(0.5); np.save(%r, fd.OF_filter_along_Z(v, k, %d, %d, v.mean()))"
            % (ROOT, str(tmp_path / "v.npy"), str(tmp_path / "o.npy"), l, w))
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, "FDN_STRICT_ORDER": "1"}, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(tmp_path / "o.npy")
    k = fdn.get_gaussian_kernel(0.5)
    want = oracle.filter_along_axis(vol, 0, k, l, w, vol.mean(), nthreads=16)
    assert np.array_equal(got, want)


def test_volume_statistics_slice_form_and_nan(fdn):
    """fdn_stats_slices_dev / Handle.stats_volume (the header statistics of the output file, seq:562-564, taken slice by slice so
    that a multi-GPU run reproduces them bit for bit): numpy's min / max exactly, mean and rms to float64 round-off, the same four
    numbers whether the slices are reduced in one call or slab by slab; a NaN voxel makes min and max NaN as numpy's do (seq:566's
    `np.max(filtered) < 256` then picks uint16), in fdn_stats_dev as well."""
    from flowdenoising_amd import _lib
    from flowdenoising_amd.operators import handle
    rng = np.random.default_rng(5)
    v = (rng.standard_normal((9, 37, 53)) * 40 + 7).astype(np.float32)
    h = handle()
    d = h.malloc(v.nbytes)
    try:
        h.h2d(d, v)
        st = h.stats_volume(d, v.shape)
        v64 = v.astype(np.float64)
        assert st["min"] == v.min() and st["max"] == v.max()
        assert abs(st["mean"] - v64.mean()) < 1e-12 and abs(st["std"] - v64.std()) < 1e-11
        per = v[0].size
        rows = np.concatenate([h.stats_slices_dev(d, 4, per, 0.0), h.stats_slices_dev(d + 4 * per * 4, 5, per, 0.0)])      # two "slabs"
        assert np.array_equal(rows, h.stats_slices_dev(d, 9, per, 0.0))
        assert _lib.combine_slice_stats(rows, v.size)["mean"] == st["mean"]
        v[4, 5, 6] = np.nan
        h.h2d(d, v)
        for st in (h.stats_volume(d, v.shape), h.stats_dev(d, v.size)):
            assert np.isnan(st["min"]) and np.isnan(st["max"]) and np.isnan(st["mean"])
    finally:
        h.free(d)


# ---- the native transports (libflowdn_rccl.so): no torch in any of these processes ---------------------------------------
@pytest.mark.gpu_subprocess
@pytest.mark.parametrize("with_torch", [False, True])
def test_native_rccl_world_size_1_loopback(fdn, tmp_path, with_torch):
    """One real RCCL communicator made by libflowdn_rccl.so itself (ncclGetUniqueId / ncclCommInitRank, no torch in the
    process): fdn_filter_3d_sharded in loopback mode sends the blocks a rank keeps to itself inside the group -- ncclSend /
    ncclRecv under ncclGroupStart / End, the calls of an N > 1 run -- and takes the mean through ncclAllGather; the result
    equals the single-GPU OF_filter bit for bit."""
    vol = _vol((12, 70, 150), seed=35)
    np.save(tmp_path / "v.npy", vol)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "FDN_RANK", "FDN_WORLD", "FDN_RDV")}
    if with_torch:       # bench.py's process: torch imported first, so libflowdn_rccl.so must sit on torch's bundled HIP runtime and RCCL
        pytest.importorskip("torch")
        env["FDN_TEST_IMPORT_TORCH"] = "1"
    else:                # the CLI's rank processes: no torch, /opt/rocm's runtime
        env["FDN_SYSTEM_ROCM"] = "1"
    sig = "1.0,0.5,1.0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_native_worker.py"), str(tmp_path / "v.npy"), str(tmp_path / "o"),
                        sig, "0", "0", "5", "loopback"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    print(r.stdout.strip())
    assert "transport: rccl" in r.stdout and "rank 0 of 1" in r.stdout
    ks = [fdn.get_gaussian_kernel(float(s)) for s in sig.split(",")]
    assert np.array_equal(np.load(f"{tmp_path}/o.0.npy"), fdn.OF_filter(vol, ks, 0, 5))


@pytest.mark.gpu_subprocess
@pytest.mark.parametrize("world,shape,sig,border,l", [(2, (12, 70, 150), "1.0,0.5,1.0", 0, 0), (3, (13, 64, 128), "1.0,-,0.5", 1, 1),
                                                      (4, (10, 66, 140), "1.5,0.5,1.0", 0, 0),
                                                      # (four ranks + this process stay under the box's limit of six GPU processes) slabs of 2-3
                                                      # slices under halos of 6: every rank's halo reaches past its neighbours, wrap-around too
                                                      (4, (9, 66, 64), "1.5,1.5,-", 1, 0)])
def test_native_transport_multi_rank(fdn, tmp_path, world, shape, sig, border, l):
    """N > 1 without torch: `world` rank processes started by flowdenoising_amd.launch.spawn run fdn_filter_3d_sharded on
    the native transport and reproduce the single-GPU OF_filter bit for bit -- over RCCL where the node has a GPU per rank
    (the driver's scaling box), through the shared-memory rehearsal transport where the ranks share GPU 0 (this box)."""
    from flowdenoising_amd import launch
    vol = _vol(shape, seed=33)
    np.save(tmp_path / "v.npy", vol)
    lines = []
    env = dict(os.environ, FDN_SYSTEM_ROCM="1")
    rc = launch.spawn([sys.executable, os.path.join(ROOT, "tests", "_native_worker.py"), str(tmp_path / "v.npy"), str(tmp_path / "o"),
                       sig, str(border), str(l), "5"], world, env=env, relay=lines.append)
    assert rc == 0, "".join(lines)
    print("".join(lines).strip())
    assert "transport:" in "".join(lines)
    got = np.concatenate([np.load(f"{tmp_path}/o.{r}.npy") for r in range(world)])
    ks = [None if s == "-" else fdn.get_gaussian_kernel(float(s)) for s in sig.split(",")]
    assert np.array_equal(got, fdn.OF_filter(vol, ks, l, 5, border_mode=border))


@pytest.mark.gpu_subprocess
@pytest.mark.parametrize("suffix,dtype,compat", [("mrc", np.float32, "seq"), ("mrc", np.int16, "seq"), ("tif", np.uint16, "seq"), ("tif", np.float32, "par")])
def test_cli_gpus_2_is_torch_free_and_writes_the_single_gpu_file(fdn, tmp_path, suffix, dtype, compat):
    """`python flowdenoising.py --gpus 2`: two rank processes (subprocess.Popen, no torch.distributed.run), each reads its
    own Z-slab, filters on the native transport and writes its slab into the output file at its byte offset; torch never
    enters sys.modules (FDN_ASSERT_NO_TORCH) and the file equals the single-GPU run's byte for byte -- header statistics
    and seq:566-571's uint8 / uint16 decision included."""
    from flowdenoising_amd import io as fio
    v = _vol((11, 40, 72), seed=12)
    if np.issubdtype(dtype, np.integer):
        v = np.round((v - v.min()) * (900.0 / (v.max() - v.min()))).astype(dtype)
    src = str(tmp_path / f"in.{suffix}")
    if suffix == "mrc":
        _write_mrc_any(src, v)
    else:
        fio.write_tiff(src, v)
    outs = []
    for gpus in (1, 2):
        dst = str(tmp_path / f"out{gpus}.{suffix}")
        env = {k: val for k, val in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "FDN_RANK", "FDN_WORLD", "FDN_RDV")}
        env["FDN_ASSERT_NO_TORCH"] = "1"
        cmd = [sys.executable, os.path.join(ROOT, "flowdenoising.py"), "-i", src, "-o", dst, "-s", "1.0", "0.5", "1.0", "--compat", compat, "-l", "0"]
        if gpus > 1:
            cmd += ["--gpus", str(gpus)]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(open(dst, "rb").read())
    assert outs[0] == outs[1]


def _write_mrc_any(path, vol):
    """An MRC of the array's own mode (the product writer only writes mode 2)."""
    import struct
    mode = {np.dtype(np.int8): 0, np.dtype(np.int16): 1, np.dtype(np.float32): 2, np.dtype(np.uint16): 6}[vol.dtype]
    nz, ny, nx = vol.shape
    h = bytearray(1024)
    struct.pack_into("<4i", h, 0, nx, ny, nz, mode)
    struct.pack_into("<3i", h, 28, nx, ny, nz)
    struct.pack_into("<3i", h, 64, 1, 2, 3)
    h[208:212] = b"MAP "
    h[212:216] = bytes([0x44, 0x44, 0, 0])
    with open(path, "wb") as f:
        f.write(h)
        f.write(np.ascontiguousarray(vol).tobytes())


# ---- the C engine at world 7 and 8: rank THREADS in this process (the boxes allow six GPU processes) -----------------------
@pytest.mark.gpu_subprocess
@pytest.mark.parametrize("world,shape,sig,border,l,w,dtype", [
    (8, (16, 66, 140), "1.0,0.5,1.0", 0, 0, 5, np.float32),      # two slices per rank, mean-padded
    (8, (19, 70, 64), "1.5,1.5,-", 1, 0, 5, np.float32),         # wrap-around; K//2 = 6 exceeds every slab (2-3 slices): halos from three ranks away
    (7, (16, 66, 140), "1.0,1.0,0.5", 0, 1, 7, np.float32),      # a pyramid level on the Z pass's images (66 x 140), uneven slabs
    (8, (17, 40, 72), "1.0,0.5,1.0", 0, 0, 5, np.int16),         # seq on an integer volume: the float64 padded volume through every exchange
    (7, (15, 64, 66), "0.5,1.0,1.0", 1, 0, 15, np.int16),        # par on an integer volume, the one-iteration kernel
])
def test_native_engine_world_7_and_8_as_rank_threads(fdn, tmp_path, world, shape, sig, border, l, w, dtype):
    """fdn_filter_3d_sharded -- plan, packing, one exchange per pass, the exact mean, the passes -- at world = 7 and 8 on one
    GPU: every rank a thread with its own handle, stream and shared-memory transport (tests/_thread_ranks.py); the
    concatenated slabs equal the single-GPU OF_filter bit for bit.  (src/flowdenoising.py:181-206 is the decomposition this
    replaces; the driver's 8-GPU box runs the same engine over RCCL.)  The rank threads live in a fresh process
    (conftest.run_in_fresh_process); the single-GPU reference is computed here."""
    import json
    from conftest import run_in_fresh_process
    from flowdenoising_amd import _lib
    vol = _vol(shape, seed=41)
    if dtype is not np.float32:
        vol = np.round((vol - vol.min()) * (3000.0 / (vol.max() - vol.min())) - 700).astype(dtype)
    ks = [None if s == "-" else fdn.get_gaussian_kernel(float(s)) for s in sig.split(",")]
    want = fdn.OF_filter(vol, ks, l, w, border_mode=border)
    code = ("import json, _thread_ranks\n"
            "from flowdenoising_amd import _lib, operators\n"
            "ks = [None if s == '-' else _lib.gaussian_kernel(float(s)) for s in str(sig).split(',')]\n"
            "p = operators.integer_semantics(vol, operators._params(int(l), int(w), True, int(border), True))\n"
            "got, info = _thread_ranks.run(vol.astype(np.float32), ks, p, int(world))\n"
            "out['got'] = got\nprint('INFO ' + json.dumps(info))\n")
    res, stdout = run_in_fresh_process(code, dict(vol=vol, sig=sig, l=l, w=w, border=border, world=world), tmp_path)
    info = json.loads([ln for ln in stdout.splitlines() if ln.startswith("INFO ")][-1][5:])
    print(info)
    assert info["count"] == world and len(info["devices"]) == world and len(set(info["devices"])) == 1      # one GPU: a rehearsal, and it says so
    got = res["got"]
    assert got.dtype == np.float32 and np.array_equal(got, want.astype(np.float32))


def _bench_line(stdout):
    import json
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu_subprocess
def test_bench_line_proves_its_ranks_and_devices(fdn):
    """The N > 1 line says who was there as the communicator reports it (`n_ranks_seen` = ncclCommCount over RCCL, the
    distinct ranks heard from over shared memory) and which GPU every rank sat on (`devices`: all-gathered PCI bus ids):
    N distinct strings or the line calls itself a REHEARSAL.  cpu_baseline is null with a pointer to the N = 1 line; the
    roofline block stays."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--shape", "24,96,160", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _bench_line(r.stdout)
    assert d["n_ranks_seen"] == 3 and len(d["devices"]) == 3 and d["distinct_devices"] == len(set(d["devices"]))
    assert ("REHEARSAL" in d["config"]["parallelism"]) == (d["distinct_devices"] < 3)
    assert d["cpu_baseline"] is None and "N = 1" in d["cpu_baseline_see"] and d["roofline"]["frac"] > 0
    assert "transport_fallback" not in d and d["checked"]["ok"]


@pytest.mark.gpu_subprocess
@pytest.mark.parametrize("how", ["bare", "torchrun"])
def test_bench_falls_back_to_fresh_ranks_when_the_native_job_fails(fdn, how):
    """A native start-up failure must not leave an `rc != 0` record without a number.  From a bare shell the parent -- which
    never touched a GPU -- starts N fresh children once more on the torch.distributed engine; under torch.distributed.run
    every rank runs its native rank in one child of its own, and when that job fails anywhere all ranks (still fresh) carry
    on with the torch.distributed engine in themselves.  Either way the line says `transport_fallback` and why, and still
    checks its output.  (FDN_TEST_FAIL_NATIVE=1: native rank 1 raises after the transport is up -- rank 0 is then waiting
    in a collective and has to be told.)"""
    import socket
    pytest.importorskip("torch")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "FDN_RANK", "FDN_WORLD", "FDN_RDV")}
    env["FDN_TEST_FAIL_NATIVE"] = "1"
    args = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shape", "24,96,160", "--steps", "1", "--warmup", "0"]
    if how == "torchrun":
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)] + args
    else:
        cmd = [sys.executable] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _bench_line(r.stdout)
    assert "injected failure" in d["transport_fallback"] and d["backend"] in ("nccl", "gloo")
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["value"] > 0 and len(d["devices"]) == 2
    assert d["checked"]["ok"] and d["checked"]["sharded_output_equals_single_gpu_rerun"]
    # and without the fallback the failure is the exit code, promptly
    env["FDN_BENCH_NO_FALLBACK"] = "1"
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "injected failure" in r.stderr


@pytest.mark.gpu_subprocess
def test_cli_gpus_refuses_more_ranks_than_slices_and_a_failing_rank_stops_the_job(fdn, tmp_path):
    """`--gpus N` with an axis shorter than N is refused by the parent, before any rank starts; a rank that fails (rank 0
    cannot create the output file) tells the others, which leave their barrier at once instead of waiting for the timeout."""
    import time
    from flowdenoising_amd import io as fio
    v = _vol((3, 40, 72), seed=12)
    src = str(tmp_path / "in.mrc")
    fio.write_volume(src, v)
    env = {k: val for k, val in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "FDN_RANK", "FDN_WORLD", "FDN_RDV")}
    cli = [sys.executable, os.path.join(ROOT, "flowdenoising.py"), "-i", src, "-s", "1.0", "0.5", "1.0"]
    r = subprocess.run(cli + ["-o", str(tmp_path / "o.mrc"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "--gpus 4" in r.stderr
    t0 = time.perf_counter()
    r = subprocess.run(cli + ["-o", str(tmp_path / "no_such_dir" / "o.mrc"), "--gpus", "2"], env=dict(env, FDN_RDV_TIMEOUT="120"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and time.perf_counter() - t0 < 60, r.stderr[-2000:]


# ---- the targets of a batch as two sub-batches on two streams (fdn_set_option sub_batches; automatic on small grids) ----------
@pytest.mark.parametrize("case", [
    dict(shape=(9, 130, 300), sig=(1.0, None, 1.0), l=0, w=5, border=0),               # 3-iteration kernel; 9 and 300 targets: odd and even halves
    dict(shape=(8, 90, 200), sig=(1.0, 1.0, 0.5), l=2, w=5, border=1, chained=False),  # ... pyramid (each half its own part of the flow pyramid), wrap, --recompute_flow
    dict(shape=(9, 130, 300), sig=(1.0, None, 1.0), l=2, w=11, border=1),              # one-iteration kernel with a pyramid, wrapped ends
    dict(shape=(3, 70, 160), sig=(1.5, 1.0, None), l=1, w=15, border=0),               # three targets: halves of two and one
    dict(shape=(1, 70, 160), sig=(1.0, None, None), l=0, w=5, border=0),               # one target: nothing to split (last_sub_batches says 1)
    dict(shape=(10, 48, 56), sig=(1.0, 1.0, 0.5), l=0, w=5, border=0, dtype=np.int16),   # seq on an integer volume (float64 padded volume)
    dict(shape=(10, 48, 56), sig=(1.0, 0.5, 1.0), l=1, w=15, border=1, dtype=np.uint8),  # par on uint8: fixed-point remap
    dict(shape=(12, 64, 96), sig=(1.0, 1.0, 1.0), l=1, w=5, border=0, limit=6 << 20),    # a workspace limit: several batches per pass, each split
])
def test_sub_batches_on_two_streams_give_the_same_bits(fdn, case):
    """fdn_set_option("sub_batches", 2): the target slices of a batch run as two independent halves, each the complete chain
    of both sides on a stream of its own (the tails of small launches are filled by the other half).  Disjoint slices of the
    same buffers, the same kernels: the same bits as one stream, on both Farneback kernels, with pyramids, both border rules,
    integer volumes, and under a workspace limit."""
    from flowdenoising_amd.operators import handle
    vol = _vol(case["shape"], seed=91)
    dt = case.get("dtype")
    if dt is not None:
        top = 255 if dt is np.uint8 else 3000
        vol = np.round((vol - vol.min()) * (top / (vol.max() - vol.min())) - (0 if dt is np.uint8 else 700)).astype(dt)
    ks = [None if s is None else fdn.get_gaussian_kernel(s) for s in case["sig"]]
    h = handle()
    outs = {}
    try:
        if case.get("limit"):
            h.set_workspace_limit(case["limit"])
        for sb in (1, 2, 0):
            h.set_option("sub_batches", sb)
            assert h.get_option("sub_batches") == sb
            outs[sb] = fdn.OF_filter(vol, ks, case["l"], case["w"], border_mode=case["border"], chained=case.get("chained", True))
            ran = h.get_option("last_sub_batches")
            n_last = [n for n, s in zip(case["shape"], case["sig"]) if s is not None][-1]
            if sb == 1 or n_last < 2:
                assert ran == 1, (sb, ran)
            elif sb == 2 and not case.get("limit"):        # (under a limit the last batch of the last pass may hold one target)
                assert ran == 2, (sb, ran)
    finally:
        h.set_option("sub_batches", 0)
        if case.get("limit"):
            h.set_workspace_limit(0)
    assert np.array_equal(outs[1], outs[2]) and np.array_equal(outs[1], outs[0])


def test_handles_used_from_threads_in_this_process(fdn, oracle):
    """Multi-threaded use of the library INSIDE the long-lived test process (most multi-threaded GPU tests run in processes
    of their own, see conftest): four threads share the process-wide handle through the pair operators (the
    C ABI takes the handle's lock), three more own a handle each -- created in the thread, closed by the main thread, the
    out-of-core mode's pattern -- and run sweeps at the same time; then the out-of-core mode itself; then an ordinary
    single-threaded filter of a fresh 1.1 MB volume, the kind of call round 5's sessions aborted in (NOTES_r06.md section 2).
    Everything must equal the single-threaded results bit for bit."""
    import threading
    from flowdenoising_amd import _lib, streaming
    from flowdenoising_amd.operators import _params
    vol = _vol((12, 96, 160), seed=5)
    k = fdn.get_gaussian_kernel(1.0)
    r = k.size // 2
    params = _params(0, 5)
    S, H, W = vol.shape[0] - 2 * r, vol.shape[1], vol.shape[2]
    want_flow = [fdn.get_flow(vol[i + 1], vol[i], 0, 5, np.zeros((H, W, 2), np.float32)) for i in range(4)]
    want_warp = [fdn.warp_slice(vol[i + 1], want_flow[i]) for i in range(4)]
    want_sweep = fdn.OF_filter_along_Z(vol, k, 0, 5, vol.mean())
    got, made, errors = {}, [], []

    def pair_user(i):
        try:
            for _ in range(5):
                f = fdn.get_flow(vol[i + 1], vol[i], 0, 5, np.zeros((H, W, 2), np.float32))
                got[("flow", i)], got[("warp", i)] = f, fdn.warp_slice(vol[i + 1], f)
        except BaseException as e:      # noqa: BLE001
            errors.append(e)

    def handle_owner(i):
        try:
            h = _lib.Handle(0)
            made.append(h)
            d_stack, d_out = h.malloc(vol.nbytes), h.malloc(S * H * W * 4)
            for _ in range(3):
                h.h2d(d_stack, vol)
                h.sweep_stack_dev(d_stack, d_out, S, H, W, k, params)
                out = np.empty((S, H, W), np.float32)
                h.d2h(out, d_out)
            h.free(d_stack)
            h.free(d_out)
            got[("sweep", i)] = out
        except BaseException as e:      # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=pair_user, args=(i,)) for i in range(4)] + [threading.Thread(target=handle_owner, args=(i,)) for i in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for h in made:
        h.close()
    assert not errors, errors
    for i in range(4):
        assert np.array_equal(got[("flow", i)], want_flow[i]) and np.array_equal(got[("warp", i)], want_warp[i])
    for i in range(3):
        assert np.array_equal(got[("sweep", i)], want_sweep[r:r + S])
    streamed = streaming.filter_streamed(vol, [k, k, None], 0, 5, chunk_slices=3)
    again = streaming.filter_streamed(vol, [k, k, None], 0, 5, chunk_slices=5, workers=2)      # the kept workers, another chunking
    streaming.release_workers()
    assert np.array_equal(streamed, fdn.OF_filter(vol, [k, k, None], 0, 5)) and np.array_equal(again, streamed)
    fresh = _vol((7, 130, 300), seed=33)
    out = fdn.OF_filter(fresh, [k, None, k], 0, 5)
    assert np.array_equal(out, oracle.OF_filter(fresh, [k, None, k], 0, 5, nthreads=8))


# ---- the two cv2 unknowns as product options: HIP and oracle agree bit for bit in every setting -----------------
@pytest.mark.parametrize("mode,lanes", [(1, 8), (2, 8), (2, 4), (2, 16)])
@pytest.mark.parametrize("w,path", [(5, 0), (15, 0), (7, 1)])
def test_opencv_fma_modes_match_the_oracle(fdn, oracle, mode, lanes, w, path):
    """fdn_set_option("opencv_fma", mode) / fdo_set_fma(mode): every multiply-add of the pyramid's blur and vertical resize
    fused (1) or fused on the vector body of a row only (2, with the lane count) -- on the 3-iteration kernel (w = 5), the
    one-iteration kernel (w = 15) and the per-stage kernels (path 1), Z and X passes of a volume whose pyramid rows are not
    multiples of the lane counts.  The result differs from mode 0 and equals the oracle in the same mode."""
    from flowdenoising_amd.operators import handle
    vol = _vol((6, 70, 150), seed=17)
    ks = [fdn.get_gaussian_kernel(1.0), None, fdn.get_gaussian_kernel(0.5)]
    h = handle()
    try:
        h.set_option("path", path)
        plain = fdn.OF_filter(vol, ks, 2, w)
        h.set_option("opencv_fma", mode)
        h.set_option("opencv_fma_lanes", lanes)
        assert (h.get_option("opencv_fma"), h.get_option("opencv_fma_lanes")) == (mode, lanes)
        got = fdn.OF_filter(vol, ks, 2, w)
        level0 = fdn.OF_filter(vol, ks, 0, w)
    finally:
        h.set_option("opencv_fma", 0)
        h.set_option("opencv_fma_lanes", 8)
        h.set_option("path", 0)
    try:
        oracle.set_fma(mode, lanes)
        want = oracle.OF_filter(vol, ks, 2, w, nthreads=8)
    finally:
        oracle.set_fma(0)
    assert np.array_equal(got, want), rel_err(got, want)
    assert not np.array_equal(got, plain) and rel_err(got, plain) < 5e-2
    assert np.array_equal(level0, fdn.OF_filter(vol, ks, 0, w))                       # no pyramid: nothing to fuse


@pytest.mark.parametrize("case", [
    dict(l=0, w=5, border=0), dict(l=2, w=5, border=1), dict(l=1, w=15, border=0), dict(l=0, w=5, border=0, path=1),
    dict(l=0, w=5, border=0, dtype=np.int16), dict(l=1, w=5, border=1, dtype=np.int16), dict(l=0, w=5, border=1, dtype=np.uint8),
])
def test_remap_model_matches_the_oracle(fdn, oracle, case):
    """fdn_set_option("remap_model", 1) / fdo_set_remap_model(1): every warp of a sweep as unquantised float32 bilinear
    interpolation instead of cv2's classic 1/32-pixel table -- in the Farneback kernels' final stage, in k_sweep_side (path 1)
    and in the integer-volume semantics (float64 padded volume: doubles; 16-bit images: rounded; 8-bit: fixed point as before)."""
    from flowdenoising_amd.operators import handle
    vol = _vol((8, 48, 76), seed=23)
    dt = case.get("dtype")
    if dt is not None:
        top = 255 if dt is np.uint8 else 3000
        vol = np.round((vol - vol.min()) * (top / (vol.max() - vol.min())) - (0 if dt is np.uint8 else 700)).astype(dt)
    ks = [fdn.get_gaussian_kernel(1.0), fdn.get_gaussian_kernel(0.5), fdn.get_gaussian_kernel(1.0)]
    h = handle()
    try:
        h.set_option("path", case.get("path", 0))
        classic = fdn.OF_filter(vol, ks, case["l"], case["w"], border_mode=case["border"])
        h.set_option("remap_model", 1)
        got = fdn.OF_filter(vol, ks, case["l"], case["w"], border_mode=case["border"])
    finally:
        h.set_option("remap_model", 0)
        h.set_option("path", 0)
    try:
        oracle.set_remap_model(1)
        if dt is None:
            want = oracle.OF_filter(vol, ks, case["l"], case["w"], border_mode=case["border"], nthreads=8)
        elif case["border"] == 0:
            want = oracle.OF_filter_integer_input(vol, ks, case["l"], case["w"], nthreads=8)
        else:
            want = oracle.filter_par_integer_input(vol, ks, case["l"], case["w"], nthreads=8)
    finally:
        oracle.set_remap_model(0)
    assert np.array_equal(got, np.asarray(want, dtype=got.dtype)), rel_err(got, want)
    if dt is not np.uint8:
        assert not np.array_equal(got, classic)
    else:
        assert np.array_equal(got, classic)                          # 8-bit images: the fixed-point table in either model


def test_remap_model_on_the_pair_operators(fdn, oracle):
    from flowdenoising_amd.operators import handle
    rng = np.random.default_rng(9)
    ref = (rng.standard_normal((40, 44)) * 100).astype(np.float32)
    flow = (rng.standard_normal((40, 44, 2)) * 1.3).astype(np.float32)
    m = np.stack([(flow[..., 0].astype(np.float64) + np.arange(44)[None, :]).astype(np.float32),
                  (flow[..., 1].astype(np.float64) + np.arange(40)[:, None]).astype(np.float32)], axis=-1)
    h = handle()
    try:
        h.set_option("remap_model", 1)
        oracle.set_remap_model(1)
        for img in (ref, ref.astype(np.float64), np.round(ref).astype(np.int16), np.clip(np.round(ref + 128), 0, 255).astype(np.uint8)):
            got = fdn.warp_slice(img, flow)
            assert got.dtype == img.dtype and np.array_equal(got, oracle.remap_any(img, m)), img.dtype
    finally:
        h.set_option("remap_model", 0)
        oracle.set_remap_model(0)
