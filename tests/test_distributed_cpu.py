"""The N > 1 decomposition (flowdenoising_amd/distributed.py) on CPU: world_size 2 and 3 with the
gloo backend, the pass compute injected from the oracle.  What is under test is the host logic:
slab plan, halo exchange (mean-pad and wrap), the Z->Y->X->Z repartition, the global mean.
The sharded result must equal the single-process oracle bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleBackend:
    """Pass compute for the CPU tests (the product uses HipBackend)."""

    def __init__(self, O):
        self.O = O

    def sweep_stack(self, stack, out, S, H, W, kernel, params):
        r = kernel.size // 2
        res = self.O.filter_axis_range(stack.numpy(), 0, kernel, params.levels, params.winsize, 0.0, r, r + S,
                                       use_of=bool(params.use_of), chained=bool(params.chained))
        out.copy_(__import__("torch").from_numpy(res[r:r + S]))

    def pack(self, src_view, dst):
        dst.copy_(src_view)

    def chunk_sums(self, t):
        a = t.numpy().reshape(-1)      # numpy's own pairwise sum of each 8192-element chunk
        return np.array([a[s:s + 8192].sum(dtype=np.float32) for s in range(0, a.size, 8192)], dtype=np.float32)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, shape, sigmas, border_mode, use_of, q, loopback=False):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from flowdenoising_amd import _lib
    from flowdenoising_amd.distributed import SlabEngine, SlabPlan
    from flowdenoising_amd.synth import make_volume
    from oracle import oracle as O
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        vol = make_volume(shape, seed=21, amplitude=100.0)
        plan = SlabPlan(shape, world, rank)
        slab = torch.from_numpy(vol[plan.z0:plan.z0 + plan.zlen].copy())
        kernels = [None if s is None else O.get_gaussian_kernel(s) for s in sigmas]
        params = _lib.SweepParams(0, 5, 3, 5, 1.2, border_mode, 1, int(use_of))
        eng = SlabEngine(plan, OracleBackend(O), dist, loopback=loopback)
        mean_auto = eng.global_mean(slab)
        first = eng.filter_3d(slab, kernels, params, mean=vol.mean()).clone()
        out = eng.filter_3d(slab, kernels, params, mean=vol.mean())      # second step: the persistent buffers are reused
        assert torch.equal(first, out)
        assert set(eng.phase_times()) == set(eng.PHASES) and eng.phase_times()["compute"] > 0
        full = eng.gather_z_slabs(out, 0)
        if rank == 0:
            q.put((full.numpy().copy(), float(mean_auto)))
    finally:
        dist.destroy_process_group()


def _run(world, shape, sigmas, border_mode=0, use_of=True, loopback=False):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, sigmas, border_mode, use_of, q, loopback)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("world,shape", [(2, (12, 34, 36)), (3, (11, 34, 37)), (4, (10, 33, 35)), (8, (16, 40, 42))])
def test_sharded_of_filter_equals_single_process(oracle, world, shape):
    from flowdenoising_amd.synth import make_volume
    sig = (1.0, 0.5, 1.0)
    got, mean_auto = _run(world, shape, sig)
    vol = make_volume(shape, seed=21, amplitude=100.0)
    want = oracle.OF_filter(vol, [oracle.get_gaussian_kernel(s) for s in sig], 0, 5)
    assert np.array_equal(got, want)
    assert np.float32(mean_auto) == vol.mean()


@pytest.mark.parametrize("world,shape", [(3, (7, 64, 128)), (3, (9, 70, 90)), (2, (5, 100, 131)), (4, (9, 20, 22))])
def test_sharded_mean_is_numpys_float32_mean(oracle, world, shape):
    """seq:420 from Z-slabs: numpy's float32 mean of the whole volume, bit for bit -- slabs that start at chunk
    boundaries (Y*X a multiple of 8192, as for 1024 x 1024 slices), slabs that do not (the straddling chunk is
    completed with the next rank's leading elements) and volumes so small that a chunk spans several slabs."""
    from flowdenoising_amd.synth import make_volume
    _, mean_auto = _run(world, shape, (0.5, None, None))
    vol = make_volume(shape, seed=21, amplitude=100.0)
    assert np.float32(mean_auto) == vol.mean()


def test_sharded_wrap_borders_and_axis_subset(oracle):
    from flowdenoising_amd.synth import make_volume
    shape, sig = (8, 34, 36), (1.0, None, 0.5)
    got, _ = _run(2, shape, sig, border_mode=1)
    vol = make_volume(shape, seed=21, amplitude=100.0)
    ks = [None if s is None else oracle.get_gaussian_kernel(s) for s in sig]
    want = oracle.OF_filter(vol, ks, 0, 5, border_mode=1)
    assert np.array_equal(got, want)


def test_sharded_no_of(oracle):
    from flowdenoising_amd.synth import make_volume
    shape, sig = (9, 20, 22), (1.5, 1.0, 1.0)   # halo (6) larger than a slab (4-5 slices)
    got, _ = _run(2, shape, sig, use_of=False)
    vol = make_volume(shape, seed=21, amplitude=100.0)
    want = oracle.no_OF_filter(vol, [oracle.get_gaussian_kernel(s) for s in sig])
    assert np.array_equal(got, want)


@pytest.mark.parametrize("world,shape,border", [(1, (6, 34, 36), 0), (1, (3, 64, 128), 1), (2, (7, 34, 36), 0)])
def test_loopback_routes_own_blocks_through_the_transport(oracle, world, shape, border):
    """SlabEngine(loopback=True): the blocks a rank keeps are sent to itself inside the batched group and the mean
    goes through the all_gather even with ONE rank -- the mode the GPU suite uses to put real RCCL calls under a
    world-size-1 run (tests/test_gpu_full.py::test_rccl_world_size_1_carries_the_slab_engine).  Same bits."""
    from flowdenoising_amd.synth import make_volume
    sig = (1.0, 0.5, 1.0)
    got, mean_auto = _run(world, shape, sig, border_mode=border, loopback=True)
    vol = make_volume(shape, seed=21, amplitude=100.0)
    want = oracle.OF_filter(vol, [oracle.get_gaussian_kernel(s) for s in sig], 0, 5, border_mode=border)
    assert np.array_equal(got, want)
    assert np.float32(mean_auto) == vol.mean()


def test_exchange_schedule_is_consistent():
    """Sender and receiver derive the same block list; the blocks a rank receives tile its halo-extended stack
    exactly once (positions outside a mean-padded volume excepted)."""
    from flowdenoising_amd.distributed import ORIENT, SlabPlan
    for world, shape, r, wrap in [(3, (7, 9, 11), 2, False), (2, (5, 6, 7), 3, True), (4, (9, 8, 10), 1, True), (3, (6, 7, 8), 0, False),
                                  (8, (16, 24, 17), 4, False), (8, (9, 8, 16), 8, True)]:      # the scaling run's rank count; halos spanning several slabs
        for A in range(3):
            for B in range(3):
                for j in range(world):
                    plan = SlabPlan(shape, world, j)
                    s, e = plan.parts[B][j]
                    dims = [shape[a] for a in ORIENT[B]]
                    cover = np.zeros((e - s + 2 * r, dims[1], dims[2]), int)
                    for i in range(world):
                        for p0, rng in plan.blocks(A, B, r, wrap, i, j):
                            n = rng[B][1] - rng[B][0]
                            cover[p0:p0 + n, rng[ORIENT[B][1]][0]:rng[ORIENT[B][1]][1], rng[ORIENT[B][2]][0]:rng[ORIENT[B][2]][1]] += 1
                    inside = np.array([wrap or 0 <= s - r + p < shape[B] for p in range(e - s + 2 * r)])
                    assert (cover[inside] == 1).all() and (cover[~inside] == 0).all(), (world, shape, r, wrap, A, B, j)


def test_halo_runs_cover_every_halo_slice():
    from flowdenoising_amd.distributed import SlabPlan
    for world, n, r, wrap in [(2, 10, 3, False), (3, 7, 4, True), (4, 9, 2, True), (1, 5, 3, True)]:
        plan = SlabPlan((n, n, n), world, 0)
        runs = plan.halo_runs(0, r, wrap)
        for d, (s, e) in enumerate(plan.parts[0]):
            got = {}
            for src, dst, s_loc, d_pos, cnt in runs:
                if dst == d:
                    for i in range(cnt):
                        got[d_pos + i] = plan.parts[0][src][0] + s_loc + i
            for p in list(range(r)) + list(range(r + e - s, 2 * r + e - s)):
                g = s - r + p
                if wrap:
                    assert got[p] == g % n
                elif 0 <= g < n:
                    assert got[p] == g
                else:
                    assert p not in got
