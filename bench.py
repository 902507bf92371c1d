#!/usr/bin/env python3
"""bench.py -- Mvoxels/s of the full flow-driven denoise (Z, Y and X passes, sigma=2, levels=0,
winsize=5) on a synthetic 1024x1024x512 float32 volume (BASELINE.json configs[2]).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one complete OF_filter of the volume (mean + three axis sweeps), input and output
resident in HBM.  PyTorch is used for device memory, the synthetic generator and (N > 1)
torch.distributed/RCCL; all arithmetic of the step runs in libflowdn.so.

Rank 0 prints ONE JSON line (contract in the task statement) including
  "roofline":     dominant kernel, algorithmic bytes / HIP-event time vs the 8 TB/s HBM peak
  "cpu_baseline": the CPU oracle (oracle/, a port of the reference's arithmetic parallelised
                  over target slices like src/flowdenoising.py:181-206) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--shape", default="512,1024,1024", help="Z,Y,X")
    ap.add_argument("--sigma", type=float, default=2.0)
    ap.add_argument("--axes", default="zyx", help="subset of zyx (configs[1] is 'z')")
    ap.add_argument("--amplitude", type=float, default=100.0)
    ap.add_argument("--levels", type=int, default=0, help="pyramid levels (-l); configs[4] uses 3")
    ap.add_argument("--winsize", type=int, default=5, help="Farneback window (-w); configs[4] uses 15")
    ap.add_argument("--cpu-targets", type=int, default=0, help="target slices of the CPU sample (0 = four per core)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-timers", action="store_true", help="skip per-kernel HIP-event timing")
    return ap.parse_args()


def algorithmic_bytes(timers, shape, K, axes, levels=0):
    """SURVEY.md 8(d) per-unit figures x the units each launch processed.

    staged path : FarnebackUpdateFlow_Blur launch = M read 20 B + flow write 8 B per pixel, plus,
                  on the first iters-1 launches of a chain step, the matrix refresh 68 B
                  -> (3*28 + 2*68)/3 = 73.33 B per pixel per launch on average.
    fused path  : one launch = one whole Farneback pair + warp/accumulate = 288 + 12 B per pixel.
    Every launch covers all target slices of the pass: Z*Y*X pixels."""
    nvox = shape[0] * shape[1] * shape[2]
    per_px = {"update_flow": (3 * 28 + 2 * 68) / 3.0, "fused": 300.0}
    best = None
    for name in ("fused", "update_flow"):
        ms, cnt = timers.get(name, (0.0, 0))
        if cnt > 0 and (best is None or ms > best[1]):
            best = (name, ms, cnt)
    if best is None:
        return None
    name, ms, cnt = best
    # with a pyramid a chain step is one launch per level; level k has 4^-k of the pixels
    bytes_per_launch = per_px[name] * nvox * sum(0.25 ** k for k in range(levels + 1)) / (levels + 1)
    avg_ms = ms / cnt
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
    traffic, traffic_note = measured_traffic(name, nvox)
    extra = {}
    if traffic:   # what the DRAM counters saw, and what actually limits the kernel (committed PMC pass)
        extra["hbm_measured"] = {"GBps": round(traffic / (avg_ms * 1e-3) / 1e9, 1),
                                 "frac": round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        try:
            with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
                v = json.load(f).get("valu")
            if v:
                extra["limiter"] = {"unit": "VALU issue", "utilisation": v["valu_issue_utilisation"], "source": v["source"]}
        except OSError:
            pass
    return {**_roofline_core(name, achieved, traffic, traffic_note, bytes_per_launch, avg_ms, cnt), **extra}


def _roofline_core(name, achieved, traffic, traffic_note, bytes_per_launch, avg_ms, cnt):
    return {"bound": "hbm", "kernel": {"fused": "k_farneback_fused", "update_flow": "k_update_flow_scan"}[name],
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_note": traffic_note,
            "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": round(avg_ms, 4),
            "launches": cnt}


def measured_traffic(name, nvox):
    """HBM bytes per launch of the dominant kernel from the committed PMC pass (profiles/), scaled to
    this run's pixels per launch; None when no PMC data exists for the kernel."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if name != "fused" or not os.path.exists(path):
        return None, "no PMC pass for this kernel"
    with open(path) as f:
        t = json.load(f)
    scale = nvox / t["pixels_per_launch"]
    return round(t["bytes_per_launch"] * scale), (
        f"(2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch, separate rocprofv3 --pmc passes ({os.path.basename(path)}): "
        f"{t['bytes_per_pixel']:.0f} B per pixel per launch against 300 algorithmic (M and the per-iteration flows never "
        "leave the chip) and 68 compulsory")


def cpu_baseline(vol_t, shape, kernel, mean, n_targets):
    """Time the oracle's Z pass on `n_targets` target slices taken from the middle of the same
    volume, one chunk of slices per core (par:181-206), and scale to the full three-pass job."""
    from oracle import oracle as O
    O.build()
    # the GPU box gives one GPU a share of 16 host cores; FDN_BENCH_CORES overrides
    cap = int(os.environ.get("FDN_BENCH_CORES", "16"))
    cores = max(1, min(len(os.sched_getaffinity(0)), O.max_threads(), cap))
    if n_targets <= 0:
        n_targets = 4 * cores      # about 10 s of wall time on the GPU box's 16 cores
    Z, Y, X = shape
    r = kernel.size // 2
    n_targets = min(n_targets, Z)
    z0 = max(0, Z // 2 - n_targets // 2 - r)
    z1 = min(Z, z0 + n_targets + 2 * r)
    slab = vol_t[z0:z1].cpu().numpy()
    s0 = min(r, slab.shape[0] - n_targets)
    t0 = time.perf_counter()
    O.filter_axis_range(slab, 0, kernel, 0, 5, mean, s0, s0 + n_targets, nthreads=cores)
    dt = time.perf_counter() - t0
    vox = n_targets * Y * X
    per_axis = vox / dt / 1e6
    return {"value": round(per_axis / 3.0, 4), "unit": "Mvoxels/s", "cores": cores, "kind": "port",
            "sample": f"oracle Z pass on {n_targets} target slices ({Y}x{X}, 16 Farneback pairs each) of the same "
                      f"volume in {dt:.1f} s with {cores} threads; per-axis rate {per_axis:.3f} Mvox/s divided by 3 "
                      "for the Z+Y+X job (same pixel-pair count per axis)"}


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from flowdenoising_amd import _lib, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    shape = tuple(int(v) for v in a.shape.split(","))
    Z, Y, X = shape
    kernel = _lib.gaussian_kernel(a.sigma)
    kernels = [kernel if c in a.axes else None for c in "zyx"]
    naxes = sum(k is not None for k in kernels)
    params = _lib.SweepParams(a.levels, a.winsize, 3, 5, 1.2, _lib.BORDER_MEAN_PAD, 1, 1)

    h = _lib.Handle(local_rank)
    h.set_stream(torch.cuda.current_stream().cuda_stream)

    if world == 1:
        vol = synth.make_volume(shape, seed=1234 + 3, amplitude=a.amplitude, xp=torch, device=dev)
        out = torch.empty_like(vol)

        def step():
            mean = h.mean_dev(vol.data_ptr(), vol.numel())       # seq:420
            h.filter_3d_dev(vol.data_ptr(), out.data_ptr(), shape, kernels, mean, params)
            return mean
        parallelism = "1 GPU"
    else:
        from flowdenoising_amd import distributed as fd
        plan = fd.SlabPlan(shape, world, rank)
        vol = synth.make_volume(shape, seed=1234 + 3, amplitude=a.amplitude, xp=torch, device=dev,
                                z0=plan.z0, zlen=plan.zlen)
        eng = fd.SlabEngine(plan, h, dist)

        def step():
            return eng.filter_3d(vol, kernels, params)
        out = None
        parallelism = f"{world} Z-slabs, halo exchange + all-to-all repartition per pass (RCCL)"

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if not a.no_timers:
        h.enable_timers(True)
        h.timers(reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mean = None
    for _ in range(a.steps):
        mean = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    timers = h.timers() if not a.no_timers else {}
    h.enable_timers(False)

    if rank == 0:
        nvox = Z * Y * X
        ms_per_step = dt / a.steps * 1e3
        value = nvox / (dt / a.steps) / 1e6
        res = {
            "metric": "Mvoxels/s denoised (sigma=2) on 1024x1024x512 float32",
            "value": round(value, 3), "unit": "Mvoxels/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{X}x{Y}x{Z} float32, sigma={a.sigma:g} (K={kernel.size}), levels={a.levels}, winsize={a.winsize}, "
                                   f"OF along {a.axes.upper()}, mean-padded borders (BASELINE.json configs[2])",
                       "axes": a.axes, "parallelism": parallelism, "amplitude": a.amplitude},
        }
        # whole path against the SURVEY 8(d) algorithmic traffic (4832 B/voxel/axis at sigma=2)
        per_axis_bytes = 24 + (kernel.size - 1) * 300 + 8
        res["whole_path"] = {"algorithmic_GBps": round(per_axis_bytes * naxes * nvox / (dt / a.steps) / 1e9 / world, 1),
                             "frac_of_hbm_peak_per_gpu": round(per_axis_bytes * naxes * nvox / (dt / a.steps) / 1e9 / world / HBM_PEAK_GBS, 4)}
        if timers:
            res["roofline"] = algorithmic_bytes(timers, shape if world == 1 else (Z // world, Y, X), kernel.size, a.axes, a.levels)
            res["kernel_ms_per_step"] = {k: round(v[0] / a.steps, 2) for k, v in timers.items() if v[1]}
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(vol, shape, kernel, mean, a.cpu_targets)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
