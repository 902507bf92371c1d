#!/usr/bin/env python3
"""Per-rank overhead of the sharded filter, measured on ONE GPU: rank r of an N-rank plan runs
fdn_filter_3d_sharded on the NULL transport (include/flowdn_rccl.h: nothing travels, receive buffers are zero-filled), so
every buffer, block list, pack / unpack launch and the mean's chunk sums have the shapes of the real N-GPU run; the library's
timer table (HIP events on the handle's stream) gives milliseconds per step for packing + unpacking (FDN_TIMER_PERMUTE), the
mean (FDN_TIMER_MEAN) and the zero-fill standing in for the exchange (FDN_TIMER_COLLECTIVE).  Compute times are NOT
representative (the passes run on zero-filled halos) and are reported only for scale.

usage: rank_emulation.py [--shape Z,Y,X] [--sigma S] [--steps K] [--out file.json]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="512,1024,1024")
    ap.add_argument("--sigma", type=float, default=2.0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--worlds", default="2,4,8")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r04_rank_emulation.json"))
    a = ap.parse_args()
    from flowdenoising_amd import _lib, synth
    from flowdenoising_amd.distributed import split
    shape = tuple(int(v) for v in a.shape.split(","))
    Z, Y, X = shape
    ks = [_lib.gaussian_kernel(a.sigma)] * 3
    params = _lib.SweepParams(0, 5, 3, 5, 1.2, _lib.BORDER_MEAN_PAD, 1, 1)
    h = _lib.Handle(0)
    rows = []
    for world in [int(v) for v in a.worlds.split(",")]:
        parts = split(Z, world)
        for rank in sorted({0, world // 2, world - 1}):
            z0, z1 = parts[rank]
            slab = synth.make_volume(shape, seed=1237, z0=z0, zlen=z1 - z0)
            d_in, d_out = h.malloc(slab.nbytes), h.malloc(slab.nbytes)
            h.h2d(d_in, slab)
            tr = _lib.Transport("null", rank, world, 0)
            h.filter_3d_sharded(d_in, d_out, shape, ks, params, tr)          # warm-up: allocations
            h.synchronize()
            h.enable_timers(True)
            h.timers(reset=True)
            for _ in range(a.steps):
                h.filter_3d_sharded(d_in, d_out, shape, ks, params, tr)
            h.synchronize()
            t = h.timers(reset=True)
            h.enable_timers(False)
            per = {k: round(v[0] / a.steps, 3) for k, v in t.items() if v[1]}
            row = {"world": world, "rank": rank, "slab_slices": z1 - z0,
                   "pack_unpack_ms_per_step": per.get("permute", 0.0), "pack_unpack_launches_per_step": t["permute"][1] // a.steps,
                   "mean_ms_per_step": per.get("mean", 0.0), "zero_fill_in_place_of_exchange_ms_per_step": per.get("collective", 0.0),
                   "overhead_ms_per_step": round(per.get("permute", 0.0) + per.get("mean", 0.0), 3),
                   "compute_ms_per_step_not_representative": round(sum(per.get(k, 0.0) for k in ("fused", "iter", "polyexp", "warp", "update_flow", "update_matrices")), 2)}
            rows.append(row)
            print(json.dumps(row), flush=True)
            tr.close()
            h.free(d_in)
            h.free(d_out)
    res = {"what": "per-rank overhead of fdn_filter_3d_sharded outside its pass kernels, rank r of an N-rank plan emulated on one GPU with the NULL transport",
           "shape": list(shape), "sigma": a.sigma, "steps": a.steps, "rows": rows,
           "note": "overhead = pack + unpack (fdn_permute kernels, four exchanges per step) + the global mean; the exchange itself is not measured here"}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
