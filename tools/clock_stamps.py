#!/usr/bin/env python3
"""In-kernel clock of k_farneback_fused: runs the bench workload's Z pass back to back for a few seconds
on a DIAGNOSTIC build of the library (tools/build_variant.sh clock "-DFDN_CLOCK_STAMPS -DFDN_ONLY_MH2": every workgroup's stage-A
wave stamps s_memtime and s_memrealtime around its row loop into a buffer nothing else reads), then reads the stamps of the
last launch: clock = delta s_memtime / delta s_memrealtime x 100 MHz, median over workgroups (MI355X_MICROARCH.md, DVFS
give-back item 6).  Also prints cycles per row step of a workgroup.

usage: clock_stamps.py build_variants/lib_clock.so [seconds]"""
import ctypes
import json
import os
import shutil
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    lib_path = os.path.abspath(sys.argv[1])
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    from flowdenoising_amd import _lib, synth
    _lib.LIB_PATH = lib_path                       # the diagnostic build instead of the product library
    lib = _lib.load()
    h = _lib.Handle(0)
    shape = (512, 1024, 1024)
    vol = synth.make_volume(shape, seed=1237, amplitude=100.0)
    k = _lib.gaussian_kernel(2.0)
    params = _lib.SweepParams(0, 5, 3, 5, 1.2, _lib.BORDER_MEAN_PAD, 1, 1)
    d_in, d_out = h.malloc(vol.nbytes), h.malloc(vol.nbytes)
    h.h2d(d_in, vol)
    mean = h.mean_dev(d_in, vol.size)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < secs:         # >= 2 s of back-to-back launches on random data before the stamps are read
        h.filter_axis_dev(d_in, d_out, shape, 0, k, mean, params)
        h.synchronize()
        n += 1
    buf = (ctypes.c_ulonglong * (2 * 16384))()
    got = lib.fdn_debug_clock_stamps(buf, 2 * 16384)
    assert got > 0, "not a -DFDN_CLOCK_STAMPS build"
    a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 2).astype(np.float64)
    a = a[(a[:, 0] > 0) & (a[:, 1] > 0)]
    ghz = a[:, 0] / a[:, 1] * 0.1
    rows = 1024 + 9
    res = {"what": "in-kernel clock of k_farneback_fused, stage-A wave of every workgroup of the last launch (10 240 workgroups; stamps of 16 384 slots)",
           "workgroups_stamped": int(a.shape[0]), "passes_run": n, "seconds": round(time.perf_counter() - t0, 2),
           "clock_ghz_median": round(float(np.median(ghz)), 4), "clock_ghz_p5_p95": [round(float(np.percentile(ghz, 5)), 4), round(float(np.percentile(ghz, 95)), 4)],
           "shader_cycles_per_workgroup_median": float(np.median(a[:, 0])), "shader_cycles_per_row_step_median": round(float(np.median(a[:, 0])) / rows, 1),
           "workgroup_lifetime_us_median": round(float(np.median(a[:, 1])) / 100.0, 1)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
