#!/usr/bin/env python3
"""End-to-end wall time of the drop-in CLI (flowdenoising.py), phase by phase: the product is the
command line, not fdn_filter_3d_dev.  Writes a synthetic input file, runs `python flowdenoising.py -i ... -o ...` as a
child process with FDN_CLI_TIMING set, and prints one JSON record: file read / H2D / compute / D2H / file write seconds
(from the CLI's own clock), the process's total wall time (interpreter start, imports and library load included) and
the same volume's device-resident time for comparison.

usage: cli_wall.py [config2|config4] [out.json] [gpus]      (gpus > 1: `--gpus N`; on a one-GPU box the ranks share it)"""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "config2"
    out_json = sys.argv[2] if len(sys.argv) > 2 else None
    gpus = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    import torch
    from flowdenoising_amd import io as fio, synth
    dev = torch.device("cuda", 0)
    # files in memory-backed /dev/shm when there is room: the boxes' scratch disks differ by 5x in read speed, and what is
    # measured here is the program, not the disk (FDN_WALL_DIR overrides)
    base = os.environ.get("FDN_WALL_DIR") or ("/dev/shm" if os.path.isdir("/dev/shm") and os.statvfs("/dev/shm").f_bavail * os.statvfs("/dev/shm").f_frsize > (20 << 30) else os.environ.get("TMPDIR", "/tmp"))
    with tempfile.TemporaryDirectory(dir=base) as td:
        if which == "config2":        # BASELINE configs[2]: 1024 x 1024 x 512 float32 MRC, sigma 2, defaults
            shape, args = (512, 1024, 1024), ["-s", "2", "2", "2"]
            vol = synth.make_volume(shape, seed=1237, amplitude=100.0, xp=torch, device=dev).cpu().numpy()
            src, dst = os.path.join(td, "in.mrc"), os.path.join(td, "out.mrc")
            fio.write_mrc(src, vol)
        else:                         # BASELINE configs[4]: 2048 x 2048 x 512 uint16 TIFF stack, sigma 2 2 4, -l 3 -w 15
            shape, args = (512, 2048, 2048), ["-s", "2", "2", "4", "-l", "3", "-w", "15"]
            v = synth.make_volume(shape, seed=1239, amplitude=100.0, xp=torch, device=dev)
            lo, hi = float(v.min()), float(v.max())
            vol = torch.round((v - lo) / (hi - lo) * 4095).to(torch.int32).cpu().numpy().astype(np.uint16)
            del v
            src, dst = os.path.join(td, "in.tif"), os.path.join(td, "out.tif")
            fio.write_tiff(src, vol)
        nvox = int(np.prod(shape))
        del vol
        torch.cuda.empty_cache()
        if gpus > 1:
            args = args + ["--gpus", str(gpus)]
        rec = {"workload": which, "shape": list(shape), "cli_args": args, "input_bytes": os.path.getsize(src), "file_dir": base,
               "gpus_visible": torch.cuda.device_count()}
        runs = []
        for it in range(2):           # the second run has the input in the page cache and the library's code object cached
            tj = os.path.join(td, "t.json")
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable, os.path.join(ROOT, "flowdenoising.py"), "-i", src, "-o", dst] + args,
                               env={**os.environ, "FDN_CLI_TIMING": tj}, capture_output=True, text=True)
            total = time.perf_counter() - t0
            if r.returncode:
                sys.exit(r.stderr[-2000:])
            with open(tj) as f:
                ph = json.load(f)
            ph = {k: round(v, 3) for k, v in ph.items()}
            ph["process_total"] = round(total, 3)
            ph["startup_and_other"] = round(total - ph["read"] - ph["filter"] - ph["write"], 3)
            ph["Mvoxels_per_s_end_to_end"] = round(nvox / total / 1e6, 1)
            runs.append(ph)
            print(json.dumps(ph), flush=True)
        rec["runs"] = runs
        rec["output_bytes"] = os.path.getsize(dst)
        best = runs[-1]
        pcie_plus_compute = best["h2d"] + best["compute"] + best.get("d2h", best.get("d2h_write", 0.0))      # (--gpus N: rank 0's phases; h2d includes its slab read, d2h its file write)
        rec["pcie_plus_compute_s"] = round(pcie_plus_compute, 3)
        rec["wall_over_pcie_plus_compute"] = round(best["process_total"] / pcie_plus_compute, 2)
        rec["note"] = ("read / write: file <-> host array (page cache); h2d / compute / d2h: inside filter, synchronised between phases; "
                       "filter = their sum + statistics + allocation; startup_and_other: interpreter, numpy, libflowdn.so load, GPU context")
    print(json.dumps(rec))
    if out_json:
        with open(out_json, "w") as f:
            json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()
