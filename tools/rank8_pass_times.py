#!/usr/bin/env python3
"""Per-rank pass times of an 8-GPU run of BASELINE configs[2] measured on ONE GPU, on real data: the Z
pass of a 64-slice Z-slab, the Y pass of a 128-slice Y-slab and the X pass of a 128-slice X-slab of a synthetic volume, each
through bench.py (its own post-run check included), on one stream (--sub-batches 1) and with the batch's targets as two
sub-batches on two streams (--sub-batches 2; the automatic choice for these grids).
Writes profiles-style JSON: usage  rank8_pass_times.py [out.json]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [("64,1024,1024", "z", "Z pass of a 64-slice Z-slab (1 280 workgroups per launch, 1 024 rows each)"),
          ("512,128,1024", "y", "Y pass of a 128-slice Y-slab (2 560 workgroups per launch, 512 rows each)"),
          ("512,1024,128", "x", "X pass of a 128-slice X-slab (2 560 workgroups per launch, 512 rows each)")]


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "rank8_pass_times.json")
    rows = []
    for two in (1, 2):
        for shape, axes, what in SHAPES:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--shape", shape, "--axes", axes, "--steps", "5", "--warmup", "2",
                                "--no-cpu-baseline", "--sub-batches", str(two)], capture_output=True, text=True, timeout=600)
            if r.returncode:
                sys.exit(r.stderr[-2000:])
            d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            row = {"sub_batches": two, "shape": shape, "axes": axes, "what": what, "ms_per_step": d["ms_per_step"],
                   "kernel_ms_per_step": d["kernel_ms_per_step"], "avg_launch_ms": d["roofline"]["avg_launch_ms"],
                   "checked_bit_equal": d["checked"]["bit_equal"]}
            rows.append(row)
            print(json.dumps(row), flush=True)
    res = {"what": "pass times of one rank of an 8-rank run of configs[2] (512 x 1024 x 1024, sigma 2), measured on one GPU on real data",
           "ideal_ms_per_rank": "single-GPU ms per step / 8", "rows": rows}
    for two in (1, 2):
        res[f"sum_ms_sub_batches_{two}"] = round(sum(r["ms_per_step"] for r in rows if r["sub_batches"] == two), 2)
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k.startswith("sum")}))


if __name__ == "__main__":
    main()
