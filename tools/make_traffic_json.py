#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc passes of tools/profile_round.sh into profiles/<round>_traffic.json, the file bench.py
reads for roofline.traffic / roofline.limiter.  It is stamped with the hash of the kernel sources, the commit and
the workload it was taken on; bench.py ignores it (traffic = null, loud warning) when any of them differs.

usage: make_traffic_json.py <kernel substring> <out.json> <commit> [pmc glob] [levels winsize [Z,Y,X [sigma(s) [bench args]]]]"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_hash, the one definition)

kern, out, commit = sys.argv[1], sys.argv[2], sys.argv[3]
pat = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "gpurun_out/pmc*/**/*_counter_collection.csv")
acc = collections.defaultdict(list)
for f in sorted(glob.glob(pat, recursive=True)):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = {k: sum(v) / len(v) for k, v in acc.items()}
if "FETCH_SIZE" not in mean or "WRITE_SIZE" not in mean:
    sys.exit(f"no FETCH_SIZE/WRITE_SIZE rows for {kern} under {pat}")
levels, winsize = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (0, 5)
shape = [int(v) for v in sys.argv[7].split(",")] if len(sys.argv) > 7 else [512, 1024, 1024]
sigma = (float(sys.argv[8]) if "," not in sys.argv[8] else sys.argv[8]) if len(sys.argv) > 8 else 2.0     # as bench.py's run_cfg has it
bench_args = sys.argv[9] if len(sys.argv) > 9 else (f" --levels {levels} --winsize {winsize}" if (levels, winsize) != (0, 5) else "")
# pixels of an AVERAGE launch, as bench.py prices it: with a pyramid a chain step is one launch (three for the one-iteration
# kernel) per level and level k has 4^-k of the pixels; `kern` must then select the launches of every level
px = shape[0] * shape[1] * shape[2] * sum(0.25 ** k for k in range(levels + 1)) / (levels + 1)
# FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE x 2 is the guide's gfx950 correction (128-B fabric reads tallied at 64 B)
nbytes = (2 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024
res = {
    "kernel": kern,
    "kernel_source_sha": bench.kernel_source_hash(kern),
    "kernel_sources": list(bench.kernel_sources(kern)),
    "commit": commit,
    "command": "rocprofv3 --pmc <set> -- python bench.py" + bench_args
               + " --steps 1 --warmup 0 --no-cpu-baseline --no-timers --no-check "
               "(separate passes per counter set, tools/profile_round.sh)",
    "workload": {"shape": shape, "winsize": winsize, "levels": levels, "sigma": sigma, "axes": "zyx"},
    "launches_averaged": len(acc["FETCH_SIZE"]),
    "fetch_size_kib_per_launch": mean["FETCH_SIZE"], "write_size_kib_per_launch": mean["WRITE_SIZE"],
    "correction": "FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM section; re-checked on known byte counts by "
                  "tools/ubench/fetch_calib.hip); WRITE_SIZE as is",
    "bytes_per_launch": nbytes, "pixels_per_launch": px, "bytes_per_pixel": nbytes / px,
}
model_path = os.environ.get("FDN_VALU_MODEL", os.path.join(ROOT, "profiles", "r04_valu_model.json"))
if "SQ_INSTS_VALU" in mean and "GRBM_GUI_ACTIVE" in mean and os.path.exists(model_path) and kern.startswith("k_farneback_fused"):
    # Per-class issue model (tools/valu_model.py): the kernel's own instruction mix priced with the cycles each opcode was
    # MEASURED to hold a SIMD at 4 waves per SIMD (tools/ubench/rates.hip, profiles/r04_valu_rates_256cus.txt) -- 2.89 cycles
    # per VALU instruction on average for this kernel, not the flat 4 of rounds 1-3 (nor the guide's 2, which holds for
    # v_add / v_mul / v_mov only).  GRBM_GUI_ACTIVE is summed over the 8 XCDs.
    model = json.load(open(model_path))
    per = model["per_band_row_step"]["issue_cycles"] / model["per_band_row_step"]["valu_instructions"]
    cycles = mean["GRBM_GUI_ACTIVE"] / 8
    util = mean["SQ_INSTS_VALU"] * per / (1024 * cycles)
    res["limiter"] = {"unit": "VALU issue (per-class model)", "utilisation": round(util, 3),
                      "cycles_per_valu_instruction": round(per, 3),
                      "sq_insts_valu_per_launch": mean["SQ_INSTS_VALU"], "gpu_cycles_per_launch": cycles,
                      "lds_bank_conflict_share": round(mean.get("SQ_LDS_BANK_CONFLICT", 0) / max(mean.get("SQ_LDS_IDX_ACTIVE", 1), 1), 3),
                      "source": "SQ_INSTS_VALU (executed, --pmc) x the kernel's mean issue cycles per instruction (profiles/r04_valu_model.json: "
                                "static per-stage opcode counts x measured per-opcode cycles) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); "
                                "in-kernel clock 2.05 GHz by s_memtime / s_memrealtime stamps (profiles/r04_clock_stamps.json)"}
res["all_counters_mean_per_launch"] = mean
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("kernel", "bytes_per_pixel", "kernel_source_sha")}), res.get("limiter"))
