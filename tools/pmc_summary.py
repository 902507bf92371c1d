#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean counter value per dispatch."""
import csv, glob, sys, collections
pat = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc*/**/*_counter_collection.csv"
kern = sys.argv[2] if len(sys.argv) > 2 else "k_farneback_fused"
for f in sorted(glob.glob(pat, recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{f.split('/')[1]:8s} {k:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
