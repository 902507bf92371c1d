#!/bin/bash
mkdir -p gpurun_out
for occ in 3 4 5; do
for cfg in "512,1024,1024 z" "64,1024,1024 z" "128,1024,1024 z" "512,128,1024 y"; do
  set -- $cfg
  echo "== occ $occ $1 $2" | tee -a gpurun_out/occ.log
  FDN_FUSED_OCC=$occ timeout -k 10 120 python bench.py --shape $1 --axes $2 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'].get('avg_launch_ms'))" | tee -a gpurun_out/occ.log
done; done
