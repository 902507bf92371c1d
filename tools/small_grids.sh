#!/bin/bash
# per-rank pass shapes of an 8-GPU run of 1024x1024x512 (Z / Y / X slabs), timed on one GPU
mkdir -p gpurun_out
for cfg in "64,1024,1024 z" "512,128,1024 y" "512,1024,128 x" "128,1024,1024 z" "256,1024,1024 z"; do
  set -- $cfg
  echo "== $1 $2" | tee -a gpurun_out/small.log
  timeout -k 10 120 python bench.py --shape $1 --axes $2 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline'].get('avg_launch_ms'), d.get('kernel_ms_per_step'))" | tee -a gpurun_out/small.log
done
