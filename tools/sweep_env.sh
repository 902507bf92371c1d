#!/bin/bash
# usage: sweep_env.sh VAR v1 v2 ... : bench the built library under each value of an env var
set -e
mkdir -p gpurun_out
var=$1; shift
for v in "$@"; do
  echo "== $var=$v" | tee -a gpurun_out/sweep_env.log
  env $var=$v timeout -k 10 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'].get('avg_launch_ms'), d.get('kernel_ms_per_step'))" | tee -a gpurun_out/sweep_env.log
done
