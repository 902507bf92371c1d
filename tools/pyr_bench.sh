#!/bin/bash
mkdir -p gpurun_out
for cfg in "3 5" "3 15" "0 15"; do
  set -- $cfg
  echo "== levels $1 winsize $2" | tee -a gpurun_out/pyr.log
  timeout -k 10 300 python bench.py --levels $1 --winsize $2 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('kernel_ms_per_step'))" | tee -a gpurun_out/pyr.log
done
