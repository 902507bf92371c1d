#!/bin/bash
# Bench every prebuilt kernel variant in build_variants/ (scratch copy on the GPU box only).
# BENCH_ARGS: extra bench.py arguments (e.g. "--levels 3 --winsize 15"); CHECK=1 keeps bench.py's oracle check on;
# PARITY=0 skips the parity tests on the first variant (timing experiments with deliberately wrong kernels), PARITY=all runs them on every variant.
set -e
mkdir -p gpurun_out
cp flowdenoising_amd/libflowdn.so /tmp/lib_default.so
first=1
nocheck=--no-check
[ "$CHECK" = 1 ] && nocheck=
for f in build_variants/lib_*.so; do
  cp $f flowdenoising_amd/libflowdn.so
  if { [ $first = 1 ] && [ "$PARITY" != 0 ]; } || [ "$PARITY" = all ]; then timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/sweep_parity.log 2>&1; tail -1 gpurun_out/sweep_parity.log; first=0; fi
  echo "== $f" | tee -a gpurun_out/sweep.log
  timeout -k 10 300 python bench.py $BENCH_ARGS --steps 2 --warmup 1 --no-cpu-baseline $nocheck 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'].get('avg_launch_ms'), d.get('kernel_ms_per_step'), d.get('checked', {}).get('bit_equal'), d.get('checked', {}).get('max_rel_err'))" | tee -a gpurun_out/sweep.log
done
cp /tmp/lib_default.so flowdenoising_amd/libflowdn.so
