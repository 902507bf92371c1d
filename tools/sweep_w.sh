#!/bin/bash
# bench every prebuilt library in build_variants/ at the window sizes given as arguments
mkdir -p gpurun_out
cp flowdenoising_amd/libflowdn.so /tmp/lib_default.so
for f in build_variants/lib_*.so; do
  cp $f flowdenoising_amd/libflowdn.so
  for w in "$@"; do
    echo "== $f w=$w" | tee -a gpurun_out/sweep_w.log
    timeout -k 10 120 python bench.py --winsize $w --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('kernel_ms_per_step'))" | tee -a gpurun_out/sweep_w.log
  done
done
cp /tmp/lib_default.so flowdenoising_amd/libflowdn.so
