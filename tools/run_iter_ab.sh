set -e
timeout -k 10 400 python -m pytest tests/test_gpu_full.py -m gpu -x -q -k "wide_window_small_image or iter_kernel_window_sizes" > gpurun_out/t1.log 2>&1 || { tail -30 gpurun_out/t1.log; exit 1; }
tail -2 gpurun_out/t1.log
timeout -k 10 300 python -m pytest tests/test_ref_sweeps.py -m gpu -x -q > gpurun_out/t2.log 2>&1 || { tail -30 gpurun_out/t2.log; exit 1; }
tail -2 gpurun_out/t2.log
rm -f gpurun_out/sweep.log
PARITY=0 CHECK=1 BENCH_ARGS="--levels 3 --winsize 15" bash tools/sweep_variants.sh
PARITY=0 CHECK=1 BENCH_ARGS="--levels 3 --winsize 15" bash tools/sweep_variants.sh
