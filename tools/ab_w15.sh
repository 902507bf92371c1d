rm -f gpurun_out/sweep.log
PARITY=0 CHECK=1 BENCH_ARGS="--levels 3 --winsize 15" bash tools/sweep_variants.sh
