# Sample socket power and sclk (rocm-smi) twice a second while bench.py runs: is the fused kernel running at a power cap?
# usage: bash tools/power_trace.sh [bench args] -> gpurun_out/power_trace.log
mkdir -p gpurun_out
python bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-check "$@" > gpurun_out/power_bench.json 2> gpurun_out/power_bench.err &
bp=$!
: > gpurun_out/power_trace.log
while kill -0 $bp 2>/dev/null; do
  rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' >> gpurun_out/power_trace.log
  echo >> gpurun_out/power_trace.log
  sleep 0.5
done
wait $bp
cut -c1-200 gpurun_out/power_bench.json
sort gpurun_out/power_trace.log | uniq -c | sort -rn | head -30
