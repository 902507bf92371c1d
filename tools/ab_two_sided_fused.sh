# Same-box A/B of the two-sided chain steps on the 3-iteration kernel (headline workload): two_sided 3 (on) against 1 (off).
mkdir -p gpurun_out
for two in 3 1 3 1; do
  timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --two-sided $two > gpurun_out/ab_fused_two$two.json 2> gpurun_out/ab_fused_two$two.err || { echo "two=$two failed"; tail -5 gpurun_out/ab_fused_two$two.err; exit 1; }
  python3 - <<PY
import json
d = json.load(open("gpurun_out/ab_fused_two$two.json"))
print("headline, two_sided=$two:", d["value"], "Mvox/s", d["ms_per_step"], "ms/step", d["kernel_ms_per_step"], "launch ms", d["roofline"]["avg_launch_ms"], "checked", d["checked"]["ok"], d["checked"]["bit_equal"])
PY
done
