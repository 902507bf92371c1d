# Same-box A/B of the two-sided chain steps (fdn_set_option two_sided): the wide-window path on the bench volume and configs[4].
# usage (through gpurun, from the repo root): bash tools/ab_two_sided.sh [cfg4]
mkdir -p gpurun_out
for two in 1 0 1 0; do
  timeout -k 10 300 python bench.py --levels 3 --winsize 15 --steps 2 --warmup 1 --no-cpu-baseline --two-sided $two > gpurun_out/ab_w15_two$two.json 2> gpurun_out/ab_w15_two$two.err || { echo "w15 two=$two failed"; tail -5 gpurun_out/ab_w15_two$two.err; exit 1; }
  python3 - <<PY
import json
d = json.load(open("gpurun_out/ab_w15_two$two.json"))
print("w15 bench volume, two_sided=$two:", d["value"], "Mvox/s", d["ms_per_step"], "ms/step", d["kernel_ms_per_step"], "checked", d["checked"]["ok"], d["checked"]["bit_equal"], d["checked"].get("bit_equal_kernel_order"))
PY
done
if [ "$1" = cfg4 ]; then
for two in 1 0; do
  timeout -k 10 600 python bench.py --shape 512,2048,2048 --sigmas 2,2,4 --levels 3 --winsize 15 --steps 1 --warmup 1 --no-cpu-baseline --two-sided $two > gpurun_out/ab_cfg4_two$two.json 2> gpurun_out/ab_cfg4_two$two.err || { echo "cfg4 two=$two failed"; tail -5 gpurun_out/ab_cfg4_two$two.err; exit 1; }
  python3 - <<PY
import json
d = json.load(open("gpurun_out/ab_cfg4_two$two.json"))
print("configs[4], two_sided=$two:", d["value"], "Mvox/s", d["ms_per_step"], "ms/step", d["kernel_ms_per_step"], "checked", d["checked"]["ok"], d["checked"]["bit_equal"], d["checked"].get("bit_equal_kernel_order"))
PY
done
fi
