#!/bin/bash
# Same-box A/B of the sub-batches (fdn_set_option sub_batches): one stream against two on the 8-rank
# pass shapes, configs[1], the headline and (CFG4=1) configs[4].   usage (through gpurun): bash tools/ab_sub_batches.sh
mkdir -p gpurun_out
run() {   # name, bench args...
  name=$1; shift
  for sb in ${SBS:-1 2}; do
    timeout -k 10 900 python bench.py "$@" --no-cpu-baseline --sub-batches $sb > gpurun_out/ab_sb_${name}_$sb.json 2> gpurun_out/ab_sb_${name}_$sb.err || { echo "$name sb=$sb failed"; tail -5 gpurun_out/ab_sb_${name}_$sb.err; return 1; }
    python - <<PY
import json
d = json.load(open("gpurun_out/ab_sb_${name}_$sb.json"))
print("$name sub_batches=$sb ran", d.get("sub_batches"), ":", d["value"], "Mvox/s", d["ms_per_step"], "ms/step", "launch", d["roofline"]["avg_launch_ms"], "checked", d["checked"]["ok"], d["checked"]["bit_equal"], flush=True)
PY
  done
}
run rank8z --shape 64,1024,1024 --axes z --steps 5 --warmup 2 || exit 1
run rank8y --shape 512,128,1024 --axes y --steps 5 --warmup 2 || exit 1
run rank8x --shape 512,1024,128 --axes x --steps 5 --warmup 2 || exit 1
run cfg1 --shape 256,512,512 --axes z --steps 5 --warmup 2 || exit 1
run headline --steps 3 --warmup 1 || exit 1
run w15 --levels 3 --winsize 15 --steps 2 --warmup 1 || exit 1
if [ "$CFG4" = 1 ]; then run cfg4 --shape 512,2048,2048 --sigmas 2,2,4 --levels 3 --winsize 15 --steps 1 --warmup 1 || exit 1; fi
