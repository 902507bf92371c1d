# HBM bytes per launch of the dominant kernel for the library in place: FETCH_SIZE and WRITE_SIZE in separate passes
# (2 x FETCH_SIZE + WRITE_SIZE, KiB; the gfx950 correction of MI355X_MICROARCH.md).
# usage: bash tools/pmc_bytes.sh <tag> <kernel substring> [bench args]
tag=$1; kern=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/pmcb_$tag.*
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-timers --no-check $@"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcb_$tag.$i -- $B > gpurun_out/pmcb_$tag.$i.log 2>&1 || { echo "set $i failed/timeout"; break; }
done
python3 tools/pmc_summary.py "gpurun_out/pmcb_$tag.*/**/*_counter_collection.csv" "$kern" | tee gpurun_out/pmcb_$tag.txt
python3 - <<PY
import re
v={}
for ln in open("gpurun_out/pmcb_$tag.txt"):
    m=re.search(r"(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+) mean=([0-9.e+]+)", ln)
    if m: v[m.group(1)]=float(m.group(3))
if len(v)==2: print("$tag bytes per launch %.3g = %.1f B/px of 512x1024x1024" % ((2*v["FETCH_SIZE"]+v["WRITE_SIZE"])*1024, (2*v["FETCH_SIZE"]+v["WRITE_SIZE"])*1024/536870912))
PY
