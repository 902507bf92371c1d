# Where do the waves of the one-iteration kernel stand?  rocprofv3 PC sampling (beta) of one Z pass on a small slab, aggregated
# on the box into a per-instruction / per-stall-reason table (the raw CSV is far larger than gpurun merges back).
# usage (through gpurun): bash tools/pc_sampling.sh [stochastic|host_trap] [bench args]
METHOD=${1:-stochastic}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf /tmp/pcs
ARGS=${@:---shape 96,1024,1024 --axes z --levels 0 --winsize 15}
UNIT=cycles; IVAL=1048576
if [ "$METHOD" = host_trap ]; then UNIT=time; IVAL=1000; fi
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout -k 10 170 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $METHOD --pc-sampling-unit $UNIT --pc-sampling-interval $IVAL --kernel-trace --output-format csv -d /tmp/pcs -- python3 bench.py $ARGS --steps 1 --warmup 0 --no-cpu-baseline --no-timers --no-check > gpurun_out/pcs_$METHOD.log 2>&1
echo "rocprofv3 rc=$?"; tail -3 gpurun_out/pcs_$METHOD.log | cut -c1-200
ls -la /tmp/pcs/*/ 2>/dev/null | head
python3 - <<'PY'
import csv, glob, collections, sys
files = glob.glob("/tmp/pcs/**/*pc_sampling*.csv", recursive=True)
print("files:", files)
for f in files:
    rd = csv.DictReader(open(f))
    print("columns:", rd.fieldnames)
    by = collections.Counter(); tot = 0
    for r in rd:
        tot += 1
        ins = (r.get("Instruction") or "")[:60]
        key = (ins, r.get("Instruction_Type", ""), r.get("Stall_Reason", r.get("Wave_Issued", "")))
        by[key] += 1
    out = open("gpurun_out/pcs_table_%s.txt" % ("stochastic" if "stoch" in f else "host_trap"), "w")
    out.write("samples %d  file %s\n" % (tot, f))
    for (ins, typ, why), n in by.most_common(150):
        out.write("%7d %5.2f%%  %-60s %-14s %s\n" % (n, 100.0 * n / max(tot, 1), ins, typ, why))
    out.close()
    print("samples", tot)
PY
head -40 gpurun_out/pcs_table_*.txt 2>/dev/null | cut -c1-160
