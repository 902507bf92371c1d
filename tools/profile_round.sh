# Round profile (run through gpurun from the repo root): rocprofv3 kernel-trace stats of the default bench command
# + PMC passes of the dominant kernel.  Results land in gpurun_out/; copy what is to be judged into profiles/:
#   gpurun_out/prof_trace/**/*_kernel_stats.csv -> profiles/rNN_kernel_stats.csv
#   gpurun_out/bench_prof.json                  -> profiles/rNN_bench.json
#   gpurun_out/pmc_summary.txt                  -> profiles/rNN_pmc_summary.txt
#   gpurun_out/traffic.json                     -> profiles/rNN_traffic.json   (bench.py picks the newest matching one)
# usage: bash tools/profile_round.sh <commit>
COMMIT=${1:-unknown}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/prof_trace gpurun_out/pmc*
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err || echo "trace run failed"
cat gpurun_out/bench_prof.json
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-timers --no-check"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  echo "set $i: $set"
  timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc$i -- $B > gpurun_out/pmc$i.log 2>&1 || { echo "set $i failed/timeout"; break; }
done
python3 tools/pmc_summary.py "gpurun_out/pmc*/**/*_counter_collection.csv" k_farneback_fused > gpurun_out/pmc_summary.txt; cat gpurun_out/pmc_summary.txt
python3 tools/make_traffic_json.py k_farneback_fused gpurun_out/traffic.json $COMMIT
