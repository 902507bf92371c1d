# Tracked bench records of the non-headline BASELINE configurations on one GPU, each with bench.py's
# own post-run check: results in gpurun_out/cfg{1,3,4}_bench.json -> profiles/rNN_cfg{1,3,4}_bench.json
#   configs[1]: 512 x 512 x 256, sigma 2, OF along Z only
#   configs[3]: 1024^3, sigma 4 (K = 33 on all three axes)
#   configs[4]: 2048 x 2048 x 512 (uint16 range), sigma (2, 2, 4) -> bench.py takes one sigma: run as sigma 2 on Z, Y and 4 on X through --sigmas
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --shape 256,512,512 --axes z --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/cfg1_bench.json 2> gpurun_out/cfg1_bench.err; cut -c1-200 gpurun_out/cfg1_bench.json
timeout -k 10 600 python bench.py --shape 1024,1024,1024 --sigma 4 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/cfg3_bench.json 2> gpurun_out/cfg3_bench.err; cut -c1-200 gpurun_out/cfg3_bench.json
timeout -k 10 900 python bench.py --shape 512,2048,2048 --sigmas 2,2,4 --levels 3 --winsize 15 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/cfg4_bench.json 2> gpurun_out/cfg4_bench.err; cut -c1-200 gpurun_out/cfg4_bench.json
# configs[4] at its own shape under the counters (its record carries `traffic`): separate --pmc passes, then the
# traffic file bench.py matches by kernel, workload and kernel-source hash; the bench record is re-taken afterwards so that it picks it up
if [ "$CFG4_PMC" = 1 ]; then
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  rm -rf gpurun_out/cfg4_pmc*
  B="python3 bench.py --shape 512,2048,2048 --sigmas 2,2,4 --levels 3 --winsize 15 --steps 1 --warmup 0 --no-cpu-baseline --no-timers --no-check"
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d gpurun_out/cfg4_pmc$i -- $B > gpurun_out/cfg4_pmc$i.log 2>&1 || { echo "cfg4 set $i failed"; break; }
  done
  python3 tools/make_traffic_json.py "k_farneback_iter<7" gpurun_out/cfg4_traffic.json ${1:-unknown} "gpurun_out/cfg4_pmc*/**/*_counter_collection.csv" 3 15 512,2048,2048 2,2,4 " --shape 512,2048,2048 --sigmas 2,2,4 --levels 3 --winsize 15"
  cp gpurun_out/cfg4_traffic.json profiles/${ROUND:-r05}_cfg4_traffic.json
  timeout -k 10 900 python bench.py --shape 512,2048,2048 --sigmas 2,2,4 --levels 3 --winsize 15 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/cfg4_bench.json 2> gpurun_out/cfg4_bench.err; cut -c1-200 gpurun_out/cfg4_bench.json
fi
