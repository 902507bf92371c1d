# PMC passes + kernel stats of the wide-window path (bench.py --levels 3 --winsize 15); run through gpurun from the repo root.
# usage: bash tools/profile_w15.sh <commit>   (results in gpurun_out/w15_*; copy into profiles/r02_w15_*)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/w15_trace gpurun_out/w15_pmc*
B="python3 bench.py --levels 3 --winsize 15 --steps 1 --warmup 0 --no-cpu-baseline --no-timers --no-check"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/w15_trace -- python3 bench.py --levels 3 --winsize 15 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/w15_bench.json 2> gpurun_out/w15_bench.err || echo "trace failed"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/w15_pmc$i -- $B > gpurun_out/w15_pmc$i.log 2>&1 || { echo "set $i failed"; break; }
done
python3 tools/pmc_summary.py "gpurun_out/w15_pmc*/**/*_counter_collection.csv" "k_farneback_iter<7, 1, true" > gpurun_out/w15_pmc_summary.txt
python3 tools/pmc_summary.py "gpurun_out/w15_pmc*/**/*_counter_collection.csv" "k_farneback_iter<7, 1, false" >> gpurun_out/w15_pmc_summary.txt
cat gpurun_out/w15_pmc_summary.txt
# what bench.py reads: the mean over the launches of every level and kind (its roofline line averages the same way) ...
python3 tools/make_traffic_json.py "k_farneback_iter<7" gpurun_out/w15_traffic.json ${1:-unknown} "gpurun_out/w15_pmc*/**/*_counter_collection.csv" 3 15
# ... and the level-0 warping launch on its own (512 x 1024 x 1024 pixels per launch), for DESIGN.md
python3 tools/make_traffic_json.py "k_farneback_iter<7, 1, true" gpurun_out/w15_traffic_level0_warp.json ${1:-unknown} "gpurun_out/w15_pmc*/**/*_counter_collection.csv" 0 15
