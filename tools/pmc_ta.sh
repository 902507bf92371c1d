# Vector-memory path (texture addresser / L1) counters of the one-iteration kernel: is the gather of R1 at p + flow --
# one cache line per lane and instruction where flows differ from lane to lane -- what paces the row step?
# usage (through gpurun): bash tools/pmc_ta.sh [bench args, default -l 3 -w 15]   -> gpurun_out/pmcta.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/pmcta.*
ARGS="${@:---levels 3 --winsize 15}"
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-timers --no-check $ARGS"
rocprofv3 --list-avail > gpurun_out/pmc_avail.txt 2>&1 || true
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcta.$i -- $B > gpurun_out/pmcta.$i.log 2>&1 || { echo "set $i ($set) failed"; tail -3 gpurun_out/pmcta.$i.log; }
done
for k in "k_farneback_iter<7, 1, false" "k_farneback_iter<7, 1, true" "k_farneback_fused"; do
  echo "== $k" | tee -a gpurun_out/pmcta.txt
  python3 tools/pmc_summary.py "gpurun_out/pmcta.*/**/*_counter_collection.csv" "$k" | tee -a gpurun_out/pmcta.txt
done
