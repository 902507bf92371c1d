# HBM bytes (FETCH_SIZE, WRITE_SIZE) and L2 hits / misses of the one-iteration kernel with and without the two-sided chain steps.
# usage (through gpurun): bash tools/pmc_two_sided.sh   -> gpurun_out/pmc2s_*.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for two in 1 0; do
  rm -rf gpurun_out/pmc2s_$two.*
  B="python3 bench.py --levels 3 --winsize 15 --steps 1 --warmup 0 --no-cpu-baseline --no-timers --no-check --two-sided $two"
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc2s_$two.$i -- $B > gpurun_out/pmc2s_$two.$i.log 2>&1 || { echo "two=$two set $i ($set) failed"; tail -3 gpurun_out/pmc2s_$two.$i.log; }
  done
  for k in "k_farneback_iter<7, 1, false" "k_farneback_iter<7, 1, true"; do
    echo "== two_sided=$two  $k" | tee -a gpurun_out/pmc2s_$two.txt
    python3 tools/pmc_summary.py "gpurun_out/pmc2s_$two.*/**/*_counter_collection.csv" "$k" | tee -a gpurun_out/pmc2s_$two.txt
  done
done
