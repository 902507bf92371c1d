# SQ instruction / activity counters of the one-iteration kernel for a library of build_variants/ (tools/build_variant.sh) put in
# place of the product's for the run.  usage (through gpurun): bash tools/pmc_variant.sh <name> [bench args, default -l 3 -w 15]
# -> gpurun_out/pmcv_<name>.txt
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/pmcv_$name.*
cp flowdenoising_amd/libflowdn.so /tmp/lib_default.so
cp build_variants/lib_$name.so flowdenoising_amd/libflowdn.so
ARGS="${@:---levels 3 --winsize 15}"
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-timers --no-check $ARGS"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcv_$name.$i -- $B > gpurun_out/pmcv_$name.$i.log 2>&1 || { echo "set $i ($set) failed"; tail -3 gpurun_out/pmcv_$name.$i.log; }
done
cp /tmp/lib_default.so flowdenoising_amd/libflowdn.so
for k in "k_farneback_iter<7, 1, false" "k_farneback_iter<7, 1, true"; do
  echo "== $name  $k" | tee -a gpurun_out/pmcv_$name.txt
  python3 tools/pmc_summary.py "gpurun_out/pmcv_$name.*/**/*_counter_collection.csv" "$k" | tee -a gpurun_out/pmcv_$name.txt
done
