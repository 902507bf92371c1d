# Two PMC passes (SQ activity, instruction counts + cycles) of the fused kernel for the library in place.
# usage: [KERN=k_polyexp] bash tools/pmc_quick.sh <tag> [bench args]   -> gpurun_out/pmcq_<tag>.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/pmcq_$tag.*
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-timers --no-check $@"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcq_$tag.$i -- $B > gpurun_out/pmcq_$tag.$i.log 2>&1 || { echo "set $i failed/timeout"; break; }
done
python3 tools/pmc_summary.py "gpurun_out/pmcq_$tag.*/**/*_counter_collection.csv" ${KERN:-k_farneback_fused} > gpurun_out/pmcq_$tag.txt; cat gpurun_out/pmcq_$tag.txt
