cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 1 --warmup 0 --shape 256,1024,1024 --axes z --no-cpu-baseline"
rm -rf gpurun_out/pmc*
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  echo "set $i: $set"
  timeout -k 10 90 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc$i -- $B --no-timers > gpurun_out/pmc$i.log 2>&1 || { echo "set $i failed/timeout"; break; }
done
python tools/pmc_summary.py
