cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 1 --warmup 0 --shape 256,1024,1024 --axes z --no-cpu-baseline --no-timers"
rm -rf gpurun_out/pmc*
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_REQ_sum" "TCC_READ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_TAG_STALL_sum"; do
  i=$((i+1)); echo "set $i: $set"
  timeout -k 10 90 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc$i -- $B > gpurun_out/pmc$i.log 2>&1 || { echo "set $i failed/timeout"; tail -3 gpurun_out/pmc$i.log; }
done
python tools/pmc_summary.py
