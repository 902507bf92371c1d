# The in-process parity tests under the debugging allocator (FDN_GUARD_ALLOC, fdn_api.hip): every device buffer of the library ends at an
# unmapped page -- a kernel that addresses past the end of an operand faults at once -- and fresh memory is NaNs (2) or zeros (3): a
# result that depends on the fill is a read of memory nobody wrote.  usage (through gpurun): bash tools/guard_parity.sh -> gpurun_out/guard_*
# the in-process parity tests with every device buffer ending at an unmapped page, fresh memory as NaNs and as zeros
for mode in 2 3; do
  echo "== FDN_GUARD_ALLOC=$mode"
  FDN_GUARD_ALLOC=$mode FDN_TEST_TRACE=gpurun_out/guard_trace_$mode.txt AMD_LOG_LEVEL=1 timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_ref_sweeps.py tests/test_gpu_integer.py tests/test_gpu_full.py -m "gpu and not gpu_subprocess" -x -q -p no:cacheprovider --deselect tests/test_gpu_full.py::test_wide_kernel_on_a_gib_volume_spot_parity -k "not full_size and not config and not reserve and not workspace and not statistics and not gib" > gpurun_out/guard_$mode.out 2> gpurun_out/guard_$mode.err; echo "rc=$?"; tail -3 gpurun_out/guard_$mode.out | cut -c1-200; grep -i -E "fault|core dump" gpurun_out/guard_$mode.err | head -3
done
