# The round-end sequence on one box: the GPU suite, smoke(), the default bench line.
# usage (through gpurun): bash tools/final_check.sh   -> gpurun_out/final_{tests,smoke,bench}.log
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/final_tests.log 2>&1; rc=$?
tail -3 gpurun_out/final_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1 || { tail -5 gpurun_out/final_smoke.log; exit 1; }
tail -2 gpurun_out/final_smoke.log
timeout -k 10 300 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err || { tail -5 gpurun_out/final_bench.err; exit 1; }
python -c "import json; d=json.loads(open('gpurun_out/final_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['checked']['bit_equal'], d['cpu_baseline']['value'])"
