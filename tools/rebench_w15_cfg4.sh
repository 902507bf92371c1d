mkdir -p gpurun_out
timeout -k 10 300 python bench.py --levels 3 --winsize 15 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/w15_bench.json 2> gpurun_out/w15_bench.err || exit 1
timeout -k 10 900 python bench.py --shape 512,2048,2048 --sigmas 2,2,4 --levels 3 --winsize 15 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/cfg4_bench.json 2> gpurun_out/cfg4_bench.err || exit 1
python -c "
import json
for f in ('w15','cfg4'):
    d=json.loads(open('gpurun_out/%s_bench.json'%f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['roofline'].get('traffic_frac'), (d['roofline'].get('limiter') or {}).get('utilisation'), d['checked']['bit_equal'])
"
