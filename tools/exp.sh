for v in NOSHFL NOGATHER; do
  cp flowdenoising_amd/libflowdn.so /tmp/orig.so
  cp flowdenoising_amd/libflowdn_$v.so flowdenoising_amd/libflowdn.so
  echo $v; python bench.py --steps 1 --warmup 0 --shape 256,1024,1024 --axes z --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms_per_step'], d['roofline']['avg_launch_ms'])"
  cp /tmp/orig.so flowdenoising_amd/libflowdn.so
done
echo BASE; python bench.py --steps 1 --warmup 0 --shape 256,1024,1024 --axes z --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms_per_step'], d['roofline']['avg_launch_ms'])"
