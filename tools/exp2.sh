for shape in 64,1024,1024 128,1024,1024 512,1024,1024; do for r in 0 1; do
echo "shape $shape R0_RING=$r"; FDN_R0_RING=$r python bench.py --steps 1 --warmup 1 --no-cpu-baseline --shape $shape --axes z 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'])"
done; done
