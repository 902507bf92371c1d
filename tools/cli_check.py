import sys, subprocess, numpy as np
sys.path.insert(0, ".")
from flowdenoising_amd import io as fio
from flowdenoising_amd.synth import make_volume
v = make_volume((24, 96, 128), seed=9, amplitude=100.0)
v16 = ((v - v.min()) / (v.max() - v.min()) * 4095).astype(np.uint16)
fio.write_volume("/tmp/in.tif", v16)
r = subprocess.run([sys.executable, "flowdenoising.py", "-i", "/tmp/in.tif", "-o", "/tmp/out.tif", "-s", "2", "2", "4", "-l", "3", "-w", "15", "-v", "1"], capture_output=True, text=True)
print(r.returncode, r.stderr[-600:])
o = fio.read_volume("/tmp/out.tif")
print(o.dtype, o.shape, float(o.mean()), float(v16.mean()))
from oracle import oracle as O
ks = [O.get_gaussian_kernel(s) for s in (2, 2, 4)]
want = O.OF_filter(v16.astype(np.float32), ks, 3, 15, nthreads=8)
print("max |diff| vs oracle after uint16 cast:", np.abs(o.astype(np.float32) - want.astype(np.uint16).astype(np.float32)).max())
