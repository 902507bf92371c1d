"""How often does a 64-lane row segment of the fused kernel contain a lane whose flow leaves the LDS
window?  Flows of the bench volume (chained, as the sweep does) for neighbour distances 1..8."""
import numpy as np, sys
sys.path.insert(0, ".")
import flowdenoising_amd as fd
from flowdenoising_amd.synth import make_volume
vol = make_volume((20, 1024, 1024), seed=1234 + 3, amplitude=100.0)
t = 10
flow = np.zeros((1024, 1024, 2), np.float32)
for d in range(1, 9):
    flow = fd.get_flow(vol[t + d], vol[t], 0, 5, flow)   # seq:98 chained
    fx, fy = flow[..., 0], flow[..., 1]
    line = [f"d={d} |f|max={np.abs(flow).max():.1f} p99={np.percentile(np.abs(flow), 99):.2f}"]
    for (D, DX) in ((4, 5), (7, 5), (8, 8), (10, 8), (7, 8), (12, 12)):
        # x1 = floor(x+fx): window columns [x-DX, x+DX-1] for the 2x2 footprint -> floor(fx) in [-DX, DX-2]
        mx = (np.floor(fx) < -DX) | (np.floor(fx) > DX - 2)
        my = (np.floor(fy) < -D) | (np.floor(fy) > D - 2)
        miss = mx | my
        seg = miss[:, :1024 - 1024 % 52].reshape(1024, -1, 52).any(axis=2)
        line.append(f"{D}x{DX}: lane {miss.mean()*100:.2f}% wave {seg.mean()*100:.1f}%")
    print(" | ".join(line), flush=True)
