"""BASELINE configs[4] on ONE GPU (the reference case is 8 GPUs): 2048 x 2048 x 512 voxels (uint16 range), sigma = (2, 2, 4),
-l 3 -w 15, volume resident in HBM.  Checks that the workspace logic copes with 8 GiB volumes and reports the time."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from flowdenoising_amd import _lib, synth
shape = (512, 2048, 2048)
dev = torch.device("cuda", 0)
h = _lib.Handle(0)
h.set_stream(torch.cuda.current_stream().cuda_stream)
vol = synth.make_volume(shape, seed=1234 + 5, amplitude=100.0, xp=torch, device=dev)
vol = ((vol - vol.min()) / (vol.max() - vol.min()) * 4095).round()      # what a uint16 stack holds, as f32
out = torch.empty_like(vol)
ks = [_lib.gaussian_kernel(2.0), _lib.gaussian_kernel(2.0), _lib.gaussian_kernel(4.0)]
params = _lib.SweepParams(3, 15, 3, 5, 1.2, _lib.BORDER_MEAN_PAD, 1, 1)
mean = h.mean_dev(vol.data_ptr(), vol.numel())
h.enable_timers(True)
for it in range(2):
    h.timers(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h.filter_3d_dev(vol.data_ptr(), out.data_ptr(), shape, ks, mean, params)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    tm = h.timers()
    print(f"run {it}: {dt:.2f} s = {vol.numel() / dt / 1e6:.1f} Mvox/s; " + ", ".join(f"{n}={v[0]:.0f}" for n, v in tm.items() if v[1]), flush=True)
    print("free/total GiB", [round(v / 2**30, 1) for v in torch.cuda.mem_get_info()], flush=True)
print("finite:", bool(torch.isfinite(out).all()), "mean in/out", float(vol.mean()), float(out.mean()))
