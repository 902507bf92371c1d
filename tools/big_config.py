"""BASELINE configs[3] on one GPU: 1024 x 1024 x 1024 f32, sigma = 4 (K = 33): time the three passes with the volume
resident in HBM and spot-check one Z-pass target slice against the oracle run on its 33-slice sub-volume."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from flowdenoising_amd import _lib, synth
from oracle import oracle as O
shape = (1024, 1024, 1024)
dev = torch.device("cuda", 0)
h = _lib.Handle(0)
h.set_stream(torch.cuda.current_stream().cuda_stream)
vol = synth.make_volume(shape, seed=1234 + 4, amplitude=100.0, xp=torch, device=dev)
out = torch.empty_like(vol)
k = _lib.gaussian_kernel(4.0)
params = _lib.SweepParams(0, 5, 3, 5, 1.2, _lib.BORDER_MEAN_PAD, 1, 1)
mean = h.mean_dev(vol.data_ptr(), vol.numel())
for axes in ("z", "zyx"):
    ks = [k if c in axes else None for c in "zyx"]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h.filter_3d_dev(vol.data_ptr(), out.data_ptr(), shape, ks, mean, params)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"axes {axes}: {dt:.3f} s = {vol.numel() / dt / 1e6:.1f} Mvox/s", flush=True)
    if axes == "z":
        t = 500
        sub = vol[t - 16:t + 17].cpu().numpy()
        O.build()
        want = O.filter_axis_range(sub, 0, k, 0, 5, mean, 16, 17, nthreads=16)
        got = out[t].cpu().numpy()
        err = np.abs(got - want[16]).max() / np.abs(want[16]).max()
        print("spot parity (Z pass, target 500): rel err", err, "bit-equal", np.array_equal(got, want[16]), flush=True)
        assert err < 2e-6
print("free/total GiB", [round(v / 2**30, 1) for v in torch.cuda.mem_get_info()])
