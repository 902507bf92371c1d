"""Where the out-of-core mode's time goes on the bench volume (BASELINE configs[2], 512 x 1024 x 1024, sigma 2): chunks of 64 / 128 /
256 target slices with 2 / 3 / 4 workers, against the resident path.  Writes gpurun_out/stream_profile.json (-> profiles/rNN_stream_profile.json).
usage (through gpurun): python tools/stream_profile.py [quick]"""
import json, sys, time, numpy as np
sys.path.insert(0, ".")
import flowdenoising_amd as fd
from flowdenoising_amd import _lib, streaming
from flowdenoising_amd.synth import make_volume
vol = make_volume((512, 1024, 1024), seed=1237, amplitude=100.0)
k = fd.get_gaussian_kernel(2.0)
t0 = time.perf_counter(); m = _lib.mean_host(vol); t_mean = time.perf_counter() - t0
print(f"mean_host: {t_mean:.3f} s", flush=True)
rows = []
grid = [(128, 3), (171, 3), (128, 3)] if len(sys.argv) > 1 and sys.argv[1] == "quick" else [(128, 2), (128, 3), (64, 2), (64, 3), (256, 2), (96, 3)]
out = None
for chunk, workers in grid:
    best = None
    for rep in range(2):
        out = None                      # (the previous result's 2 GiB go back before the clock starts: not part of the call)
        t0 = time.perf_counter()
        out = streaming.OF_filter_streamed(vol, [k, k, k], 0, 5, chunk, mean=m, workers=workers)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    rows.append({"chunk_slices": chunk, "workers": workers, "seconds": round(best, 3), "Mvox_per_s": round(vol.size / best / 1e6, 1)})
    print(rows[-1], flush=True)
t0 = time.perf_counter(); ref = fd.OF_filter(vol, [k, k, k], 0, 5); t_res = time.perf_counter() - t0
same = bool(np.array_equal(out, ref))
print(f"resident OF_filter (numpy in, numpy out): {t_res:.2f} s; bit-identical: {same}", flush=True)
json.dump({"what": "out-of-core mode (streaming.filter_streamed, host volume -> host result, mean given) on 512 x 1024 x 1024, sigma 2, Z + Y + X",
           "mean_host_s": round(t_mean, 3), "rows": rows, "resident_numpy_in_numpy_out_s": round(t_res, 3), "bit_identical_to_resident": same},
          open("gpurun_out/stream_profile.json", "w"), indent=1)
