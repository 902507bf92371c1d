"""Throughput of the drop-in pair operators under par's thread pool (src/flowdenoising.py:187-193: P threads, each calling
get_flow and warp_slice per neighbour): pairs per second at P = 1, 2, 4, 8, 16 on images of one size, host arrays in and out.
usage: pair_seam_rate.py [H] [W] [pairs per thread]      FDN_PAIR_HANDLES=1 shows the one-handle (serialised) rate"""
import sys, time, json, os
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, ".")
import flowdenoising_amd as fdn
from flowdenoising_amd.synth import make_volume

H = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n = int(sys.argv[3]) if len(sys.argv) > 3 else 24
vol = make_volume((18, H, W), seed=3, amplitude=100.0)


def chain(t):
    """one thread's work: a chain of n pairs on its own target (flow carried along, as par:312-326)"""
    target = vol[t % 16 + 1]
    flow = np.zeros((H, W, 2), np.float32)
    acc = np.zeros((H, W), np.float32)
    for i in range(n):
        ref = vol[(t + i) % 18]
        flow = fdn.get_flow(ref, target, 0, 5, flow)
        acc += fdn.warp_slice(ref, flow) * np.float32(0.1)
    return acc


want = chain(0)
res = {"image": [H, W], "pairs_per_thread": n, "pair_handles": int(os.environ.get("FDN_PAIR_HANDLES", "8")), "rates": {}}
for P in (1, 2, 4, 8, 16):
    with ThreadPoolExecutor(max_workers=P) as pool:
        list(pool.map(chain, range(P)))                  # warm: handles, workspaces
        t0 = time.perf_counter()
        outs = list(pool.map(chain, range(P)))
        dt = time.perf_counter() - t0
    assert np.array_equal(outs[0], want)
    res["rates"][str(P)] = {"pairs_per_s": round(P * n / dt, 1), "ms_per_pair_per_thread": round(dt / n * 1e3, 3)}
print(json.dumps(res))
