#!/usr/bin/env python3
"""bench.py -- Mvoxels/s of the full flow-driven denoise (Z, Y and X passes, sigma=2, levels=0,
winsize=5) on a synthetic 1024x1024x512 float32 volume (BASELINE.json configs[2]).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one complete OF_filter of the volume (mean + three axis sweeps), input and output
resident in HBM.  PyTorch is used for device memory and the synthetic generator only; all arithmetic of the
step runs in libflowdn.so, and with N > 1 every exchange goes through libflowdn_rccl.so (ncclSend / ncclRecv over
xGMI, include/flowdn_rccl.h) -- torch.distributed is not initialised (--engine python keeps the torch.distributed
slab engine of flowdenoising_amd/distributed.py for comparison).

Rank 0 prints ONE JSON line (contract in the task statement) including
  "roofline":     the dominant kernel against the 8 TB/s HBM peak, priced on the bytes that kernel MUST move
                  (never a fraction above 1); the SURVEY 8(d) stage-list figure sits beside it under
                  "unfused_algorithmic"; PMC traffic / limiter only from a committed profile of the same sources
  "sweep":        the warped-Gaussian sweep as its own kernel (north_star's 50 % target), timed after the run
  "checked":      the timed output re-derived pass by pass and spot-checked against the oracle (after the timed region)
  "cpu_baseline": the CPU oracle (oracle/, a port of the reference's arithmetic parallelised
                  over target slices like src/flowdenoising.py:181-206) on a bounded sample.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
KERNEL_SOURCES = ("fdn_fused.hip", "fdn_iter.hip", "fdn_device.h", "fdn_kernels.hip")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--shape", default="512,1024,1024", help="Z,Y,X")
    ap.add_argument("--sigma", type=float, default=2.0)
    ap.add_argument("--sigmas", default="", help="per-axis sigmas Z,Y,X (configs[4] is 2,2,4); overrides --sigma")
    ap.add_argument("--axes", default="zyx", help="subset of zyx (configs[1] is 'z')")
    ap.add_argument("--amplitude", type=float, default=100.0)
    ap.add_argument("--levels", type=int, default=0, help="pyramid levels (-l); configs[4] uses 3")
    ap.add_argument("--winsize", type=int, default=5, help="Farneback window (-w); configs[4] uses 15")
    ap.add_argument("--integer", choices=("", "seq", "par"), default="", help="time the integer-volume semantics instead (not the headline): "
                    "seq = float64 padded volume, par = integer images; N = 1 only")
    ap.add_argument("--engine", choices=("native", "python"), default="native", help="N > 1: fdn_filter_3d_sharded on the native transport "
                    "(libflowdn_rccl.so: RCCL, or shared memory when ranks share a GPU), or the torch.distributed slab engine above the C ABI (distributed.py)")
    ap.add_argument("--path", type=int, default=0, help="fdn_set_option path: 0 auto, 1 staged, 2 per-iteration kernels")
    ap.add_argument("--sub-batches", type=int, default=0, help="fdn_set_option sub_batches: 0 automatic (two streams on small grids), 1 one stream, 2 two")
    ap.add_argument("--cpu-targets", type=int, default=0, help="target slices of the CPU sample (0 = four per core)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the post-run oracle spot check and the sweep-kernel line")
    ap.add_argument("--no-timers", action="store_true", help="skip per-kernel HIP-event timing")
    ap.add_argument("--no-sampling", action="store_true", help="do not sample shader clock and socket power during the timed region")
    return ap.parse_args()


# the sources a kernel is compiled from: a committed PMC pass of one kernel stays valid when another kernel's file changes
KERNEL_SOURCES_OF = {"k_farneback_fused": ("fdn_fused.hip", "fdn_device.h"), "k_farneback_iter": ("fdn_iter.hip", "fdn_device.h"),
                     "k_update_flow_scan": ("fdn_kernels.hip", "fdn_device.h")}


def kernel_sources(kernel=None):
    for name, files in KERNEL_SOURCES_OF.items():
        if kernel and str(kernel).startswith(name):
            return files
    return KERNEL_SOURCES


def kernel_source_hash(kernel=None):
    """sha of the sources `kernel` is compiled from (all kernel sources when kernel is None)."""
    h = hashlib.sha256()
    for fn in kernel_sources(kernel):
        p = os.path.join(ROOT, "flowdenoising_amd", "csrc", fn)
        if os.path.exists(p):
            with open(p, "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


# Bytes one launch of each dominant kernel must move per pixel of the batch:
#   fused (3 iterations + warp in one launch): R0 20 + R1 20 + neighbour image 4 + accumulator 8 = 52, plus flow in 8 /
#       flow out 8 on the chain steps that have them ((K-3)/(K-1) of the launches each)
#   iter (one Farneback iteration per launch): R0 20 + R1 20 + flow in 8 + flow out 8 = 56; the last of three also warps: +12
#   update_flow (staged FarnebackUpdateFlow_Blur): M 20 + flow 8, plus on the refreshing launches R0 20 + R1 20 + flow 8 + M 20
def compulsory_bytes_per_px(name, K):
    if name == "fused":
        return 52.0 + 16.0 * (K - 3) / (K - 1)
    if name == "iter":
        return 56.0 + 12.0 / 3 - 8.0 / 3 * 2.0 / (K - 1)     # the first launch of each side's chain has no flow to read
    return (3 * 28 + 2 * 68) / 3.0


def roofline(timers, nvox, K, levels, run_cfg, sub_batches=1):
    """The dominant kernel against the HBM peak, priced on the bytes it has to move."""
    names = {"fused": "k_farneback_fused", "iter": "k_farneback_iter", "update_flow": "k_update_flow_scan"}
    best = None
    for name in names:
        ms, cnt = timers.get(name, (0.0, 0))
        if cnt > 0 and (best is None or ms > best[1]):
            best = (name, ms, cnt)
    if best is None:
        return None
    name, ms, cnt = best
    # with a pyramid a chain step is one launch (three for `iter`) per level; level k has 4^-k of the pixels
    px_per_launch = nvox * sum(0.25 ** k for k in range(levels + 1)) / (levels + 1)     # pixels x pairs of a launch
    avg_s = ms / cnt * 1e-3
    concurrent = None
    if sub_batches > 1 and timers.get("chains", (0, 0))[1]:
        # The batch's targets ran as sub-batches on two streams: a launch covers 1 / sub_batches of the pixels and the launches
        # of the two streams overlap, so their own durations add up to more than the wall clock.  Price the kernel on the span
        # from fork to join instead (FDN_TIMER_CHAINS: HIP events on the main stream; it also holds the centre-tap axpy of
        # each sub-batch and, with a pyramid, the flow-shrink kernels -- a few per cent, charged to the kernel here): one
        # "launch" below = the sub_batches concurrent launches of one chain step and level.
        concurrent = {"sub_batches": sub_batches, "own_launch_ms": round(ms / cnt, 4), "launches_on_both_streams": cnt,
                      "note": "launches of the two streams overlap: avg_launch_ms = (fork-to-join span of the chains) / (launches / sub_batches)"}
        cnt = cnt // sub_batches
        avg_s = timers["chains"][0] / cnt * 1e-3
        ms = timers["chains"][0]
    bpp = compulsory_bytes_per_px(name, K)
    achieved = bpp * px_per_launch / avg_s / 1e9
    unfused = {"fused": 300.0, "iter": 100.0, "update_flow": (3 * 28 + 2 * 68) / 3.0}[name]
    r = {"bound": "hbm", "kernel": names[name], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(achieved / HBM_PEAK_GBS, 4),
         "bytes_per_px": round(bpp, 2),
         "bytes_model": "compulsory HBM bytes of one launch (inputs read once, outputs written once; matrices and "
                        "intermediate flows that stay on chip are not charged)",
         "px_per_launch": int(px_per_launch), "avg_launch_ms": round(ms / cnt, 4), "launches": cnt,
         "unfused_algorithmic": {"bytes_per_px": round(unfused, 1), "GBps": round(unfused * px_per_launch / avg_s / 1e9, 1),
                                 "note": "SURVEY 8(d) stage list with every named intermediate through HBM; exceeds the "
                                         "peak exactly when fusion removed that traffic -- not an HBM fraction"},
         "traffic": None}
    if concurrent:
        r["concurrent_launches"] = concurrent
    t = committed_profile(names[name], run_cfg)
    r["traffic_source"] = t["note"]
    if t.get("bytes_per_px") is not None:
        traffic = t["bytes_per_px"] * px_per_launch
        r["traffic"] = round(traffic)
        r["traffic_frac"] = round(traffic / avg_s / 1e9 / HBM_PEAK_GBS, 4)
        r["overfetch"] = round(t["bytes_per_px"] / bpp, 3)
        if t.get("limiter"):
            r["limiter"] = t["limiter"]
    return r


def committed_profile(kernel, run_cfg):
    """PMC traffic of the dominant kernel from profiles/*_traffic.json (tools/profile_round.sh, tools/profile_w15.sh).
    The file whose kernel, kernel sources (sha) and workload all match this run is used, newest round first;
    otherwise traffic stays null and the reason is reported."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True)
    if not files:
        return {"note": "no committed PMC pass"}
    sha = kernel_source_hash(kernel)
    why = []
    for fn in files:
        with open(fn) as f:
            t = json.load(f)
        base = os.path.basename(fn)
        if not str(t.get("kernel", "")).startswith(kernel):
            why.append(f"{base}: kernel {t.get('kernel')}")
            continue
        bad = [key for key in ("shape", "winsize", "levels", "sigma", "axes") if t.get("workload", {}).get(key) != run_cfg.get(key)]
        if bad:
            why.append(f"{base}: another workload ({', '.join(bad)})")
            continue
        if t.get("kernel_source_sha") != sha:
            why.append(f"{base}: STALE (kernel sources {t.get('kernel_source_sha')} != {sha})")
            continue
        return {"bytes_per_px": t["bytes_per_pixel"], "limiter": t.get("limiter"),
                "note": f"source: committed profile {base} (commit {t.get('commit')}, `{t.get('command')}`): "
                        "(2 x FETCH_SIZE + WRITE_SIZE) per launch from separate rocprofv3 --pmc passes, not measured in this run"}
    if any("STALE" in w for w in why):
        print("bench.py: WARNING: the committed PMC pass of this kernel and workload was taken on other kernel sources: "
              "roofline.traffic left null; rerun tools/profile_round.sh", file=sys.stderr)
    return {"note": "no committed PMC pass matches this run -- " + "; ".join(why)}


def sweep_line(h, vol, shape, kernel, params, mean):
    """north_star states its 50 % target on the warped-Gaussian sweep.  In the product that sweep is fused into the
    Farneback kernel (its 12 B/px are inside `roofline`); as a kernel of its own it exists on the per-stage path
    (k_sweep_side: one launch folds the K//2 warped neighbours of one side into the accumulator, which stays in a
    register): time it there, on a Z pass over the first 256 slices."""
    import torch
    Z, Y, X = shape
    n = min(256, Z)
    r = kernel.size // 2
    out = torch.empty((n, Y, X), dtype=torch.float32, device=vol.device)
    h.set_option("path", 1)
    try:
        h.timers(reset=True)
        h.filter_axis_dev(vol.data_ptr(), out.data_ptr(), (n, Y, X), 0, kernel, mean, params)
        torch.cuda.synchronize()
        tm = h.timers(reset=True)
    finally:
        h.set_option("path", 0)
    ms, cnt = tm["warp"]
    if not cnt:
        return None
    px = n * Y * X
    avg_s = ms / cnt * 1e-3
    bpp = 12.0 * r + 8.0            # flow 8 + neighbour 4 per pair; accumulator read + written once per side
    return {"kernel": "k_sweep_side", "bound": "hbm", "bytes_per_px": bpp,
            "achieved": round(bpp * px / avg_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(bpp * px / avg_s / 1e9 / HBM_PEAK_GBS, 4),
            "avg_launch_ms": round(ms / cnt, 4), "launches": cnt,
            "note": f"one launch = the {r} warped neighbours of one side of every target slice ({n} slices of {Y}x{X}); bytes = "
                    "SURVEY 8(d)'s sweep model, which is also what the kernel moves; per-stage path (fdn_set_option path = 1)"}


def to_host(h, t):
    """A device tensor's values as a numpy array, copied by the library (fdn_memcpy_d2h: page-locked for the call or bounced,
    DESIGN.md 2) and not by `tensor.cpu()`: a pageable copy of this size is page-locked by the HIP runtime on the fly, the
    path the GPU memory fault of profiles/history/NOTES_r06.md section 2 was caught in."""
    import torch
    t = t.contiguous()
    torch.cuda.synchronize()
    arr = np.empty(tuple(t.shape), dtype=np.float32)
    h.d2h(arr, t.data_ptr())
    return arr


def check_output(h, vol, out, shape, kernels, params, mean):
    """After the timed region: redo the step pass by pass (so that each pass's input exists), require the same bits
    as the timed output, and recompute one target slice of every pass with the oracle from that pass's own input."""
    import torch
    from oracle import oracle as O
    O.build()
    cur = vol
    worst = 0.0
    exact = True
    exact_kernel_order = True
    passes = []
    for axis in range(3):
        k = kernels[axis]
        if k is None:
            continue
        nxt = torch.empty_like(vol)
        h.filter_axis_dev(cur.data_ptr(), nxt.data_ptr(), shape, axis, k, mean, params)
        torch.cuda.synchronize()
        n, r = shape[axis], k.size // 2
        t = n // 2 + 3
        lo, hi = max(0, t - r), min(n, t + r + 1)
        idx = [slice(None)] * 3
        idx[axis] = slice(lo, hi)
        sub = to_host(h, cur[tuple(idx)])
        idx[axis] = t
        got = to_host(h, nxt[tuple(idx)])
        nthr = min(16, len(os.sched_getaffinity(0)))      # the oracle spreads the slice's Farneback pairs over threads
        want = np.take(O.filter_axis_range(sub, axis, k, params.levels, params.winsize, mean, t - lo, t - lo + 1, nthreads=nthr),
                       t - lo, axis=axis)
        err = float(np.abs(got.astype(np.float64) - want).max() / max(float(np.abs(want).max()), 1e-30))
        worst = max(worst, err)
        same_bits = bool(np.array_equal(got, want))
        exact = exact and same_bits
        if not same_bits:       # not OpenCV's f64 summation order: then it must be the kernels' own order, bit for bit
            mode = 2 if params.winsize // 2 <= 4 else 4
            want_k = np.take(O.filter_axis_range(sub, axis, k, params.levels, params.winsize, mean, t - lo, t - lo + 1, nthreads=nthr,
                                                 box_mode=mode), t - lo, axis=axis)
            exact_kernel_order = exact_kernel_order and bool(np.array_equal(got, want_k))
        passes.append(f"{'ZYX'[axis]}[{t}]")
        if cur is not vol:
            del cur
        cur = nxt
    same = bool(torch.equal(cur, out))
    return {"timed_output_equals_pass_by_pass_rerun": same, "slices": passes, "max_rel_err": worst, "bit_equal": exact,
            "bit_equal_kernel_order": exact_kernel_order,
            "tolerance": 1e-4, "ok": bool(same and worst < 1e-4),
            "how": "one target slice per pass recomputed by the CPU oracle from the GPU's input of that pass"}


def cpu_baseline(h, vol_t, shape, kernel, mean, n_targets, levels, winsize):
    """Time the oracle's Z pass on `n_targets` target slices taken from the middle of the same
    volume, one chunk of slices per core (par:181-206), and scale to the full three-pass job."""
    from oracle import oracle as O
    O.build()
    # the GPU box gives one GPU a share of 16 host cores; FDN_BENCH_CORES overrides
    cap = int(os.environ.get("FDN_BENCH_CORES", "16"))
    cores = max(1, min(len(os.sched_getaffinity(0)), O.max_threads(), cap))
    if n_targets <= 0:
        n_targets = 4 * cores      # about 10 s of wall time on the GPU box's 16 cores
    Z, Y, X = shape
    r = kernel.size // 2
    n_targets = min(n_targets, Z)
    z0 = max(0, Z // 2 - n_targets // 2 - r)
    z1 = min(Z, z0 + n_targets + 2 * r)
    slab = to_host(h, vol_t[z0:z1])
    s0 = min(r, slab.shape[0] - n_targets)
    t0 = time.perf_counter()
    O.filter_axis_range(slab, 0, kernel, levels, winsize, mean, s0, s0 + n_targets, nthreads=cores)
    dt = time.perf_counter() - t0
    vox = n_targets * Y * X
    per_axis = vox / dt / 1e6
    return {"value": round(per_axis / 3.0, 4), "unit": "Mvoxels/s", "cores": cores, "kind": "port",
            "sample": f"oracle Z pass on {n_targets} target slices ({Y}x{X}, {kernel.size - 1} Farneback pairs each) of the same "
                      f"volume in {dt:.1f} s with {cores} threads; per-axis rate {per_axis:.3f} Mvox/s divided by 3 "
                      "for the Z+Y+X job (same pixel-pair count per axis).  A C port: it leaves out what the reference's "
                      "Python adds per pair (the numpy grid build of seq:53-55, 42 ms per 1024x1024 call), so the "
                      "reference itself would be slower than this"}


class GpuSampler:
    """Shader clock and socket power of one GPU while the timed region runs: a side thread reads librocm_smi64 (the library
    behind `rocm-smi --showclocks --showpower`: sysfs reads, nothing on the GPU's queues) ten times a second.  The fused
    kernel runs at the clock the socket's power cap leaves it, and boxes differ by 2-3 %: with these fields
    a slower line can be read as a slower clock or as slower code.  Any failure leaves the fields null with the reason."""

    def __init__(self, pci_id, hz=10.0):
        import ctypes
        import threading
        self.period = 1.0 / hz
        self.clk, self.pw, self.cap_w, self.error = [], [], None, None
        self._stop = threading.Event()
        self._thread = None
        try:
            self._lib = lib = ctypes.CDLL("librocm_smi64.so")
            if lib.rsmi_init(ctypes.c_uint64(0)) != 0:
                raise OSError("rsmi_init failed")
            n = ctypes.c_uint32()
            lib.rsmi_num_monitor_devices(ctypes.byref(n))
            want = None
            try:                                        # "0000:05:00.0" -> rocm_smi's BDFID
                dom, bus, rest = pci_id.split(":")
                devn, fn = rest.split(".")
                want = (int(dom, 16) << 32) | (int(bus, 16) << 8) | (int(devn, 16) << 3) | int(fn, 16)
            except (ValueError, AttributeError):
                pass
            self._dv = None
            for i in range(n.value):
                b = ctypes.c_uint64()
                if lib.rsmi_dev_pci_id_get(ctypes.c_uint32(i), ctypes.byref(b)) == 0 and (want is None or (b.value & 0xffffffffffff) == (want & 0xffffffffffff)):
                    self._dv = i
                    break
            if self._dv is None:
                raise OSError(f"no rocm_smi device with PCI id {pci_id}")

            class Freqs(ctypes.Structure):
                _fields_ = [("has_deep_sleep", ctypes.c_bool), ("num_supported", ctypes.c_uint32), ("current", ctypes.c_uint32),
                            ("frequency", ctypes.c_uint64 * 33)]
            self._Freqs = Freqs
            cap = ctypes.c_uint64()
            if lib.rsmi_dev_power_cap_get(ctypes.c_uint32(self._dv), ctypes.c_uint32(0), ctypes.byref(cap)) == 0:
                self.cap_w = cap.value / 1e6
            self._thread = threading.Thread(target=self._run, daemon=True)
        except (OSError, AttributeError) as e:
            self.error = f"{type(e).__name__}: {e}"

    def _sample(self):
        import ctypes
        f = self._Freqs()
        if self._lib.rsmi_dev_gpu_clk_freq_get(ctypes.c_uint32(self._dv), ctypes.c_int(0), ctypes.byref(f)) == 0 and f.current < 33:
            self.clk.append(f.frequency[f.current] / 1e6)
        pw, kind = ctypes.c_uint64(), ctypes.c_int()
        if self._lib.rsmi_dev_power_get(ctypes.c_uint32(self._dv), ctypes.byref(pw), ctypes.byref(kind)) == 0:
            self.pw.append(pw.value / 1e6)

    def _run(self):
        while not self._stop.is_set():
            self._sample()
            self._stop.wait(self.period)

    def start(self):
        if self._thread is not None:
            self._thread.start()

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join()

    def fields(self):
        def stat(v):
            return None if not v else {"mean": round(sum(v) / len(v), 1), "min": round(min(v), 1), "max": round(max(v), 1), "samples": len(v)}
        c, p = stat(self.clk), stat(self.pw)
        out = {"gpu_clock_mhz_mean": c and c["mean"], "power_w_mean": p and p["mean"], "power_cap_w": self.cap_w,
               "gpu_clock_mhz": c, "power_w": p,
               "sampling": "rank 0's GPU, librocm_smi64 (rsmi_dev_gpu_clk_freq_get SYS / rsmi_dev_power_get) every "
                           f"{self.period * 1e3:.0f} ms in a side thread over the timed region"}
        if self.error:
            out["sampling_error"] = self.error
        return out


def relay_json(line):
    """Only the JSON line goes to stdout (gloo, the rehearsal backend, prints its connection banner there too)."""
    out = sys.stdout if line.lstrip().startswith("{") else sys.stderr
    out.write(line)
    out.flush()


def run_torch_distributed(a, extra_env=None):
    """N ranks of the torch.distributed slab engine (--engine python) as children of this process: torch.distributed.run."""
    port = int(os.environ.get("MASTER_PORT", 0)) or 29500 + os.getpid() % 2000
    args = [v for v in sys.argv[1:] if v not in ("--engine", "native", "python")] + ["--engine", "python"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + args
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "2")
    env.update(extra_env or {})
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    for line in proc.stdout:       # stderr goes straight through
        relay_json(line)
    return proc.wait()


def self_launch(a):
    """`python bench.py --gpus N` from a bare shell (how the driver calls it): start the N ranks as CHILD processes --
    before anything here imports torch or touches the GPU, and never by exec --, relay rank 0's JSON line and return
    the first non-zero exit code.  native engine: plain subprocess.Popen ranks that meet through libflowdn_rccl.so
    (flowdenoising_amd/launch.py); python engine: torch.distributed.run, as that engine needs a process group.
    If the native job fails, this parent -- which has never touched a GPU -- starts N FRESH children once more on the
    torch.distributed engine, and the line they print says so (`transport_fallback`)."""
    if a.engine != "native":
        return run_torch_distributed(a)
    from flowdenoising_amd import launch
    errors = []
    rc = launch.spawn([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], a.gpus, relay=relay_json, errors=errors)
    if rc == 0 or os.environ.get("FDN_BENCH_NO_FALLBACK") == "1":
        return rc
    why = f"native job failed (exit code {rc}" + ("; " + " | ".join(errors) if errors else "") + ")"
    print(f"bench.py: WARNING: {why}; starting {a.gpus} fresh ranks on the torch.distributed slab engine", file=sys.stderr)
    return run_torch_distributed(a, {"FDN_BENCH_FALLBACK": why[:1500]})


def workload_names(shape, sig3, axes, levels, winsize):
    """(metric, workload tag): BASELINE.json's own words for the configuration it quotes the metric on, and the index of
    the BASELINE config this run is, if it is one of them."""
    Z, Y, X = shape
    cfg = None
    table = {1: ((256, 512, 512), [2.0] * 3, "z", 0, 5), 2: ((512, 1024, 1024), [2.0] * 3, "zyx", 0, 5),
             3: ((1024, 1024, 1024), [4.0] * 3, "zyx", 0, 5), 4: ((512, 2048, 2048), [2.0, 2.0, 4.0], "zyx", 3, 15),
             0: ((64, 128, 128), [2.0] * 3, "zyx", 0, 5)}
    for i, (sh, sg, ax, lv, ws) in table.items():
        used = [sg[k] for k, c in enumerate("zyx") if c in ax]
        mine = [sig3[k] for k, c in enumerate("zyx") if c in axes]
        if tuple(shape) == sh and axes == ax and mine == used and levels == lv and winsize == ws:
            cfg = i
    sg = format(sig3[0], "g") if len(set(sig3)) == 1 else ",".join(format(v, "g") for v in sig3)
    if cfg == 2:
        metric = "Mvoxels/s denoised (sigma=2) on 1024x1024x512 float32"          # BASELINE.json "metric"
    else:
        metric = f"Mvoxels/s denoised (sigma={sg}) on {X}x{Y}x{Z} float32"
    return metric, (f"BASELINE.json configs[{cfg}]" if cfg is not None else "not one of BASELINE.json's configs")


def check_sharded(h, move, dev, rank, world, shape, kernels, params, amplitude, slab_out, levels):
    """After the timed region of an N > 1 run: every rank sends its output slab to rank 0, which regenerates the whole
    synthetic volume, filters it on its own GPU alone (fdn_filter_3d_dev), requires the gathered sharded output to equal
    that single-GPU output bit for bit, and has the oracle recompute one target slice per pass (check_output).
    move: (send_to_0(tensor), recv_from(rank, tensor), barrier) of the engine in use.  Returns the block on rank 0."""
    import torch
    from flowdenoising_amd import distributed as fd, synth
    send_to_0, recv_from, barrier = move
    Z, Y, X = shape
    parts = fd.split(Z, world)
    torch.cuda.synchronize()
    if rank != 0:
        send_to_0(slab_out)
        torch.cuda.synchronize()
        barrier()                         # rank 0's single-GPU rerun and oracle check happen behind this barrier
        return None
    full = torch.empty(shape, dtype=torch.float32, device=dev)
    full[parts[0][0]:parts[0][1]].copy_(slab_out)
    recv_from([(r, full[z0:z1]) for r, (z0, z1) in enumerate(parts) if r != 0])
    torch.cuda.synchronize()
    # the volume the ranks filtered: the generator seeds a slab's noise by its first slice, so the whole is their concatenation
    vol = torch.cat([synth.make_volume(shape, seed=1234 + 3, amplitude=amplitude, xp=torch, device=dev, z0=z0, zlen=z1 - z0) for z0, z1 in parts])
    single = torch.empty_like(vol)
    torch.cuda.synchronize()
    mean = h.mean_dev(vol.data_ptr(), vol.numel())
    h.filter_3d_dev(vol.data_ptr(), single.data_ptr(), shape, kernels, mean, params)
    torch.cuda.synchronize()
    same = bool(torch.equal(full, single))
    del full
    res = check_output(h, vol, single, shape, kernels, params, mean)
    res["sharded_output_equals_single_gpu_rerun"] = same
    res["ok"] = bool(res["ok"] and same)
    res["how"] = (f"the {world} ranks' output slabs gathered on rank 0 and compared with a single-GPU fdn_filter_3d_dev of the same "
                  "synthetic volume (bit for bit); that single-GPU output then checked as at N = 1: " + res["how"])
    barrier()
    return res


def main():
    a = parse()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # before torch or any HIP library is loaded: both launch routes alike
    from flowdenoising_amd import launch
    job = launch.job()
    if a.gpus > 1 and job is None:
        sys.exit(self_launch(a))
    fallback = os.environ.get("FDN_BENCH_FALLBACK") or None
    if job is not None and a.engine == "native" and launch.started_by_torchrun():
        # A rank torch.distributed.run started (the round driver's N > 1 launch).  It has not touched a GPU: the native
        # job's rank runs in ONE child of this process; if the native job fails anywhere, every rank gets here with a
        # reason, still fresh, and carries the run on the torch.distributed engine instead (said so in the line).
        rc, why = launch.supervise_rank([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], job, relay=relay_json)
        if rc == 0:
            if job[0] == 0:
                launch.remove_derived_rendezvous(job[3])
            sys.exit(0)
        if os.environ.get("FDN_BENCH_NO_FALLBACK") == "1":
            sys.exit(rc or 1)
        fallback = ("native job failed (" + " | ".join(why) + ")")[:1500]
        if job[0] == 0:
            print(f"bench.py: WARNING: {fallback}; the ranks go on with the torch.distributed slab engine", file=sys.stderr)
        a.engine = "python"
    try:
        run(a, job, fallback)
    except BaseException as e:      # noqa: BLE001 -- a rank that cannot go on says why, tells the others, and leaves at once
        if isinstance(e, SystemExit) and not e.code:
            raise
        if job is None:
            raise
        import signal
        import traceback
        traceback.print_exc()
        launch.report_failure(job[3], job[0], f"{type(e).__name__}: {e}")
        # leave in an orderly way -- transport, handle, then the interpreter's own teardown (torch's context) -- but under an
        # alarm: a communicator whose peers are gone may never come back from its teardown, and then the default action of
        # SIGALRM ends the process
        signal.alarm(20)
        tr = _live.pop("tr", None)
        if tr is not None:
            tr.abort()
            tr.close()
        h = _live.pop("h", None)
        if h is not None:
            h.close()
        sys.stdout.flush()
        sys.stderr.flush()
        sys.exit(1)


_live = {}                          # the rank's transport, for the failure path above


def run(a, job, fallback):
    import torch
    from flowdenoising_amd import _lib, launch, synth

    rank, world, local_rank, rdv = job if job is not None else (0, 1, 0, None)
    if world != a.gpus:          # started by torch.distributed.run with another rank count: the environment decides
        a.gpus = world
    ngpu = torch.cuda.device_count()
    tr = dist = None
    if world > 1 and a.engine == "native":
        tr, device = launch.make_transport(rank, world, local_rank, rdv)      # RCCL, or shared memory when ranks share a GPU
        _live["tr"] = tr
        if os.environ.get("FDN_TEST_FAIL_NATIVE") == str(rank):               # tests: this native rank gives up here
            raise RuntimeError("FDN_TEST_FAIL_NATIVE: injected failure of the native engine")
    else:
        device = local_rank % max(ngpu, 1)
    torch.cuda.set_device(device)
    dev = torch.device("cuda", device)
    devices = None
    if tr is not None:
        n_seen, devices = tr.count(), tr.devices()
    elif world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # which GPUs the ranks really sit on decides the backend: gloo + host staging when they share one (a rehearsal)
        sub = os.path.join(rdv, "py_engine")
        os.makedirs(sub, mode=0o700, exist_ok=True)
        devices = launch._exchange_lines(sub, rank, world, _lib.device_pci_id(device))
        limit = datetime.timedelta(seconds=float(os.environ.get("FDN_RDV_TIMEOUT", "300")))
        if len(set(devices)) < world:
            dist.init_process_group("gloo", timeout=limit)
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=limit)
        n_seen = dist.get_world_size()
    else:
        n_seen = 1
    rehearsal = world > 1 and len(set(devices)) < world     # ranks share GPUs (a one-GPU box): every line of the N > 1 path runs,
                                                            # exchanges go through host memory, and no number of it means anything

    shape = tuple(int(v) for v in a.shape.split(","))
    Z, Y, X = shape
    sig3 = [float(v) for v in a.sigmas.split(",")] if a.sigmas else [a.sigma] * 3
    if len(sig3) != 3:
        sys.exit("--sigmas takes three values: Z,Y,X")
    ks3 = [_lib.gaussian_kernel(v) for v in sig3]
    first_axis = next((i for i, c in enumerate("zyx") if c in a.axes), 0)
    kernel = ks3[first_axis]         # the auxiliary lines (roofline byte model, sweep kernel, CPU sample) use the first pass's taps
    kernels = [ks3[i] if c in a.axes else None for i, c in enumerate("zyx")]
    params = _lib.SweepParams(a.levels, a.winsize, 3, 5, 1.2, _lib.BORDER_MEAN_PAD, 1, 1)
    if a.integer == "par":
        params.border_mode, params.warp_mode, params.round_lo, params.round_hi = _lib.BORDER_WRAP, _lib.WARP_ROUND_INT, -32768.0, 32767.0

    h = _lib.Handle(device)
    _live["h"] = h
    if tr is None:
        h.set_stream(torch.cuda.current_stream().cuda_stream)
    # (the native N > 1 engine keeps the handle's own non-blocking stream, as the CLI's ranks and the RCCL tests do: the
    #  exchanges then never sit on the legacy default stream; torch only generates the input, with a device-wide
    #  synchronisation on either side of the timed region)
    if a.path:
        h.set_option("path", a.path)
    if a.sub_batches:
        h.set_option("sub_batches", a.sub_batches)

    eng = None
    out_slab = None
    if world == 1:
        vol = synth.make_volume(shape, seed=1234 + 3, amplitude=a.amplitude, xp=torch, device=dev)
        out = torch.empty_like(vol)

        if a.integer:
            vol = torch.round(vol)                               # integer values, as an int16 MRC would hold

        def step():
            mean = h.mean_dev(vol.data_ptr(), vol.numel())       # seq:420
            if a.integer == "seq":
                params.warp_mode, params.pad64 = _lib.WARP_F64_PADDED, float(mean)
            h.filter_3d_dev(vol.data_ptr(), out.data_ptr(), shape, kernels, mean, params)
            return mean
        parallelism = "1 GPU"
        backend = "none"
    else:
        from flowdenoising_amd import distributed as fd
        plan = fd.SlabPlan(shape, world, rank)
        vol = synth.make_volume(shape, seed=1234 + 3, amplitude=a.amplitude, xp=torch, device=dev,
                                z0=plan.z0, zlen=plan.zlen)
        out = None
        if a.engine == "native":
            out_slab = torch.empty_like(vol)

            def step():
                h.filter_3d_sharded(vol.data_ptr(), out_slab.data_ptr(), shape, kernels, params, tr)
                return None
            backend = tr.describe().split(",")[0]
            parallelism = (f"{world} Z-slabs, one exchange per pass (halos + repartition): fdn_filter_3d_sharded below the C ABI, "
                           "point-to-point over RCCL (ncclSend / ncclRecv in one group per exchange, libflowdn_rccl.so)")
        else:
            eng = fd.SlabEngine(plan, h, dist)

            def step():
                _live["py_out"] = eng.filter_3d(vol, kernels, params)      # a view of an engine-owned buffer
                return None
            backend = dist.get_backend()
            parallelism = f"{world} Z-slabs, one exchange per pass (halos + repartition), torch.distributed point-to-point ({backend}); slab engine above the C ABI"
        if rehearsal:
            parallelism = (f"REHEARSAL: {world} ranks on {len(set(devices))} distinct GPU(s) ({', '.join(sorted(set(devices)))}), exchanges "
                           "staged through host memory (not a measurement)")

    def barrier():
        if tr is not None:
            tr.barrier()
        elif dist is not None:
            dist.barrier()

    def gather_f64(values):
        """(world, n) array of every rank's float64 values."""
        arr = np.asarray(values, dtype=np.float64)
        if tr is not None:
            return tr.allgather_array(arr)
        if dist is not None:
            t = torch.tensor(arr, dtype=torch.float64, device="cpu" if rehearsal else dev)
            allr = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(allr, t)
            return np.stack([r.cpu().numpy() for r in allr])
        return arr[None]

    torch.cuda.synchronize()          # the generator's kernels (torch's stream) before the library's own stream reads the volume
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if not a.no_timers:
        h.enable_timers(True)
        h.timers(reset=True)
        if eng is not None:
            eng.reset_phase_times()
    sampler = None
    if rank == 0 and not a.no_sampling:
        try:
            sampler = GpuSampler(_lib.device_pci_id(device))
        except Exception as e:      # noqa: BLE001 -- the sampler is a side line: it never stops the measurement
            sampler = None
            print(f"bench.py: clock / power sampling unavailable: {e}", file=sys.stderr)
    barrier()
    torch.cuda.synchronize()
    if sampler is not None:
        sampler.start()
    t0 = time.perf_counter()
    mean = None
    for _ in range(a.steps):
        mean = step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if sampler is not None:
        sampler.stop()
    dt = float(gather_f64([dt]).max())                    # the MAX over ranks
    timers = h.timers() if not a.no_timers else {}
    sub_batches = h.get_option("last_sub_batches")       # what the last pass of the timed steps ran with (automatic on small grids)
    phases = None
    if world > 1 and not a.no_timers:
        if eng is not None:
            mine = eng.phase_times()          # ms per category on this rank, over the timed steps
        else:                                 # the library's own timer table (HIP events on the stream the work runs on)
            # (with sub-batches the launches of the two streams overlap: the fork-to-join span of the chains is what they took)
            chain = timers["chains"][0] if timers["chains"][1] else timers["fused"][0] + timers["iter"][0]
            kern = sum(timers[k][0] for k in ("polyexp", "update_matrices", "update_flow", "warp")) + chain
            mine = {"compute": kern, "pack_unpack_permute": timers["permute"][0], "exchange": timers["collective"][0], "mean": timers["mean"][0]}
        names = sorted(mine)
        allr = gather_f64([mine[n] for n in names])
        phases = {n: [round(float(allr[r][i]) / a.steps, 2) for r in range(world)] for i, n in enumerate(names)}

    nvox = Z * Y * X
    res = None
    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = nvox / (dt / a.steps) / 1e6
        metric, tag = workload_names(shape, sig3, a.axes, a.levels, a.winsize)
        res = {
            "metric": metric,
            "value": round(value, 3), "unit": "Mvoxels/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "n_ranks_seen": n_seen,
            "n_ranks_seen_source": ("ncclCommCount of the live communicator" if tr is not None and tr.kind == "rccl" else
                                    "distinct ranks the shared-memory transport heard from" if tr is not None else
                                    "torch.distributed.get_world_size()" if dist is not None else "single process"),
            "backend": backend,
            "config": {"workload": f"{X}x{Y}x{Z} float32, sigma={a.sigmas or format(a.sigma, 'g')} (K={','.join(str(k.size) for k in ks3)}), levels={a.levels}, winsize={a.winsize}, "
                                   f"OF along {a.axes.upper()}, mean-padded borders ({tag})",
                       "axes": a.axes, "parallelism": parallelism, "amplitude": a.amplitude,
                       "amplitude_note": "BASELINE.md's unit-range generator scaled by 100: OpenCV's absolute +1e-3 regulariser zeroes "
                                         "every flow on unit-range data.  Throughput depends on it a little: larger flows leave the "
                                         "kernel's LDS window more often"},
        }
        if sampler is not None:
            res.update(sampler.fields())
        if tr is not None:
            res["transport"] = tr.describe()
        if devices is not None:
            res["devices"] = devices         # hipDeviceGetPCIBusId of every rank, all-gathered: N distinct strings = N distinct GPUs
            res["distinct_devices"] = len(set(devices))
        if fallback:
            res["transport_fallback"] = fallback
        # whole path against the SURVEY 8(d) stage list (4832 B/voxel/axis at sigma=2): what an UNFUSED implementation
        # would have to move; above the HBM peak it measures traffic removed by fusion, not bandwidth
        path_bytes = sum(24 + (k.size - 1) * 300 + 8 for k in kernels if k is not None)
        res["whole_path"] = {"unfused_algorithmic_GBps": round(path_bytes * nvox / (dt / a.steps) / 1e9 / world, 1),
                             "ratio_to_hbm_peak_per_gpu": round(path_bytes * nvox / (dt / a.steps) / 1e9 / world / HBM_PEAK_GBS, 4),
                             "note": "SURVEY 8(d) bytes of the unfused stage list / wall time; a ratio above 1 = traffic that fusion removed"}
        if timers:
            run_cfg = {"shape": list(shape), "winsize": a.winsize, "levels": a.levels, "sigma": a.sigma if not a.sigmas else a.sigmas, "axes": a.axes}
            res["roofline"] = roofline(timers, nvox // world, kernel.size, a.levels, run_cfg, sub_batches)
            res["kernel_ms_per_step"] = {k: round(v[0] / a.steps, 2) for k, v in timers.items() if v[1]}
            res["sub_batches"] = sub_batches
        if phases:
            res["phase_ms_per_step_per_rank"] = phases
        if a.integer:
            res["config"]["workload"] += f"; INTEGER-VOLUME SEMANTICS ({a.integer}) -- not the headline configuration"
    if not a.no_check and not a.integer:
        h.enable_timers(False)
        h.set_workspace_limit(0)      # gives the handle's buffers back (a limit of 0 is "none"): the check allocates volumes of its own,
                                      # and on configs[4] the two do not fit side by side
        if world == 1:
            res["checked"] = check_output(h, vol, out, shape, kernels, params, mean)
            if a.levels == 0:
                h.enable_timers(True)
                res["sweep"] = sweep_line(h, vol, shape, kernel, params, mean)
        else:
            if tr is not None:
                move = (lambda t: tr.exchange([(t.data_ptr(), t.numel() * 4, 0, True)], 0),
                        lambda lst: tr.exchange([(t.data_ptr(), t.numel() * 4, r, False) for r, t in lst], 0), barrier)
            else:
                host = dist.get_backend() == "gloo"

                def send_to_0(t):
                    dist.send(t.cpu() if host else t, 0)

                def recv_from(lst):
                    for r, t in lst:
                        if host:
                            buf = torch.empty(t.shape, dtype=t.dtype)
                            dist.recv(buf, r)
                            t.copy_(buf)
                        else:
                            dist.recv(t, r)
                move = (send_to_0, recv_from, barrier)
                if out_slab is None:
                    out_slab = _live["py_out"].contiguous()
            chk = check_sharded(h, move, dev, rank, world, shape, kernels, params, a.amplitude, out_slab, a.levels)
            if rank == 0:
                res["checked"] = chk
    h.enable_timers(False)
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(h, vol, shape, kernel, mean, a.cpu_targets, a.levels, a.winsize)
        elif world > 1:
            res["cpu_baseline"] = None
            res["cpu_baseline_see"] = "the N = 1 line: the CPU sample is timed on rank 0 at N = 1 only"
        print(json.dumps(res), flush=True)
    barrier()
    if dist is not None:
        dist.destroy_process_group()
    if tr is not None:
        tr.close()
        _live.pop("tr", None)
    if rank == 0 and rdv is not None:
        launch.remove_derived_rendezvous(rdv)       # a directory derived under torch.distributed.run is ours to remove


if __name__ == "__main__":
    main()
