"""Helper of the GPU tests (not a test module): the N ranks of fdn_filter_3d_sharded as THREADS of one process.

The GPU boxes of this pipeline allow six processes on the card, so a world of 7 or 8 cannot be rehearsed with one process
per rank.  Here every rank is a thread with its own fdn_handle (own stream, own scratch), its own shared-memory transport
(FDN_TRANSPORT_SHM: the rehearsal transport of ranks that share a GPU) and its own slab -- ctypes drops the GIL around
every call into libflowdn.so / libflowdn_rccl.so, the libraries' error strings are thread-local, and the transport's
control block is the same shared mapping whether its ranks are processes or threads.  What runs is the C engine's
schedule for `world` ranks: plan, packing, exchanges, exact mean, passes."""
import os
import shutil
import sys
import tempfile
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run(vol, kernels, params, world, device=0, steps=2):
    """fdn_filter_3d_sharded of the float32 volume `vol` on `world` rank threads; returns the output slabs concatenated
    and rank 0's transport description.  `params`: a _lib.SweepParams (integer semantics already applied by the caller)."""
    from flowdenoising_amd import _lib
    from flowdenoising_amd.distributed import split
    vol = np.ascontiguousarray(vol, dtype=np.float32)
    parts = split(vol.shape[0], world)
    rdv = tempfile.mkdtemp(prefix="fdn_thr_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    outs, errors, info = [None] * world, [], {}
    os.environ.setdefault("FDN_RDV_TIMEOUT", "120")

    def rank_main(r):
        tr = h = None
        try:
            tr = _lib.Transport("shm", r, world, device, rdv)
            h = _lib.Handle(device)
            z0, z1 = parts[r]
            slab = np.ascontiguousarray(vol[z0:z1])
            d_in, d_out = h.malloc(slab.nbytes), h.malloc(slab.nbytes)
            h.h2d(d_in, slab)
            for _ in range(steps):                # a second step reuses every buffer of the first
                h.filter_3d_sharded(d_in, d_out, vol.shape, kernels, params.copy(), tr)
            out = np.empty_like(slab)
            h.d2h(out, d_out)
            outs[r] = out
            if r == 0:
                info["describe"], info["count"] = tr.describe(), tr.count()
            devs = tr.devices()
            if r == 0:
                info["devices"] = devs
            tr.barrier()
            h.free(d_in)
            h.free(d_out)
        except BaseException as e:                # noqa: BLE001 -- reported by the caller; the other ranks must stop waiting
            errors.append((r, e))
            if tr is not None:
                tr.abort()
        finally:
            if tr is not None:
                tr.close()
            if h is not None:
                h.close()

    threads = [threading.Thread(target=rank_main, args=(r,), name=f"rank{r}") for r in range(world)]
    try:
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        shutil.rmtree(rdv, ignore_errors=True)
    if errors:
        r, e = sorted(errors, key=lambda x: "another rank failed" in str(x[1]))[0]
        raise RuntimeError(f"rank thread {r} of {world}: {type(e).__name__}: {e}") from e
    return np.concatenate(outs), info
