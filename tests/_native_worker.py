"""Worker of the native multi-GPU tests (not a test module): one rank of fdn_filter_3d_sharded on the transports of
libflowdn_rccl.so -- RCCL when every rank has a GPU (and always for the world-size-1 loopback run), the shared-memory
rehearsal transport when ranks share one.  No torch anywhere in this process (asserted at the end)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    vol_path, out_path, sig, border, levels, winsize = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    loopback = len(sys.argv) > 7 and sys.argv[7] == "loopback"
    with_torch = os.environ.get("FDN_TEST_IMPORT_TORCH") == "1"
    if with_torch:                        # bench.py's situation: torch (its bundled HIP runtime and librccl) is in the process first
        import torch  # noqa: F401
    from flowdenoising_amd import _lib, launch
    from flowdenoising_amd.distributed import split
    job = launch.job()
    if job is None:                       # one rank, one real RCCL communicator
        import tempfile
        rdv = tempfile.mkdtemp(prefix="fdn_rdv_")
        rank, world, local = 0, 1, 0
        tr, device = launch.make_transport(0, 1, 0, rdv, kind="rccl")
    else:
        rank, world, local, rdv = job
        tr, device = launch.make_transport(rank, world, local, rdv)
    vol = np.load(vol_path, mmap_mode="r")
    z0, z1 = split(vol.shape[0], world)[rank]
    slab = np.ascontiguousarray(vol[z0:z1], dtype=np.float32)
    h = _lib.Handle(device)
    if loopback:
        h.set_option("shard_loopback", 1)
    kernels = [None if s == "-" else _lib.gaussian_kernel(float(s)) for s in sig.split(",")]
    params = _lib.SweepParams(levels, winsize, 3, 5, 1.2, border, 1, 1)
    d_in, d_out = h.malloc(slab.nbytes), h.malloc(slab.nbytes)
    h.h2d(d_in, slab)
    out = np.empty_like(slab)
    for _ in range(2):                    # a second step reuses every buffer of the first
        h.filter_3d_sharded(d_in, d_out, vol.shape, kernels, params, tr)
    h.d2h(out, d_out)
    np.save(f"{out_path}.{rank}.npy", out)
    # the transport's own entry points on device memory: a ring of messages (to self with one rank), the host all-gather
    a = (np.arange(1 << 14, dtype=np.float32) + 1000.0 * rank)
    d_a, d_b = h.malloc(a.nbytes), h.malloc(a.nbytes)
    h.h2d(d_a, a)
    h.set_stream(0)                       # the legacy default stream: the copies below order behind the exchange on it
    tr.exchange([(d_b, a.nbytes, (rank - 1) % world, False), (d_a, a.nbytes, (rank + 1) % world, True)], 0)
    got = np.empty_like(a)
    h.d2h(got, d_b)
    assert np.array_equal(got, np.arange(1 << 14, dtype=np.float32) + 1000.0 * ((rank - 1) % world)), got[:4]
    assert tr.allgather_host(bytes([rank]) * 3) == b"".join(bytes([r]) * 3 for r in range(world))
    tr.barrier()
    if rank == 0:
        print("transport:", tr.describe(), flush=True)
    for p in (d_a, d_b, d_in, d_out):
        h.free(p)
    tr.close()
    h.close()
    assert with_torch or "torch" not in sys.modules
    if job is None:
        import shutil
        shutil.rmtree(rdv, ignore_errors=True)


if __name__ == "__main__":
    main()
