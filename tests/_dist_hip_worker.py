"""Worker of tests/test_gpu_full.py::test_slab_engine_on_hip_backend_multi_rank (not a test module).
One rank of a world-size-N SlabEngine run on the HIP backend.  With at least N GPUs every rank takes its own
GPU and the exchange runs over RCCL ("nccl"); on a one-GPU box the ranks share GPU 0 and the exchange is staged
through the host over gloo -- same engine code, same pack / unpack kernels, only the transport differs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    vol_path, out_path, sig, border, levels, winsize = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    loopback = len(sys.argv) > 7 and sys.argv[7] == "loopback"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist
    from flowdenoising_amd import _lib
    from flowdenoising_amd.distributed import SlabEngine, SlabPlan
    ngpu = torch.cuda.device_count()
    local = rank if ngpu >= world else 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if ngpu >= world:
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    vol = np.load(vol_path, mmap_mode="r")
    plan = SlabPlan(vol.shape, world, rank)
    slab = torch.from_numpy(np.ascontiguousarray(vol[plan.z0:plan.z0 + plan.zlen])).to(dev)
    h = _lib.Handle(local)
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    kernels = [None if s == "-" else _lib.gaussian_kernel(float(s)) for s in sig.split(",")]
    params = _lib.SweepParams(levels, winsize, 3, 5, 1.2, border, 1, 1)
    eng = SlabEngine(plan, h, dist, loopback=loopback)
    eng.filter_3d(slab, kernels, params)                      # a first step, so that the second reuses every buffer
    out = eng.filter_3d(slab, kernels, params).cpu().numpy()
    mean = eng.global_mean(slab)
    full = eng.gather_z_slabs(torch.from_numpy(out).to(dev), 0)
    np.save(f"{out_path}.{rank}.npy", out)
    if rank == 0:
        np.save(f"{out_path}.gathered.npy", full.cpu().numpy())
    # the same job through the C-level entry point (fdn_filter_3d_sharded): schedule, packing and mean in libflowdn.so,
    # the transport behind the two fdn_comm callbacks (TorchComm: RCCL, or gloo staged through the host)
    from flowdenoising_amd.distributed import TorchComm
    res = torch.empty_like(slab)
    torch.cuda.synchronize()
    h.filter_3d_sharded(slab.data_ptr(), res.data_ptr(), vol.shape, kernels, params, TorchComm(dist, dev))
    h.synchronize()
    np.save(f"{out_path}.c.{rank}.npy", res.cpu().numpy())
    if loopback:
        # TorchComm's own plumbing on the real backend: raw device pointers wrapped as tensors, the handle's stream, one
        # batched group with a send to self and its receive (RCCL with one rank), the host all-gather
        comm = TorchComm(dist, dev)
        a = torch.arange(1 << 16, dtype=torch.float32, device=dev) * 0.5
        bt = torch.zeros_like(a)
        torch.cuda.synchronize()
        comm.exchange([(bt.data_ptr(), bt.numel() * 4, rank, False), (a.data_ptr(), a.numel() * 4, rank, True)], torch.cuda.current_stream().cuda_stream)
        assert torch.equal(a, bt)
        blob = bytes(range(200)) * 3
        assert comm.allgather_host(blob) == blob * world
    if rank == 0:
        np.save(f"{out_path}.mean.npy", np.float32(mean))
        print("backend", dist.get_backend(), "world", dist.get_world_size(), "phases", {k: round(v, 2) for k, v in eng.phase_times().items()}, flush=True)
    dist.barrier()
    dist.destroy_process_group()
    h.close()


if __name__ == "__main__":
    main()
