"""Worker of tests/test_transport_cpu.py (not a test module): one rank of a shared-memory transport job whose "device"
buffers are host memory (device -1), so the message matching, the all-gather and the barriers of libflowdn_rccl.so run
without a GPU.  Every rank sends rank j a block filled with 1000 * me + j (sizes differ per pair), twice in a row (the
outboxes are reused), receives the others' blocks and checks them."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from flowdenoising_amd import _lib, launch
    rank, world, local, rdv = launch.job()
    sub = os.path.join(rdv, "t0")
    os.makedirs(sub, exist_ok=True)
    t = _lib.Transport("shm", rank, world, -1, sub)
    assert "shm" in t.describe()
    for rnd in range(3):
        size = lambda i, j: 1000 * (rnd + 1) + 37 * i + 11 * j          # noqa: E731  floats from i to j
        sends = {j: np.full(size(rank, j), 1000.0 * rank + j + rnd, dtype=np.float32) for j in range(world) if j != rank or rnd == 2}
        recvs = {i: np.zeros(size(i, rank), dtype=np.float32) for i in range(world) if i != rank or rnd == 2}
        msgs = [(a.ctypes.data, a.nbytes, i, False) for i, a in recvs.items()] + [(a.ctypes.data, a.nbytes, j, True) for j, a in sends.items()]
        if rnd == 1 and rank == world - 1:
            msgs = []                            # a rank with nothing to move still takes part (n = 0)
            recvs = {}
        elif rnd == 1:
            msgs = [m for m in msgs if m[2] != world - 1]
            recvs.pop(world - 1, None)
        t.exchange(msgs, 0)
        for i, a in recvs.items():
            assert np.all(a == np.float32(1000.0 * i + rank + rnd)), (rnd, rank, i, a[:4])
        got = t.allgather_array(np.array([rank * 10 + rnd, rank], dtype=np.int64))
        assert got.tolist() == [[r * 10 + rnd, r] for r in range(world)]
        t.barrier()
    blob = bytes([rank]) * 5
    assert t.allgather_host(blob) == b"".join(bytes([r]) * 5 for r in range(world))
    t.close()
    print(f"rank {rank} ok", flush=True)


if __name__ == "__main__":
    main()
