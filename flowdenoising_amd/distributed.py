"""Multi-GPU decomposition of OF_filter: one process per GPU.

Every axis pass shards along ITS OWN axis (target slices are independent; each needs K//2
neighbour slices either side), so the volume moves through three partitions:

    Z-slabs --Z pass--> Z-slabs --all-to-all--> Y-slabs --Y pass--> --all-to-all--> X-slabs
            --X pass--> --all-to-all--> Z-slabs (output, same partition as the input)

with a K//2-slice halo exchange between neighbouring ranks before each pass (the outermost
slabs pad with the global mean, src/flowdenoising_sequential.py:88, obtained by one scalar
all-reduce, seq:420; wrap-around borders of src/flowdenoising.py:312 come from the far ranks).
The Y/X passes cannot use fixed halos along Z because their images span Z (seq:255, seq:333),
hence the repartition.

Collectives go through torch.distributed point-to-point ops (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests): they are neighbour / all-pairs exchanges of contiguous
blocks, all links busy at once, no ring collective and no reduction of bulk data.

The compute of a pass is delegated to a backend object:
    backend.sweep_stack(stack, out, S, H, W, kernel, params)   stack: (S + 2r, H, W) tensor
    backend.local_sum(tensor) -> float
HipBackend (below) runs it in libflowdn.so; the CPU tests inject a backend built on the oracle.
"""
import numpy as np


def split(n, parts):
    """Near-equal contiguous split of range(n): [(start, stop)] * parts."""
    base, rem = divmod(n, parts)
    out, s = [], 0
    for i in range(parts):
        e = s + base + (1 if i < rem else 0)
        out.append((s, e))
        s = e
    return out


class SlabPlan:
    def __init__(self, shape, world, rank):
        Z, Y, X = shape
        if world > min(Z, Y, X):
            raise ValueError(f"{world} ranks need every axis >= {world}, got {shape}")
        self.shape = tuple(shape)
        self.world, self.rank = world, rank
        self.parts = [split(n, world) for n in shape]  # per axis: [(start, stop)] per rank
        self.z0, z1 = self.parts[0][rank]
        self.zlen = z1 - self.z0

    def owner(self, axis, idx):
        for r, (s, e) in enumerate(self.parts[axis]):
            if s <= idx < e:
                return r
        raise IndexError(idx)

    def halo_runs(self, axis, r, wrap):
        """Runs (src_rank, dst_rank, src_local_start, dst_stack_start, count) that fill every rank's
        two halos of `r` slices along `axis`.  Stack position p of rank d holds global slice
        start_d - r + p.  Mean-padded borders (wrap=False) produce no run for out-of-range slices."""
        n = self.shape[axis]
        runs = []
        for d, (s, e) in enumerate(self.parts[axis]):
            for lo_p, lo_g in ((0, s - r), (r + (e - s), e)):
                q = 0
                while q < r:
                    g = lo_g + q
                    if wrap:
                        g %= n
                    elif g < 0 or g >= n:
                        q += 1
                        continue
                    src = self.owner(axis, g)
                    ss, se = self.parts[axis][src]
                    cnt = 1  # extend the run while contiguous, same owner and in range
                    while q + cnt < r:
                        g2 = lo_g + q + cnt
                        g2 = g2 % n if wrap else g2
                        if not (0 <= g2 < n) or g2 != g + cnt or not (ss <= g2 < se):
                            break
                        cnt += 1
                    runs.append((src, d, g - ss, lo_p + q, cnt))
                    q += cnt
        return runs


class HipBackend:
    """Pass compute in libflowdn.so on torch CUDA tensors (device pointers through the C ABI)."""

    def __init__(self, handle):
        self.h = handle

    def sweep_stack(self, stack, out, S, H, W, kernel, params):
        assert stack.is_contiguous() and out.is_contiguous()
        self.h.sweep_stack_dev(stack.data_ptr(), out.data_ptr(), S, H, W, kernel, params)

    def local_sum(self, t):
        return self.h.sum_dev(t.data_ptr(), t.numel())

    def chunk_sums(self, t):
        return self.h.np_chunk_sums_dev(t.data_ptr(), t.numel())


class SlabEngine:
    """Distributed OF_filter / no_OF_filter on Z-slabs.  `dist` is torch.distributed (or None for a
    single rank).  Tensors live wherever the backend computes (CUDA for HipBackend)."""

    # orientation of a slab partitioned along `axis`: stack dims (slices, H, W) as global axes
    ORIENT = {0: (0, 1, 2), 1: (1, 0, 2), 2: (2, 0, 1)}  # Z: (z|y,x)  Y: (y|z,x)  X: (x|z,y)

    def __init__(self, plan, backend, dist=None):
        import torch
        self.torch = torch
        self.plan, self.backend, self.dist = plan, backend, dist
        if hasattr(backend, "sweep_stack_dev"):  # a bare fdn handle was passed
            self.backend = HipBackend(backend)

    # -- communication helpers -----------------------------------------------------------
    def _exchange(self, sends, recvs):
        """sends: [(dst_rank, tensor)], recvs: [(src_rank, tensor)] in matching per-pair order.
        Self-pairs are copied locally."""
        torch, dist, me = self.torch, self.dist, self.plan.rank
        local_s = [t for r, t in sends if r == me]
        local_r = [t for r, t in recvs if r == me]
        for s, d in zip(local_s, local_r):
            d.copy_(s)
        ops = []
        if dist is not None and self.plan.world > 1:
            for r, t in recvs:
                if r != me:
                    ops.append(dist.P2POp(dist.irecv, t, r))
            for r, t in sends:
                if r != me:
                    ops.append(dist.P2POp(dist.isend, t, r))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()

    def global_mean(self, vol):
        """seq:420 for a sharded volume.  numpy reduces a float32 volume as pairwise sums of 8192-element
        chunks accumulated left to right in float32; when every slab starts at a chunk boundary (Y*X a
        multiple of 8192, as for 1024 x 1024 slices) the ranks form their chunks' sums, gather them and
        accumulate in order: numpy's value exactly.  Otherwise: float64 sum per rank, one scalar
        all-reduce (within 1 ulp)."""
        torch, dist = self.torch, self.dist
        Z, Y, X = self.plan.shape
        if (Y * X) % 8192 == 0 and hasattr(self.backend, "chunk_sums"):
            mine = np.ascontiguousarray(self.backend.chunk_sums(vol), dtype=np.float32)
            if dist is not None and self.plan.world > 1:
                per = [(e - s) * (Y * X // 8192) for s, e in self.plan.parts[0]]
                buf = torch.zeros(max(per), dtype=torch.float32, device=vol.device)
                buf[:mine.size] = torch.from_numpy(mine).to(vol.device)
                got = [torch.empty_like(buf) for _ in range(self.plan.world)]
                dist.all_gather(got, buf)
                allsums = np.concatenate([g[:n].cpu().numpy() for g, n in zip(got, per)])
            else:
                allsums = mine
            tot = np.cumsum(allsums, dtype=np.float32)[-1]     # sequential float32 accumulation
            return np.float32(tot / np.float32(Z * Y * X))
        s = float(self.backend.local_sum(vol))
        if dist is not None and self.plan.world > 1:
            t = torch.tensor([s], dtype=torch.float64, device=vol.device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            s = float(t.item())
        Z, Y, X = self.plan.shape
        return np.float32(s / (Z * Y * X))

    def _fill_halos(self, stack, axis, r, mean, wrap):
        """stack: (len + 2r, H, W) with its interior already in place."""
        plan, me = self.plan, self.plan.rank
        s, e = plan.parts[axis][me]
        n_loc = e - s
        if not wrap:
            stack[:r].fill_(float(mean))
            stack[r + n_loc:].fill_(float(mean))
        interior = stack[r:r + n_loc]
        sends, recvs = [], []
        for src, dst, s_loc, d_pos, cnt in plan.halo_runs(axis, r, wrap):
            if src == me:
                sends.append((dst, interior[s_loc:s_loc + cnt] if dst != me else interior[s_loc:s_loc + cnt].clone()))
            if dst == me:
                recvs.append((src, stack[d_pos:d_pos + cnt]))
        self._exchange(sends, recvs)

    def _repartition(self, slab, from_axis, to_axis, r):
        """slab: (len_from, H, W) oriented for `from_axis`, holding this rank's part of the volume.
        Returns the stack (len_to + 2r, H', W') oriented for `to_axis` with its interior filled."""
        torch, plan, me = self.torch, self.plan, self.plan.rank
        of, ot = self.ORIENT[from_axis], self.ORIENT[to_axis]
        ts, te = plan.parts[to_axis][me]
        dims_to = [plan.shape[a] for a in ot]
        stack = torch.empty((te - ts + 2 * r, dims_to[1], dims_to[2]), dtype=slab.dtype, device=slab.device)
        interior = stack[r:r + (te - ts)]
        # view of my slab / my stack interior indexed by GLOBAL axes order (z, y, x)
        slab_g = slab.permute(*[of.index(a) for a in (0, 1, 2)])
        inter_g = interior.permute(*[ot.index(a) for a in (0, 1, 2)])
        fs, fe = plan.parts[from_axis][me]
        sends, recvs = [], []
        for j in range(plan.world):
            # block I send to j: my from-range x j's to-range
            js, je = plan.parts[to_axis][j]
            idx = [slice(None)] * 3
            idx[to_axis] = slice(js, je)
            sends.append((j, slab_g[tuple(idx)].contiguous()))
            # block I receive from j: j's from-range x my to-range
            gs, ge = plan.parts[from_axis][j]
            shp = list(plan.shape)
            shp[from_axis] = ge - gs
            shp[to_axis] = te - ts
            recvs.append((j, torch.empty(shp, dtype=slab.dtype, device=slab.device)))
        self._exchange(sends, recvs)
        for j, blk in recvs:
            gs, ge = plan.parts[from_axis][j]
            idx = [slice(None)] * 3
            idx[from_axis] = slice(gs, ge)
            inter_g[tuple(idx)].copy_(blk)
        return stack

    # -- the filter ------------------------------------------------------------------------
    def filter_3d(self, vol, kernels, params, mean=None):
        """vol: this rank's Z-slab (zlen, Y, X).  Returns the filtered Z-slab (same partition).
        kernels = [kz, ky, kx]; None skips an axis.  params: _lib.SweepParams."""
        torch, plan = self.torch, self.plan
        if tuple(vol.shape) != (plan.zlen, plan.shape[1], plan.shape[2]):
            raise ValueError(f"rank {plan.rank} expects a {(plan.zlen,) + plan.shape[1:]} slab, got {tuple(vol.shape)}")
        wrap = params.border_mode == 1
        if mean is None:
            mean = self.global_mean(vol) if not wrap else np.float32(0)
        cur, cur_axis = vol, 0
        for axis in (0, 1, 2):
            k = kernels[axis]
            if k is None:
                continue
            k = np.ascontiguousarray(k, dtype=np.float64)
            r = k.size // 2
            if cur_axis == axis:
                n_loc = cur.shape[0]
                stack = torch.empty((n_loc + 2 * r,) + tuple(cur.shape[1:]), dtype=cur.dtype, device=cur.device)
                stack[r:r + n_loc].copy_(cur)
            else:
                stack = self._repartition(cur, cur_axis, axis, r)
                n_loc = stack.shape[0] - 2 * r
            self._fill_halos(stack, axis, r, mean, wrap)
            out = torch.empty((n_loc,) + tuple(stack.shape[1:]), dtype=cur.dtype, device=cur.device)
            self.backend.sweep_stack(stack, out, n_loc, stack.shape[1], stack.shape[2], k, params)
            cur, cur_axis = out, axis
        if cur_axis != 0:
            cur = self._repartition(cur, cur_axis, 0, 0)
        elif cur is vol:
            cur = vol.clone()
        return cur
