"""Multi-GPU decomposition of OF_filter: one process per GPU.

Every axis pass shards along ITS OWN axis (target slices are independent; each needs K//2
neighbour slices either side), so the volume moves through three partitions:

    Z-slabs --Z pass--> Z-slabs --exchange--> Y-slabs --Y pass--> --exchange--> X-slabs
            --X pass--> --exchange--> Z-slabs (output, same partition as the input)

The Y/X passes cannot use fixed halos along Z because their images span Z (seq:255, seq:333),
hence the repartition.  ONE exchange per pass delivers both the new partition and its halos:
rank i sends rank j the part of j's halo-extended range that i holds (for the Z pass: the
K//2 boundary slices of the neighbouring slabs; for Y and X: an all-to-all of
(own range) x (j's range + halo) blocks), so a pass costs one round of point-to-point
messages and no second synchronisation for halos.  Slices outside the volume are filled
with the global mean (src/flowdenoising_sequential.py:88; one small gather, seq:420) or come
from the far ranks (wrap-around, src/flowdenoising.py:312).

Collectives go through torch.distributed point-to-point ops (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests): one batched group per exchange, all pairs at once -- all
links busy, no ring, no reduction of bulk data.  The exchange stays on this side of the C ABI
(INTEGRATION.md 3): libflowdn.so is single-device; it supplies the pass (fdn_sweep_stack_dev),
the pack kernel (fdn_permute_dev) and the reduction pieces of the mean.

Buffers (stack, pass output, send and receive staging) are allocated once per engine and reused
every step.  Blocks are packed on the sender INTO THE RECEIVER'S ORIENTATION by the LDS-tiled
permute kernel; the receiver drops them into its stack with row-contiguous strided copies.

The compute of a pass is delegated to a backend object:
    backend.sweep_stack(stack, out, S, H, W, kernel, params)   stack: (S + 2r, H, W) tensor
    backend.pack(src_view, dst)                                dst contiguous <- permuted/strided view
    backend.chunk_sums(flat_tensor) -> float32 array           numpy's pairwise sums of 8192-element chunks
HipBackend (below) runs them in libflowdn.so; the CPU tests inject a backend built on the oracle.
"""
import time

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


# orientation of a slab partitioned along `axis`: stack dims (slices, H, W) as global axes
ORIENT = {0: (0, 1, 2), 1: (1, 0, 2), 2: (2, 0, 1)}  # Z: (z|y,x)  Y: (y|z,x)  X: (x|z,y)


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

    def stack_runs(self, axis, j, r, wrap):
        """Rank j's stack along `axis` holds positions p = 0 .. len_j + 2r - 1, position p being global slice
        s_j - r + p.  Returns maximal runs (p0, g0, count) of consecutive positions whose global slices are
        consecutive and inside the volume (wrap: taken modulo the axis length); positions outside the volume
        of a mean-padded pass belong to no run."""
        n = self.shape[axis]
        s, e = self.parts[axis][j]
        runs = []
        for p in range(e - s + 2 * r):
            g = s - r + p
            if wrap:
                g %= n
            elif g < 0 or g >= n:
                continue
            if runs and runs[-1][0] + runs[-1][2] == p and runs[-1][1] + runs[-1][2] == g:
                runs[-1][2] += 1
            else:
                runs.append([p, g, 1])
        return [tuple(x) for x in runs]

    def blocks(self, A, B, r, wrap, i, j):
        """What rank i (holding its slab of the partition along A) sends rank j for the pass along B:
        [(p0, {axis: (lo, hi)})] -- global index ranges of the block and the stack position of its first
        B-slice.  Sender and receiver derive the same list, in the same order."""
        si, ei = self.parts[A][i]
        out = []
        for p0, g0, cnt in self.stack_runs(B, j, r, wrap):
            rng = {ax: (0, self.shape[ax]) for ax in range(3)}
            if A == B:                     # same partition: i contributes the slices of the run it owns
                lo, hi = max(g0, si), min(g0 + cnt, ei)
                if lo >= hi:
                    continue
                rng[B] = (lo, hi)
                out.append((p0 + lo - g0, rng))
            else:                          # repartition: i holds every B-slice of its own A-range
                rng[B] = (g0, g0 + cnt)
                rng[A] = (si, ei)
                out.append((p0, rng))
        return out

    # kept for callers of the round-1 interface (tests of the halo geometry)
    def halo_runs(self, axis, r, wrap):
        """Runs (src_rank, dst_rank, src_local_start, dst_stack_start, count) filling every rank's two halos."""
        runs = []
        for d, (s, e) in enumerate(self.parts[axis]):
            for p0, g0, cnt in self.stack_runs(axis, d, r, wrap):
                for q in range(cnt):
                    p, g = p0 + q, g0 + q
                    if r <= p < r + (e - s):
                        continue
                    src = self.owner(axis, g)
                    ss = self.parts[axis][src][0]
                    last = runs[-1] if runs else None
                    if last and last[0] == src and last[1] == d and last[2] + last[4] == g - ss and last[3] + last[4] == p:
                        last[4] += 1
                    else:
                        runs.append([src, d, g - ss, p, 1])
        return [tuple(x) for x in runs]


class TorchComm:
    """The two callbacks of fdn_filter_3d_sharded (include/flowdn.h: fdn_comm) over torch.distributed: with the `nccl`
    backend (RCCL) the device buffers go into one batched group of isend / irecv as they are; with `gloo` (CPU tests,
    rehearsals of several ranks on one GPU) they are staged through the host.  The C++ engine behind that entry point
    runs the same schedule as SlabEngine below."""

    def __init__(self, dist, device):
        import torch
        self.torch, self.dist, self.device = torch, dist, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    class _Ptr:
        def __init__(self, ptr, nbytes):
            self.__cuda_array_interface__ = {"shape": (nbytes // 4,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}

    def _view(self, ptr, nbytes):
        return self.torch.as_tensor(self._Ptr(ptr, nbytes), device=self.device)

    def exchange(self, msgs, stream):
        torch, dist = self.torch, self.dist
        if not msgs:
            return
        views = [(self._view(p, n), peer, snd) for p, n, peer, snd in msgs]
        s = torch.cuda.ExternalStream(int(stream or 0), device=self.device) if stream else torch.cuda.current_stream(self.device)
        with torch.cuda.stream(s):
            if dist.get_backend() == "nccl":
                ops = [dist.P2POp(dist.isend if snd else dist.irecv, t, peer) for t, peer, snd in views]
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            else:
                host = [(t.cpu() if snd else torch.empty(t.shape, dtype=t.dtype), t, peer, snd) for t, peer, snd in views]
                ops = [dist.P2POp(dist.isend if snd else dist.irecv, ht, peer) for ht, _, peer, snd in host]
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
                for ht, t, _, snd in host:
                    if not snd:
                        t.copy_(ht)
            s.synchronize()

    def allgather_host(self, send):
        torch, dist = self.torch, self.dist
        mine = torch.frombuffer(bytearray(send), dtype=torch.uint8)
        dev = self.device if dist.get_backend() == "nccl" else "cpu"
        got = [torch.empty(mine.numel(), dtype=torch.uint8, device=dev) for _ in range(self.world)]
        dist.all_gather(got, mine.to(dev))
        return b"".join(bytes(g.cpu().numpy().tobytes()) for g in got)


class HipBackend:
    """Pass compute in libflowdn.so on torch CUDA tensors (device pointers through the C ABI)."""

    def __init__(self, handle):
        self.h = handle

    def sweep_stack(self, stack, out, S, H, W, kernel, params):
        assert stack.is_contiguous() and out.is_contiguous()
        self.h.sweep_stack_dev(stack.data_ptr(), out.data_ptr(), S, H, W, kernel, params)

    def pack(self, src_view, dst):
        """dst (contiguous, same shape) <- src_view (a permuted / sliced view with one unit stride): fdn_permute_dev,
        LDS-tiled so that both sides stay coalesced."""
        st = src_view.stride()
        if dst.numel() == 0:
            return
        if 1 not in st:                       # a one-element dim carries an arbitrary stride: let torch copy
            dst.copy_(src_view)
            return
        self.h.permute_dev(src_view.data_ptr(), dst.data_ptr(), tuple(src_view.shape), tuple(st))

    def chunk_sums(self, t):
        return self.h.np_chunk_sums_dev(t.data_ptr(), t.numel())


class SlabEngine:
    """Distributed OF_filter / no_OF_filter on Z-slabs.  `dist` is torch.distributed (or None for a
    single rank).  Tensors live wherever the backend computes (CUDA for HipBackend)."""

    ORIENT = ORIENT
    PHASES = ("compute", "pack", "exchange", "unpack", "mean")

    def __init__(self, plan, backend, dist=None, loopback=False):
        """loopback: route the blocks a rank keeps for itself through the transport too (a send to self inside the
        batched group) and take the mean through the all_gather even with one rank -- so that a world-size-1 run on
        one GPU carries real RCCL calls in exactly the shapes an N > 1 run issues (a test switch; slower)."""
        import torch
        self.torch = torch
        self.plan, self.backend, self.dist = plan, backend, dist
        self.loopback = bool(loopback) and dist is not None
        if hasattr(backend, "sweep_stack_dev"):  # a bare fdn handle was passed
            self.backend = HipBackend(backend)
        self._bufs = {}
        self._sched = {}
        self._ms = {p: 0.0 for p in self.PHASES}
        self._pending = []          # (phase, start event, stop event) not yet resolved
        self.timing = True

    # -- persistent buffers and phase timers ----------------------------------------------------
    def _buf(self, name, numel, like):
        """A flat float32 buffer of at least `numel` elements, allocated once and reused every step."""
        b = self._bufs.get(name)
        if b is None or b.numel() < numel or b.device != like.device:
            b = self.torch.empty(max(int(numel), 1), dtype=like.dtype, device=like.device)
            self._bufs[name] = b
        return b

    class _Phase:
        def __init__(self, eng, name, cuda):
            self.eng, self.name, self.cuda = eng, name, cuda and eng.timing

        def __enter__(self):
            if self.cuda:
                self.a = self.eng.torch.cuda.Event(enable_timing=True)
                self.a.record()
            else:
                self.t0 = time.perf_counter()
            return self

        def __exit__(self, *exc):
            if self.cuda:
                b = self.eng.torch.cuda.Event(enable_timing=True)
                b.record()
                self.eng._pending.append((self.name, self.a, b))
            elif self.eng.timing:
                self.eng._ms[self.name] += (time.perf_counter() - self.t0) * 1e3
            return False

    def _phase(self, name, like):
        return self._Phase(self, name, like.is_cuda)

    def phase_times(self):
        """Milliseconds per phase on this rank since the last reset: the pass kernels ("compute"), packing blocks
        into the receiver's orientation, the point-to-point exchange (waiting for the slowest peer included),
        unpacking, and the global mean.  GPU phases are timed with events on the stream that runs them."""
        if self._pending:
            self.torch.cuda.synchronize()
            h = getattr(self.backend, "h", None)
            for name, a, b in self._pending:
                ms = a.elapsed_time(b)
                self._ms[name] += ms
                if name == "exchange" and h is not None:          # the collectives, in the library's own timer table too
                    h.add_timer("collective", ms)
            self._pending = []
        return dict(self._ms)

    def reset_phase_times(self):
        self.phase_times()
        self._ms = {p: 0.0 for p in self.PHASES}

    # -- communication ----------------------------------------------------------------------------
    def _host_staged(self, t):
        """gloo moves host memory only: with it, device tensors are staged through the host (the CPU tests, and the
        world-size-2 rehearsal of the HIP backend on a one-GPU box; RCCL needs one GPU per rank)."""
        return t.is_cuda and self.dist is not None and self.dist.get_backend() == "gloo"

    def _p2p(self, recvs, sends):
        """recvs: [(tensor, src rank)], sends: [(tensor, dst rank)] -- one batched group, every pair at once."""
        dist = self.dist
        if dist.get_backend() != "nccl":     # RCCL carries a send to self inside a group; gloo has no pair to oneself:
            me = self.plan.rank              # match those messages here, in order (loopback mode only)
            own_r = [t for t, r in recvs if r == me]
            own_s = [t for t, r in sends if r == me]
            assert len(own_r) == len(own_s)
            for d, s_ in zip(own_r, own_s):
                d.copy_(s_)
            recvs = [(t, r) for t, r in recvs if r != me]
            sends = [(t, r) for t, r in sends if r != me]
        if not recvs and not sends:
            return
        if (recvs and self._host_staged(recvs[0][0])) or (sends and self._host_staged(sends[0][0])):
            host_r = [(self.torch.empty(t.shape, dtype=t.dtype), t, r) for t, r in recvs]
            ops = [dist.P2POp(dist.irecv, h, r) for h, _, r in host_r] + [dist.P2POp(dist.isend, t.cpu(), r) for t, r in sends]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            for h, t, _ in host_r:
                t.copy_(h)
            return
        ops = [dist.P2POp(dist.irecv, t, r) for t, r in recvs] + [dist.P2POp(dist.isend, t, r) for t, r in sends]
        for w in dist.batch_isend_irecv(ops):
            w.wait()

    def _all_gather(self, t):
        """[t of rank 0, t of rank 1, ...] (same shape on every rank)."""
        if self._host_staged(t):
            got = [self.torch.empty(t.shape, dtype=t.dtype) for _ in range(self.plan.world)]
            self.dist.all_gather(got, t.cpu())
            return got
        got = [self.torch.empty_like(t) for _ in range(self.plan.world)]
        self.dist.all_gather(got, t)
        return got

    def _schedule(self, A, B, r, wrap):
        """Send and receive lists of this rank for the exchange A-partition -> pass along B (cached)."""
        key = (A, B, r, wrap)
        sc = self._sched.get(key)
        if sc is None:
            plan, me = self.plan, self.plan.rank
            sends = [(j, plan.blocks(A, B, r, wrap, me, j)) for j in range(plan.world)]
            recvs = [(i, plan.blocks(A, B, r, wrap, i, me)) for i in range(plan.world)]
            sc = self._sched[key] = (sends, recvs)
        return sc

    @staticmethod
    def _numel(rng):
        n = 1
        for lo, hi in rng.values():
            n *= hi - lo
        return n

    def _exchange(self, slab, A, B, r, wrap, stack):
        """Fill `stack` (oriented for B, K//2 = r halo slices either side) from the slabs of the A-partition."""
        torch, dist, plan, me = self.torch, self.dist, self.plan, self.plan.rank
        oa, ob = ORIENT[A], ORIENT[B]
        sends, recvs = self._schedule(A, B, r, wrap)
        n_send = sum(self._numel(rng) for _, bl in sends for _, rng in bl)
        loop = self.loopback
        n_recv = sum(self._numel(rng) for i, bl in recvs if i != me or loop for _, rng in bl)
        sendbuf = self._buf("send", n_send, slab)
        recvbuf = self._buf("recv", n_recv, slab)
        sa = plan.parts[A][me][0]
        # 1. pack every block into the receiver's orientation
        soff, send_regions = 0, {}
        with self._phase("pack", slab):
            for j, bl in sends:
                start = soff
                for _, rng in bl:
                    idx = tuple(slice(rng[ax][0] - (sa if ax == A else 0), rng[ax][1] - (sa if ax == A else 0)) for ax in oa)
                    view = slab[idx].permute(*[oa.index(ax) for ax in ob])
                    n = view.numel()
                    self.backend.pack(view, sendbuf[soff:soff + n].view(view.shape))
                    soff += n
                send_regions[j] = (start, soff)
        # 2. one batched group of point-to-point messages: every pair at once
        roff, recv_regions = 0, {}
        for i, bl in recvs:
            if i == me and not loop:
                continue
            n = sum(self._numel(rng) for _, rng in bl)
            recv_regions[i] = (roff, roff + n)
            roff += n
        with self._phase("exchange", slab):
            if dist is not None and (plan.world > 1 or loop):
                self._p2p([(recvbuf[lo:hi], i) for i, (lo, hi) in recv_regions.items() if hi > lo],
                          [(sendbuf[lo:hi], j) for j, (lo, hi) in send_regions.items() if (j != me or loop) and hi > lo])
        # 3. unpack: row-contiguous strided copies into the stack (the local block straight from the send buffer)
        with self._phase("unpack", slab):
            for i, bl in recvs:
                buf, off = (sendbuf, send_regions[me][0]) if i == me and not loop else (recvbuf, recv_regions[i][0])
                for p0, rng in bl:
                    shp = [rng[ax][1] - rng[ax][0] for ax in ob]
                    n = shp[0] * shp[1] * shp[2]
                    dst = stack[p0:p0 + shp[0], rng[ob[1]][0]:rng[ob[1]][1], rng[ob[2]][0]:rng[ob[2]][1]]
                    dst.copy_(buf[off:off + n].view(shp))
                    off += n

    def _fill_pad(self, stack, B, r, mean):
        """Mean-padded pass: stack positions whose slice lies outside the volume (seq:88-89)."""
        s, e = self.plan.parts[B][self.plan.rank]
        n = self.plan.shape[B]
        lo = max(0, r - s)                       # positions 0 .. lo-1 are before slice 0
        hi = min(e - s + 2 * r, n - s + r)       # positions hi .. are after slice n-1
        if lo > 0:
            stack[:lo].fill_(float(mean))
        if hi < e - s + 2 * r:
            stack[hi:].fill_(float(mean))
        return lo, e - s + 2 * r - hi       # leading / trailing pad slices

    # -- seq:420 for a sharded volume ----------------------------------------------------------------
    def global_mean(self, vol):
        """numpy's float32 vol.mean() of the WHOLE volume, bit for bit, from Z-slabs: numpy reduces a float32 array
        as pairwise sums of 8192-element chunks accumulated left to right in float32.  Every chunk is summed (in
        numpy's order, backend.chunk_sums) by the rank that holds its first element; a chunk that straddles a slab
        boundary gets its missing elements from the following rank (at most 8191 of them).  The chunk sums are then
        gathered and accumulated in order.  (The padded volume ends move by 1.2e-4 of the range per ulp of this
        value, DESIGN.md 4.4 -- hence exactly numpy's value and not just a good mean.)"""
        torch, dist, plan = self.torch, self.dist, self.plan
        Z, Y, X = plan.shape
        ntot = Z * Y * X
        flat = vol.reshape(-1)
        with self._phase("mean", vol):
            if dist is None or (plan.world == 1 and not self.loopback):
                sums = np.asarray(self.backend.chunk_sums(flat), dtype=np.float32)
                return np.float32(np.cumsum(sums, dtype=np.float32)[-1] / np.float32(ntot))
            starts = [s * Y * X for s, _ in plan.parts[0]] + [ntot]
            if min(starts[k + 1] - starts[k] for k in range(plan.world)) < 8192:
                # tiny volume: a chunk may span several slabs; gather the whole thing (it is small)
                m = max(starts[k + 1] - starts[k] for k in range(plan.world))
                pad = torch.zeros(m, dtype=flat.dtype, device=flat.device)
                pad[:flat.numel()] = flat
                got = self._all_gather(pad)
                whole = torch.cat([g[:starts[k + 1] - starts[k]] for k, g in enumerate(got)]).to(flat.device)
                sums = np.asarray(self.backend.chunk_sums(whole), dtype=np.float32)
                return np.float32(np.cumsum(sums, dtype=np.float32)[-1] / np.float32(ntot))
            me = plan.rank
            up = lambda v: -(-v // 8192) * 8192           # noqa: E731  next chunk boundary
            first = [min(up(starts[k]), ntot) for k in range(plan.world)] + [ntot]   # first element of rank k's own chunks
            head = first[me] - starts[me]                 # my leading elements belong to the previous rank's last chunk
            tail = first[me + 1] - starts[me + 1]         # elements of my last chunk held by the next rank
            tail_buf = torch.empty(tail, dtype=flat.dtype, device=flat.device) if tail > 0 else None
            self._p2p([(tail_buf, me + 1)] if tail > 0 else [], [(flat[:head].contiguous(), me - 1)] if head > 0 else [])
            own = flat[head:]
            if tail > 0:
                nfull = own.numel() // 8192 * 8192
                parts = [np.asarray(self.backend.chunk_sums(own[:nfull]), dtype=np.float32)] if nfull else []
                parts.append(np.asarray(self.backend.chunk_sums(torch.cat([own[nfull:], tail_buf])), dtype=np.float32))
                mine = np.concatenate(parts)
            else:
                mine = np.asarray(self.backend.chunk_sums(own), dtype=np.float32) if own.numel() else np.zeros(0, np.float32)
            per = [(first[k + 1] - first[k] + 8191) // 8192 for k in range(plan.world)]
            assert mine.size == per[me], (mine.size, per)
            buf = torch.zeros(max(max(per), 1), dtype=torch.float32, device=vol.device)
            if mine.size:
                buf[:mine.size] = torch.from_numpy(mine).to(vol.device)
            got = self._all_gather(buf)
            allsums = np.concatenate([g[:n].cpu().numpy() for g, n in zip(got, per)])
            return np.float32(np.cumsum(allsums, dtype=np.float32)[-1] / np.float32(ntot))

    def gather_z_slabs(self, slab, dst=0):
        """The whole volume on rank `dst` (None on the others): every rank sends its Z-slab there and only there."""
        torch, plan, me = self.torch, self.plan, self.plan.rank
        if plan.world == 1 or self.dist is None:
            return slab
        if me != dst:
            self._p2p([], [(slab.contiguous(), dst)])
            return None
        full = torch.empty(plan.shape, dtype=slab.dtype, device=slab.device)
        full[plan.z0:plan.z0 + plan.zlen].copy_(slab)
        self._p2p([(full[s:e], r) for r, (s, e) in enumerate(plan.parts[0]) if r != dst], [])
        return full

    # -- the filter ------------------------------------------------------------------------
    def filter_3d(self, vol, kernels, params, mean=None):
        """vol: this rank's Z-slab (zlen, Y, X).  Returns the filtered Z-slab (same partition) -- a view of an
        engine-owned buffer that the next call overwrites.  kernels = [kz, ky, kx]; None skips an axis."""
        plan = self.plan
        if tuple(vol.shape) != (plan.zlen, plan.shape[1], plan.shape[2]):
            raise ValueError(f"rank {plan.rank} expects a {(plan.zlen,) + plan.shape[1:]} slab, got {tuple(vol.shape)}")
        if not vol.is_contiguous():
            vol = vol.contiguous()
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
            s, e = plan.parts[axis][plan.rank]
            dims = [plan.shape[a] for a in ORIENT[axis]]
            n_loc, H, W = e - s, dims[1], dims[2]
            stack = self._buf(f"stack{axis}", (n_loc + 2 * r) * H * W, vol)[:(n_loc + 2 * r) * H * W].view(n_loc + 2 * r, H, W)
            self._exchange(cur, cur_axis, axis, r, wrap, stack)
            pp = params
            if not wrap:
                npl, nph = self._fill_pad(stack, axis, r, mean)
                if getattr(params, "warp_mode", 0) == 1:      # FDN_WARP_F64_PADDED: the library needs to know the pad slices
                    pp = params.copy()
                    pp.pad_lo, pp.pad_hi = npl, nph
            out = self._buf(f"out{axis}", n_loc * H * W, vol)[:n_loc * H * W].view(n_loc, H, W)
            with self._phase("compute", vol):
                self.backend.sweep_stack(stack, out, n_loc, H, W, k, pp)
            cur, cur_axis = out, axis
        if cur_axis != 0:
            res = self._buf("result", vol.numel(), vol)[:vol.numel()].view(vol.shape)
            self._exchange(cur, cur_axis, 0, 0, False, res)
            cur = res
        elif cur is vol:
            res = self._buf("result", vol.numel(), vol)[:vol.numel()].view(vol.shape)
            res.copy_(vol)
            cur = res
        return cur
