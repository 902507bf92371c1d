"""Command line of flowdenoising.py, MI355X build.

Same options as the reference (union of src/flowdenoising_sequential.py:446-473 and
src/flowdenoising.py:384-415): -i/--input, -o/--output, -s/--sigma (Z Y X), -l/--levels,
-w/--winsize, -v/--verbosity, -n/--no_OF, -m/--memory_map, -p/--number_of_processes,
--recompute_flow, --show_fingerprint (and, ignored, flowdenoising_GPU.py's --use_GPU / --use_threads).  File-type dispatch and output dtypes follow seq:508-571.

Defaults follow the parity oracle, flowdenoising_sequential.py: levels = 0, volume ends padded
with the global mean, full Z->Y->X result.  `--compat par` switches to flowdenoising.py's own
behaviour: levels = 3 by default, wrap-around volume ends (par:312), float32 TIFF output
(par:548).  (par's loss of its X pass, par:290 + par:520, is not reproduced: the full Z->Y->X
result is written, as src/flowdenoising_GPU.py:460 does.)

New options: --device N (GPU index), --gpus N (shard over N GPUs of this node: N rank processes started with
subprocess.Popen, exchanges through libflowdn_rccl.so -- RCCL over xGMI; PyTorch is not involved), --chunk_slices N (out-of-core mode for volumes larger than the GPU's
memory: the volume stays on the host, N slices of a pass at a time on the GPU; -1 = as many as fit),
--strict_order (OpenCV's own f64 summation order in the box filter; slow, for verification).
"""
import argparse
import hashlib
import logging
import os
import sys
import threading
import time

import numpy as np

LOGGING_FORMAT = "[%(asctime)s] (%(levelname)s) %(message)s"  # seq:26
SIGMA = 2.0
OF_WINDOW_SIZE = 5


def int_or_str(text):
    """seq:433-438."""
    try:
        return int(text)
    except ValueError:
        return text


def build_parser():
    p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter,
                                description="3D Gaussian filtering controlled by the optical flow (MI355X build).")
    p.add_argument("-i", "--input", type=int_or_str, help="Input a MRC-file or a multi-image TIFF-file",
                   default="./volume.mrc")
    p.add_argument("-o", "--output", type=int_or_str, help="Output a MRC-file or a multi-image TIFF-file",
                   default="./denoised_volume.mrc")
    p.add_argument("-s", "--sigma", nargs="+", help="Gaussian sigma for each dimension in the order (Z, Y, X)",
                   default=(SIGMA, SIGMA, SIGMA))
    p.add_argument("-l", "--levels", type=int_or_str, default=None,
                   help="Number of levels of the Gaussian pyramid used by the optical flow estimator "
                        "(default 0; 3 with --compat par)")
    p.add_argument("-w", "--winsize", type=int_or_str, help="Size of the window used by the optical flow estimator",
                   default=OF_WINDOW_SIZE)
    p.add_argument("-v", "--verbosity", type=int_or_str, help="Verbosity level", default=0)
    p.add_argument("-n", "--no_OF", action="store_true", help="Disable optical flow compensation")
    p.add_argument("-m", "--memory_map", action="store_true", help="Enable memory-mapping (only for MRC files)")
    p.add_argument("-p", "--number_of_processes", type=int_or_str, default=os.cpu_count(),
                   help="Accepted for compatibility; slices are batched on the GPU instead of a process pool")
    p.add_argument("--recompute_flow", action="store_true", help="Disable the use of adjacent optical flow fields")
    p.add_argument("--show_fingerprint", action="store_true", help="Show a hash of this program")
    # src/flowdenoising_GPU.py:597-598: accepted so that its command lines keep working; the flow always runs on the GPU here
    p.add_argument("--use_GPU", action="store_true", help="Accepted for compatibility (the optical flow always runs on the GPU)")
    p.add_argument("--use_threads", action="store_true", help="Accepted for compatibility (no process pool here)")
    p.add_argument("--compat", choices=("seq", "par"), default="seq",
                   help="seq: flowdenoising_sequential.py semantics (mean-padded ends, levels 0); "
                        "par: flowdenoising.py semantics (wrap-around ends, levels 3, float32 TIFF)")
    p.add_argument("--chunk_slices", type=int, default=0,
                   help="Out-of-core mode: keep the volume on the host and process this many slices of a pass at a "
                        "time on the GPU (-1: as many as fit); for volumes larger than GPU memory")
    p.add_argument("--strict_order", action="store_true",
                   help="Run OpenCV's serial horizontal running sum in the box filter instead of the direct window sum "
                        "(the one f64 summation order in which the fast kernels differ from OpenCV, ~1e-16; about 20x slower)")
    # the two things a particular cv2 build may do differently from the modelled one: switches, so that matching a
    # given opencv-python is an option and not a new kernel
    p.add_argument("--opencv_fma", type=int, choices=(0, 1, 2), default=0,
                   help="How the pyramid's blur and resize multiply-adds round (levels > 0 only): 0 two roundings (OpenCV's scalar code), "
                        "1 fused, 2 fused on the vector body of a row (see --opencv_fma_lanes) and not on its tail")
    p.add_argument("--opencv_fma_lanes", type=int, default=8, help="SIMD lanes of --opencv_fma 2 (8: AVX2, 4: SSE / NEON, 16: AVX-512)")
    p.add_argument("--remap_model", type=int, choices=(0, 1), default=0,
                   help="cv2.remap on float maps: 0 the classic 1/32-pixel coordinate table, 1 unquantised float32 bilinear")
    p.add_argument("--device", type=int, default=0, help="GPU index")
    p.add_argument("--gpus", type=int, default=1, help="Shard the volume over this many GPUs of the node")
    return p


def _feedback(state):
    while True:
        logging.info(f"{state['stage']}")
        time.sleep(1)


def _run_single(args, vol, kernels, l, w, device, stats, as_float32, timing=None, wait_for=None):
    from . import _lib
    from .operators import _params, filter_3d_own_mean
    border = _lib.BORDER_WRAP if args.compat == "par" else _lib.BORDER_MEAN_PAD
    params = _params(l, w, use_of=not args.no_OF, border_mode=border, chained=not args.recompute_flow)
    if args.chunk_slices:
        from .streaming import filter_streamed
        if wait_for is not None:
            wait_for()
        if as_float32:
            vol = np.asarray(vol, dtype=np.float32)     # seq:517 (main() has usually converted a TIFF stack already: no second copy)
        return filter_streamed(vol, kernels, l, w, None if args.chunk_slices < 0 else args.chunk_slices,
                               use_of=not args.no_OF, border_mode=border, chained=not args.recompute_flow, device=device)
    # mean = vol.mean() (seq:420) and the statistics the reference logs / mrcfile writes: taken on the GPU
    from . import io as fio
    downcast = not fio.is_mrc_output(args.output) and args.compat != "par"     # seq:566-571's uint8 / uint16 TIFF, cast on the GPU
    shape = vol.shape
    tiff32 = args.compat == "par"

    def sink(dtype, st_out):        # the output file, written slab by slab while the rest of the result is still downloading
        stats["streamed"] = True
        return fio.VolumeWriter(args.output, shape, np.float32 if (fio.is_mrc_output(args.output) or tiff32) else dtype, st_out)
    mapped = None
    if fio.is_mrc_output(args.output) and not (os.path.exists(args.output) and os.path.exists(args.input) and os.path.samefile(args.input, args.output)):
        # an MRC output is mapped and made ready (pages faulted in, page-locked where the kernel allows) while the passes run:
        # the result then lands in the file's own pages in one copy (io.MappedMrcWriter; seq:558-564 writes after the last pass)
        def mapped():
            return fio.MappedMrcWriter(args.output, shape)
    return filter_3d_own_mean(vol, kernels, params, device, stats=stats, float32_semantics=as_float32, tiff_downcast=downcast,
                              timing=timing, sink=sink, wait_for=wait_for, mapped_out=mapped)


def _reserve(args, shape, dtype, Ks, l, w):
    """While the file is being read: create the GPU context and allocate every buffer the filter will use
    (fdn_reserve_3d) -- 25 GB of hipMalloc at configs[2], 0.4 s that would otherwise sit between upload and compute."""
    try:
        from . import _lib
        from . import io as fio
        from .operators import _params, handle, integer_semantics
        border = _lib.BORDER_WRAP if args.compat == "par" else _lib.BORDER_MEAN_PAD
        params = _params(l, w, use_of=not args.no_OF, border_mode=border, chained=not args.recompute_flow)
        if fio.is_mrc_input(args.input):          # an integer MRC keeps its dtype (seq:513): other kernels, other buffers
            params = integer_semantics(np.zeros((1, 1, 1), dtype=dtype), params)
        handle(args.device).reserve_3d(shape, Ks, params)
    except Exception as e:                        # an optimisation only: the real call reports real problems
        logging.debug(f"reserve: {e}")


def _gather_slice_rows(tr, rows, parts):
    """Per-slice rows (zlen, 4) of every rank -> (Z, 4) in slice order (slabs differ in length: padded for the all-gather)."""
    m = max(e - s_ for s_, e in parts)
    pad = np.zeros((m, 4), dtype=np.float64)
    pad[:rows.shape[0]] = rows
    got = tr.allgather_array(pad)
    return np.concatenate([got[r, :e - s_] for r, (s_, e) in enumerate(parts)])


def _run_sharded(args, shape, kernels, l, w, stats, job, wall):
    """One rank of `--gpus N` (flowdenoising_amd/launch.py started it; no PyTorch anywhere): reads ITS OWN Z-slab of the
    input file, runs fdn_filter_3d_sharded on the native transport (RCCL over xGMI with one GPU per rank, the shared-memory
    rehearsal transport when ranks share a GPU), and writes its slab of the result into the output file at the byte offset
    those slices have there -- no rank ever holds the whole volume.  The header statistics (seq:562-564) and seq:566-571's
    uint8 / uint16 decision come from per-slice reductions all-gathered in slice order: the file is the single-GPU run's,
    byte for byte.  The partition is src/flowdenoising.py:181-206's: near-equal contiguous chunks of target slices."""
    from . import _lib, launch
    from . import io as fio
    from .distributed import split
    from .operators import _download_into, _params, integer_semantics
    rank, world, local, rdv = job
    tr, device = launch.make_transport(rank, world, local, rdv)
    logging.info(f"rank {rank}: {tr.describe()}")
    try:
        return _run_sharded_rank(args, shape, kernels, l, w, stats, wall, tr, device, rank, world)
    except BaseException as e:      # noqa: BLE001 -- a reader, a writer or the filter failed on THIS rank: the others are told
        launch.report_failure(rdv, rank, f"{type(e).__name__}: {e}")      # before this rank leaves (they would wait in an
        tr.abort()                                                       # exchange or at a barrier until their timeout)
        raise
    finally:
        tr.close()
        if rank == 0:
            launch.remove_derived_rendezvous(rdv)


def _run_sharded_rank(args, shape, kernels, l, w, stats, wall, tr, device, rank, world):
    from . import _lib
    from . import io as fio
    from .distributed import split
    from .operators import _download_into, _params, integer_semantics
    h = _lib.Handle(device)
    Z, Y, X = shape
    parts = split(Z, world)
    z0, z1 = parts[rank]
    t0 = time.perf_counter()
    raw = fio.read_slab(args.input, z0, z1)
    as_float32 = not fio.is_mrc_input(args.input)          # seq:517: only an MRC keeps an integer dtype
    border = _lib.BORDER_WRAP if args.compat == "par" else _lib.BORDER_MEAN_PAD
    params = _params(l, w, use_of=not args.no_OF, border_mode=border, chained=not args.recompute_flow)
    if not as_float32:
        params = integer_semantics(raw, params, mean_on_device=True)
    raw_int = raw.dtype.kind in "iu" and raw.dtype.itemsize <= 2 and raw.dtype.isnative
    src = np.ascontiguousarray(raw) if raw_int else np.ascontiguousarray(raw, dtype=np.float32)
    n = src.size
    d_in = h.malloc(n * 4)
    d_out = h.malloc(n * 4)
    try:
        h.h2d(d_out if raw_int else d_in, src)             # raw integers into the (larger) output buffer first
        if raw_int:
            h.convert_dev(d_out, src.dtype, d_in, n)
        h.synchronize()
        wall["h2d"] = time.perf_counter() - t0
        verbose = logging.getLogger().isEnabledFor(logging.INFO)
        if verbose or params.pad64 != params.pad64:
            rows = _gather_slice_rows(tr, h.stats_slices_dev(d_in, z1 - z0, Y * X, 0.0), parts)
            st_in = _lib.combine_slice_stats(rows, Z * Y * X)
            if stats is not None:
                stats["in"] = st_in
            if params.pad64 != params.pad64:               # an integer MRC's float64 mean (seq:420): exact sums of integers
                params.pad64 = st_in["mean"]
        t0 = time.perf_counter()
        h.filter_3d_sharded(d_in, d_out, shape, kernels, params, tr)
        h.synchronize()
        wall["compute"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        mrc_out = fio.is_mrc_output(args.output)
        downcast = not mrc_out and args.compat != "par"    # seq:566-571's uint8 / uint16 TIFF, cast on the GPU
        first = _gather_slice_rows(tr, h.stats_slices_dev(d_out, z1 - z0, Y * X, 0.0), parts)
        st_out = _lib.combine_slice_stats(first, Z * Y * X)
        if mrc_out or verbose:
            second = _gather_slice_rows(tr, h.stats_slices_dev(d_out, z1 - z0, Y * X, st_out["mean"]), parts)
            st_out = _lib.combine_slice_stats(first, Z * Y * X, second)
        if stats is not None:
            stats["out"] = st_out
        d_res, dtype = d_out, np.dtype(np.float32)
        if downcast:
            dtype = np.dtype(np.uint8 if st_out["max"] < 256 else np.uint16)
            h.truncate_dev(d_out, dtype, d_in, n)          # the input's device copy is no longer needed
            d_res = d_in
        out = np.empty((z1 - z0, Y, X), dtype=dtype)
        if rank == 0:                                      # the file and its header exist before anybody else opens it
            writer = fio.VolumeWriter(args.output, shape, dtype, st_out, z0=0, create=True)
        tr.barrier()
        if rank != 0:
            writer = fio.VolumeWriter(args.output, shape, dtype, st_out, z0=z0, create=False)
        pin = out.nbytes >= (8 << 20) and h.host_register(out)
        try:
            _download_into(h, out, d_res, writer)
        finally:
            if pin:
                h.host_unregister(out)
        tr.barrier()                                       # every slab is in the file
        wall["d2h_write"] = time.perf_counter() - t0
        if stats is not None:
            stats["streamed"] = True
            stats["dtype"] = dtype
    finally:
        h.free(d_out)
        h.free(d_in)
        h.close()
    return None


def main(argv=None):
    parser = build_parser()
    args = parser.parse_args(argv)

    if args.strict_order:
        os.environ["FDN_STRICT_ORDER"] = "1"   # read by libflowdn.so on its first sweep
    if args.opencv_fma:
        os.environ["FDN_OPENCV_FMA"] = str(args.opencv_fma)             # read by every handle at fdn_create (the ranks of --gpus N inherit them)
        os.environ["FDN_OPENCV_FMA_LANES"] = str(args.opencv_fma_lanes)
    if args.remap_model:
        os.environ["FDN_REMAP_MODEL"] = str(args.remap_model)

    if args.show_fingerprint:  # par:425-431 hashes the script; here: the library that does the work
        from . import _lib
        h = hashlib.sha256()
        for fn in (os.path.abspath(__file__), _lib.LIB_PATH):
            with open(fn, "rb") as f:
                while chunk := f.read(1 << 16):
                    h.update(chunk)
        print("fingerprint =", h.hexdigest())

    level = {2: logging.DEBUG, 1: logging.INFO}.get(args.verbosity, logging.CRITICAL)  # seq:480-487
    logging.basicConfig(format=LOGGING_FORMAT, level=level)
    if args.verbosity in (1, 2):
        logging.info(f"Verbosity level = {args.verbosity}")

    from . import launch
    job = launch.job() if args.gpus > 1 else None
    sharded = job is not None
    if args.gpus > 1 and (args.chunk_slices or args.device):
        parser.error("--gpus shards the volume over GPUs 0..N-1 and keeps every slab resident: "
                     "it cannot be combined with --chunk_slices or --device")
    if args.gpus > 1 and not sharded:
        from . import io as fio
        try:
            hshape, _ = fio.volume_info(args.input)
        except Exception as e:
            parser.error(f"cannot read {args.input}: {e}")
        if args.gpus > min(hshape):
            parser.error(f"--gpus {args.gpus}: every axis of the volume must have at least one slice per rank, its shape is {tuple(hshape)}")
        # the parent: N rank processes of this same command, started with plain subprocess.Popen before anything here has
        # touched a GPU (never an exec); they meet through the native transport (include/flowdn_rccl.h) -- no PyTorch
        script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flowdenoising.py")
        env = dict(os.environ)
        env.setdefault("FDN_SYSTEM_ROCM", "1")     # the ranks never import torch: /opt/rocm's HIP runtime and RCCL
        return launch.spawn([sys.executable, script] + (sys.argv[1:] if argv is None else list(argv)), args.gpus, env=env)
    if sharded and job[1] != args.gpus:
        parser.error(f"--gpus {args.gpus} inside a job of {job[1]} ranks")
    rank = job[0] if sharded else 0

    sigma = [float(i) for i in args.sigma]
    logging.info(f"sigma={tuple(sigma)}")
    l = args.levels if args.levels is not None else (3 if args.compat == "par" else 0)
    w = args.winsize

    state = {"stage": "reading"}
    t = threading.Thread(target=_feedback, args=(state,), daemon=True)  # seq:489-491
    t.start()

    from . import io as fio
    from .operators import get_gaussian_kernel
    logging.info(f"reading \"{args.input}\"")
    t0 = time.perf_counter()
    if sharded:                      # header only: every rank reads its own slab in _run_sharded
        shape, dtype = fio.volume_info(args.input)
        vol = None
    else:
        prep = None
        if not args.chunk_slices:
            try:
                hshape, hdtype = fio.volume_info(args.input)        # header / page directory only
                Ks = [2 * int(4.0 * s + 0.5) + 1 for s in sigma[:3]]  # seq:30-41: taps of get_gaussian_kernel(sigma)
                prep = threading.Thread(target=_reserve, args=(args, hshape, hdtype, Ks, l, w), daemon=True)
                prep.start()
            except Exception:
                prep = None
        # an MRC is memory-mapped, never copied: its pages go from the page cache to the GPU (the reference reads it into a
        # fresh array, seq:513; -m asks for the map explicitly, seq:510-512); the buffers are being reserved meanwhile
        vol = fio.read_volume(args.input, mmap=args.memory_map or (prep is not None and fio.is_mrc_input(args.input)))
        # seq:517, par:475: a TIFF is float32 from here on; an MRC keeps its dtype (seq:513).  An 8- or 16-bit TIFF stack
        # stays as read and is converted on the GPU (the same values, a quarter or half of the bytes on the host and the wire)
        as_float32 = not fio.is_mrc_input(args.input)
        if as_float32 and not (vol.dtype.kind in "iu" and vol.dtype.itemsize <= 2 and not args.chunk_slices):
            vol = vol.astype(np.float32)
        shape, dtype = vol.shape, (np.dtype(np.float32) if as_float32 else vol.dtype)
    wall = {"read": time.perf_counter() - t0}      # phase record for tools/cli_wall.py (FDN_CLI_TIMING=<file>)
    logging.info(f"read \"{args.input}\" in {wall['read']} seconds")
    logging.info(f"shape of the input volume (Z, Y, X) = {shape}")
    logging.info(f"type of the volume = {dtype}")
    verbose = logging.getLogger().isEnabledFor(logging.INFO)       # seq:529-532 computes these eagerly; nobody reads them at -v 0
    stats = {}
    if vol is not None and verbose and args.chunk_slices:          # the streamed mode never holds the volume on the device
        logging.info(f"{args.input} max = {vol.max()}")
        logging.info(f"{args.input} min = {vol.min()}")
        logging.info(f"Input vol average = {vol.mean()}")

    kernels = [get_gaussian_kernel(sigma[0]), get_gaussian_kernel(sigma[1]), get_gaussian_kernel(sigma[2])]  # seq:534-537
    logging.info(f"length of each filter (Z, Y, X) = {[len(i) for i in kernels]}")

    state["stage"] = "filtering"
    t0 = time.perf_counter()
    if sharded:
        filtered = _run_sharded(args, shape, kernels, l, w, stats, job, wall)
    else:
        filtered = _run_single(args, vol, kernels, l, w, args.device, stats, as_float32, timing=wall,
                               wait_for=prep.join if prep is not None else None)
    wall["filter"] = time.perf_counter() - t0
    logging.info(f"Volume filtered in {wall['filter']} seconds")
    if rank != 0:
        _assert_no_torch()
        return 0
    if "in" in stats:            # seq:529-532, from the device copy
        logging.info(f"{args.input} max = {stats['in']['max']}")
        logging.info(f"{args.input} min = {stats['in']['min']}")
        logging.info(f"Input vol average = {stats['in']['mean']}")

    logging.info(f"shape of the denoised volume (Z, Y, X) = {filtered.shape if filtered is not None else tuple(shape)}")
    logging.info(f"{args.output} type = {filtered.dtype if filtered is not None else stats.get('dtype')}")
    if "out" in stats:           # seq:547-550
        logging.info(f"{args.output} max = {stats['out']['max']}")
        logging.info(f"{args.output} min = {stats['out']['min']}")
        logging.info(f"Output vol average = {stats['out']['mean']}")
    elif verbose:
        logging.info(f"{args.output} max = {filtered.max()}")
        logging.info(f"{args.output} min = {filtered.min()}")
        logging.info(f"Output vol average = {filtered.mean()}")
    state["stage"] = "writing"
    t0 = time.perf_counter()
    if not stats.get("streamed"):          # (the resident single-GPU path wrote the file while it downloaded the result)
        fio.write_volume(args.output, filtered, tiff_float32=(args.compat == "par"), stats=stats.get("out"))
    wall["write"] = time.perf_counter() - t0
    logging.info(f"written \"{args.output}\" in {wall['write']} seconds")
    if os.environ.get("FDN_CLI_TIMING"):
        import json
        with open(os.environ["FDN_CLI_TIMING"], "w") as f:
            json.dump(wall, f)
    _assert_no_torch()
    return 0


def _assert_no_torch():
    """FDN_ASSERT_NO_TORCH=1 (tests): the drop-in -- single GPU or `--gpus N` -- must finish without PyTorch ever having
    been imported into the process (BASELINE north_star: "PyTorch-ROCm is not needed here")."""
    if os.environ.get("FDN_ASSERT_NO_TORCH") == "1" and "torch" in sys.modules:
        raise AssertionError("torch was imported by the product path")


if __name__ == "__main__":
    sys.exit(main())
