/*
 * flowdn.h -- C ABI of libflowdn.so: the MI355X (gfx950) implementation of
 * FlowDenoising's hot path (Farneback optical flow between neighbouring slices +
 * flow-warped separable 3-D Gaussian).
 *
 * The reference has no FFI layer of its own: its hot path is reached through Python
 * callables that forward to OpenCV.  Each entry point below names the reference
 * interface it replaces ("seq" = src/flowdenoising_sequential.py, "par" =
 * src/flowdenoising.py, "gpu" = src/flowdenoising_GPU.py in the reference tree).
 *
 * Conventions
 *   - plain C symbols, `int` status: 0 = ok, negative = error; text via fdn_last_error()
 *     (thread-local).  No exceptions cross the boundary.
 *   - volumes are (Z, Y, X) row-major float32, X fastest (numpy C order, seq:514).
 *   - "_dev" entry points take DEVICE pointers owned by the caller (hipMalloc'd, or a
 *     torch tensor's data_ptr()); the others take HOST pointers and stage through
 *     library-owned device memory.  All work is enqueued on the handle's HIP stream;
 *     host-pointer calls return after the result has been copied back.
 *   - thread safety: every entry point that takes a handle holds the handle's own lock for the whole call, so ONE
 *     handle may be shared by any number of host threads -- par calls its pair operators from P pool threads at once
 *     (par:187-193, 299-327); their calls then run one after the other on the handle's stream, each with the handle's
 *     staging buffers to itself.  Threads that want their calls to OVERLAP use a handle each (handles are independent;
 *     device-wide runtime operations -- allocation, stream and event creation -- are serialised across the handles of a
 *     process by the library).  fdn_destroy waits for a call in flight; no call may START on a handle once fdn_destroy
 *     has been called for it.  fdn_last_error() is per thread.  The `fdn_comm` callbacks of fdn_filter_3d_sharded are
 *     called with the handle's lock held: they must not call back into the same handle from another thread and wait.
 */
#ifndef FLOWDN_H
#define FLOWDN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fdn_ctx* fdn_handle;

/* cv2.OPTFLOW_USE_INITIAL_FLOW (seq:62).  cv2.OPTFLOW_FARNEBACK_GAUSSIAN (256) is never
 * set by the reference and is rejected. */
#define FDN_USE_INITIAL_FLOW 4

/* border handling of the axis sweeps */
#define FDN_BORDER_MEAN_PAD 0 /* seq:88-89: K slices of `pad_value` around the volume */
#define FDN_BORDER_WRAP 1     /* par:312: neighbour index modulo the axis length      */

/* parameters of one axis sweep; mirrors the arguments threaded through
 * OF_filter_along_* (seq:78) and the module constants seq:43-49 */
typedef struct fdn_sweep_params {
    int levels;      /* -l, OF_LEVELS (seq:44)                                  */
    int winsize;     /* -w, OF_WINDOW_SIZE (seq:45)                             */
    int iters;       /* OF_ITERS = 3 (seq:46)                                   */
    int poly_n;      /* OF_POLY_N = 5 (seq:47)                                  */
    double poly_sigma; /* OF_POLY_SIGMA = 1.2 (seq:48)                          */
    int border_mode; /* FDN_BORDER_*                                            */
    int chained;     /* 1: previous flow seeds the next (seq:97-98);
                        0: --recompute_flow, zero initial flow (par:89-114)     */
    int use_of;      /* 0: -n/--no_OF plain separable Gaussian (seq:426-431)    */
    /* Integer volumes.  Both reference programs keep an integer MRC's dtype (seq:513, par:472), and what cv2.remap
     * and numpy then do differs from the float32 case (all zero = float32 semantics):
     *   FDN_WARP_F64_PADDED (seq): vol.mean() is a float64 (seq:420), so np.full makes the padded volume float64
     *     (seq:88-89) in all three passes: cv2.remap weights its taps in double and does not round to float32, and
     *     pad slices hold the float64 mean `pad64` (Farneback still sees float32 images: pad_value = (float)pad64);
     *   FDN_WARP_ROUND_INT (par): the neighbour slices ARE integer images: cv2.remap's result is rounded half to
     *     even and saturated to the type's range [round_lo, round_hi], and every pass's result is truncated toward
     *     zero into the integer volume (par:131, par:287-289);
     *   FDN_WARP_FIXED_U8 (par, uint8 volume): the same, except that cv2.remap interpolates 8-bit images in fixed point
     *     (16-bit integer weights, (sum + 2^14) >> 15); range [0, 255].
     * The caller converts the volume to float32 (exact for 8/16-bit types) and says which it was. */
    int warp_mode;   /* FDN_WARP_*                                              */
    int pad_lo, pad_hi; /* fdn_sweep_stack_dev with FDN_WARP_F64_PADDED: how many leading / trailing stack slices are
                        pad slices (the other entry points know where they padded and ignore these)            */
    double pad64;    /* FDN_WARP_F64_PADDED: the float64 mean                    */
    double round_lo, round_hi; /* FDN_WARP_ROUND_INT: e.g. -32768, 32767 for mode-1 MRC */
} fdn_sweep_params;
#define FDN_WARP_F32 0
#define FDN_WARP_F64_PADDED 1
#define FDN_WARP_ROUND_INT 2
#define FDN_WARP_FIXED_U8 3   /* par on a uint8 volume: cv2.remap interpolates 8-bit images in fixed point (16-bit integer weights
                                 = table weights x 2^15, (sum + 2^14) >> 15); every pass truncated into [0, 255] (par:131, 287-289) */

/* ---- lifetime ------------------------------------------------------------------ */
/* Binds a handle to HIP device `device` and creates its stream.  Replaces the implicit
 * "OpenCV is loaded" state of the reference; gpu:92-103 (GPU_flower.__init__) is the
 * closest reference analogue. */
int fdn_create(int device, fdn_handle* out);
/* HIP devices visible to this process (0 when there is none or the runtime cannot start): what a rank of a multi-GPU run
 * compares with the number of ranks to choose between one GPU each (RCCL) and sharing (include/flowdn_rccl.h). */
int fdn_device_count(int* count_out);
/* PCI bus id of device `device` ("0000:05:00.0"): what tells ranks on one GPU from ranks on GPUs of their own when a
 * launcher has narrowed each rank's view to one device (HIP_VISIBLE_DEVICES). */
int fdn_device_pci_id(int device, char* buf, int cap);
int fdn_destroy(fdn_handle h);
const char* fdn_last_error(void);
/* Enqueue on an external HIP stream instead (e.g. torch.cuda.current_stream().cuda_stream, so that
 * library kernels and torch/RCCL ops order on one stream).  NULL means the legacy default stream.
 * fdn_reset_stream goes back to the handle's own (non-blocking) stream. */
int fdn_set_stream(fdn_handle h, void* hip_stream);
int fdn_reset_stream(fdn_handle h);
int fdn_synchronize(fdn_handle h);
/* Cap on ALL device memory the handle owns (stacks, re-oriented pass outputs, intermediate volumes, polynomial
 * expansions, flows, matrices); 0 = no cap: whole passes at once, the flow batch bounded by the free device memory less
 * a reserve of max(1/16 of the device, 6 GiB) that is never taken; a buffer more than four times (and 1 GiB) larger
 * than its next use is given back.
 * With a cap every pass is cut into chunks of target slices (each with its K/2 halo slices either side) that fit;
 * results do not depend on it, bit for bit.  Too small a cap for one target slice is an error, not a fallback.
 * Setting it releases what the handle holds.  fdn_workspace_bytes reports what it holds now. */
int fdn_set_workspace_limit(fdn_handle h, size_t bytes);
int fdn_workspace_bytes(fdn_handle h, size_t* bytes_out);
/* free and total device memory (hipMemGetInfo): what the out-of-core mode sizes its chunks by */
int fdn_mem_info(fdn_handle h, size_t* free_out, size_t* total_out);

/* Switches of a live handle (tests and experiments; fdn_create reads the same from the environment:
 * FDN_STRICT_ORDER, FDN_PATH / FDN_FORCE_STAGED, FDN_FUSED_OCC, FDN_LDS_PAD, FDN_SUB_BATCHES, FDN_OPENCV_FMA,
 * FDN_OPENCV_FMA_LANES, FDN_REMAP_MODEL).  Every path gives the same
 * bits except strict_order:
 *   "strict_order" 0/1  OpenCV's serial f64 running sum along x in FarnebackUpdateFlow_Blur instead of the
 *                       direct window sum (about 20x slower; errors out, never falls back, when a row does
 *                       not fit the LDS)
 *   "path"         0 automatic, 1 one kernel per Farneback stage, 2 one kernel per iteration
 *   "fused_occ"    0 automatic, 3..5 one-band workgroups per CU of the 3-iteration fused kernel, 8 two bands per
 *                  8-wave workgroup (two workgroups per CU)
 *   "lds_pad"      bytes of extra dynamic LDS per workgroup of that kernel (occupancy curves)
 *   "shard_loopback" 0/1  fdn_filter_3d_sharded: the blocks a rank keeps for itself travel through the transport too
 *                  (a send to self inside the group; the mean through allgather_host even with one rank), so that
 *                  ONE rank on one GPU issues the calls of an N > 1 run
 *   "sub_batches"  0 automatic, 1 one stream, 2 two: the target slices of a batch are independent (seq:92), so a pass whose
 *                  launches fill the GPU's workgroup slots only a few times over runs them as two halves, each the whole
 *                  chain of both sides on a stream of its own -- the tail of one half's launch is filled by the other
 *                  half's work (the reference's remainder round, par:194-206, keeps its workers busy the same way).
 *                  Automatic: two when a launch of the pass, at any pyramid level, is under four rounds of slots.
 *   "opencv_fma"   how the multiply-adds of cv::GaussianBlur's two passes and of cv::resize's vertical pass round -- pyramid levels
 *                  >= 1 only: 0 (default) two roundings, the reading of OpenCV's scalar code; 1 fused everywhere; 2 fused on the
 *                  vector body of a row, its first (width / lanes) * lanes elements, and not on the tail ("opencv_fma_lanes", 8 =
 *                  AVX2).  opencv-python is unpinned in the reference and absent here: which of these a given wheel
 *                  does is unknown, and at levels > 0 it moves results by up to 2e-3 -- the day a cv2 is at hand the matching
 *                  reading is a switch, on the oracle (fdo_set_fma) and here, bit-equal to each other in every mode.
 *   "remap_model"  cv2.remap on a float map: 0 (default) the classic path, coordinates rounded to 1/32 pixel and weights from
 *                  the 32 x 32 table; 1 unquantised float32 bilinear interpolation (two lerps), a model of the reworked remap
 *                  of newer OpenCV releases; 8-bit images keep their fixed-point table either way.  tests/test_cv2_pin.py tells
 *                  the two apart on a real cv2.
 * The first group changes speed only (same bits); "opencv_fma" and "remap_model" change results: they select which cv2 is matched.
 * No counterpart in the reference (cv2 has no such switches). */
int fdn_set_option(fdn_handle h, const char* name, long value);
/* Reads an option back; also the read-only "last_sub_batches" (what the last sweep ran with) and "compute_units". */
int fdn_get_option(fdn_handle h, const char* name, long* value_out);

/* ---- device memory helpers (so that a host program needs no other HIP binding) ----- */
int fdn_malloc(fdn_handle h, size_t bytes, void** dptr);
int fdn_free(fdn_handle h, void* dptr);
int fdn_memcpy_h2d(fdn_handle h, void* dst_dev, const void* src_host, size_t bytes);
int fdn_memcpy_d2h(fdn_handle h, void* dst_host, const void* src_dev, size_t bytes);
int fdn_memset_f32(fdn_handle h, float* dst_dev, float value, size_t count);
/* strided copies: `height` rows of `width_bytes`, row r at base + r * pitch (a slab volume[:, y0:y1, :] or
 * volume[:, :, x0:x1] of a host volume moves without a host-side gather; the out-of-core mode uses them) */
int fdn_memcpy2d_h2d(fdn_handle h, void* dst_dev, size_t dst_pitch, const void* src_host, size_t src_pitch,
                     size_t width_bytes, size_t height);
int fdn_memcpy2d_d2h(fdn_handle h, void* dst_host, size_t dst_pitch, const void* src_dev, size_t src_pitch,
                     size_t width_bytes, size_t height);
/* How host memory travels.  fdn_memcpy_h2d / _d2h / fdn_memcpy2d_* and the host-pointer entry points (fdn_filter_3d, ...)
 * never hand PAGEABLE memory to the copy engines: memory that was page-locked through fdn_host_register moves in one DMA;
 * a contiguous block of 8 MB or more is page-locked by the library for the duration of the call; everything else goes
 * through a page-locked bounce buffer of the handle (one host memcpy more).  The reason is in the HIP runtime: a copy of
 * pageable memory above about 1 MB is page-locked by the runtime on the fly, that registration outlives the call, and a
 * later copy that meets one whose pages the application has freed in the meantime ends in a GPU memory fault and an abort
 * of the process (caught in round 6; profiles/history/NOTES_r06.md, section 2). */
/* page-lock / release a host buffer of the caller (hipHostRegister): copies from and to it then run at PCIe speed.
 * Release it BEFORE the memory is freed.  A registration belongs to the process: fdn_host_unregister accepts any handle,
 * and NULL. */
int fdn_host_register(fdn_handle h, void* ptr, size_t bytes);
int fdn_host_unregister(fdn_handle h, void* ptr);

/* ---- a-1  get_gaussian_kernel(sigma)  (seq:30-41, par:34-45) ---------------------- */
/* Writes K = 2*int(4*sigma+0.5)+1 float64 taps; returns K, or -K if cap < K. Host only. */
int fdn_gaussian_kernel(double sigma, double* out, int cap);

/* ---- a-2/a-3  cv2.calcOpticalFlowFarneback as called by get_flow (seq:59-67,
 *      par:65-114, gpu:155-177) ------------------------------------------------------ */
/* prev/next: H x W float32, contiguous HOST images; flow: H x W x 2 float32 HOST, read as
 * the initial flow when flags & FDN_USE_INITIAL_FLOW and overwritten with the result
 * (cv2 updates `flow` in place, seq:98).  pyr_scale is fixed at 0.5 (seq:62). */
int fdn_farneback(fdn_handle h, const float* prev, const float* next, float* flow_inout,
                  int H, int W, int levels, int winsize, int iters, int poly_n,
                  double poly_sigma, int flags);

/* The same with the images given as VIEWS, as the reference passes them: padded_vol[z + i] (seq:97),
 * padded_vol[:, y + i, :] (seq:255: row stride Y*X), padded_vol[:, :, x + i] (seq:333: row stride Y*X, column
 * stride X).  Pixel (r, c) of an image is base[r * row_stride + c * col_stride]; strides in ELEMENTS.  flow is
 * contiguous.  The views are gathered by the library into its own pinned staging buffer: no copy on the caller's
 * side (numpy hands over arr.ctypes.data and arr.strides). */
int fdn_farneback_strided(fdn_handle h, const float* prev, ptrdiff_t prev_row_stride, ptrdiff_t prev_col_stride,
                          const float* next, ptrdiff_t next_row_stride, ptrdiff_t next_col_stride,
                          float* flow_inout, int H, int W, int levels, int winsize, int iters, int poly_n,
                          double poly_sigma, int flags);
/* ... and on DEVICE memory (views of a volume already in HBM; the flow stays on the device between the calls of
 * a chain, seq:97-98): no host round trip, nothing synchronises. */
int fdn_farneback_dev(fdn_handle h, const float* d_prev, ptrdiff_t prev_row_stride, ptrdiff_t prev_col_stride,
                      const float* d_next, ptrdiff_t next_row_stride, ptrdiff_t next_col_stride,
                      float* d_flow_inout, int H, int W, int levels, int winsize, int iters, int poly_n,
                      double poly_sigma, int flags);

/* ---- a-4  warp_slice(reference, flow)  (seq:51-57, par:55-63) --------------------- */
/* dst(y,x) = cv2.remap(reference, float32(flow + grid), INTER_LINEAR, BORDER_REPLICATE):
 * 1/32-pixel quantised bilinear gather.  HOST pointers, contiguous. */
int fdn_warp(fdn_handle h, const float* reference, const float* flow, float* dst, int H, int W);
/* reference as a view (strides in elements, as above); flow and dst contiguous */
int fdn_warp_strided(fdn_handle h, const float* reference, ptrdiff_t row_stride, ptrdiff_t col_stride,
                     const float* flow, float* dst, int H, int W);
/* DEVICE pointers; d_dst must not alias the reference */
int fdn_warp_dev(fdn_handle h, const float* d_reference, ptrdiff_t row_stride, ptrdiff_t col_stride,
                 const float* d_flow, float* d_dst, int H, int W);

/* ---- the pair operators on images that are not float32 ------------------------------------------------------
 * The reference hands cv2 whatever its arrays are: slices of an integer MRC (par:312, dtype of the file) or of the
 * float64 padded volume seq:88-89 builds from one.  cv2.calcOpticalFlowFarneback converts both images to float32
 * (convertTo(CV_32F)); cv2.remap computes per depth and returns the image's type: CV_64F weights in double without
 * rounding to float, CV_16S / CV_16U in float and then rounds half to even and saturates; CV_8U in fixed point (16-bit integer
 * weights, (sum + 2^14) >> 15); CV_8S is not supported by cv2.remap (error here too).  HOST pointers; strides
 * in elements of the image's type; flow (H x W x 2 float32) and dst (H x W, the reference's type) contiguous. */
#define FDN_DEPTH_F32 0
#define FDN_DEPTH_F64 1
#define FDN_DEPTH_I16 2
#define FDN_DEPTH_U16 3
#define FDN_DEPTH_I8 4
#define FDN_DEPTH_U8 5
int fdn_farneback_typed(fdn_handle h, const void* prev, int prev_depth, ptrdiff_t prev_row_stride, ptrdiff_t prev_col_stride,
                        const void* next, int next_depth, ptrdiff_t next_row_stride, ptrdiff_t next_col_stride,
                        float* flow_inout, int H, int W, int levels, int winsize, int iters, int poly_n, double poly_sigma, int flags);
int fdn_warp_typed(fdn_handle h, const void* reference, int depth, ptrdiff_t row_stride, ptrdiff_t col_stride,
                   const float* flow, void* dst, int H, int W);

/* ---- a-5/6/7/10  OF_filter_along_{Z,Y,X} (seq:78-130, 235-288, 313-364),
 *      no_OF_filter_along_* (seq:171-192, 290-311, 396-417),
 *      FlowDenoising.filter_along_*_slice (par:306-373) ----------------------------- */
/* axis: 0 = Z, 1 = Y, 2 = X.  kernel: K float64 taps (K odd).  pad_value: the volume mean
 * of seq:420 (ignored for FDN_BORDER_WRAP).  in/out must not alias. */
int fdn_filter_axis_dev(fdn_handle h, const float* d_in, float* d_out, int Z, int Y, int X,
                        int axis, const double* kernel, int K, float pad_value,
                        const fdn_sweep_params* p);
int fdn_filter_axis(fdn_handle h, const float* in, float* out, int Z, int Y, int X,
                    int axis, const double* kernel, int K, float pad_value,
                    const fdn_sweep_params* p);

/* ---- a-8/a-9  OF_filter / no_OF_filter (seq:419-431); GaussianDenoising.filter
 *      (par:285-290) ------------------------------------------------------------------ */
/* Z pass, then Y, then X, each consuming the previous output; kernels[a] == NULL (or
 * K[a] == 0) skips axis a.  pad_value is ONE scalar for all passes (seq:420). */
/* Allocate every device buffer fdn_filter_3d_dev will use for a volume of this shape, these tap counts and parameters,
 * and launch nothing: a caller that knows the shape before it has the voxels (a file header, seq:508-517) runs this
 * while it reads the file, so that the first real call does not pay for 25 GB of hipMalloc (0.4 s at configs[2]). */
int fdn_reserve_3d(fdn_handle h, int Z, int Y, int X, const int K[3], const fdn_sweep_params* p);
int fdn_filter_3d_dev(fdn_handle h, const float* d_in, float* d_out, int Z, int Y, int X,
                      const double* const kernels[3], const int K[3], float pad_value,
                      const fdn_sweep_params* p);
int fdn_filter_3d(fdn_handle h, const float* in, float* out, int Z, int Y, int X,
                  const double* const kernels[3], const int K[3], float pad_value,
                  const fdn_sweep_params* p);

/* vol.mean() of seq:420 for a contiguous float32 HOST volume, bit-identical to numpy's (float32
 * pairwise summation).  The padded borders make the result sensitive to the last bit of this
 * value, so use it (or numpy itself) whenever bit-faithfulness to the reference matters. */
int fdn_mean_host(const float* in, size_t count, float* mean_out);
/* the same mean of a DEVICE volume, equally bit-identical to numpy's: the pairwise sums of the
 * 8192-element chunks are formed on the GPU in numpy's order, their left-to-right float32
 * accumulation on the host */
int fdn_mean_dev(fdn_handle h, const float* d_in, size_t count, float* mean_out);
/* those chunk sums themselves: sums_out (HOST, ceil(count / 8192) floats; the last chunk may be partial).
 * Ranks whose slabs start at multiples of 8192 elements concatenate them to get numpy's mean of the
 * whole volume exactly. */
int fdn_np_chunk_sums_dev(fdn_handle h, const float* d_in, size_t count, float* sums_out);
/* sum only (float64): the fallback for the multi-GPU mean when slabs do not start at chunk boundaries
 * (at most 1 ulp from numpy's value) */
int fdn_sum_dev(fdn_handle h, const float* d_in, size_t count, double* sum_out);
/* {min, max, mean, standard deviation} of a DEVICE float32 array in float64 (two passes): the header statistics
 * mrcfile's set_data computes for the output file (seq:562-564: dmin, dmax, dmean, rms) and what seq:529-532 / 547-550
 * log about the input and output volumes -- taken where the volume already is instead of by numpy on the host
 * (six reductions over 2 GiB cost the reference's CLI about 2 s). */
int fdn_stats_dev(fdn_handle h, const float* d_in, size_t count, double* out4);   /* a NaN voxel makes min and max NaN, as numpy's do */
/* The same per slice, in a form that does not depend on how the volume is split into slabs: out[4 s ..] = {min, max, sum,
 * sum of squared deviations from `centre`} of slice s (`slice_elems` floats each), every slice reduced in one fixed order.
 * The caller adds the slices up in order (pass 1: centre = 0 -> the mean; pass 2: centre = mean -> the rms).  A rank of a
 * multi-GPU run computes its slab's slices, the ranks all-gather them, and the header of the output file (seq:562-564)
 * is the single-GPU one bit for bit. */
int fdn_stats_slices_dev(fdn_handle h, const float* d_in, int nslices, size_t slice_elems, double centre, double* out);
/* d_dst[i] = (float)d_src[i] for an 8- or 16-bit integer DEVICE array (depth: FDN_DEPTH_I16 / U16 / I8 / U8):
 * seq:517's `vol.astype(np.float32)` of a TIFF stack, and the device copy of an integer MRC, without a float32 copy
 * on the host (a 2048 x 2048 x 512 uint16 stack is 2 GiB on the wire instead of 8). */
int fdn_convert_dev(fdn_handle h, const void* d_src, int depth, float* d_dst, size_t count);
/* the reverse for the TIFF output of seq:566-571 (`filtered.astype(np.uint8)` if max < 256 else `np.uint16`):
 * d_dst[i] = (uint8 / uint16) d_src[i], C's truncating cast as numpy performs it (depth FDN_DEPTH_U8 / U16);
 * d_dst must not overlap d_src. */
int fdn_truncate_dev(fdn_handle h, const float* d_src, int depth, void* d_dst, size_t count);

/* ---- the sharded filter below the ABI (SURVEY 8b put multi-GPU under fdn_filter_3d) -------------------------
 * OF_filter / no_OF_filter (seq:419-431) of a volume held as Z-slabs, one rank per GPU: `d_slab_in` is this rank's slab
 * of the near-equal contiguous split of Z over comm->world ranks (rank r holds slices [r Z / world ...): base = Z / world
 * slices, the first Z % world ranks one more -- src/flowdenoising.py:181-206 splits its chunks the same way), `d_slab_out`
 * receives the filtered slab of the same partition.  Every pass shards along its own axis: Z-slabs -> Z pass -> Y-slabs
 * -> Y pass -> X-slabs -> X pass -> Z-slabs, ONE exchange per pass delivering partition and K//2 halos together; volume
 * ends are padded with the global mean (seq:88, seq:420: numpy's float32 mean of the whole volume, assembled from the
 * ranks' chunk sums) or wrap around (par:312).  The library owns schedule, packing (fdn_permute_dev's kernels) and the
 * passes; the CALLER owns the communicator and supplies two callbacks:
 *   exchange        one batched group of point-to-point messages on DEVICE buffers, all of them at once: with RCCL
 *                   ncclGroupStart(); ncclSend / ncclRecv per message on `stream`; ncclGroupEnd();  with MPI
 *                   hipStreamSynchronize(stream), MPI_Isend / MPI_Irecv per message, MPI_Waitall.  Every rank calls it
 *                   the same number of times; a rank with nothing to move in a round passes n = 0.  A message to
 *                   oneself occurs only under the "shard_loopback" test switch (fdn_set_option).  Return 0 on success.
 *   allgather_host  every rank contributes `bytes` from `send` (HOST memory); `recv` (HOST, world * bytes) receives all
 *                   contributions in rank order (ncclAllGather on a staging buffer, MPI_Allgather, ...).
 * libflowdn_rccl.so (include/flowdn_rccl.h) implements both callbacks natively: RCCL over xGMI, a shared-memory
 * rehearsal transport for ranks that share a GPU, and a null transport for per-rank overhead measurements.
 * The Python engine flowdenoising_amd/distributed.py (torch.distributed; the gloo rehearsals of the CPU tests) implements
 * the same schedule above the ABI; the two give the same bits.  Integer-volume modes (fdn_sweep_params.warp_mode) are supported with pad64 given by the
 * caller; otherwise the mean is computed here. */
typedef struct fdn_msg {
    void* d_buf;        /* DEVICE memory of this rank */
    size_t bytes;
    int peer;           /* the other rank */
    int is_send;        /* 1: send d_buf to peer, 0: receive from peer into d_buf */
} fdn_msg;
typedef struct fdn_comm {
    void* ctx;
    int rank, world;
    int (*exchange)(void* ctx, int n, const fdn_msg* msgs, void* stream);
    int (*allgather_host)(void* ctx, const void* send, void* recv, size_t bytes);
} fdn_comm;
int fdn_filter_3d_sharded(fdn_handle h, const float* d_slab_in, float* d_slab_out, int Z, int Y, int X,
                          const double* const kernels[3], const int K[3], const fdn_sweep_params* p,
                          const fdn_comm* comm);

/* ---- slab primitives (multi-GPU decomposition; the reviewer variant
 *      tests/flowdenoising_reviewer_solution2.py:496-508 keeps "chunk + kernel.size"
 *      slices resident the same way) ------------------------------------------------- */
/* Sweep along the OUTER axis of a stack of S + 2*(K/2) images (H x W each): stack slice
 * s + K/2 is target s; the K/2 slices either side of the interior are halos the caller
 * has filled (neighbour data, pad value, or wrapped copies).  Writes S images to d_out. */
int fdn_sweep_stack_dev(fdn_handle h, const float* d_stack, float* d_out, int S, int H, int W,
                        const double* kernel, int K, const fdn_sweep_params* p);
/* Allocate what fdn_sweep_stack_dev will use for a stack of this shape, tap count and parameters, and launch nothing (the
 * fdn_reserve_3d of the slab primitive): the out-of-core mode sizes every worker's buffers before its first chunk, so that
 * no chunk waits for gigabytes of hipMalloc. */
int fdn_reserve_stack(fdn_handle h, int S, int H, int W, int K, const fdn_sweep_params* p);
/* out[a][b][c] = in[a*sa + b*sb + c*sc]; out is contiguous with dims (A,B,C); strides in
 * elements.  Used for the Z->Y->X slab re-orientation between passes. */
int fdn_permute_dev(fdn_handle h, const float* d_in, float* d_out, int A, int B, int C,
                    int64_t sa, int64_t sb, int64_t sc);

/* ---- timers (the taxonomy of gpu:47-53: OF estimation / warping+convolution /
 *      transfers) ---------------------------------------------------------------------- */
#define FDN_TIMER_POLYEXP 0          /* blur + polynomial expansion, once per slice per pass   */
#define FDN_TIMER_UPDATE_MATRICES 1  /* initial FarnebackUpdateMatrices of a chain step         */
#define FDN_TIMER_UPDATE_FLOW 2      /* box filter + 2x2 solve (+ matrix refresh), per iteration */
#define FDN_TIMER_WARP 3             /* warp + accumulate (and the no-OF taps)                  */
#define FDN_TIMER_PERMUTE 4          /* slab re-orientation / halo fill                         */
#define FDN_TIMER_TRANSFER 5         /* H2D / D2H                                               */
#define FDN_TIMER_FUSED 6            /* fused Farneback chain-step kernel (fast path)           */
#define FDN_TIMER_ITER 7             /* one-iteration Farneback kernel (wide windows)            */
#define FDN_TIMER_COLLECTIVE 8       /* multi-GPU exchanges: fdn_filter_3d_sharded times its fdn_comm.exchange calls on the stream
                                        (time waiting for the slowest peer included); a host-level engine adds its own (fdn_add_timer) */
#define FDN_TIMER_MEAN 9             /* fdn_filter_3d_sharded: the global mean (seq:420) from the ranks' chunk sums                     */
#define FDN_TIMER_CHAINS 10         /* a batch's complete chains of both sides as ONE span on the handle's stream, sub-batch fork to join:
                                        with sub-batches the launches of the two streams overlap and their own times add up to more
                                        than the wall clock -- this span is what the batch took (bench.py prices the roofline on it) */
#define FDN_TIMER_COUNT 11
/* HIP-event timing of the phases above on the handle's stream.  Event pairs are recorded
 * asynchronously (no host sync inside the timed work) and resolved by fdn_get_timers, which
 * returns accumulated milliseconds and the number of timed launches per category. */
int fdn_enable_timers(fdn_handle h, int on);
int fdn_get_timers(fdn_handle h, double* ms_out /* FDN_TIMER_COUNT */,
                   long long* count_out /* FDN_TIMER_COUNT */, int reset);
/* Adds time measured outside the library to a category (the slab engine reports its RCCL exchanges as
 * FDN_TIMER_COLLECTIVE, so that one table holds the whole taxonomy of gpu:47-53 plus the collectives). */
int fdn_add_timer(fdn_handle h, int which, double ms, long long count);

/* library/version string, e.g. "flowdn 0.1 gfx950" */
const char* fdn_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FLOWDN_H */
