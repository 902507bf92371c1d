// fdn_internal.h -- shared declarations between the C-ABI layer (fdn_api.hip) and the
// gfx950 kernels (fdn_kernels.hip).  Not installed; the public surface is include/flowdn.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace fdn {

// The "opencv_fma" option: how the multiply-adds of cv::GaussianBlur's two passes and of cv::resize's vertical pass round
// (the pyramid of levels > 0 only; level 0's blur taps are powers of two and fusing changes nothing there):
//   0  two roundings everywhere (the reading of OpenCV's scalar code; the default)
//   1  fused everywhere (a build whose vector loops use FMA and cover whole rows)
//   2  fused on the vector body of a row -- its first (width / lanes) * lanes elements --, two roundings on the tail
//      (v_muladd in the SIMD loop, plain C after it; lanes = 8 for AVX2, 4 for SSE / NEON, 16 for AVX-512)
struct FmaMode { int mode = 0; int lanes = 8; };

// Polynomial-expansion constants (OpenCV FarnebackPrepareGaussian), passed by value.
struct PolyConsts {
    int n;                       // half-width: taps -n..n
    float g[16], xg[16], xxg[16]; // index k = 0..n (g is even, xg odd, xxg even in k)
    double ig11, ig03, ig33, ig55;
};

void prepare_poly_consts(int n, double sigma, PolyConsts* pc);

// Runtime-tap Gaussian blur kernel (cv::getGaussianKernel, f32).  Level k of the pyramid uses
// sigma = (2^k - 1)/2 and ~5 sigma taps: 159 taps at level 6 (images of 2048 pixels and more).
constexpr int FDN_MAX_BLUR_TAPS = 191;
struct BlurTaps {
    int n;
    float k[FDN_MAX_BLUR_TAPS + 1];
};
void prepare_blur_taps(int n, double sigma, BlurTaps* bt);

// ---- launchers: all asynchronous on `st`, all pointers are device pointers ----------

// R[s] = polyexp(blur3x3(img[s])) for s in [0, nslices): img slices are H*W apart,
// R slices are 5*H*W floats apart, each in the RImage layout of fdn_device.h
// ([(c0,c1) x HW][(c2,c3) x HW][c4 x HW]).
void launch_blur3_polyexp(const float* img, float* R, int nslices, int H, int W,
                          const PolyConsts& pc, hipStream_t st);
// same without the fused 3x3 blur (pyramid levels: the input is already blurred+resized)
void launch_polyexp(const float* img, float* R, int nslices, int H, int W,
                    const PolyConsts& pc, hipStream_t st);

// Batched pair descriptors: for pair b in [0, npairs): target slice index t0 + b,
// neighbour slice index t0 + b + d, both indices into the R / image stack.
struct PairBatch {
    int npairs;
    int t0;      // stack index of the first target
    int d;       // neighbour offset (signed)
};

// M[b] = UpdateMatrices(R[t], R[n], flow[b])    (M planar 5 x H x W per pair; flow == nullptr: zero flow)
void launch_update_matrices(const float* Rstack, const float* flow, float* M, PairBatch pb,
                            int H, int W, hipStream_t st);
// flow[b] = solve(box_w(Min[b])); if Mout: Mout[b] = UpdateMatrices(R[t], R[n], flow[b])
// the same with OpenCV's serial horizontal running sum (strict_order); false: a row of W columns does not fit
// the LDS (strict_order_supported tells beforehand)
bool strict_order_supported(int W, int winsize);
bool launch_update_flow_strict(const float* Rstack, const float* Min, float* Mout, float* flow, PairBatch pb,
                               int H, int W, int winsize, hipStream_t st);
void launch_update_flow(const float* Rstack, const float* Min, float* Mout, float* flow,
                        PairBatch pb, int H, int W, int winsize, hipStream_t st);
// The whole side of a pass in one launch: for step s = 0 .. nsteps-1 (nearest neighbour first), neighbour
// stack[t0 + b + pb.d * (first_step + s + 1)] warped by flows[s][b] and folded into acc[b] with weights[s]; pb.d = -1 / +1.
// How a neighbour is sampled and folded in (fdn_sweep_params.warp_mode): kind 0 = float32 (seq / par on float data);
// 1 = the padded volume is float64 (seq on an integer MRC): remap weights in double, stack slices [0, pad_lo) and
// [pad_hi, ...) hold the float64 value pad64; 2 = the neighbour is an integer image (par on an integer MRC): remap's
// result is rounded half-to-even and saturated to [lo, hi].
// fixed8 (with kind 2): the neighbour is a uint8 image: remap in OpenCV's 8-bit fixed point instead of float + rounding.
// model (the "remap_model" option): 0 = cv2.remap's classic 1/32-pixel coordinate table, 1 = unquantised float32 bilinear.
struct WarpMode { int kind = 0; int pad_lo = 0, pad_hi = 1 << 30; double pad64 = 0.; float lo = 0.f, hi = 0.f; int fixed8 = 0; int model = 0; };
void launch_sweep_side(const float* stack, const float* flows, float* acc, PairBatch pb, int nsteps, int first_step,
                       int H, int W, const double* weights, hipStream_t st, const WarpMode& wm = WarpMode());
// acc[b] = f32( f64(acc[b]) + f64(stack[t0 + b + d]) * weight )   (centre tap, no-OF taps)
void launch_axpy_slices(const float* stack, float* acc, PairBatch pb, int H, int W,
                        double weight, hipStream_t st, const WarpMode& wm = WarpMode());
// v = trunc(v) clamped to [lo, hi]: a float32 result stored into an integer volume (par:131, par:287)
void launch_trunc_clamp(float* v, size_t count, float lo, float hi, hipStream_t st);
// dst(y,x) = remap(src, flow)  single image (fdn_warp)
void launch_warp(const float* src, const float* flow, float* dst, int H, int W, hipStream_t st, int model = 0);
// the same for a CV_8U image (values 0..255 held as floats): cv2.remap's 8-bit fixed-point interpolation
void launch_warp_u8(const float* src, const float* flow, float* dst, int H, int W, hipStream_t st);
// the same for a CV_64F image: double in, double out (cv2.remap's Cast<double, double> path)
void launch_warp_f64(const double* src, const float* flow, double* dst, int H, int W, hipStream_t st, int model = 0);

// Per-handle switches: read from the environment once, at fdn_create, and changed with fdn_set_option
// (tests and experiments; every path gives the same bits except strict_order).
struct Tuning {
    int strict_order = 0;    // FDN_STRICT_ORDER: OpenCV's serial horizontal running sum in the box filter
    int path = 0;            // FDN_PATH: 0 = automatic, 1 = staged per-stage kernels (also FDN_FORCE_STAGED=1),
                             //           2 = one-iteration kernels (k_farneback_iter) whatever the window
    int fused_occ = 0;       // FDN_FUSED_OCC: pin the 3-iteration fused kernel's occupancy build (3, 4, 5)
    unsigned lds_pad = 0;    // FDN_LDS_PAD: extra dynamic LDS per workgroup (occupancy curves)
    int cus = 256;           // compute units of the handle's device
    int shard_loopback = 0;  // fdn_filter_3d_sharded: the blocks a rank keeps also travel through the transport (send to self)
    int sub_batches = 0;     // FDN_SUB_BATCHES: 0 = automatic (two when a launch of the pass is under four rounds of workgroup slots),
                             // 1 = one stream, 2 = the target slices of a batch as two independent sub-batches on two streams
    FmaMode fma;             // "opencv_fma" / "opencv_fma_lanes" (FDN_OPENCV_FMA, FDN_OPENCV_FMA_LANES)
    int remap_model = 0;     // "remap_model" (FDN_REMAP_MODEL): WarpMode::model of every warp
    long occ_blocks = 0;     // set by the sweep while sub-batches run: the workgroups of BOTH sub-batches' launches, which is
                             // what the occupancy choice of the 3-iteration kernel goes by (0: the launch's own grid)
};

// Fused chain step (fdn_fused.hip): for every pair of the batch, the whole level-0 Farneback
// (initial matrices + iters x [box filter, solve, refresh]) seeded by flow_in (NULL = zero
// flow), then acc += weight * remap(stack[n], flow); flow_out (NULL = not needed) receives
// the final flow for the next chain step.  flow_in and flow_out must be different buffers.
bool fused_supported(int winsize, int iters, int H, int W);
void launch_farneback_fused(const float* Rstack, const float* stack, const float* flow_in, float* flow_out,
                            float* acc, PairBatch pb, int H, int W, int winsize, int iters, double weight,
                            hipStream_t st, const Tuning& tn, int coarse_h = 0, int coarse_w = 0, const WarpMode& wm = WarpMode());

// One Farneback iteration per launch, any window (fdn_iter.hip): flow_out = solve(box_w(UpdateMatrices(flow_in)));
// acc != nullptr: also acc += weight * remap(stack[n], flow_out) (the last iteration of the finest level; flow_out may
// then be nullptr).  flow_in nullptr = zero flow; coarse_h/w > 0: flow_in is the next coarser level's flow of that size.
// Returns 0, or -1 when the launch could not be configured.
bool iter_supported(int winsize, int H, int W);
size_t iter_lds_bytes(int mh, bool acc);
int launch_farneback_iter(const float* Rstack, const float* stack, const float* flow_in, float* flow_out, float* acc,
                          PairBatch pb, int H, int W, int winsize, double weight, hipStream_t st,
                          int coarse_h = 0, int coarse_w = 0, const WarpMode& wm = WarpMode(), const FmaMode& fm = FmaMode());

// where the fused kernel's initial flow comes from when it is the next coarser pyramid level's result
struct FlowSource { int h, w; double sx, sy; FmaMode fm; };   // h == 0: flow_in has the image's own size; fm: how the upsampling rounds

void launch_fill(float* dst, float value, size_t count, hipStream_t st);
// out[r][c] (contiguous H x W) = in[r * rs + c * cs]  (strides in elements; a slice view of a volume)
void launch_copy_strided(const float* in, int64_t rs, int64_t cs, float* out, int H, int W, hipStream_t st);
// out[a][b][c] = in[a*sa + b*sb + c*sc]; out element (a, b, c) at out + a*oa + b*ob + c (oa = ob = 0: contiguous)
void launch_permute(const float* in, float* out, int A, int B, int C, int64_t sa, int64_t sb,
                    int64_t sc, hipStream_t st, int64_t oa = 0, int64_t ob = 0);
// partial sums (f64) into `partials` (nblocks entries); returns nblocks used
// sums[c] = numpy's float32 pairwise sum of in[8192 c .. 8192 c + 8191]
void launch_np_chunk_sums(const float* in, size_t nchunks, float* sums, hipStream_t st);
int launch_stats_partials(const float* in, size_t count, double centre, double* partials, int max_blocks, hipStream_t st);
// per slice {min, max, sum, sum of squared deviations from centre}: FDN_STATS_BLOCKS_PER_SLICE partials of 4 doubles per slice
constexpr int FDN_STATS_BLOCKS_PER_SLICE = 16;
void launch_stats_slices(const float* in, int nslices, size_t slice_elems, double centre, double* partials, hipStream_t st);
void launch_convert_f32(const void* in, int depth, float* out, size_t count, hipStream_t st);
void launch_truncate_from_f32(const float* in, int depth, void* out, size_t count, hipStream_t st);
int launch_sum_partials(const float* in, size_t count, double* partials, int max_blocks,
                        hipStream_t st);

// pyramid pieces (levels > 0)
void launch_gaussian_blur(const float* in, float* tmp, float* out, int nimg, int H, int W,
                          const BlurTaps& bt, hipStream_t st, const FmaMode& fm = FmaMode());
// small = resize(GaussianBlur(in, bt), (dw, dh), INTER_LINEAR) with the blur evaluated only where the resize reads it
// (a strict shrink: dw < W, dh < H); tmp: nimg * H * 2 dw floats
void launch_blur_resize(const float* in, float* tmp, float* small, int nimg, int H, int W, int dh, int dw,
                        const BlurTaps& bt, hipStream_t st, const FmaMode& fm = FmaMode());
// cv::resize as FarnebackOpticalFlowImpl::calc uses it, nimg images of cn interleaved
// channels: interp 1 = INTER_LINEAR (an exact 2x2 shrink is promoted to area, as cv::resize
// does), 3 = INTER_AREA (integer ratios on this path).  If apply_ps, each result is then
// multiplied by ps in f64 (the "flow *= scale" of calc).
void resize_images(const float* in, int sh, int sw, float* out, int dh, int dw, int cn, int nimg,
                   int interp, bool apply_ps, double ps, hipStream_t st, const FmaMode& fm = FmaMode());

// true when resize_images can do this (interp, size pair) itself; INTER_AREA with a non-integer
// shrink ratio needs the table form below
bool resize_needs_tables(int sh, int sw, int dh, int dw, int interp);
void resize_area_tab(const float* in, int sh, int sw, float* out, int dh, int dw, int cn, int nimg,
                     const int* x_si, const float* x_alpha, const int* x_start,
                     const int* y_si, const float* y_alpha, const int* y_start, bool apply_ps, double ps,
                     hipStream_t st);

} // namespace fdn
