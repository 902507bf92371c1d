// fdn_kernels.hip -- gfx950 (MI355X, wave64) kernels of the FlowDenoising hot path.
//
// General ("staged") path: one kernel per Farneback stage, batched over all target
// slices of a sweep step.  Works for any winsize / image size and is what the pyramid
// levels and the pair-level entry points use.  The fused fast path for the default
// configuration lives in fdn_fused.hip.
//
// Arithmetic follows OpenCV's operand types expression by expression (see
// oracle/fdn_oracle.c for the restatement and the reference call sites,
// src/flowdenoising_sequential.py:56,62); the file is compiled with -ffp-contract=off so
// that the compiler does not fuse multiplies and adds the CPU code keeps separate.
// Layouts: images [H][W] f32; polynomial expansion R: channel pairs (0,1), (2,3) interleaved +
// channel 4 planar (RImage, fdn_device.h); matrices M: 5 planes of [H][W]; flow: interleaved
// (x,y) float2 as in cv2.
#include "../../include/flowdn.h"
#include "fdn_internal.h"
#include "fdn_device.h"

namespace fdn {

// ---------------------------------------------------------------------------------
// Polynomial expansion (FarnebackPolyExp), optionally fused with the level-0 3x3
// binomial blur (GaussianBlur(ksize 3, sigma 0) -> taps .25 .5 .25, reflect-101).
// Block = 256 threads, output tile 64 x 16; LDS: blurred tile with halo n, then the
// three vertical-pass rows.
// ---------------------------------------------------------------------------------
constexpr int PE_TW = 64, PE_TH = 16, PE_MAXN = 7;

static __device__ __forceinline__ float blur3_at(const float* __restrict__ img, int H, int W, int cy, int cx)
{
    int xl = reflect101(cx - 1, W), xr = reflect101(cx + 1, W);
    float t[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        int yy = reflect101(cy + r - 1, H);
        const float* S = img + (size_t)yy * W;
        t[r] = S[cx] * 0.5f + (S[xl] + S[xr]) * 0.25f;
    }
    return 0.5f * t[1] + 0.25f * (t[2] + t[0]);
}

// NT: the neighbourhood half-width when known at compile time (5: the reference's poly_n; loops unroll,
// index divisions become multiplications), 0: taken from pc.n.
template <bool FUSE_BLUR3, int NT>
__global__ __launch_bounds__(256) void k_polyexp(const float* __restrict__ img_base, float* __restrict__ R_base,
                                                 int H, int W, PolyConsts pc)
{
    __shared__ float sB[(PE_TH + 2 * PE_MAXN) * (PE_TW + 2 * PE_MAXN)];
    __shared__ float sRow[3][PE_TH][PE_TW + 2 * PE_MAXN];
    const int n = NT ? NT : pc.n;
    const int LW = PE_TW + 2 * n, LH = PE_TH + 2 * n;
    const size_t HW = (size_t)H * W;
    const float* img = img_base + (size_t)blockIdx.z * HW;
    float* R = R_base + (size_t)blockIdx.z * 5 * HW;
    const int x0 = blockIdx.x * PE_TW, y0 = blockIdx.y * PE_TH;

    // (staging the raw pixels of interior tiles in LDS -- one coalesced load per element instead of nine per blurred
    // value -- changes nothing: 16.0 -> 16.6 ms per step; the kernel is bound by its f64 horizontal pass)
    for (int idx = threadIdx.x; idx < LW * LH; idx += 256) {
        int ty = idx / LW, tx = idx - ty * LW;
        int cy = clampi(y0 - n + ty, 0, H - 1), cx = clampi(x0 - n + tx, 0, W - 1);
        sB[ty * LW + tx] = FUSE_BLUR3 ? blur3_at(img, H, W, cy, cx) : img[(size_t)cy * W + cx];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < LW * PE_TH; idx += 256) {
        int ry = idx / LW, tx = idx - ry * LW;
        const float* c = sB + (ry + n) * LW + tx;
        float r0 = c[0] * pc.g[0], r1 = 0.f, r2 = 0.f;
#pragma unroll
        for (int k = 1; k <= n; k++) {
            float s0 = c[-k * LW], s1 = c[k * LW];
            float p = s0 + s1;
            r0 = r0 + pc.g[k] * p;
            r1 = r1 + pc.xg[k] * (s1 - s0);
            r2 = r2 + pc.xxg[k] * p;
        }
        sRow[0][ry][tx] = r0; sRow[1][ry][tx] = r1; sRow[2][ry][tx] = r2;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < PE_TW * PE_TH; idx += 256) {
        int ry = idx / PE_TW, ox = idx - ry * PE_TW;
        int x = x0 + ox, y = y0 + ry;
        if (x >= W || y >= H) continue;
        const float* a0 = &sRow[0][ry][ox + n];
        const float* a1 = &sRow[1][ry][ox + n];
        const float* a2 = &sRow[2][ry][ox + n];
        float g0 = pc.g[0];
        double b1 = (double)(a0[0] * g0), b2 = 0, b3 = (double)(a1[0] * g0), b4 = 0, b5 = (double)(a2[0] * g0), b6 = 0;
#pragma unroll
        for (int k = 1; k <= n; k++) {
            float gk = pc.g[k], xgk = pc.xg[k], xxgk = pc.xxg[k];
            double tg = (double)(a0[k] + a0[-k]);
            b1 += tg * (double)gk;
            b4 += tg * (double)xxgk;
            b2 += (double)((a0[k] - a0[-k]) * xgk);
            b3 += (double)((a1[k] + a1[-k]) * gk);
            b6 += (double)((a1[k] - a1[-k]) * xgk);
            b5 += (double)((a2[k] + a2[-k]) * gk);
        }
        size_t o = (size_t)y * W + x;
        // channel pairs (0,1), (2,3) interleaved, channel 4 planar (RImage, fdn_device.h)
        ((float2*)R)[o] = make_float2((float)(b3 * pc.ig11), (float)(b2 * pc.ig11));
        ((float2*)(R + 2 * HW))[o] = make_float2((float)(b1 * pc.ig03 + b5 * pc.ig33), (float)(b1 * pc.ig03 + b4 * pc.ig33));
        R[4 * HW + o] = (float)(b6 * pc.ig55);
    }
}

void launch_blur3_polyexp(const float* img, float* R, int nslices, int H, int W, const PolyConsts& pc, hipStream_t st)
{
    if (nslices <= 0) return;
    dim3 grid((W + PE_TW - 1) / PE_TW, (H + PE_TH - 1) / PE_TH, nslices);
    if (pc.n == 5) hipLaunchKernelGGL((k_polyexp<true, 5>), grid, dim3(256), 0, st, img, R, H, W, pc);
    else hipLaunchKernelGGL((k_polyexp<true, 0>), grid, dim3(256), 0, st, img, R, H, W, pc);
}
void launch_polyexp(const float* img, float* R, int nslices, int H, int W, const PolyConsts& pc, hipStream_t st)
{
    if (nslices <= 0) return;
    dim3 grid((W + PE_TW - 1) / PE_TW, (H + PE_TH - 1) / PE_TH, nslices);
    if (pc.n == 5) hipLaunchKernelGGL((k_polyexp<false, 5>), grid, dim3(256), 0, st, img, R, H, W, pc);
    else hipLaunchKernelGGL((k_polyexp<false, 0>), grid, dim3(256), 0, st, img, R, H, W, pc);
}

__global__ __launch_bounds__(256) void k_update_matrices(const float* __restrict__ Rstack, const float* __restrict__ flow_base,
                                                         float* __restrict__ M_base, PairBatch pb, int H, int W)
{
    const size_t HW = (size_t)H * W;
    const int b = blockIdx.z;
    const RImage R0 = r_image(Rstack + (size_t)(pb.t0 + b) * 5 * HW, HW);
    const RImage R1 = r_image(Rstack + (size_t)(pb.t0 + b + pb.d) * 5 * HW, HW);
    const float2* flow = flow_base ? (const float2*)flow_base + (size_t)b * HW : nullptr;   // nullptr: zero initial flow
    float* M = M_base + (size_t)b * 5 * HW;
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    size_t o = (size_t)y * W + x;
    float m[5];
    float2 f = flow ? flow[o] : make_float2(0.f, 0.f);
    compute_M(R0, R1, H, W, x, y, f.x, f.y, m);
#pragma unroll
    for (int c = 0; c < 5; c++) M[c * HW + o] = m[c];
}

void launch_update_matrices(const float* Rstack, const float* flow, float* M, PairBatch pb, int H, int W, hipStream_t st)
{
    if (pb.npairs <= 0) return;
    dim3 grid((W + 63) / 64, (H + 3) / 4, pb.npairs);
    hipLaunchKernelGGL(k_update_matrices, grid, dim3(256), 0, st, Rstack, flow, M, pb, H, W);
}

// ---------------------------------------------------------------------------------
// FarnebackUpdateFlow_Blur, column-streaming form.
//
// OpenCV keeps, per column and channel, a vertical RUNNING sum in f64 that is fed by f32
// row differences: vsum += (float)(M[y+m] - M[y-m-1]).  The f32 rounding of those
// differences random-walks down the image and, through the ill-conditioned 2x2 solve,
// moves the flow by up to ~1e-4 relative on ordinary data -- so a tile-local box sum
// cannot meet the 1e-4 parity bar.  Here one wave owns a band of 64 columns and marches
// down the rows carrying vsum[5] in registers, which reproduces that arithmetic exactly.
// The horizontal (2m+1)-sum is taken across lanes with wave shuffles in f64 (OpenCV's own
// horizontal running sum is f64 too; the two differ by ~1e-16 relative).
//   band b covers columns [b*BW - m, b*BW - m + 64), BW = 64 - 2m; lanes [m, 64-m) own outputs;
//   out-of-image columns act as replicas of the clamped column (BORDER_REPLICATE of vsum).
// Rows may be split into segments for parallelism on small batches: a segment first
// replays the cheap vertical recurrence from row 0 so that its running sum is the same.
// ---------------------------------------------------------------------------------
// vsum before row 0: M[0]*(m+2) (an f32 product) + rows 1..m-1 (clamped)
static __device__ __forceinline__ void vsum_init(const float* __restrict__ Mc, size_t HW, int H, int W, int xc, int m, double vs[5])
{
#pragma unroll
    for (int c = 0; c < 5; c++) {
        const float* p = Mc + c * HW + xc;
        double v = (double)(p[0] * (float)(m + 2));
        for (int y = 1; y < m; y++) v += (double)p[(size_t)(y < H - 1 ? y : H - 1) * W];
        vs[c] = v;
    }
}

__global__ __launch_bounds__(256) void k_update_flow_scan(const float* __restrict__ Rstack, const float* __restrict__ Min_base,
                                                          float* __restrict__ Mout_base, float* __restrict__ flow_base,
                                                          PairBatch pb, int H, int W, int m, double scale,
                                                          int nbands, int rows_per_seg)
{
    // per-wave exchange buffer for the horizontal window (8-byte LDS reads: ~2 cycles each on the LDS
    // pipe against 2 x 24 for a ds_bpermute pair, and a 15-wide window needs 70 of them per row)
    __shared__ double xch[4][5][64];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int band = blockIdx.x * 4 + wv;
    if (band >= nbands) return;               // whole wave leaves; no block-level sync below
    const int BW = 64 - 2 * m;
    const int x = band * BW - m + lane;
    const int xc = clampi(x, 0, W - 1);
    const size_t HW = (size_t)H * W;
    const int b = blockIdx.z;
    const float* Min = Min_base + (size_t)b * 5 * HW;
    const int ys = blockIdx.y * rows_per_seg;
    const int ye = ys + rows_per_seg < H ? ys + rows_per_seg : H;

    double vs[5];
    vsum_init(Min, HW, H, W, xc, m, vs);
    for (int y = 0; y < ys; y++) { // replay of the vertical recurrence for rows above the segment
        const float* p1 = Min + (size_t)(y + m < H - 1 ? y + m : H - 1) * W + xc;
        const float* p0 = Min + (size_t)(y - m - 1 > 0 ? y - m - 1 : 0) * W + xc;
#pragma unroll
        for (int c = 0; c < 5; c++) vs[c] += (double)(p1[c * HW] - p0[c * HW]);
    }
    const bool owner = lane >= m && lane < 64 - m && x < W;
    const RImage R0 = r_image(Rstack + (size_t)(pb.t0 + b) * 5 * HW, HW);
    const RImage R1 = r_image(Rstack + (size_t)(pb.t0 + b + pb.d) * 5 * HW, HW);
    float2* flow = (float2*)flow_base + (size_t)b * HW;
    float* Mout = Mout_base ? Mout_base + (size_t)b * 5 * HW : nullptr;

    for (int y = ys; y < ye; y++) {
        const float* p1 = Min + (size_t)(y + m < H - 1 ? y + m : H - 1) * W + xc;
        const float* p0 = Min + (size_t)(y - m - 1 > 0 ? y - m - 1 : 0) * W + xc;
        double a[5];
#pragma unroll
        for (int c = 0; c < 5; c++) {
            vs[c] += (double)(p1[c * HW] - p0[c * HW]);
            xch[wv][c][lane] = vs[c];
        }
        // only this wave reads what it wrote and a wave's LDS operations execute in order; the fences keep the
        // compiler from moving the reads across the writes (other lanes' addresses), as in k_update_flow_scan_t
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int jlo = lane - m < 0 ? 0 : lane - m, jhi = lane + m > 63 ? 63 : lane + m;
#pragma unroll
        for (int c = 0; c < 5; c++) {
            // the 2m+1 terms left to right, starting from the first (not from 0.0: the oracle's order)
            double s = xch[wv][c][jlo];
            for (int j = lane - m + 1; j < jlo; j++) s += xch[wv][c][0];  // clamped lanes left of the wave
            for (int j = (lane - m < jlo ? jlo : jlo + 1); j <= jhi; j++) s += xch[wv][c][j];
            for (int j = jhi + 1; j <= lane + m; j++) s += xch[wv][c][63]; // and right of it
            a[c] = s;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    // next row's writes stay behind these reads
        __builtin_amdgcn_wave_barrier();
        if (owner) {
            float2 f = solve_flow(a, scale);
            size_t o = (size_t)y * W + x;
            flow[o] = f;
            if (Mout) {
                float mm[5];
                compute_M(R0, R1, H, W, x, y, f.x, f.y, mm);
#pragma unroll
                for (int c = 0; c < 5; c++) Mout[c * HW + o] = mm[c];
            }
        }
    }
}

// The same kernel with the window half-width known at compile time: the 2M+1 LDS reads of a row
// become immediate-offset ds_read2_b64 pairs.  Lanes outside [M, 64-M) only feed other lanes'
// windows (their own results are never stored), so their out-of-range terms need no clamping:
// the exchange row is padded by M entries either side and they read whatever is there.
template <int M>
__global__ __launch_bounds__(256) void k_update_flow_scan_t(const float* __restrict__ Rstack, const float* __restrict__ Min_base,
                                                            float* __restrict__ Mout_base, float* __restrict__ flow_base,
                                                            PairBatch pb, int H, int W, double scale,
                                                            int nbands, int rows_per_seg)
{
    __shared__ double xch[4][5][64 + 2 * M];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int band = blockIdx.x * 4 + wv;
    if (band >= nbands) return;               // whole wave leaves; no block-level sync below
    constexpr int BW = 64 - 2 * M;
    const int x = band * BW - M + lane;
    const int xc = clampi(x, 0, W - 1);
    const size_t HW = (size_t)H * W;
    const int b = blockIdx.z;
    const float* Min = Min_base + (size_t)b * 5 * HW;
    const int ys = blockIdx.y * rows_per_seg;
    const int ye = ys + rows_per_seg < H ? ys + rows_per_seg : H;

    double vs[5];
    vsum_init(Min, HW, H, W, xc, M, vs);
    for (int y = 0; y < ys; y++) { // replay of the vertical recurrence for rows above the segment
        const float* p1 = Min + (size_t)(y + M < H - 1 ? y + M : H - 1) * W + xc;
        const float* p0 = Min + (size_t)(y - M - 1 > 0 ? y - M - 1 : 0) * W + xc;
#pragma unroll
        for (int c = 0; c < 5; c++) vs[c] += (double)(p1[c * HW] - p0[c * HW]);
    }
    const bool owner = lane >= M && lane < 64 - M && x < W;
    const RImage R0 = r_image(Rstack + (size_t)(pb.t0 + b) * 5 * HW, HW);
    const RImage R1 = r_image(Rstack + (size_t)(pb.t0 + b + pb.d) * 5 * HW, HW);
    float2* flow = (float2*)flow_base + (size_t)b * HW;
    float* Mout = Mout_base ? Mout_base + (size_t)b * 5 * HW : nullptr;

    for (int y = ys; y < ye; y++) {
        const float* p1 = Min + (size_t)(y + M < H - 1 ? y + M : H - 1) * W + xc;
        const float* p0 = Min + (size_t)(y - M - 1 > 0 ? y - M - 1 : 0) * W + xc;
        double a[5];
#pragma unroll
        for (int c = 0; c < 5; c++) {
            vs[c] += (double)(p1[c * HW] - p0[c * HW]);
            xch[wv][c][lane + M] = vs[c];
        }
        // only this wave reads what it wrote and a wave's LDS operations execute in order; the fences
        // keep the compiler from moving the reads across the writes (other lanes' addresses)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 5; c++) {
            double s = xch[wv][c][lane];      // the 2M+1 terms left to right, starting from the first
#pragma unroll
            for (int k = 1; k <= 2 * M; k++) s += xch[wv][c][lane + k];
            a[c] = s;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (owner) {
            float2 f = solve_flow(a, scale);
            size_t o = (size_t)y * W + x;
            flow[o] = f;
            if (Mout) {
                float mm[5];
                compute_M(R0, R1, H, W, x, y, f.x, f.y, mm);
#pragma unroll
                for (int c = 0; c < 5; c++) Mout[c * HW + o] = mm[c];
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// Strict mode (FDN_STRICT_ORDER=1): FarnebackUpdateFlow_Blur with OpenCV's HORIZONTAL running sum
// too -- g += vsum[x+m] - vsum[x-m-1], one serial f64 chain along each row -- instead of the
// window summed directly.  The two differ by f64 rounding only (~1e-16), which the near-singular
// solve can turn into a different f32 flow once in ~10^10 pixels; this kernel exists to show that
// this is the only difference: with it the GPU reproduces the OpenCV-order oracle bit for bit.
// One workgroup per pair marches down the rows: all threads update the row's vertical running sums
// in LDS, five threads (one per channel) walk the row serially, all threads solve.  Slow by design.
// The row is walked in segments of S columns -- the chain's running value stays in its thread's register from one
// segment to the next, the window sums of a segment are solved before the chain moves on -- so that the LDS holds the
// row of vertical sums (40 B per column) plus ONE segment of window sums: rows up to about 3 800 columns fit
// (configs[4]'s 2048-pixel rows in two segments; until round 4 the whole row's window sums were kept: W <= 2040).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_update_flow_strict(const float* __restrict__ Rstack, const float* __restrict__ Min_base,
                                                            float* __restrict__ Mout_base, float* __restrict__ flow_base,
                                                            PairBatch pb, int H, int W, int m, double scale, int S)
{
    extern __shared__ double sh[];
    const int P = W + 2 * (m + 1);               // a row of vsum with m+1 replicated entries either side
    double* vs = sh;                              // [5][P], entry x + m + 1 = column x
    double* G = sh + 5 * (size_t)P;               // [5][S]: the window sums of the segment being solved
    const size_t HW = (size_t)H * W;
    const int b = blockIdx.x;
    const float* Mp = Min_base + (size_t)b * 5 * HW;
    const RImage R0 = r_image(Rstack + (size_t)(pb.t0 + b) * 5 * HW, HW);
    const RImage R1 = r_image(Rstack + (size_t)(pb.t0 + b + pb.d) * 5 * HW, HW);
    float2* flow = (float2*)flow_base + (size_t)b * HW;
    float* Mout = Mout_base ? Mout_base + (size_t)b * 5 * HW : nullptr;

    for (int x = threadIdx.x; x < W; x += 256) {
        double v[5];
        vsum_init(Mp, HW, H, W, x, m, v);
#pragma unroll
        for (int c = 0; c < 5; c++) vs[c * P + x + m + 1] = v[c];
    }
    __syncthreads();
    for (int y = 0; y < H; y++) {
        const float* p1 = Mp + (size_t)(y + m < H - 1 ? y + m : H - 1) * W;
        const float* p0 = Mp + (size_t)(y - m - 1 > 0 ? y - m - 1 : 0) * W;
        for (int x = threadIdx.x; x < W; x += 256) {
#pragma unroll
            for (int c = 0; c < 5; c++) vs[c * P + x + m + 1] += (double)(p1[c * HW + x] - p0[c * HW + x]);
        }
        __syncthreads();
        double g = 0.;                            // the chain's running value (threads 0..4: one channel each), kept across segments
        if (threadIdx.x < 5) {
            double* v = vs + threadIdx.x * P + m + 1;
            for (int k = 1; k <= m + 1; k++) { v[-k] = v[0]; v[W - 1 + k] = v[W - 1]; }
            g = v[0] * (double)(m + 2);
            for (int x = 1; x < m; x++) g += v[x];
        }
        for (int x0 = 0; x0 < W; x0 += S) {
            const int x1 = x0 + S < W ? x0 + S : W;
            if (threadIdx.x < 5) {                // the serial chain of one channel, columns [x0, x1)
                const double* v = vs + threadIdx.x * P + m + 1;
                double* out = G + threadIdx.x * (size_t)S - x0;
                for (int x = x0; x < x1; x++) {
                    g += v[x + m] - v[x - m - 1];
                    out[x] = g;
                }
            }
            __syncthreads();
            for (int x = x0 + threadIdx.x; x < x1; x += 256) {
                double a[5];
#pragma unroll
                for (int c = 0; c < 5; c++) a[c] = G[c * (size_t)S + (x - x0)];
                const float2 f = solve_flow(a, scale);
                const size_t o = (size_t)y * W + x;
                flow[o] = f;
                if (Mout) {
                    float mm[5];
                    compute_M(R0, R1, H, W, x, y, f.x, f.y, mm);
#pragma unroll
                    for (int c = 0; c < 5; c++) Mout[c * HW + o] = mm[c];
                }
            }
            __syncthreads();                      // the segment's window sums are consumed before the chain overwrites them;
        }                                         // and the next row's vsum update comes after this row's last chain read
    }
}

// LDS: the row of vertical sums, and window sums for as long a segment as the rest of the 160 KB holds
static int strict_segment(int W, int winsize)
{
    const int m = winsize / 2;
    const long P = (long)W + 2 * (m + 1);
    const long room = (160 * 1024) / (5 * (long)sizeof(double)) - P;      // doubles per channel left for a segment
    return (int)(room < W ? room : W);
}
// (a segment shorter than the row is worth its two barriers only from 64 columns up; a narrow row is one segment)
bool strict_order_supported(int W, int winsize) { const int S = strict_segment(W, winsize); return S >= (W < 64 ? W : 64) && S >= 1; }

// returns false when the row does not fit the LDS (the caller reports the error: strict mode never falls back silently)
bool launch_update_flow_strict(const float* Rstack, const float* Min, float* Mout, float* flow, PairBatch pb,
                               int H, int W, int winsize, hipStream_t st)
{
    if (pb.npairs <= 0) return true;
    const int m = winsize / 2;
    const int S = strict_segment(W, winsize);
    if (!strict_order_supported(W, winsize)) return false;
    const size_t bytes = (5 * (size_t)(W + 2 * (m + 1)) + 5 * (size_t)S) * sizeof(double);
    // per launch: the attribute belongs to the device's code object and handles may sit on different devices
    if (hipFuncSetAttribute((const void*)k_update_flow_strict, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
    const double scale = 1. / ((double)winsize * winsize);
    hipLaunchKernelGGL(k_update_flow_strict, dim3(pb.npairs), dim3(256), bytes, st, Rstack, Min, Mout, flow, pb, H, W, m, scale, S);
    return true;
}

void launch_update_flow(const float* Rstack, const float* Min, float* Mout, float* flow, PairBatch pb,
                        int H, int W, int winsize, hipStream_t st)
{
    if (pb.npairs <= 0) return;
    int m = winsize / 2;
    double scale = 1. / ((double)winsize * winsize);
    int BW = 64 - 2 * m;
    int nbands = (W + BW - 1) / BW;
    // enough waves to fill 256 CUs x 8 waves; otherwise split rows (each segment replays the
    // vertical recurrence above it, so keep segments few)
    long waves = (long)nbands * pb.npairs;
    int nseg = 1;
    if (waves < 4096) nseg = (int)((4096 + waves - 1) / waves);
    int max_seg = (H + 31) / 32;
    if (nseg > max_seg) nseg = max_seg;
    int rows_per_seg = (H + nseg - 1) / nseg;
    nseg = (H + rows_per_seg - 1) / rows_per_seg;
    dim3 grid((nbands + 3) / 4, nseg, pb.npairs);
    // compile-time window for the usual sizes (winsize 15: BASELINE configs[4]); others take the general kernel
#define FDN_SCAN_T(MM) hipLaunchKernelGGL(k_update_flow_scan_t<MM>, grid, dim3(256), 0, st, Rstack, Min, Mout, flow, pb, H, W, scale, nbands, rows_per_seg)
    switch (m) {
    case 1: FDN_SCAN_T(1); break;
    case 3: FDN_SCAN_T(3); break;
    case 4: FDN_SCAN_T(4); break;
    case 5: FDN_SCAN_T(5); break;
    case 7: FDN_SCAN_T(7); break;
    default:
        hipLaunchKernelGGL(k_update_flow_scan, grid, dim3(256), 0, st, Rstack, Min, Mout, flow, pb, H, W, m, scale, nbands, rows_per_seg);
    }
#undef FDN_SCAN_T
}

// ---------------------------------------------------------------------------------
// warp_slice: map = f32(f64(flow) + grid) (numpy's f32 + int64 promotion), then
// cv2.remap INTER_LINEAR / BORDER_REPLICATE: coordinates rounded half-even to 1/32 px,
// weights (1-a/32)(1-b/32)..., four clamped taps, f32 left-to-right sum.
// ---------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------
// The warped-Gaussian sweep of one side of a pass as ONE kernel (flow 8 + neighbour 4 bytes per pixel
// and pair, the accumulator read and written once): for every target of the batch, the `nsteps` neighbours on one
// side in the reference's order (nearest first, seq:95 / seq:110), each warped by its own flow (seq:106) and folded
// into the accumulator, acc = f32(f64(acc) + f64(v) w) (seq:107) -- the accumulator stays in a register.
// flows: [nsteps][npairs][H][W][2] (the flows of a chain are all kept by the per-stage path).
// Loads of U steps are issued together (flows, then their 4 taps each) before the serial fold.
// ---------------------------------------------------------------------------------
constexpr int SWEEP_MAX_STEPS = 48;
struct SweepWeights { double w[SWEEP_MAX_STEPS]; };

template <int U, int MODE>
__global__ __launch_bounds__(256) void k_sweep_side(const float* __restrict__ stack, const float* __restrict__ flows,
                                                    float* __restrict__ acc_base, PairBatch pb, int nsteps, int first_step,
                                                    int H, int W, SweepWeights sw, WarpMode wm)
{
    const size_t HW = (size_t)H * W;
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t o = (size_t)y * W + x;
    float* acc = acc_base + (size_t)b * HW;
    float a = acc[o];
    const size_t step_stride = (size_t)pb.npairs * HW;     // float2 elements between consecutive steps' flows
    const float2* fl = (const float2*)flows + (size_t)b * HW + o;
    const int dir = pb.d;                                   // -1: the back side, +1: the forward side
    const bool unq = wm.model == 1 && !(MODE == 2 && wm.fixed8);     // the "remap_model" option (8-bit images keep their fixed-point table)
    for (int s0 = 0; s0 < nsteps; s0 += U) {
        float2 f[U];
        RemapTaps t[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int s = s0 + u < nsteps ? s0 + u : nsteps - 1;
            f[u] = fl[(size_t)s * step_stride];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int s = s0 + u < nsteps ? s0 + u : nsteps - 1;
            const float* src = stack + (size_t)(pb.t0 + b + dir * (first_step + s + 1)) * HW;
            remap_issue<true>(src, H, W, x, y, f[u], t[u], unq);
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            if (s0 + u < nsteps) {
                if (MODE == 1) {          // float64 padded volume (seq on integer input)
                    const int q = pb.t0 + b + dir * (first_step + s0 + u + 1);
                    a = (float)((double)a + remap_finish_f64(t[u], q < wm.pad_lo || q >= wm.pad_hi, wm.pad64, unq) * sw.w[s0 + u]);
                } else if (MODE == 2) {   // integer neighbour image (par on integer input): saturate_cast<T>(float) = cvRound, clamped
                    const float v = wm.fixed8 ? remap_finish_u8(t[u]) : fminf(fmaxf(rintf(remap_finish(t[u], unq)), wm.lo), wm.hi);
                    a = (float)((double)a + (double)v * sw.w[s0 + u]);
                } else
                    a = (float)((double)a + (double)remap_finish(t[u], unq) * sw.w[s0 + u]);
            }
    }
    acc[o] = a;
}

// acc[b] <- fold of steps first_step .. first_step + nsteps - 1 of side pb.d (= -1 or +1); weights[s] belongs to step
// first_step + s; flows points at step first_step's flows.
void launch_sweep_side(const float* stack, const float* flows, float* acc, PairBatch pb, int nsteps, int first_step,
                       int H, int W, const double* weights, hipStream_t st, const WarpMode& wm)
{
    if (pb.npairs <= 0) return;
    dim3 grid((W + 63) / 64, (H + 3) / 4, pb.npairs);
    const size_t step_stride = (size_t)pb.npairs * H * W * 2;
    for (int s0 = 0; s0 < nsteps; s0 += SWEEP_MAX_STEPS) {
        const int n = nsteps - s0 < SWEEP_MAX_STEPS ? nsteps - s0 : SWEEP_MAX_STEPS;
        SweepWeights sw;
        for (int i = 0; i < n; i++) sw.w[i] = weights[s0 + i];
        const float* fl = flows + (size_t)s0 * step_stride;
        if (n == 1) {   // one step at a time (the integer-volume modes behind the Farneback kernels): no unrolled group to fill
            if (wm.kind == 1) hipLaunchKernelGGL((k_sweep_side<1, 1>), grid, dim3(256), 0, st, stack, fl, acc, pb, n, first_step + s0, H, W, sw, wm);
            else if (wm.kind == 2) hipLaunchKernelGGL((k_sweep_side<1, 2>), grid, dim3(256), 0, st, stack, fl, acc, pb, n, first_step + s0, H, W, sw, wm);
            else hipLaunchKernelGGL((k_sweep_side<1, 0>), grid, dim3(256), 0, st, stack, fl, acc, pb, n, first_step + s0, H, W, sw, wm);
        } else if (wm.kind == 1) hipLaunchKernelGGL((k_sweep_side<8, 1>), grid, dim3(256), 0, st, stack, fl, acc, pb, n, first_step + s0, H, W, sw, wm);
        else if (wm.kind == 2) hipLaunchKernelGGL((k_sweep_side<8, 2>), grid, dim3(256), 0, st, stack, fl, acc, pb, n, first_step + s0, H, W, sw, wm);
        else hipLaunchKernelGGL((k_sweep_side<8, 0>), grid, dim3(256), 0, st, stack, fl, acc, pb, n, first_step + s0, H, W, sw, wm);
    }
}

__global__ __launch_bounds__(256) void k_warp(const float* __restrict__ src, const float* __restrict__ flow_base,
                                              float* __restrict__ dst, int H, int W, int model)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    size_t o = (size_t)y * W + x;
    dst[o] = remap_sample(src, H, W, x, y, ((const float2*)flow_base)[o], model == 1);
}
void launch_warp(const float* src, const float* flow, float* dst, int H, int W, hipStream_t st, int model)
{
    dim3 grid((W + 63) / 64, (H + 3) / 4, 1);
    hipLaunchKernelGGL(k_warp, grid, dim3(256), 0, st, src, flow, dst, H, W, model);
}

__global__ __launch_bounds__(256) void k_warp_u8(const float* __restrict__ src, const float* __restrict__ flow_base,
                                                 float* __restrict__ dst, int H, int W)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    size_t o = (size_t)y * W + x;
    RemapTaps r;
    remap_issue<false>(src, H, W, x, y, ((const float2*)flow_base)[o], r);
    dst[o] = remap_finish_u8(r);
}
void launch_warp_u8(const float* src, const float* flow, float* dst, int H, int W, hipStream_t st)
{
    dim3 grid((W + 63) / 64, (H + 3) / 4, 1);
    hipLaunchKernelGGL(k_warp_u8, grid, dim3(256), 0, st, src, flow, dst, H, W);
}

__global__ __launch_bounds__(256) void k_warp_f64(const double* __restrict__ src, const float* __restrict__ flow_base,
                                                  double* __restrict__ dst, int H, int W, int model)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    size_t o = (size_t)y * W + x;
    dst[o] = remap_sample_f64(src, H, W, x, y, ((const float2*)flow_base)[o], model == 1);
}
void launch_warp_f64(const double* src, const float* flow, double* dst, int H, int W, hipStream_t st, int model)
{
    dim3 grid((W + 63) / 64, (H + 3) / 4, 1);
    hipLaunchKernelGGL(k_warp_f64, grid, dim3(256), 0, st, src, flow, dst, H, W, model);
}

__global__ __launch_bounds__(256) void k_axpy_slices(const float* __restrict__ stack, float* __restrict__ acc_base,
                                                     PairBatch pb, size_t HW, double weight, WarpMode wm)
{
    const int b = blockIdx.y;
    const int q = pb.t0 + b + pb.d;
    const float* src = stack + (size_t)q * HW;
    float* acc = acc_base + (size_t)b * HW;
    if (wm.kind == 1 && (q < wm.pad_lo || q >= wm.pad_hi)) {     // a pad slice of a float64 padded volume
        const double term = wm.pad64 * weight;
        for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < HW; o += (size_t)gridDim.x * 256)
            acc[o] = (float)((double)acc[o] + term);
        return;
    }
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < HW; o += (size_t)gridDim.x * 256)
        acc[o] = (float)((double)acc[o] + (double)src[o] * weight);
}
void launch_axpy_slices(const float* stack, float* acc, PairBatch pb, int H, int W, double weight, hipStream_t st, const WarpMode& wm)
{
    if (pb.npairs <= 0) return;
    size_t HW = (size_t)H * W;
    int gx = (int)((HW + 255) / 256); if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_axpy_slices, dim3(gx, pb.npairs), dim3(256), 0, st, stack, acc, pb, HW, weight, wm);
}

__global__ __launch_bounds__(256) void k_trunc_clamp(float* __restrict__ v, size_t count, float lo, float hi)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256)
        v[i] = fminf(fmaxf(truncf(v[i]), lo), hi);
}
void launch_trunc_clamp(float* v, size_t count, float lo, float hi, hipStream_t st)
{
    if (!count) return;
    size_t g = (count + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_trunc_clamp, dim3((unsigned)g), dim3(256), 0, st, v, count, lo, hi);
}

__global__ __launch_bounds__(256) void k_fill(float* __restrict__ dst, float value, size_t count)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) dst[i] = value;
}
void launch_fill(float* dst, float value, size_t count, hipStream_t st)
{
    if (!count) return;
    size_t g = (count + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)g), dim3(256), 0, st, dst, value, count);
}

// one image given as a strided view (the pair-level entry points; the sweeps re-orient whole volumes with the
// tiled permute below instead)
__global__ __launch_bounds__(256) void k_copy_strided(const float* __restrict__ in, int64_t rs, int64_t cs, float* __restrict__ out, int H, int W)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x < W && y < H) out[(size_t)y * W + x] = in[(int64_t)y * rs + (int64_t)x * cs];
}
void launch_copy_strided(const float* in, int64_t rs, int64_t cs, float* out, int H, int W, hipStream_t st)
{
    if (H <= 0 || W <= 0) return;
    hipLaunchKernelGGL(k_copy_strided, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, st, in, rs, cs, out, H, W);
}

// ---------------------------------------------------------------------------------
// permute: out[a][b][c] = in[a*sa + b*sb + c*sc], out contiguous (A,B,C).
// sc == 1: row copies.  Otherwise a 32x32 LDS-tiled transpose over the out dims
// (u, C) where u is the out dim whose in-stride is 1, so both sides stay coalesced.
// ---------------------------------------------------------------------------------
// out[a][b][c] lives at out + a*oa + b*ob + c: rows of C contiguous elements; (oa, ob) = (B*C, C) for a contiguous
// result, anything else to drop a block into a larger array (a chunk of a pass back into the volume).
__global__ __launch_bounds__(256) void k_permute_rows(const float* __restrict__ in, float* __restrict__ out,
                                                      int A, int B, int C, int64_t sa, int64_t sb, int64_t oa, int64_t ob)
{
    int64_t ab = blockIdx.x;
    int a = (int)(ab / B), b = (int)(ab - (int64_t)a * B);
    const float* src = in + a * sa + b * sb;
    float* dst = out + a * oa + b * ob;
    for (int c = threadIdx.x; c < C; c += 256) dst[c] = src[c];
}
// MODE 0: in-stride-1 dim is B (tile over b,c; grid.z = a).  MODE 1: it is A (tile over a,c; grid.z = b).
template <int MODE>
__global__ __launch_bounds__(256) void k_permute_tiled(const float* __restrict__ in, float* __restrict__ out,
                                                       int A, int B, int C, int64_t sa, int64_t sb, int64_t sc,
                                                       int64_t oa, int64_t ob)
{
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
    const int U = MODE == 0 ? B : A;
    const int u0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int other = blockIdx.z;
    const int64_t base_in = MODE == 0 ? (int64_t)other * sa : (int64_t)other * sb;
    const int64_t su = MODE == 0 ? sb : sa; // == 1
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        int u = u0 + tx, c = c0 + ty + r;
        if (u < U && c < C) tile[ty + r][tx] = in[base_in + (int64_t)u * su + (int64_t)c * sc];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        int u = u0 + ty + r, c = c0 + tx;
        if (u < U && c < C) {
            int a = MODE == 0 ? other : u, b = MODE == 0 ? u : other;
            out[(int64_t)a * oa + (int64_t)b * ob + c] = tile[tx][ty + r];
        }
    }
}
void launch_permute(const float* in, float* out, int A, int B, int C, int64_t sa, int64_t sb, int64_t sc, hipStream_t st,
                    int64_t oa, int64_t ob)
{
    if (A <= 0 || B <= 0 || C <= 0) return;
    if (oa == 0 && ob == 0) { oa = (int64_t)B * C; ob = C; }
    if (sc == 1) {
        hipLaunchKernelGGL(k_permute_rows, dim3((unsigned)((int64_t)A * B)), dim3(256), 0, st, in, out, A, B, C, sa, sb, oa, ob);
    } else if (sb == 1) {
        hipLaunchKernelGGL(k_permute_tiled<0>, dim3((B + 31) / 32, (C + 31) / 32, A), dim3(256), 0, st, in, out, A, B, C, sa, sb, sc, oa, ob);
    } else { // sa == 1 (validated by the caller)
        hipLaunchKernelGGL(k_permute_tiled<1>, dim3((A + 31) / 32, (C + 31) / 32, B), dim3(256), 0, st, in, out, A, B, C, sa, sb, sc, oa, ob);
    }
}

// ---------------------------------------------------------------------------------
// sum of a volume in f64 (for vol.mean(), seq:420)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sum_partials(const float* __restrict__ in, size_t count, double* __restrict__ partials)
{
    __shared__ double sh[4];
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) s += (double)in[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
// numpy's float32 pairwise sum (loops.c.src @TYPE@_pairwise_sum) of every full 8192-element chunk: the
// recursion halves 8192 down to 64 leaves of 128 elements; a leaf is 8 interleaved accumulators over its
// 16 groups of 8, combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)); leaves combine left + right up the tree.
// One wave per chunk, one lane per leaf.
// ALIGNED: `in` is 16-byte aligned (float4 loads); otherwise dword loads (a slab view that starts mid-chunk)
template <bool ALIGNED>
__global__ __launch_bounds__(256) void k_np_chunk_sums(const float* __restrict__ in, size_t nchunks, float* __restrict__ sums)
{
    const size_t chunk = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (chunk >= nchunks) return;
    const int lane = threadIdx.x & 63;
    const float* base = in + chunk * 8192 + (size_t)lane * 128;
    auto ld4 = [&](int q) -> float4 {
        if (ALIGNED) return ((const float4*)base)[q];
        return make_float4(base[4 * q], base[4 * q + 1], base[4 * q + 2], base[4 * q + 3]);
    };
    float4 lo = ld4(0), hi = ld4(1);
    float r[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int i = 1; i < 16; i++) {
        lo = ld4(2 * i); hi = ld4(2 * i + 1);
        r[0] += lo.x; r[1] += lo.y; r[2] += lo.z; r[3] += lo.w;
        r[4] += hi.x; r[5] += hi.y; r[6] += hi.z; r[7] += hi.w;
    }
    float s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) s = s + __shfl_down(s, d, 64);   // lanes that are multiples of 2d hold valid sums
    if (lane == 0) sums[chunk] = s;
}
void launch_np_chunk_sums(const float* in, size_t nchunks, float* sums, hipStream_t st)
{
    if (!nchunks) return;
    if (((uintptr_t)in & 15) == 0) hipLaunchKernelGGL(k_np_chunk_sums<true>, dim3((unsigned)((nchunks + 3) / 4)), dim3(256), 0, st, in, nchunks, sums);
    else hipLaunchKernelGGL(k_np_chunk_sums<false>, dim3((unsigned)((nchunks + 3) / 4)), dim3(256), 0, st, in, nchunks, sums);
}

int launch_sum_partials(const float* in, size_t count, double* partials, int max_blocks, hipStream_t st)
{
    size_t g = (count + 255) / 256;
    if (g > (size_t)max_blocks) g = max_blocks;
    if (g == 0) g = 1;
    hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)g), dim3(256), 0, st, in, count, partials);
    return (int)g;
}

// ---------------------------------------------------------------------------------
// Header statistics of an output volume (what mrcfile's set_data / update_header_stats computes on the host for the
// file seq:562-564 writes: dmin, dmax, dmean, rms) and the statistics seq:529-532 / 547-550 log: per block
// {min, max, sum, sum of squared deviations from `centre`} in f64; two launches (the second with the mean as centre).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stats_partials(const float* __restrict__ in, size_t count, double centre, double* __restrict__ partials)
{
    __shared__ double sh[4][4];
    double mn = __builtin_inf(), mx = -__builtin_inf(), s = 0, q = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        const double v = (double)in[i];
        // numpy's min / max propagate a NaN (seq:566's `np.max(filtered) < 256` is then False -> uint16; mrcfile's dmin /
        // dmax are NaN): once an extreme is NaN every later comparison is false and it stays
        mn = (v < mn || v != v) ? v : mn; mx = (v > mx || v != v) ? v : mx;
        s += v;
        q += (v - centre) * (v - centre);
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double a = __shfl_down(mn, off, 64), b = __shfl_down(mx, off, 64);
        mn = (a < mn || a != a) ? a : mn; mx = (b > mx || b != b) ? b : mx;
        s += __shfl_down(s, off, 64); q += __shfl_down(q, off, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[w][0] = mn; sh[w][1] = mx; sh[w][2] = s; sh[w][3] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; k++) {
            sh[0][0] = (sh[k][0] < sh[0][0] || sh[k][0] != sh[k][0]) ? sh[k][0] : sh[0][0];
            sh[0][1] = (sh[k][1] > sh[0][1] || sh[k][1] != sh[k][1]) ? sh[k][1] : sh[0][1];
            sh[0][2] += sh[k][2]; sh[0][3] += sh[k][3];
        }
        for (int k = 0; k < 4; k++) partials[(size_t)blockIdx.x * 4 + k] = sh[0][k];
    }
}

// The same per SLICE, in a form that does not depend on how many slices the caller holds: slice s is reduced by
// FDN_STATS_BLOCKS_PER_SLICE blocks of 256 threads striding over ITS elements only, partials[(s * B + b) * 4 ..]; the host
// adds a slice's B partials in block order.  A Z-slab of a sharded volume therefore yields, slice for slice, the very
// doubles the whole volume yields on one GPU -- the header statistics of a multi-GPU run are the single-GPU ones bit for bit.
__global__ __launch_bounds__(256) void k_stats_slices(const float* __restrict__ in, size_t slice_elems, double centre, double* __restrict__ partials)
{
    __shared__ double sh[4][4];
    const float* sl = in + (size_t)blockIdx.y * slice_elems;
    double mn = __builtin_inf(), mx = -__builtin_inf(), s = 0, q = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < slice_elems; i += (size_t)gridDim.x * 256) {
        const double v = (double)sl[i];
        mn = (v < mn || v != v) ? v : mn; mx = (v > mx || v != v) ? v : mx;
        s += v;
        q += (v - centre) * (v - centre);
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double a = __shfl_down(mn, off, 64), b = __shfl_down(mx, off, 64);
        mn = (a < mn || a != a) ? a : mn; mx = (b > mx || b != b) ? b : mx;
        s += __shfl_down(s, off, 64); q += __shfl_down(q, off, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[w][0] = mn; sh[w][1] = mx; sh[w][2] = s; sh[w][3] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; k++) {
            sh[0][0] = (sh[k][0] < sh[0][0] || sh[k][0] != sh[k][0]) ? sh[k][0] : sh[0][0];
            sh[0][1] = (sh[k][1] > sh[0][1] || sh[k][1] != sh[k][1]) ? sh[k][1] : sh[0][1];
            sh[0][2] += sh[k][2]; sh[0][3] += sh[k][3];
        }
        for (int k = 0; k < 4; k++) partials[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + k] = sh[0][k];
    }
}
void launch_stats_slices(const float* in, int nslices, size_t slice_elems, double centre, double* partials, hipStream_t st)
{
    hipLaunchKernelGGL(k_stats_slices, dim3(FDN_STATS_BLOCKS_PER_SLICE, (unsigned)nslices), dim3(256), 0, st, in, slice_elems, centre, partials);
}
int launch_stats_partials(const float* in, size_t count, double centre, double* partials, int max_blocks, hipStream_t st)
{
    size_t g = (count + 255) / 256;
    if (g > (size_t)max_blocks) g = max_blocks;
    if (g == 0) g = 1;
    hipLaunchKernelGGL(k_stats_partials, dim3((unsigned)g), dim3(256), 0, st, in, count, centre, partials);
    return (int)g;
}

// float32(volume) of an integer volume on the device (seq:517's astype(np.float32) of a TIFF stack; the device copy
// of an integer MRC): exact for 8- and 16-bit types.  16 bytes of source per thread.
template <typename T>
__global__ __launch_bounds__(256) void k_convert_f32(const T* __restrict__ in, float* __restrict__ out, size_t count)
{
    constexpr int N = 16 / sizeof(T);
    const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * N;
    if (i0 + N <= count && ((uintptr_t)in & 15) == 0) {
        const uint4 raw = *(const uint4*)(in + i0);
        const T* v = (const T*)&raw;
#pragma unroll
        for (int k = 0; k < N; k++) out[i0 + k] = (float)v[k];
    } else {
        for (size_t i = i0; i < count && i < i0 + N; i++) out[i] = (float)in[i];
    }
}
void launch_convert_f32(const void* in, int depth, float* out, size_t count, hipStream_t st)
{
    if (!count) return;
    auto go = [&](auto tag) {
        using T = decltype(tag);
        constexpr int N = 16 / sizeof(T);
        hipLaunchKernelGGL(k_convert_f32<T>, dim3((unsigned)((count + 256 * N - 1) / (256 * N))), dim3(256), 0, st, (const T*)in, out, count);
    };
    switch (depth) {
    case FDN_DEPTH_I16: go((int16_t)0); break;
    case FDN_DEPTH_U16: go((uint16_t)0); break;
    case FDN_DEPTH_I8: go((int8_t)0); break;
    case FDN_DEPTH_U8: go((uint8_t)0); break;
    default: break;
    }
}

// numpy's `filtered.astype(np.uint8 / np.uint16)` (seq:566-571: the TIFF output) on the device: C's float -> integer
// cast as x86 compilers emit it for the narrow types: truncate toward zero to a 32-bit integer, keep the low bits
// (a slightly negative voxel wraps to 65535 exactly as in the reference; NaN and |v| >= 2^31 are undefined there).
template <typename T>
__global__ __launch_bounds__(256) void k_truncate_from_f32(const float* __restrict__ in, T* __restrict__ out, size_t count)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        const float v = in[i];
        const int iv = v != v ? (int)0x80000000 : (v >= 2147483648.f || v < -2147483648.f) ? (int)0x80000000 : (int)v;   // cvttss2si
        out[i] = (T)(unsigned)iv;
    }
}
void launch_truncate_from_f32(const float* in, int depth, void* out, size_t count, hipStream_t st)
{
    if (!count) return;
    const unsigned g = (unsigned)std::min<size_t>((count + 255) / 256, 65536);
    if (depth == FDN_DEPTH_U16) hipLaunchKernelGGL(k_truncate_from_f32<uint16_t>, dim3(g), dim3(256), 0, st, in, (uint16_t*)out, count);
    else if (depth == FDN_DEPTH_U8) hipLaunchKernelGGL(k_truncate_from_f32<uint8_t>, dim3(g), dim3(256), 0, st, in, (uint8_t*)out, count);
}

// ---------------------------------------------------------------------------------
// Pyramid pieces (levels > 0): cv::GaussianBlur with runtime taps, cv::resize
// ---------------------------------------------------------------------------------
// one pixel of the horizontal pass (RowFilter / SymmRowSmallFilter), the multiply-adds rounding as `fused` says
static __device__ __forceinline__ float blur_row_at(const float* __restrict__ S, int x, int W, const BlurTaps& bt, bool fused)
{
    const int n = bt.n, c = n / 2;
    float s0;
    if (n == 3) {
        s0 = madf(fused, S[reflect101(x - 1, W)] + S[reflect101(x + 1, W)], bt.k[2], S[x] * bt.k[1]);
    } else if (n == 5) {
        s0 = madf(fused, S[reflect101(x - 1, W)] + S[reflect101(x + 1, W)], bt.k[3], S[x] * bt.k[2]);
        s0 = madf(fused, S[reflect101(x - 2, W)] + S[reflect101(x + 2, W)], bt.k[4], s0);
    } else {
        s0 = bt.k[0] * S[reflect101(x - c, W)];
        for (int j = 1; j < n; j++) s0 = madf(fused, bt.k[j], S[reflect101(x - c + j, W)], s0);
    }
    return s0;
}
__global__ __launch_bounds__(256) void k_blur_h(const float* __restrict__ in, float* __restrict__ out, int H, int W, BlurTaps bt, FmaMode fm)
{
    const size_t HW = (size_t)H * W;
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const float* S = in + (size_t)blockIdx.z * HW + (size_t)y * W;
    out[(size_t)blockIdx.z * HW + (size_t)y * W + x] = blur_row_at(S, x, W, bt, fma_at(fm, x, W));
}
__global__ __launch_bounds__(256) void k_blur_v(const float* __restrict__ in, float* __restrict__ out, int H, int W, BlurTaps bt, FmaMode fm)
{
    const size_t HW = (size_t)H * W;
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const float* T = in + (size_t)blockIdx.z * HW;
    const int c = bt.n / 2;
    const bool fused = fma_at(fm, x, W);
    float d = bt.k[c] * T[(size_t)y * W + x];
    for (int j = 1; j <= c; j++) {
        float sp = T[(size_t)reflect101(y + j, H) * W + x], sm = T[(size_t)reflect101(y - j, H) * W + x];
        d = madf(fused, bt.k[c + j], sp + sm, d);
    }
    out[(size_t)blockIdx.z * HW + (size_t)y * W + x] = d;
}
// ---------------------------------------------------------------------------------
// A pyramid level's image in two launches: resize(GaussianBlur(img), (w_k, h_k), INTER_LINEAR) reads only two source
// columns and two source rows per destination pixel (cv::resize turns an exact 2 x 2 shrink into the 2 x 2 block mean),
// so the blur is evaluated only there: 1/4 of the pixels at level 2, 1/16 at level 3.  The arithmetic of every
// evaluated pixel is that of k_blur_h / k_blur_v / k_resize_*: same bits as blurring the whole image first.
//   k_blur_h_sel     tmp[img][y][2 dx + c] = horizontal blur at (y, column c of dx),  all H rows
//   k_blur_v_resize  out[img][dy][dx]      = the 2 x 2 combination of the vertical blur of those columns at dy's two rows
// ---------------------------------------------------------------------------------
struct ResizeSel { int area; double scale_x, scale_y; };   // area: exact 2 x 2 shrink (block mean); else INTER_LINEAR

static __device__ __forceinline__ void sel_taps(int d, int n_src, double scale, int area, int& s0, int& s1, float& f)
{
    if (area) { s0 = 2 * d; s1 = 2 * d + 1; f = 0.f; return; }
    f = (float)(((double)d + 0.5) * scale - 0.5);
    float fl = floorf(f);
    s0 = (int)fl; f -= fl;
    if (s0 < 0) { f = 0; s0 = 0; }
    if (s0 >= n_src - 1) { f = 0; s0 = n_src - 1; }
    s1 = s0 + 1 < n_src ? s0 + 1 : n_src - 1;
}

__global__ __launch_bounds__(256) void k_blur_h_sel(const float* __restrict__ in, float* __restrict__ tmp, int H, int W, int dw,
                                                    ResizeSel rs, BlurTaps bt, FmaMode fm)
{
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (j >= 2 * dw || y >= H) return;
    int x0, x1; float fx;
    sel_taps(j >> 1, W, rs.scale_x, rs.area, x0, x1, fx);
    const int x = (j & 1) ? x1 : x0;
    const float* S = in + (size_t)blockIdx.z * H * W + (size_t)y * W;
    // (the option goes by the pixel's column in the FULL image: cv::GaussianBlur filters whole rows)
    tmp[((size_t)blockIdx.z * H + y) * (2 * dw) + j] = blur_row_at(S, x, W, bt, fma_at(fm, x, W));
}

__global__ __launch_bounds__(256) void k_blur_v_resize(const float* __restrict__ tmp, float* __restrict__ out, int H, int dh, int dw,
                                                       int W, ResizeSel rs, BlurTaps bt, FmaMode fm)
{
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    int y0, y1, xa, xb; float fy, fx;
    sel_taps(dy, H, rs.scale_y, rs.area, y0, y1, fy);
    sel_taps(dx, W, rs.scale_x, rs.area, xa, xb, fx);        // only the fraction is needed here
    const float* T = tmp + (size_t)blockIdx.z * H * (2 * dw) + 2 * dx;
    const int pitch = 2 * dw, c = bt.n / 2;
    const bool fa = fma_at(fm, xa, W), fb = fma_at(fm, xb, W);      // the vertical pass, by the two source columns' places in the full row
    float v[2][2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int y = r ? y1 : y0;
        const float2 t0 = *(const float2*)(T + (size_t)y * pitch);
        float da = bt.k[c] * t0.x, db = bt.k[c] * t0.y;
        for (int q = 1; q <= c; q++) {
            const float2 sp = *(const float2*)(T + (size_t)reflect101(y + q, H) * pitch);
            const float2 sm = *(const float2*)(T + (size_t)reflect101(y - q, H) * pitch);
            da = madf(fa, bt.k[c + q], sp.x + sm.x, da);
            db = madf(fb, bt.k[c + q], sp.y + sm.y, db);
        }
        v[r][0] = da; v[r][1] = db;
    }
    float res;
    if (rs.area) {     // k_resize_area_int: f32 block sum in row-major order, times 1/4
        float sum = 0;
        sum = sum + v[0][0]; sum = sum + v[0][1]; sum = sum + v[1][0]; sum = sum + v[1][1];
        res = sum * (1.f / 4.f);
    } else {           // k_resize_linear
        const float a1 = fx, a0 = 1.f - fx, b1 = fy, b0 = 1.f - fy;
        const float r0 = v[0][0] * a0 + v[0][1] * a1;
        const float r1 = v[1][0] * a0 + v[1][1] * a1;
        res = madf(fma_at(fm, dx, dw), r0, b0, r1 * b1);             // VResizeLinear, by the place in the destination row
    }
    out[((size_t)blockIdx.z * dh + dy) * dw + dx] = res;
}

// small = resize(GaussianBlur(in, taps bt), (dw, dh), INTER_LINEAR) for nimg images; tmp: nimg * H * 2 dw floats
void launch_blur_resize(const float* in, float* tmp, float* small, int nimg, int H, int W, int dh, int dw, const BlurTaps& bt, hipStream_t st, const FmaMode& fm)
{
    if (nimg <= 0) return;
    const double scale_x = (double)W / dw, scale_y = (double)H / dh;
    const int isx = (int)scale_x, isy = (int)scale_y;
    const bool integer = fabs(scale_x - isx) < 2.220446049250313e-16 && fabs(scale_y - isy) < 2.220446049250313e-16;
    ResizeSel rs{integer && isx == 2 && isy == 2 ? 1 : 0, scale_x, scale_y};
    hipLaunchKernelGGL(k_blur_h_sel, dim3((2 * dw + 63) / 64, (H + 3) / 4, nimg), dim3(256), 0, st, in, tmp, H, W, dw, rs, bt, fm);
    hipLaunchKernelGGL(k_blur_v_resize, dim3((dw + 63) / 64, (dh + 3) / 4, nimg), dim3(256), 0, st, tmp, small, H, dh, dw, W, rs, bt, fm);
}

void launch_gaussian_blur(const float* in, float* tmp, float* out, int nimg, int H, int W, const BlurTaps& bt, hipStream_t st, const FmaMode& fm)
{
    if (nimg <= 0) return;
    dim3 grid((W + 63) / 64, (H + 3) / 4, nimg);
    hipLaunchKernelGGL(k_blur_h, grid, dim3(256), 0, st, in, tmp, H, W, bt, fm);
    hipLaunchKernelGGL(k_blur_v, grid, dim3(256), 0, st, tmp, out, H, W, bt, fm);
}

// INTER_LINEAR (HResizeLinear then VResizeLinear, f32 coefficients)
template <int CN>
__global__ __launch_bounds__(256) void k_resize_linear(const float* __restrict__ in, int sh, int sw, float* __restrict__ out,
                                                       int dh, int dw, double scale_x, double scale_y, int apply_ps, double ps, FmaMode fm)
{
    int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    const float* src = in + (size_t)blockIdx.z * sh * sw * CN;
    float* dst = out + (size_t)blockIdx.z * dh * dw * CN;
    float fx = (float)(((double)dx + 0.5) * scale_x - 0.5);
    float flx = floorf(fx); int sx = (int)flx; fx -= flx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    float fy = (float)(((double)dy + 0.5) * scale_y - 0.5);
    float fly = floorf(fy); int sy = (int)fly; fy -= fly;
    if (sy < 0) { fy = 0; sy = 0; }
    if (sy >= sh - 1) { fy = 0; sy = sh - 1; }
    int sx1 = sx + 1 < sw ? sx + 1 : sw - 1, sy1 = sy + 1 < sh ? sy + 1 : sh - 1;
    float a1 = fx, a0 = 1.f - fx, b1 = fy, b0 = 1.f - fy;
    const float* S0 = src + (size_t)sy * sw * CN;
    const float* S1 = src + (size_t)sy1 * sw * CN;
#pragma unroll
    for (int ch = 0; ch < CN; ch++) {
        float r0 = S0[sx * CN + ch] * a0 + S0[sx1 * CN + ch] * a1;
        float r1 = S1[sx * CN + ch] * a0 + S1[sx1 * CN + ch] * a1;
        float v = madf(fma_at(fm, dx * CN + ch, dw * CN), r0, b0, r1 * b1);
        if (apply_ps) v = (float)((double)v * ps);
        dst[((size_t)dy * dw + dx) * CN + ch] = v;
    }
}
// INTER_AREA, integer ratios: f32 block sum in row-major order times 1/area
template <int CN>
__global__ __launch_bounds__(256) void k_resize_area_int(const float* __restrict__ in, int sh, int sw, float* __restrict__ out,
                                                         int dh, int dw, int isx, int isy, int apply_ps, double ps)
{
    int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    const float* src = in + (size_t)blockIdx.z * sh * sw * CN;
    float* dst = out + (size_t)blockIdx.z * dh * dw * CN;
    float scale = 1.f / (float)(isx * isy);
#pragma unroll
    for (int ch = 0; ch < CN; ch++) {
        float sum = 0;
        for (int ky = 0; ky < isy; ky++)
            for (int kx = 0; kx < isx; kx++)
                sum = sum + src[((size_t)(dy * isy + ky) * sw + dx * isx + kx) * CN + ch];
        float v = sum * scale;
        if (apply_ps) v = (float)((double)v * ps);
        dst[((size_t)dy * dw + dx) * CN + ch] = v;
    }
}
// The same for a 2-channel image (a flow) shrunk by ISX = 2, 4, 8 or 16 in x -- calc()'s coarsest-level initial flow,
// once per chain step: lanes sit on SOURCE columns so that every source row is one coalesced read (a thread per output
// pixel reads 8-byte pieces 8 ISX bytes apart: 3.2 ms per 512 x 1024^2 flows, 1.35 TB/s).  The arithmetic does not
// change: one f32 chain per output pixel over its block in row-major order.  Along a row the chain hops from lane to
// lane (t = t[lane - 1] + p, a DPP row_shr:1 add; after step s the block's lane s holds the chain through column s),
// the row's end is handed to the block's first lane for the next row (row_shl:ISX-1).
template <int ISX>
__global__ __launch_bounds__(256) void k_resize_area_flow(const float2* __restrict__ in, int sh, int sw, float2* __restrict__ out,
                                                          int dh, int dw, int isy, int apply_ps, double ps)
{
    const int lane = threadIdx.x & 63;
    const int sx = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + lane;
    if (sx - lane >= sw) return;                  // the whole wave is past the row
    const int dy = blockIdx.y;
    const bool live = sx < sw;
    const float2* src = in + (size_t)blockIdx.z * sh * sw + (size_t)dy * isy * sw + (live ? sx : sw - 1);
    float tx = 0.f, ty = 0.f;
    for (int ky = 0; ky < isy; ky++) {
        const float2 p = src[(size_t)ky * sw];
        // the chain so far: the previous row's end sits ISX-1 lanes up; nothing before the first row
        float cx = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tx), 0x100 + ISX - 1, 0xf, 0xf, true));
        float cy = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ty), 0x100 + ISX - 1, 0xf, 0xf, true));
        if (ky == 0) { cx = 0.f; cy = 0.f; }
        tx = cx + p.x; ty = cy + p.y;
#pragma unroll
        for (int s = 1; s < ISX; s++) {
            tx = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tx), 0x111, 0xf, 0xf, true)) + p.x;
            ty = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ty), 0x111, 0xf, 0xf, true)) + p.y;
        }
    }
    if (live && (lane & (ISX - 1)) == ISX - 1) {
        const float scale = 1.f / (float)(ISX * isy);
        float vx = tx * scale, vy = ty * scale;
        if (apply_ps) { vx = (float)((double)vx * ps); vy = (float)((double)vy * ps); }
        out[(size_t)blockIdx.z * dh * dw + (size_t)dy * dw + sx / ISX] = make_float2(vx, vy);
    }
}

// INTER_AREA, any shrink ratio (cv::resizeArea_ with computeResizeAreaTab): per output pixel,
// rows in table order: buf = sum_k S[sy][sx_k] * alpha_k (f32, left to right, starting from 0),
// then sum = beta_0 * buf_0, sum += beta_j * buf_j.  Tables: tab_si/tab_alpha with CSR offsets.
struct AreaTabs {
    const int* x_si; const float* x_alpha; const int* x_start;   // x_start[dw + 1]
    const int* y_si; const float* y_alpha; const int* y_start;   // y_start[dh + 1]
};
template <int CN>
__global__ __launch_bounds__(256) void k_resize_area_tab(const float* __restrict__ in, int sh, int sw, float* __restrict__ out,
                                                         int dh, int dw, AreaTabs t, int apply_ps, double ps)
{
    int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    const float* src = in + (size_t)blockIdx.z * sh * sw * CN;
    float* dst = out + (size_t)blockIdx.z * dh * dw * CN;
    const int xs = t.x_start[dx], xe = t.x_start[dx + 1], ys = t.y_start[dy], ye = t.y_start[dy + 1];
#pragma unroll
    for (int ch = 0; ch < CN; ch++) {
        float sum = 0.f;
        for (int j = ys; j < ye; j++) {
            const float* S = src + (size_t)t.y_si[j] * sw * CN;
            float buf = 0.f;
            for (int k = xs; k < xe; k++) buf = buf + S[t.x_si[k] * CN + ch] * t.x_alpha[k];
            float term = t.y_alpha[j] * buf;
            sum = j == ys ? term : sum + term;
        }
        if (apply_ps) sum = (float)((double)sum * ps);
        dst[((size_t)dy * dw + dx) * CN + ch] = sum;
    }
}
void resize_area_tab(const float* in, int sh, int sw, float* out, int dh, int dw, int cn, int nimg,
                     const int* x_si, const float* x_alpha, const int* x_start,
                     const int* y_si, const float* y_alpha, const int* y_start, bool apply_ps, double ps, hipStream_t st)
{
    if (nimg <= 0) return;
    dim3 grid((dw + 63) / 64, (dh + 3) / 4, nimg);
    AreaTabs t{x_si, x_alpha, x_start, y_si, y_alpha, y_start};
    if (cn == 1) hipLaunchKernelGGL(k_resize_area_tab<1>, grid, dim3(256), 0, st, in, sh, sw, out, dh, dw, t, (int)apply_ps, ps);
    else hipLaunchKernelGGL(k_resize_area_tab<2>, grid, dim3(256), 0, st, in, sh, sw, out, dh, dw, t, (int)apply_ps, ps);
}

// typed resize entry used by the pyramid driver
bool resize_needs_tables(int sh, int sw, int dh, int dw, int interp)
{
    double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    int isx = (int)scale_x, isy = (int)scale_y;
    bool integer = fabs(scale_x - isx) < 2.220446049250313e-16 && fabs(scale_y - isy) < 2.220446049250313e-16;
    return interp == 3 && scale_x >= 1 && scale_y >= 1 && !integer && !(sh == dh && sw == dw);
}

void resize_images(const float* in, int sh, int sw, float* out, int dh, int dw, int cn, int nimg, int interp,
                   bool apply_ps, double ps, hipStream_t st, const FmaMode& fm)
{
    if (nimg <= 0) return;
    dim3 grid((dw + 63) / 64, (dh + 3) / 4, nimg);
    double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    int isx = (int)scale_x, isy = (int)scale_y;
    bool integer = fabs(scale_x - isx) < 2.220446049250313e-16 && fabs(scale_y - isy) < 2.220446049250313e-16;
    if (interp == 1 && integer && isx == 2 && isy == 2) interp = 3;
    if (interp == 3 && scale_x >= 1 && scale_y >= 1 && integer && cn == 2 && (isx == 2 || isx == 4 || isx == 8 || isx == 16)) {
        dim3 g((sw + 255) / 256, dh, nimg);
        const float2* i2 = (const float2*)in; float2* o2 = (float2*)out;
#define FDN_AREA_FLOW(N) hipLaunchKernelGGL(k_resize_area_flow<N>, g, dim3(256), 0, st, i2, sh, sw, o2, dh, dw, isy, (int)apply_ps, ps)
        if (isx == 2) FDN_AREA_FLOW(2); else if (isx == 4) FDN_AREA_FLOW(4); else if (isx == 8) FDN_AREA_FLOW(8); else FDN_AREA_FLOW(16);
#undef FDN_AREA_FLOW
    } else if (interp == 3 && scale_x >= 1 && scale_y >= 1 && integer) {
        if (cn == 1) hipLaunchKernelGGL(k_resize_area_int<1>, grid, dim3(256), 0, st, in, sh, sw, out, dh, dw, isx, isy, (int)apply_ps, ps);
        else hipLaunchKernelGGL(k_resize_area_int<2>, grid, dim3(256), 0, st, in, sh, sw, out, dh, dw, isx, isy, (int)apply_ps, ps);
    } else {
        if (cn == 1) hipLaunchKernelGGL(k_resize_linear<1>, grid, dim3(256), 0, st, in, sh, sw, out, dh, dw, scale_x, scale_y, (int)apply_ps, ps, fm);
        else hipLaunchKernelGGL(k_resize_linear<2>, grid, dim3(256), 0, st, in, sh, sw, out, dh, dw, scale_x, scale_y, (int)apply_ps, ps, fm);
    }
}

} // namespace fdn
