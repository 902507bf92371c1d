// fdn_fused.hip -- the fast path: one kernel launch = one chain step of the sweep for EVERY
// target slice of the batch: the whole cv2.calcOpticalFlowFarneback(prev=target,
// next=neighbour, flow=previous flow, one pyramid level, iterations=3; winsize 2-9) (src/flowdenoising_sequential.py:62),
// the warp of the neighbour (seq:51-57) and the weighted accumulate (seq:107), fused.
//
// Decomposition: one 256-thread workgroup = one band of 64 image columns of one (target,
// neighbour) pair, marching down the rows; its four waves are the four STAGES of a software
// pipeline staggered by MH+1 rows, and rows stream from stage to stage through LDS:
//     wave 0 (A)  row t          : M0 = UpdateMatrices(R0, R1, flow_in); also streams the next
//                                  R1 row from HBM into the LDS window
//     wave k=1,2  row t - 3k     : vsum_k += f32(M_{k-1}[y+MH] - M_{k-1}[y-MH-1])  (OpenCV's f32-fed
//                                  vertical running sum, carried in registers from row 0, hence
//                                  bit-faithful), box sum across lanes (f64), 2x2 solve -> flow_k,
//                                  M_k = UpdateMatrices(R0, R1, flow_k)
//     wave 3      row t - 9      : same box sum + solve -> final flow; store it, warp the neighbour
//                                  image (1/32-px quantised bilinear), accumulate
//   One s_barrier per row step: everything a wave reads in step t was written in an earlier step,
//   everything it writes goes to LDS slots nobody reads in step t.
//   LDS per workgroup (40.2 KB -> 4 workgroups = 16 waves per CU):
//     hand-over  [3][2 slots][5 ch][64 lanes]         row r of M_k sits in slot r & 1 for one step; the
//                consumer takes it over into a six-row delay line in VGPRs (its own column of rows
//                y-3 .. y+2 is all the running sum needs), so the matrices never reach HBM and cost
//                7.7 KB of LDS instead of 3 x 7 rows
//     R1 window  [rows t-2(MH+1)-D .. t+D+1][5 (64+2DX)]  (dynamic LDS, D = 7 rows, DX = 5 columns; a row holds
//                its pixels as 16-byte quads (c0, c1, c2, c3) + a quarter-size array of c4: four aligned
//                ds_read_b128 fetch the 2 x 2 footprint of four channels)
//                neighbour expansion: every stage gathers its bilinear taps here.  The flows of noisy
//                volumes span several pixels, so lanes of one wave read different ROWS: from global
//                memory that is one cache line per lane and instruction (27 % of the kernel's time
//                before the window); from LDS it is bank conflicts.  Wave 0 loads one R1 row per
//                step, a step ahead.  A lane whose flow leaves the window fetches its taps from
//                global memory.
//   Splitting the stages over waves keeps each wave's register state small (one running sum set and
//   one delay line: 100 VGPRs), where a single wave running all stages was latency-bound.
//   Each iteration loses MH columns of validity either side: a band yields 64 - 6 MH = 52 output
//   columns for winsize 5.  BORDER_REPLICATE of the box filter: a lane outside the image takes its
//   matrix rows over from the hand-over slot of the lane that owns the border column, so the ordinary
//   lane shifts deliver the border column's running sums to the windows that reach outside.  After the
//   lane shifts only the lanes whose result is used stay active (EXEC mask).
//   Measured (MI355X, 512 targets of 1024 x 1024): 14.4 ms per launch; the VALU stream fills 0.7-0.8 of the SIMDs'
//   issue cycles (per-opcode costs measured by tools/ubench/rates.hip: 2.89 cycles per instruction for this mix) at the
//   2.05 GHz the socket's 1400 W power cap leaves it (in-kernel s_memtime stamps; DESIGN.md 3.2 has the history and the
//   variants that lost).
#include "fdn_internal.h"
#include "fdn_device.h"
#include <stdlib.h>
#include <type_traits>
#include <algorithm>
#include <utility>
#include <vector>

#ifdef FDN_CLOCK_STAMPS
// Diagnostic build only (tools/build_variant.sh clock "-DFDN_CLOCK_STAMPS", tools/clock_stamps.py; never the product): every
// workgroup's stage-A wave stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around its row loop into a buffer
// of their own that no kernel reads: the in-kernel clock is their ratio x 100 MHz (MI355X_MICROARCH.md, DVFS give-back 6).
__device__ unsigned long long fdn_clock_stamp_buf[2 * 16384];
extern "C" __attribute__((visibility("default"))) int fdn_debug_clock_stamps(unsigned long long* out, int n)
{
    if (n > 2 * 16384) n = 2 * 16384;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fdn_clock_stamp_buf), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? n : -1;
}
#endif

namespace fdn {

// Whole-wave lane shifts of an f64 on the VALU (DPP wave_shr:1 / wave_shl:1): lane i receives the
// value of lane i-1 (i+1); the lane shifted in from outside the wave reads 0 (bound_ctrl), which only
// reaches halo lanes whose results are never used.  Measured on MI355X (tools/ubench/rates.hip): 3.2 cycles per v_mov_dpp per
// SIMD at four waves per SIMD against 19-24 cycles per ds_bpermute_b32 on the one LDS pipe the four SIMDs share.  (Also measured: the window
// through an LDS row per stage -- 5 ds_write_b64 + 10 ds_read2_b64 in place of 40 v_mov_dpp per row step, the R1
// window cut to D = 4 to make room: 18.9 ms per launch against 17.8 with D = 4 alone and 16.9 as built.)
static __device__ __forceinline__ double wave_shr1(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ double wave_shl1(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// Workgroup barrier that publishes LDS writes only.  __syncthreads() also drains vmcnt, which would
// force every in-flight global load / prefetch / store to complete at each row step.
static __device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// NB: bands per workgroup.  NB = 2 puts two independent bands (eight waves, one barrier) into a workgroup, two workgroups
// per CU: the same sixteen waves and LDS per CU as four one-band workgroups, 3 % faster on large grids (17.06 against
// 17.58 ms per launch, same box) -- the two bands are neighbours in the image and stay in step, so the columns they share
// and the R1 rows both stream arrive once per CU.
//
// (Round 5 also built "both sides of a chain step in one launch, mirror pairs in one workgroup" on top of NB = 2: bit-equal,
// half the HBM bytes per launch and 7.7 % slower -- eight waves on one barrier.  Removed in round 6; the build is kept as
// profiles/history/r06_two_sided_removed.patch, the measurements in profiles/history/NOTES_r05.md.)
template <int MH, int D, int DX, int U, int OCC, int FIN, bool ACC, int NB, int WM = 0>
// (the second argument of __launch_bounds__ is waves per SIMD: for a 4-wave workgroup that is workgroups per CU; the
// 8-wave workgroup runs two per CU, i.e. the same 4)
__global__ __launch_bounds__(256 * NB, NB == 2 ? 4 : OCC) void k_farneback_fused(const float* __restrict__ Rstack, const float* __restrict__ stack,
                                                         const float* __restrict__ flow_in_base, float* __restrict__ flow_out_base,
                                                         float* __restrict__ acc_base, PairBatch pb, int H, int W,
                                                         double scale, double weight, int nbands, FlowSource fs, WarpMode wm)
{
    constexpr int ITERS = 3;
    constexpr int STEP = MH + 1;                 // row stagger between stages
    constexpr int RSD = 2 * MH + 2;              // rows y-MH-1 .. y+MH of M_{K-1} a consumer holds in VGPRs
    constexpr int NE = U == RSD ? RSD : RSD + U - 1;   // delay-line registers per channel (U == RSD: a ring, no moves)
    constexpr int HALO = MH * ITERS;
    constexpr int BW = 64 - 2 * HALO;
    constexpr int NRP = (ITERS - 1) * STEP + 2 * D + 2;   // window rows [t-2 STEP-D, t+D] + the one being loaded
    constexpr int WC = 64 + 2 * DX;               // D: window half-height (rows), DX: half-width (columns)
#ifndef FDN_WIN_QUAD
#define FDN_WIN_QUAD 1
#endif
    // FDN_WIN_QUAD (default): a window row is [ (c0,c1,c2,c3) x WC ] 16-byte pixels in one array and [ c4 x WC ] in a
    // second one a quarter its size (so a pixel's c4 sits at a quarter of its quad's byte offset): the 2 x 2 footprint
    // of four channels is four aligned ds_read_b128 (4 LDS cycles each, 64 banks) where the pair planes of the HBM
    // layout took four ds_read2_b64 (8 cycles each, 32 banks).  0: the RImage layout, pitch 5 WCP floats.
    constexpr int WCP = FDN_WIN_QUAD ? WC : (5 * WC) % 16 == 0 ? WC + 2 : WC;
    static_assert(WC % 2 == 0, "pair planes need an even pitch");
    __shared__ float MxAll[NB][ITERS][2][5][64]; // hand-over slots: row r of M_k lives in slot r & 1 for one step
    extern __shared__ __attribute__((aligned(16))) float win_all[];   // [NB][NRP][5 WCP] (dynamic: sized by the launcher)

    const int lane = threadIdx.x & 63;
    const int half = NB == 2 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) : 0;     // which of the workgroup's bands
#ifndef FDN_MASK_HALO
#define FDN_MASK_HALO 1
#endif
#ifndef FDN_STAGE_MAP
#define FDN_STAGE_MAP 1
#endif
    int stage_ = (threadIdx.x >> 6) & 3;
    if (NB == 2 && FDN_STAGE_MAP) {
        unsigned tab = 0x8DE4u;      // waves 0..7 -> stages A 1 2 3 | 1 3 A 2
        if (FDN_STAGE_MAP == 2 && ((blockIdx.x >> 3) & 1)) tab = 0x3693u;   // 3 A 1 2 | 2 1 3 A
        stage_ = (tab >> (2 * (threadIdx.x >> 6))) & 3;
    }
    const int stage = __builtin_amdgcn_readfirstlane(stage_);
    float (*Mx)[2][5][64] = MxAll[half];
    float* const win = win_all + (size_t)half * NRP * 5 * WCP;
    char* const winq = (char*)win;                                  // quad array [NRP][WC] x 16 B
    char* const win4 = (char*)win + (size_t)NRP * WC * 16;          // c4 array   [NRP][WC] x  4 B
    // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs, so giving XCD j the
    // j-th contiguous eighth of the (pair, band) list keeps the bands of a pair -- which share
    // their 12 halo columns and their rows in time -- behind one L2.  Speed only, never correctness.
    const long nwg = gridDim.x, q8 = nwg >> 3, rem8 = nwg & 7;
    const long xcd = blockIdx.x & 7;
    long gw = (xcd * q8 + (xcd < rem8 ? xcd : rem8) + (blockIdx.x >> 3)) * NB + half;   // a bijection on [0, NB nwg)
    const long nband_total = (long)nbands * pb.npairs;
    const bool live = gw < nband_total;          // NB = 2 and an odd number of bands: the last half repeats a band, stores off
    if (!live) gw = nband_total - 1;
    const int bw = (int)(gw / nbands);              // position in the walk over the pairs (pair_walk: chains of stride |d|)
    const int band = (int)(gw - (long)bw * nbands);
    const int b = pair_walk(bw, pb.npairs, pb.d);
    const int pd = pb.d;                         // neighbour offset
    const int xb = band * BW - HALO;            // column of lane 0
    const int x = xb + lane;
    const int xc = clampi(x, 0, W - 1);
    const bool in_img = x == xc;
    const size_t HW = (size_t)H * W;
    const float* R0 = Rstack + (size_t)(pb.t0 + b) * 5 * HW;
    const float* R1 = Rstack + (size_t)(pb.t0 + b + pd) * 5 * HW;
    // opaque wave-uniform plane bases: every access is SGPR base + 32-bit per-lane byte offset
    const RImage R0i = {uniform_ptr(R0), uniform_ptr(R0 + 2 * HW), uniform_ptr(R0 + 4 * HW)};
    const RImage R1i = {uniform_ptr(R1), uniform_ptr(R1 + 2 * HW), uniform_ptr(R1 + 4 * HW)};
    const float bxx = border_factor(xc, W);
    const bool xdamp = border_test(xc, W);
    const int xw0 = xb - DX;                     // image column of window column 0

    // BORDER_REPLICATE of the box filter without a second code path: a lane outside the image takes every matrix
    // row over from the hand-over slot of the lane that owns the border column (hl), so its running sums ARE the
    // border column's and the lane shifts of the window sum deliver them to the border lanes' windows.  (Its own
    // results are never used: its window is shifted, so from the second iteration on what it computes is not the
    // border column's matrix -- which is why it must not feed its own rows back.)  Until round 3 bands touching an
    // image edge summed their windows with ds_bpermute from clamped source lanes: 120 of them per row step, 24
    // cycles each on the CU's one LDS pipe -- two bands of twenty ate two thirds of all LDS cycles.
    const int hl = clampi(xc - xb, 0, 63);

    auto row_factor = [&](int y, float& by0, float& by1) {
        by0 = y < 5 ? (y < 2 ? 0.14f : 0.4472f) : 1.f;
        by1 = y >= H - 5 ? (H - y - 1 < 2 ? 0.14f : 0.4472f) : 1.f;
    };
    // Bilinear taps for a stage working on row ys.  `need`: lanes whose result is used.  Fast path:
    // the lane's 2x2 footprint lies in window rows [ys-D, ys+D] and the window's columns.
    auto gather = [&](int ys, int x1, int y1, bool need, GatherTapsP& g) __attribute__((always_inline)) {
        const int x1c = clamp0u(x1, W - 2), y1c = clamp0u(y1, H - 2);
        int col = x1c - xw0;
        int dy = y1c - (ys - D);
        const bool inwin = (unsigned)col <= (unsigned)(WC - 2) && (unsigned)dy <= (unsigned)(2 * D - 1);
        const bool hit = need && inwin, miss = need && !inwin;
        if (!hit) { col = lane + DX; dy = clampi(ys, 0, H - 2) - (ys - D); }   // lanes not served from LDS: stay inside the window
        // window slot of row ys - D (wave-uniform, scalar unit) + dy, wrapped once: no per-lane modulo
        int sb = (ys - D) % NRP;
        sb = sb < 0 ? sb + NRP : sb;
        unsigned s0 = (unsigned)(sb + dy);
        s0 = min(s0, s0 - (unsigned)NRP);
        unsigned s1 = s0 + 1u;
        s1 = min(s1, s1 - (unsigned)NRP);
        if (FDN_WIN_QUAD) {
            const unsigned o0 = __umul24(s0, 16u * WC) + 16u * (unsigned)col, o1 = __umul24(s1, 16u * WC) + 16u * (unsigned)col;
            const fdn_v4f pa0 = *(const fdn_v4f*)(winq + o0), pb0 = *(const fdn_v4f*)(winq + o0 + 16);
            const fdn_v4f pa1 = *(const fdn_v4f*)(winq + o1), pb1 = *(const fdn_v4f*)(winq + o1 + 16);
            g.a0[0] = pa0.xy; g.a0[1] = pa0.zw; g.b0[0] = pb0.xy; g.b0[1] = pb0.zw;
            g.a1[0] = pa1.xy; g.a1[1] = pa1.zw; g.b1[0] = pb1.xy; g.b1[1] = pb1.zw;
            const float* c0 = (const float*)(win4 + (o0 >> 2));
            const float* c1 = (const float*)(win4 + (o1 >> 2));
            g.a0s = c0[0]; g.b0s = c0[1];
            g.a1s = c1[0]; g.b1s = c1[1];
        } else {
            // (a, b) = columns (col, col + 1) of both channels of a pair: 16 contiguous bytes -> ds_read2_b64
            const float* q0 = win + __umul24(s0, 5u * WCP) + 2 * col;
            const float* q1 = win + __umul24(s1, 5u * WCP) + 2 * col;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                g.a0[q] = *(const fdn_v2f*)(q0 + 2 * q * WCP); g.b0[q] = *(const fdn_v2f*)(q0 + 2 * q * WCP + 2);
                g.a1[q] = *(const fdn_v2f*)(q1 + 2 * q * WCP); g.b1[q] = *(const fdn_v2f*)(q1 + 2 * q * WCP + 2);
            }
            g.a0s = q0[4 * WCP - col]; g.b0s = q0[4 * WCP - col + 1];
            g.a1s = q1[4 * WCP - col]; g.b1s = q1[4 * WCP - col + 1];
        }
        if (miss) gather_R1_p(R1i, H, W, x1, y1, g);   // a flow that leaves the window: those lanes (only) go to global memory (skipped by an execz branch)
    };
    const float xf = (float)xc;
    struct R0Px { fdn_v2f r01, r23; float r4; };     // R0 at one pixel
    auto update_matrices = [&](int ys, float2 f, const R0Px& r0, bool need, float mm[5]) __attribute__((always_inline)) {
        int x1, y1; float fx, fy;
        flow_target(xf, (float)ys, f.x, f.y, x1, y1, fx, fy);
        GatherTapsP g;
        gather(ys, x1, y1, need, g);
        float by0, by1;
        row_factor(ys, by0, by1);
        fdn_v2f m02, m34;
        finish_M_p(r0.r01, r0.r23, r0.r4, g, H, W, x1, y1, fx, fy, f.x, f.y, bxx, by0, by1, xdamp || border_test(ys, H), m02, mm[1], m34);
        mm[0] = m02.x; mm[2] = m02.y; mm[3] = m34.x; mm[4] = m34.y;
    };

    const int T = H + ITERS * STEP;              // row steps = barriers every wave executes

    if (stage == 0) {
        // ===== wave 0: stage A + the R1 window stream ==============================================
        // FIN 0: zero flow; 1: flow_in is this level's flow; 2: flow_in is the next coarser level's
        // (fs.h x fs.w) result, resized INTER_LINEAR and doubled on the fly (calc()'s upsampling)
        const float* flow_in = FIN == 1 ? uniform_ptr(flow_in_base + (size_t)b * HW * 2) : FIN == 2 ? uniform_ptr(flow_in_base + (size_t)b * fs.h * fs.w * 2) : nullptr;
        const LinearTap ftx = FIN == 2 ? linear_tap(xc, fs.sx, fs.w) : LinearTap{};
        auto load_flow = [&](int row) __attribute__((always_inline)) -> float2 {
            if (FIN == 1) return ld_off<float2>(flow_in, ((unsigned)row * (unsigned)W + (unsigned)xc) * 8u);
            if (FIN == 2) return resize_linear_flow(flow_in, fs.w, ftx, linear_tap(row, fs.sy, fs.h), 2.0, fs.fm, xc, W);
            return make_float2(0.f, 0.f);
        };
        const int wcol0 = clampi(xw0 + lane, 0, W - 1);        // image columns this lane loads into the window
        const int wcol1 = clampi(xw0 + 64 + lane, 0, W - 1);
        struct WinRow { fdn_v2f a01, a23, b01, b23; float a4, b4; };   // columns wcol0 (a) and wcol1 (b) of one R1 row
        auto load_window_row = [&](int v, WinRow& w) __attribute__((always_inline)) {
            const unsigned vo = (unsigned)v * (unsigned)W;
            load_R(R1i, vo + wcol0, w.a01, w.a23, w.a4);
            load_R(R1i, vo + wcol1, w.b01, w.b23, w.b4);
        };
        auto store_window_row = [&](int v, const WinRow& w) __attribute__((always_inline)) {
            if (FDN_WIN_QUAD) {
                const unsigned r = (unsigned)(v % NRP) * (16u * WC);
                fdn_v4f qa; qa.xy = w.a01; qa.zw = w.a23;
                *(fdn_v4f*)(winq + r + 16u * lane) = qa;
                *(float*)(win4 + (r >> 2) + 4u * lane) = w.a4;
                if (lane < 2 * DX) {
                    fdn_v4f qb; qb.xy = w.b01; qb.zw = w.b23;
                    *(fdn_v4f*)(winq + r + 16u * (64 + lane)) = qb;
                    *(float*)(win4 + (r >> 2) + 4u * (64 + lane)) = w.b4;
                }
                return;
            }
            float* row = win + (size_t)(v % NRP) * 5 * WCP;
            *(fdn_v2f*)(row + 2 * lane) = w.a01;
            *(fdn_v2f*)(row + 2 * WCP + 2 * lane) = w.a23;
            row[4 * WCP + lane] = w.a4;
            if (lane < 2 * DX) {
                *(fdn_v2f*)(row + 2 * (64 + lane)) = w.b01;
                *(fdn_v2f*)(row + 2 * WCP + 2 * (64 + lane)) = w.b23;
                row[4 * WCP + 64 + lane] = w.b4;
            }
        };
        {   // rows 0..D before the first step
            WinRow w;
            for (int v = 0; v <= (D < H - 1 ? D : H - 1); v++) { load_window_row(v, w); store_window_row(v, w); }
        }
        // operands of row 0, then always one row ahead
        float2 fN = load_flow(0);
        R0Px r0N;
        load_R(R0i, (unsigned)xc, r0N.r01, r0N.r23, r0N.r4);
        lds_barrier();
#ifdef FDN_CLOCK_STAMPS
        const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
        for (int t = 0; t < T; t++) {
            if (t < H) {
                const float2 f = fN;
                const R0Px r0 = r0N;
                const int tn = t + 1 < H ? t + 1 : H - 1;
                const unsigned on = (unsigned)tn * (unsigned)W + (unsigned)xc;
                fN = load_flow(tn);
                load_R(R0i, on, r0N.r01, r0N.r23, r0N.r4);
                const int vnext = t + D + 1;                 // window row the next step needs
                WinRow wl;
                load_window_row(vnext < H ? vnext : H - 1, wl);
                float mm[5];
                update_matrices(t, f, r0, in_img, mm);
#pragma unroll
                for (int c = 0; c < 5; c++) Mx[0][t & 1][c][lane] = mm[c];
                if (vnext < H) store_window_row(vnext, wl);
            }
            lds_barrier();
        }
#ifdef FDN_CLOCK_STAMPS
        if (lane == 0 && half == 0) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            ::fdn_clock_stamp_buf[2 * (blockIdx.x & 16383)] = __builtin_amdgcn_s_memtime() - st_c0;
            ::fdn_clock_stamp_buf[2 * (blockIdx.x & 16383) + 1] = __builtin_amdgcn_s_memrealtime() - st_r0;
        }
#endif
        return;
    }

    // ===== waves 1..3: iteration `stage` =============================================================
    // The loop is instantiated per stage (K is compile-time inside it) so that a row step is one basic block: the
    // five channels' running-sum -> DPP -> f64-add chains then interleave instead of running one after the other.
    const bool owner = live && lane >= HALO && lane < 64 - HALO && x < W;
    const float* img1 = ACC ? uniform_ptr(stack + (size_t)(pb.t0 + b + pd) * HW) : nullptr;
    float2* flow_out = flow_out_base ? uniform_ptr((float2*)flow_out_base + (size_t)b * HW) : nullptr;
    float* acc = ACC ? uniform_ptr(acc_base + (size_t)b * HW) : nullptr;

    auto stage_loop = [&](auto KT) __attribute__((always_inline)) {
        constexpr int K = decltype(KT)::value;
        float (*Min)[5][64] = Mx[K - 1];
        const bool need = in_img && lane >= K * MH && lane < 64 - K * MH;   // lanes whose M_K feeds a valid output
        double vs[5];
        // Delay line: this lane's column of rows y+MH (newest) .. y-MH-1 of M_{K-1}.  The producer wrote
        // row n = y+MH in the previous step and rewrites that slot two steps later, so it is taken over
        // now; rows above the image replicate row 0, rows below it row H-1 (OpenCV's clamped row pointers).
        float e[NE][5];
        lds_barrier();
        int t = 0;
        for (; t < K * STEP - MH; t++) lds_barrier();          // row 0 of M_{K-1} not yet produced
#pragma unroll
        for (int c = 0; c < 5; c++) {                          // step K STEP - MH: n = 0
            const float m = Min[0][c][hl];
#pragma unroll
            for (int i = 0; i < NE; i++) e[i][c] = m;
            vs[c] = 0.;
        }
        lds_barrier();
        t++;
        // one row step; UT = position inside the unrolled group: the newest row goes to register
        // P, the row i steps older is at (P + i) [mod RSD for the ring form]
        auto row_step = [&](auto UT) __attribute__((always_inline)) {
            constexpr int u = decltype(UT)::value;
            constexpr int P = U == RSD ? (RSD - u) % RSD : U - 1 - u;
            auto at = [](int i) constexpr { return U == RSD ? (P + i) % RSD : P + i; };
            const int y = t - K * STEP;
            const int n = y + MH;
            if (y < H) {
                {
                    if (n <= H - 1) {          // take over the row the producer wrote in the previous step
#pragma unroll
                        for (int c = 0; c < 5; c++) e[at(0)][c] = Min[n & 1][c][hl];
                    } else {                   // below the image: row H-1 again
#pragma unroll
                        for (int c = 0; c < 5; c++) e[at(0)][c] = e[at(1)][c];
                    }
                }
                if (y >= 0) {
                    // final stage: the accumulator does not depend on this step's flow: load it first
                    float acc_old = 0.f;
                    const unsigned o = (unsigned)y * (unsigned)W + (unsigned)xc;
                    if (ACC && K == ITERS) acc_old = ld_off<float>(acc, o * 4u);
                    if (y == 0) { // vsum before row 0: f32(M[0]*(m+2)) + rows 1..m-1 (clamped)
#pragma unroll
                        for (int c = 0; c < 5; c++) {
                            double v = (double)(e[at(MH)][c] * (float)(MH + 2));
#pragma unroll
                            for (int yy = 1; yy < MH; yy++) v += (double)e[at(MH - yy)][c];
                            vs[c] = v;
                        }
                    }
                    // running sums and their lane shifts on every lane (a halo lane's sums feed its neighbours' windows) ...
                    double lft[5][MH], rgt[5][MH];
#pragma unroll
                    for (int c = 0; c < 5; c++) {
                        vs[c] += (double)(e[at(0)][c] - e[at(RSD - 1)][c]);
                        lft[c][0] = wave_shr1(vs[c]); rgt[c][0] = wave_shl1(vs[c]);
#pragma unroll
                        for (int i = 1; i < MH; i++) { lft[c][i] = wave_shr1(lft[c][i - 1]); rgt[c][i] = wave_shl1(rgt[c][i - 1]); }
                    }
                    // ... everything after that only on the lanes whose result is used (EXEC-masked: the part runs at the
                    // clock its power budget allows, and idle lanes cost none: FDN_MASK_HALO = 0 for the A/B)
                    if (!FDN_MASK_HALO || (K < ITERS ? need : owner)) {
                        double a[5];
#pragma unroll
                        for (int c = 0; c < 5; c++) {
                            double s = lft[c][MH - 1];     // the 2 MH + 1 terms left to right, starting from the first
#pragma unroll
                            for (int i = MH - 2; i >= 0; i--) s += lft[c][i];
                            s += vs[c];
#pragma unroll
                            for (int i = 0; i < MH; i++) s += rgt[c][i];
                            a[c] = s;
                        }
                        const float2 f = solve_flow(a, scale);
                        if (K < ITERS) {
                            float mm[5];
                            R0Px r0;
                            load_R(R0i, o, r0.r01, r0.r23, r0.r4);
                            update_matrices(y, f, r0, need, mm);
#pragma unroll
                            for (int c = 0; c < 5; c++) Mx[K < ITERS ? K : 0][y & 1][c][lane] = mm[c];
                        } else if (ACC) {
                            // (WM: the dtype semantics of an integer volume, fold_warped in fdn_device.h; the neighbour's stack index decides `pad`)
                            const int q = pb.t0 + b + pd;
                            const bool pad = (WM & 3) == 1 && (q < wm.pad_lo || q >= wm.pad_hi);
                            const float acc_new = fold_warped<(WM & 3), (WM & 4) != 0>(img1, H, W, xc, y, f, acc_old, weight, pad, wm.pad64, wm.lo, wm.hi, wm.fixed8 != 0);
                            if (owner) {
                                if (flow_out) st_off(flow_out, o * 8u, f);
                                st_off(acc, o * 4u, acc_new);
                            }
                        } else if (owner) {      // a coarser pyramid level: the flow is the result
                            st_off(flow_out, o * 8u, f);
                        }
                    }
                }
            }
            lds_barrier();
            t++;
        };
        using std::integral_constant;
        while (t < T) {
            // U row steps with compile-time register positions; the tail group is cut by the t < T tests
            row_step(integral_constant<int, 0>{});
            if (U > 1 && t < T) row_step(integral_constant<int, (U > 1 ? 1 : 0)>{});
            if (U > 2 && t < T) row_step(integral_constant<int, (U > 2 ? 2 : 0)>{});
            if (U > 3 && t < T) row_step(integral_constant<int, (U > 3 ? 3 : 0)>{});
            if (U > 4 && t < T) row_step(integral_constant<int, (U > 4 ? 4 : 0)>{});
            if (U > 5 && t < T) row_step(integral_constant<int, (U > 5 ? 5 : 0)>{});
            if (U != RSD) {     // move the line up by U registers
#pragma unroll
                for (int c = 0; c < 5; c++) {
#pragma unroll
                    for (int i = RSD - 2; i >= 0; i--) e[i + U][c] = e[i][c];
                }
            }
        }
    };
    using std::integral_constant;
    if (stage == 1) stage_loop(integral_constant<int, 1>{});
    else if (stage == 2) stage_loop(integral_constant<int, 2>{});
    else stage_loop(integral_constant<int, 3>{});
}

bool fused_supported(int winsize, int iters, int H, int W)
{
    // window half-widths 1 (winsize 2, 3), 2 (4, 5), 3 (6, 7), 4 (8, 9); the kernel addresses pixels of one
    // image / flow field by 32-bit byte offsets
    const int mh = winsize / 2;
    return mh >= 1 && mh <= 4 && iters == 3 && H >= 2 && W >= 2 && H < (1 << 24) && W < (1 << 24) && (size_t)H * W < ((size_t)1 << 29);
}

// One build of the kernel per window half-width MH and occupancy OCC: LDS window size, unroll and VGPR
// budget chosen for OCC workgroups per CU.  (ms per launch of 512 targets of 1024 x 1024 on MI355X.)
template <int MH, int OCC> struct FusedVariant;
template <> struct FusedVariant<2, 3> { static constexpr int D = 8, DX = 8, U = 3; };   // 47.0 KB [18.1]
#ifndef FDN_V24_D      // experiment switches of tools/build_variant.sh
#define FDN_V24_D 7
#endif
#ifndef FDN_V24_DX
#define FDN_V24_DX 5
#endif
#ifndef FDN_V24_U
#define FDN_V24_U 3
#endif
template <> struct FusedVariant<2, 4> { static constexpr int D = FDN_V24_D, DX = FDN_V24_DX, U = FDN_V24_U; };   // 40.2 KB [17.2]
template <> struct FusedVariant<2, 5> { static constexpr int D = 4, DX = 5, U = 1; };   // 31.4 KB, 96 VGPRs [18.0]
template <> struct FusedVariant<1, 4> { static constexpr int D = 7, DX = 5, U = 2; };   // 58 useful columns per band
template <> struct FusedVariant<3, 4> { static constexpr int D = 6, DX = 5, U = 4; };   // 46 useful columns per band
template <> struct FusedVariant<4, 3> { static constexpr int D = 7, DX = 6, U = 2; };   // 40 useful columns per band

template <int MH, int OCC, int NB = 1>
static void launch_variant(const float* Rstack, const float* stack, const float* flow_in, float* flow_out, float* acc,
                           PairBatch pb, int H, int W, double scale, double weight, FlowSource fs, hipStream_t st, unsigned lds_pad, const WarpMode& wm)
{
    constexpr int D = FusedVariant<MH, OCC>::D, DX = FusedVariant<MH, OCC>::DX, U = FusedVariant<MH, OCC>::U;
    constexpr int WC = 64 + 2 * DX, WCP = FDN_WIN_QUAD ? WC : (5 * WC) % 16 == 0 ? WC + 2 : WC;   // as in the kernel
    constexpr unsigned win_bytes = (2 * (MH + 1) + 2 * D + 2) * 5 * WCP * sizeof(float);
    // a CU's 160 KB of LDS is handed out in 2 KB granules
    static_assert(win_bytes + 3 * 2 * 5 * 64 * sizeof(float) <= (160 * 1024 / OCC) / 2048 * 2048, "LDS per workgroup");
    const int BW = 64 - 2 * MH * 3;
    const int nbands = (W + BW - 1) / BW;
    dim3 grid((unsigned)(((long)nbands * pb.npairs + NB - 1) / NB));
    auto launch = [&](auto kern) {
        const unsigned lds = NB * win_bytes + (NB == 1 ? lds_pad : 0);     // lds_pad: an occupancy experiment knob of the one-band build
        if (lds > 64 * 1024) {      // above 64 KB of dynamic LDS a kernel must be told so: once per kernel, device and host thread
            static thread_local std::vector<std::pair<const void*, int>> told;
            int dev = 0;
            (void)hipGetDevice(&dev);
            const std::pair<const void*, int> key((const void*)kern, dev);
            if (std::find(told.begin(), told.end(), key) == told.end()) {
                // refused: nothing is launched and the runtime's last error stays set -- the caller's hipGetLastError()
                // check after the launches of a chain step reports it (as fdn_iter.hip's launcher returns -1)
                if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return;
                told.push_back(key);
            }
        }
        hipLaunchKernelGGL(kern, grid, dim3(256 * NB), lds, st, Rstack, stack, flow_in, flow_out, acc, pb, H, W, scale, weight, nbands, fs, wm);
    };
    const int fin = !flow_in ? 0 : fs.h > 0 ? 2 : 1;
    if (acc && wm.kind == 1) {          // integer volumes (fdn_sweep_params.warp_mode): the accumulate in their own semantics
        if (fin == 2) launch(k_farneback_fused<MH, D, DX, U, OCC, 2, true, NB, 1>);
        else if (fin == 1) launch(k_farneback_fused<MH, D, DX, U, OCC, 1, true, NB, 1>);
        else launch(k_farneback_fused<MH, D, DX, U, OCC, 0, true, NB, 1>);
    } else if (acc && wm.kind == 2) {
        if (fin == 2) launch(k_farneback_fused<MH, D, DX, U, OCC, 2, true, NB, 2>);
        else if (fin == 1) launch(k_farneback_fused<MH, D, DX, U, OCC, 1, true, NB, 2>);
        else launch(k_farneback_fused<MH, D, DX, U, OCC, 0, true, NB, 2>);
    } else if (acc && wm.model == 1) {  // float32 volume, unquantised remap (the "remap_model" option): WM = 4
        if (fin == 2) launch(k_farneback_fused<MH, D, DX, U, OCC, 2, true, NB, 4>);
        else if (fin == 1) launch(k_farneback_fused<MH, D, DX, U, OCC, 1, true, NB, 4>);
        else launch(k_farneback_fused<MH, D, DX, U, OCC, 0, true, NB, 4>);
    } else if (acc) {
        if (fin == 2) launch(k_farneback_fused<MH, D, DX, U, OCC, 2, true, NB, 0>);
        else if (fin == 1) launch(k_farneback_fused<MH, D, DX, U, OCC, 1, true, NB, 0>);
        else launch(k_farneback_fused<MH, D, DX, U, OCC, 0, true, NB, 0>);
    } else {
        if (fin == 2) launch(k_farneback_fused<MH, D, DX, U, OCC, 2, false, NB, 0>);
        else if (fin == 1) launch(k_farneback_fused<MH, D, DX, U, OCC, 1, false, NB, 0>);
        else launch(k_farneback_fused<MH, D, DX, U, OCC, 0, false, NB, 0>);
    }
}

// Workgroups per CU for a grid of `blocks` workgroups (winsize 4-5 builds): large grids run fastest at 4 per CU (round 3,
// 5 120 / 2 560 workgroups: 7.78 / 4.04 ms at 4; 8.09 / 4.24 at 3; 8.70 / 4.57 at 5).  A grid of at most 5 workgroups per
// CU -- 1 280: the 64-slice Z slab of an 8-GPU run -- is not short of slots but of speed per workgroup, and the build
// with the largest LDS window is the fastest there: 2.18 ms at 3 per CU, 2.22 at 4, 2.28 at 5 (round 2, before the edge
// bands lost their ds_bpermute path, 5 per CU had won that grid: 2.69 against 3.21).
static int choose_occupancy(long blocks, const Tuning& tn)
{
    if (tn.fused_occ >= 3 && tn.fused_occ <= 5) return tn.fused_occ;
    if (tn.fused_occ == 8) return 4;
    return blocks <= (long)tn.cus * 5 ? 3 : 4;
}

// acc == nullptr: Farneback only (a coarser pyramid level), flow_out is required then.
// coarse_h, coarse_w > 0: flow_in holds the next coarser level's flow of that size (upsampled in the kernel).
void launch_farneback_fused(const float* Rstack, const float* stack, const float* flow_in, float* flow_out, float* acc,
                            PairBatch pb, int H, int W, int winsize, int iters, double weight, hipStream_t st,
                            const Tuning& tn, int coarse_h, int coarse_w, const WarpMode& wm)
{
    if (pb.npairs <= 0) return;
    (void)iters;
    FlowSource fs{coarse_h, coarse_w, coarse_h > 0 ? (double)coarse_w / W : 1.0, coarse_h > 0 ? (double)coarse_h / H : 1.0, tn.fma};
    const double scale = 1. / ((double)winsize * winsize);
    const int mh = winsize / 2;
#ifndef FDN_ONLY_MH2   // (experiment builds leave the other windows out: half the compile time)
    if (mh == 1) { launch_variant<1, 4>(Rstack, stack, flow_in, flow_out, acc, pb, H, W, scale, weight, fs, st, tn.lds_pad, wm); return; }
    if (mh == 3) { launch_variant<3, 4>(Rstack, stack, flow_in, flow_out, acc, pb, H, W, scale, weight, fs, st, tn.lds_pad, wm); return; }
    if (mh == 4) { launch_variant<4, 3>(Rstack, stack, flow_in, flow_out, acc, pb, H, W, scale, weight, fs, st, tn.lds_pad, wm); return; }
#endif
    const int BW = 64 - 2 * 2 * 3;
    const long blocks = tn.occ_blocks > 0 ? tn.occ_blocks : (long)((W + BW - 1) / BW) * pb.npairs;
    switch (choose_occupancy(blocks, tn)) {
    case 3: launch_variant<2, 3>(Rstack, stack, flow_in, flow_out, acc, pb, H, W, scale, weight, fs, st, tn.lds_pad, wm); break;
    case 5: launch_variant<2, 5>(Rstack, stack, flow_in, flow_out, acc, pb, H, W, scale, weight, fs, st, tn.lds_pad, wm); break;
    default:
        // tn.fused_occ = 8: two bands per workgroup (the same 4 bands per CU, less HBM traffic; the default until the
        // edge bands lost their ds_bpermute path -- since then one band per workgroup is 2 % faster on large grids too)
        if (tn.fused_occ == 8) launch_variant<2, 4, 2>(Rstack, stack, flow_in, flow_out, acc, pb, H, W, scale, weight, fs, st, tn.lds_pad, wm);
        else launch_variant<2, 4>(Rstack, stack, flow_in, flow_out, acc, pb, H, W, scale, weight, fs, st, tn.lds_pad, wm);
        break;
    }
}

} // namespace fdn
