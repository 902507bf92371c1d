// fdn_fused2.hip -- the fused chain-step kernel with TWO image columns per lane.
//
// Same algorithm, same stage-parallel workgroup as fdn_fused.hip (read its header first): one
// 256-thread workgroup marches down one band of one (target, neighbour) pair, its four waves are the
// pipeline stages A, 1, 2, 3, rows stream between them through LDS rings, one LDS-only barrier per
// row.  Here a lane owns columns xb + 2l and xb + 2l + 1, so a band is 128 columns wide:
//   * fdn_fused.hip is instruction-issue bound (~4 cycles per wave instruction of any type).  With
//     two columns per lane the f32 arithmetic of FarnebackUpdateMatrices runs on packed
//     v_pk_mul_f32 / v_pk_add_f32 (exact IEEE per component, so results do not change), ring rows
//     move as 8-byte LDS accesses, the 5-wide box sum needs half the DPP lane shifts per column
//     (two of the five window columns are in the lane itself or arrive with one shift), and the
//     scalar/address overhead of a row step is shared by twice the pixels;
//   * the three iterations cost 6 columns either side of a band: 116 of 128 columns useful (91 %)
//     instead of 52 of 64 (81 %).
//   LDS per workgroup: M rings [3][7 rows][5 ch][64 lanes] float2 = 53.8 KB -> 2 workgroups per CU
//   (the target's R0 rows are re-read from L2 by the two middle stages instead of an LDS ring).
#include "fdn_internal.h"
#include "fdn_device.h"
#include <type_traits>

namespace fdn {

static __device__ __forceinline__ double wave_shr1_2(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ double wave_shl1_2(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ void lds_barrier2()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

static __device__ __forceinline__ float2 sel2(bool c0, bool c1, float2 a, float2 b) { return make_float2(c0 ? a.x : b.x, c1 ? a.y : b.y); }

// FarnebackUpdateMatrices after the gather, for the lane's two pixels at once (packed f32; every
// operation is the per-component twin of finish_M in fdn_device.h)
static __device__ __forceinline__ void finish_M2(const float2 r0[5], const float2 s[5], bool in0, bool in1, float2 d_x, float2 d_y,
                                                 float2 bxx, float by0, float by1, bool damp0, bool damp1, float2 m[5])
{
    const float2 zero = make_float2(0.f, 0.f), half = make_float2(0.5f, 0.5f), quarter = make_float2(0.25f, 0.25f);
    float2 r2 = sel2(in0, in1, s[0], zero);
    float2 r3 = sel2(in0, in1, s[1], zero);
    float2 r4 = sel2(in0, in1, (r0[2] + s[2]) * half, r0[2]);
    float2 r5 = sel2(in0, in1, (r0[3] + s[3]) * half, r0[3]);
    float2 r6 = sel2(in0, in1, (r0[4] + s[4]) * quarter, r0[4] * half);
    r2 = (r0[0] - r2) * half;
    r3 = (r0[1] - r3) * half;
    r2 = r2 + (r4 * d_y + r6 * d_x);
    r3 = r3 + (r6 * d_y + r5 * d_x);
    const float2 sc = bxx * make_float2(by0, by0) * make_float2(by1, by1);
    const float2 scale = sel2(damp0, damp1, sc, make_float2(1.f, 1.f));
    r2 = r2 * scale; r3 = r3 * scale; r4 = r4 * scale; r5 = r5 * scale; r6 = r6 * scale;
    m[0] = r4 * r4 + r6 * r6;
    m[1] = (r4 + r5) * r6;
    m[2] = r5 * r5 + r6 * r6;
    m[3] = r4 * r2 + r6 * r3;
    m[4] = r6 * r2 + r5 * r3;
}

template <int MH, bool HAS_FIN>
__global__ __launch_bounds__(256) void k_farneback_fused2(const float* __restrict__ Rstack, const float* __restrict__ stack,
                                                          const float* __restrict__ flow_in_base, float* __restrict__ flow_out_base,
                                                          float* __restrict__ acc_base, PairBatch pb, int H, int W,
                                                          double scale, double weight, int nbands)
{
    constexpr int ITERS = 3;
    constexpr int STEP = MH + 1;
    constexpr int RSP = 2 * MH + 3;
    constexpr int HALO = MH * ITERS;
    constexpr int BW = 128 - 2 * HALO;
    static_assert(MH == 2, "written for the 5-wide box of winsize 4/5");
    __shared__ float2 Mring[ITERS][RSP][5][64];

    const int lane = threadIdx.x & 63;
    const int stage = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware order (see fdn_fused.hip): XCD j works on the j-th contiguous eighth of the (pair, band) list
    const long nwg = gridDim.x, q8 = nwg >> 3, rem8 = nwg & 7;
    const long xcd = blockIdx.x & 7;
    const long gw = xcd * q8 + (xcd < rem8 ? xcd : rem8) + (blockIdx.x >> 3);
    const int b = (int)(gw / nbands);
    const int band = (int)(gw - (long)b * nbands);
    const int xb = band * BW - HALO;             // column of lane 0's first pixel (even)
    const int x0 = xb + 2 * lane, x1 = x0 + 1;
    const int xc0 = clampi(x0, 0, W - 1), xc1 = clampi(x1, 0, W - 1);
    // block-uniform: some window column of some lane lies outside the image (or a lane's pair is split by clamping)
    const bool edge_band = xb < MH || xb + 127 + MH > W - 1;
    const size_t HW = (size_t)H * W;
    const float* R0 = Rstack + (size_t)(pb.t0 + b) * 5 * HW;
    const float* R1 = Rstack + (size_t)(pb.t0 + b + pb.d) * 5 * HW;
    const float2 bxx = make_float2(border_factor(xc0, W), border_factor(xc1, W));
    const bool xd0 = border_test(xc0, W), xd1 = border_test(xc1, W);

    auto row_factor = [&](int y, float& by0, float& by1) {
        by0 = y < 5 ? (y < 2 ? 0.14f : 0.4472f) : 1.f;
        by1 = y >= H - 5 ? (H - y - 1 < 2 ? 0.14f : 0.4472f) : 1.f;
    };
    // the lane's two values of one plane row
    auto load2 = [&](auto ET, const float* plane, int row) __attribute__((always_inline)) -> float2 {
        const float* p = plane + (size_t)row * W;
        if (!decltype(ET)::value) { float2u v = *(const float2u*)(p + x0); return make_float2(v.a, v.b); }
        return make_float2(p[xc0], p[xc1]);
    };
    // UpdateMatrices of the lane's two pixels of row ys
    auto update_matrices2 = [&](int ys, float2 f0, float2 f1, const float2 r0[5], float2 mm[5]) __attribute__((always_inline)) {
        int xa, ya, xb2, yb2; float fxa, fya, fxb, fyb;
        flow_target(xc0, ys, f0.x, f0.y, xa, ya, fxa, fya);
        flow_target(xc1, ys, f1.x, f1.y, xb2, yb2, fxb, fyb);
        GatherTaps ga, gb;
        gather_R1(R1, HW, H, W, xa, ya, ga);
        gather_R1(R1, HW, H, W, xb2, yb2, gb);
        const bool ia = (unsigned)xa < (unsigned)(W - 1) && (unsigned)ya < (unsigned)(H - 1);
        const bool ib = (unsigned)xb2 < (unsigned)(W - 1) && (unsigned)yb2 < (unsigned)(H - 1);
        const float a00 = (1.f - fxa) * (1.f - fya), a01 = fxa * (1.f - fya), a10 = (1.f - fxa) * fya, a11 = fxa * fya;
        const float b00 = (1.f - fxb) * (1.f - fyb), b01 = fxb * (1.f - fyb), b10 = (1.f - fxb) * fyb, b11 = fxb * fyb;
        float2 s[5];
#pragma unroll
        for (int c = 0; c < 5; c++) {
            s[c].x = a00 * ga.t0[c].a + a01 * ga.t0[c].b + a10 * ga.t1[c].a + a11 * ga.t1[c].b;
            s[c].y = b00 * gb.t0[c].a + b01 * gb.t0[c].b + b10 * gb.t1[c].a + b11 * gb.t1[c].b;
        }
        float by0, by1;
        row_factor(ys, by0, by1);
        const bool yd = border_test(ys, H);
        finish_M2(r0, s, ia, ib, make_float2(f0.x, f1.x), make_float2(f0.y, f1.y), bxx, by0, by1, xd0 || yd, xd1 || yd, mm);
    };

    const int T = H + ITERS * STEP;

    using std::integral_constant;
    if (stage == 0) {
        // ===== wave 0: stage A ========================================================================
        const float* flow_in = HAS_FIN ? flow_in_base + (size_t)b * HW * 2 : nullptr;
        auto stageA = [&](auto ET) __attribute__((always_inline)) {
            constexpr bool EDGE = decltype(ET)::value;
            auto load_flow2 = [&](int row, float2& f0, float2& f1) __attribute__((always_inline)) {
                if (!HAS_FIN) { f0 = f1 = make_float2(0.f, 0.f); return; }
                const float* p = flow_in + (size_t)row * W * 2;
                float2u a = *(const float2u*)(p + 2 * (EDGE ? xc0 : x0)), c = *(const float2u*)(p + 2 * (EDGE ? xc1 : x0 + 1));
                f0 = make_float2(a.a, a.b); f1 = make_float2(c.a, c.b);
            };
            float2 fN0, fN1, r0N[5];
            load_flow2(0, fN0, fN1);
#pragma unroll
            for (int c = 0; c < 5; c++) r0N[c] = load2(ET, R0 + c * HW, 0);
            lds_barrier2();
            for (int t = 0; t < T; t++) {
                if (t < H) {
                    const float2 f0 = fN0, f1 = fN1;
                    float2 r0[5];
#pragma unroll
                    for (int c = 0; c < 5; c++) r0[c] = r0N[c];
                    const int tn = t + 1 < H ? t + 1 : H - 1;
                    load_flow2(tn, fN0, fN1);
#pragma unroll
                    for (int c = 0; c < 5; c++) r0N[c] = load2(ET, R0 + c * HW, tn);
                    float2 mm[5];
                    update_matrices2(t, f0, f1, r0, mm);
                    const int s = t % RSP;
#pragma unroll
                    for (int c = 0; c < 5; c++) Mring[0][s][c][lane] = mm[c];
                }
                lds_barrier2();
            }
        };
        if (edge_band) stageA(integral_constant<bool, true>{}); else stageA(integral_constant<bool, false>{});
        return;
    }

    // ===== waves 1..3: iteration `stage` (instantiated per stage and band kind) ===========================
    const bool own_lane = 2 * lane >= HALO && 2 * lane + 1 < 128 - HALO;
    const bool own0 = own_lane && x0 < W, own1 = own_lane && x1 < W;
    const float* img1 = stack + (size_t)(pb.t0 + b + pb.d) * HW;
    float* flow_out = flow_out_base ? flow_out_base + (size_t)b * HW * 2 : nullptr;
    float* acc = acc_base + (size_t)b * HW;
    // edge bands: band-relative column of clamp(x + j) for the five window columns of each pixel
    int rel0[5], rel1[5];
#pragma unroll
    for (int j = -MH; j <= MH; j++) {
        rel0[j + MH] = clampi(clampi(x0 + j, 0, W - 1) - xb, 0, 127);
        rel1[j + MH] = clampi(clampi(x1 + j, 0, W - 1) - xb, 0, 127);
    }
    auto column_value = [&](double v0, double v1, int rel) __attribute__((always_inline)) -> double {
        const double a = __shfl(v0, rel >> 1, 64), c = __shfl(v1, rel >> 1, 64);
        return (rel & 1) ? c : a;
    };
    auto stage_loop = [&](auto KT, auto ET) __attribute__((always_inline)) {
        constexpr int K = decltype(KT)::value;
        constexpr bool EDGE = decltype(ET)::value;
        float2 (*Min)[5][64] = Mring[K - 1];
        double vs0[5], vs1[5];
#pragma unroll
        for (int c = 0; c < 5; c++) vs0[c] = vs1[c] = 0.;
        lds_barrier2();
        for (int t = 0; t < T; t++) {
            const int y = t - K * STEP;
            if (y >= 0 && y < H) {
                float2 acc_old = make_float2(0.f, 0.f);
                if (K == ITERS) acc_old = load2(ET, acc, y);
                if (y == 0) { // vsum before row 0: f32(M[0]*(m+2)) + rows 1..m-1 (clamped)
#pragma unroll
                    for (int c = 0; c < 5; c++) {
                        const float2 m0 = Min[0][c][lane];
                        double v0 = (double)(m0.x * (float)(MH + 2)), v1 = (double)(m0.y * (float)(MH + 2));
#pragma unroll
                        for (int yy = 1; yy < MH; yy++) {
                            const float2 mr = Min[(yy < H - 1 ? yy : H - 1) % RSP][c][lane];
                            v0 += (double)mr.x; v1 += (double)mr.y;
                        }
                        vs0[c] = v0; vs1[c] = v1;
                    }
                }
                const int rn = (y + MH < H - 1 ? y + MH : H - 1) % RSP;
                const int ro = (y - MH - 1 > 0 ? y - MH - 1 : 0) % RSP;
                double a0[5], a1[5];
#pragma unroll
                for (int c = 0; c < 5; c++) {
                    const float2 d = Min[rn][c][lane] - Min[ro][c][lane];
                    vs0[c] += (double)d.x;
                    vs1[c] += (double)d.y;
                    double s0 = 0, s1 = 0;
                    if (EDGE) {
#pragma unroll
                        for (int j = 0; j <= 2 * MH; j++) {
                            s0 += column_value(vs0[c], vs1[c], rel0[j]);
                            s1 += column_value(vs0[c], vs1[c], rel1[j]);
                        }
                    } else { // columns 2l-2 .. 2l+3: left lane's pair, own pair, right lane's pair
                        const double l0 = wave_shr1_2(vs0[c]), l1 = wave_shr1_2(vs1[c]);
                        const double g0 = wave_shl1_2(vs0[c]), g1 = wave_shl1_2(vs1[c]);
                        s0 += l0; s0 += l1; s0 += vs0[c]; s0 += vs1[c]; s0 += g0;
                        s1 += l1; s1 += vs0[c]; s1 += vs1[c]; s1 += g0; s1 += g1;
                    }
                    a0[c] = s0; a1[c] = s1;
                }
                const float2 f0 = solve_flow(a0, scale), f1 = solve_flow(a1, scale);
                if (K < ITERS) {
                    float2 r0[5], mm[5];
#pragma unroll
                    for (int c = 0; c < 5; c++) r0[c] = load2(ET, R0 + c * HW, y);
                    update_matrices2(y, f0, f1, r0, mm);
                    const int s = y % RSP;
#pragma unroll
                    for (int c = 0; c < 5; c++) Mring[K < ITERS ? K : 0][s][c][lane] = mm[c];
                } else {
                    const float w0 = remap_sample(img1, H, W, xc0, y, f0), w1 = remap_sample(img1, H, W, xc1, y, f1);
                    const float n0 = (float)((double)acc_old.x + (double)w0 * weight);
                    const float n1 = (float)((double)acc_old.y + (double)w1 * weight);
                    const size_t o = (size_t)y * W;
                    if (own0) { if (flow_out) *(float2u*)(flow_out + 2 * (o + x0)) = float2u{f0.x, f0.y}; acc[o + x0] = n0; }
                    if (own1) { if (flow_out) *(float2u*)(flow_out + 2 * (o + x1)) = float2u{f1.x, f1.y}; acc[o + x1] = n1; }
                }
            }
            lds_barrier2();
        }
    };
    if (stage == 1) { if (edge_band) stage_loop(integral_constant<int, 1>{}, integral_constant<bool, true>{}); else stage_loop(integral_constant<int, 1>{}, integral_constant<bool, false>{}); }
    else if (stage == 2) { if (edge_band) stage_loop(integral_constant<int, 2>{}, integral_constant<bool, true>{}); else stage_loop(integral_constant<int, 2>{}, integral_constant<bool, false>{}); }
    else { if (edge_band) stage_loop(integral_constant<int, 3>{}, integral_constant<bool, true>{}); else stage_loop(integral_constant<int, 3>{}, integral_constant<bool, false>{}); }
}

void launch_farneback_fused2(const float* Rstack, const float* stack, const float* flow_in, float* flow_out, float* acc,
                             PairBatch pb, int H, int W, int winsize, double weight, hipStream_t st)
{
    if (pb.npairs <= 0) return;
    constexpr int MH = 2;
    const int BW = 128 - 2 * MH * 3;
    int nbands = (W + BW - 1) / BW;
    long blocks = (long)nbands * pb.npairs;
    double scale = 1. / ((double)winsize * winsize);
    dim3 grid((unsigned)blocks);
    if (flow_in)
        hipLaunchKernelGGL((k_farneback_fused2<MH, true>), grid, dim3(256), 0, st,
                           Rstack, stack, flow_in, flow_out, acc, pb, H, W, scale, weight, nbands);
    else
        hipLaunchKernelGGL((k_farneback_fused2<MH, false>), grid, dim3(256), 0, st,
                           Rstack, stack, flow_in, flow_out, acc, pb, H, W, scale, weight, nbands);
}

} // namespace fdn
