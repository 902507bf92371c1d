// fdn_device.h -- device functions shared by the staged kernels (fdn_kernels.hip) and the fused
// chain-step kernel (fdn_fused.hip).  Compiled with -ffp-contract=off: every multiply and add
// below rounds separately, as in the CPU code the reference runs (see oracle/fdn_oracle.c).
#pragma once
#include "fdn_internal.h"

namespace fdn {

static __device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// clamp(v, 0, hi) for a wave-uniform hi >= 0 as ONE instruction: the compiler turns the compare/select form into
// v_cmp + v_min + v_cndmask (it forms v_med3_i32 only for two constants); hot paths of the Farneback kernels only.
static __device__ __forceinline__ int clamp0u(int v, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "s"(hi));
    return r;
}
static __device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// ---------------------------------------------------------------------------------
// Order in which a launch walks its pairs.  Slice s of the stack is the TARGET of pair s and the NEIGHBOUR of pair
// s - d: its polynomial expansion (20 B/px) is read by both.  Walking the pairs in chains t, t + |d|, t + 2 |d|, ...
// puts those two pairs next to each other in the workgroup list -- same XCD, started together, marching down the rows
// in step -- so that the second reader finds the expansion in that XCD's L2 instead of fetching it from HBM again.
// Speed (energy) only: position i of the walk -> pair index.  FDN_PAIR_CHAINS = 0: pairs in index order.
// ---------------------------------------------------------------------------------
#ifndef FDN_PAIR_CHAINS
#define FDN_PAIR_CHAINS 1
#endif
static __device__ __forceinline__ int pair_walk(int i, int npairs, int d)
{
    const int ad = d < 0 ? -d : d;
    if (!FDN_PAIR_CHAINS || ad < 1 || ad >= npairs) return i;
    const int L = npairs / ad, rem = npairs - L * ad;      // chains c < rem have L + 1 members, the others L
    const int big = rem * (L + 1);
    int c, k;
    if (i < big) { c = i / (L + 1); k = i - c * (L + 1); }
    else { const int j = i - big; c = rem + j / L; k = j - (c - rem) * L; }
    return c + k * ad;
}

// ---------------------------------------------------------------------------------
// FarnebackUpdateMatrices' 5-pixel border damping.
//   border_factor: border[x]-factor product for one coordinate, ((x<5 ? b[x] : 1) * (x>=W-5 ? b[W-1-x] : 1))
// ---------------------------------------------------------------------------------
static __device__ __forceinline__ float border_factor(int i, int n)
{
    // border[] = {0.14, 0.14, 0.4472, 0.4472, 0.4472}; product of the near-edge and far-edge factors
    float b0 = i < 5 ? (i < 2 ? 0.14f : 0.4472f) : 1.f;
    float b1 = i >= n - 5 ? (n - i - 1 < 2 ? 0.14f : 0.4472f) : 1.f;
    return b0 * b1;
}

// OpenCV's "(unsigned)(i - BORDER) >= (unsigned)(n - BORDER*2)"
static __device__ __forceinline__ bool border_test(int i, int n) { return (unsigned)(i - 5) >= (unsigned)(n - 10); }

// Global accesses as uniform base + 32-bit per-lane BYTE offset: the compiler then emits the
// "saddr + voffset" form (global_load v, v_off, s[base]) and no 64-bit per-lane address arithmetic.
// Needs every plane / image to be smaller than 4 GiB (checked by the launchers' callers).
// A wave-uniform pointer the optimiser cannot fold into other address arithmetic: keeps one SGPR
// base per plane instead of re-deriving plane addresses with 64-bit per-lane adds.
template <typename T> static __device__ __forceinline__ T* uniform_ptr(T* p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T*)(((unsigned long long)hi << 32) | lo);
}
#define FDN_GLOBAL __attribute__((address_space(1)))     // global memory, stated (uniform_ptr's integer round trip hides it)
typedef float fdn_v2f __attribute__((ext_vector_type(2)));
template <typename T> static __device__ __forceinline__ T ld_off(const void* base, unsigned byte_off);
template <> __device__ __forceinline__ float ld_off<float>(const void* base, unsigned byte_off)
{
    return *(const FDN_GLOBAL float*)((const FDN_GLOBAL char*)base + byte_off);
}
template <> __device__ __forceinline__ float2 ld_off<float2>(const void* base, unsigned byte_off)
{
    const fdn_v2f v = *(const FDN_GLOBAL fdn_v2f*)((const FDN_GLOBAL char*)base + byte_off);
    return make_float2(v.x, v.y);
}
typedef float fdn_v4f __attribute__((ext_vector_type(4)));
template <> __device__ __forceinline__ fdn_v2f ld_off<fdn_v2f>(const void* base, unsigned byte_off)
{
    return *(const FDN_GLOBAL fdn_v2f*)((const FDN_GLOBAL char*)base + byte_off);
}
// 16 bytes at an 8-byte aligned address / 8 bytes at a 4-byte aligned one (two adjacent pixels)
static __device__ __forceinline__ fdn_v4f ld_off_v4a8(const void* base, unsigned byte_off)
{
    typedef fdn_v4f __attribute__((aligned(8))) v4_a8;
    return *(const FDN_GLOBAL v4_a8*)((const FDN_GLOBAL char*)base + byte_off);
}
static __device__ __forceinline__ fdn_v2f ld_off_v2a4(const void* base, unsigned byte_off)
{
    typedef fdn_v2f __attribute__((aligned(4))) v2_a4;
    return *(const FDN_GLOBAL v2_a4*)((const FDN_GLOBAL char*)base + byte_off);
}
static __device__ __forceinline__ void st_off(void* base, unsigned byte_off, float v)
{
    *(FDN_GLOBAL float*)((FDN_GLOBAL char*)base + byte_off) = v;
}
static __device__ __forceinline__ void st_off(void* base, unsigned byte_off, float2 v)
{
    fdn_v2f w; w.x = v.x; w.y = v.y;
    *(FDN_GLOBAL fdn_v2f*)((FDN_GLOBAL char*)base + byte_off) = w;
}

// ---------------------------------------------------------------------------------
// Layout of a polynomial expansion R in HBM (one image): the five channels as two interleaved
// pairs and one plane,
//     [ (c0, c1) x HW ][ (c2, c3) x HW ][ c4 x HW ]          (20 bytes per pixel, as planar)
// so that a pixel's pair arrives in the two halves of a 64-bit register with one load and the
// packed f32 instructions (v_pk_mul/add_f32) work on two channels at a time without shuffling.
// Pixels are addressed by 32-bit byte offsets: H * W < 2^29 (checked on the host).
// ---------------------------------------------------------------------------------
struct RImage { const float* p01; const float* p23; const float* p4; };
static __device__ __forceinline__ RImage r_image(const float* base, size_t HW)
{
    RImage r; r.p01 = base; r.p23 = base + 2 * HW; r.p4 = base + 4 * HW;
    return r;
}
static __device__ __forceinline__ void load_R(const RImage& R, unsigned px, fdn_v2f& r01, fdn_v2f& r23, float& r4)
{
    r01 = ld_off<fdn_v2f>(R.p01, px * 8u);
    r23 = ld_off<fdn_v2f>(R.p23, px * 8u);
    r4 = ld_off<float>(R.p4, px * 4u);
}

// The 2 x 2 bilinear footprint of all five channels: [pair]: taps (x1, y1), (x1+1, y1), (x1, y1+1),
// (x1+1, y1+1) of channels (2 pair, 2 pair + 1); ...s: channel 4.
struct GatherTapsP {
    fdn_v2f a0[2], b0[2], a1[2], b1[2];
    float a0s, b0s, a1s, b1s;
};

// FarnebackUpdateMatrices' gather at the clamped position (the out-of-image case is a select in
// finish_M_p): two 16-byte and one 8-byte load per row.  Needs H >= 2, W >= 2, H, W < 2^24.
static __device__ __forceinline__ void gather_R1_p(const RImage& R1, int H, int W, int x1, int y1, GatherTapsP& g)
{
    const unsigned px = __umul24((unsigned)clamp0u(y1, H - 2), (unsigned)W) + (unsigned)clamp0u(x1, W - 2);
    const unsigned px1 = px + (unsigned)W;
    fdn_v4f t;
    t = ld_off_v4a8(R1.p01, px * 8u);  g.a0[0] = t.xy; g.b0[0] = t.zw;
    t = ld_off_v4a8(R1.p23, px * 8u);  g.a0[1] = t.xy; g.b0[1] = t.zw;
    t = ld_off_v4a8(R1.p01, px1 * 8u); g.a1[0] = t.xy; g.b1[0] = t.zw;
    t = ld_off_v4a8(R1.p23, px1 * 8u); g.a1[1] = t.xy; g.b1[1] = t.zw;
    fdn_v2f u;
    u = ld_off_v2a4(R1.p4, px * 4u);  g.a0s = u.x; g.b0s = u.y;
    u = ld_off_v2a4(R1.p4, px1 * 4u); g.a1s = u.x; g.b1s = u.y;
}

static __device__ __forceinline__ void flow_target(float xf, float yf, float dx, float dy, int& x1, int& y1, float& fx, float& fy)
{
    fx = xf + dx; fy = yf + dy;
    float flx = floorf(fx), fly = floorf(fy);
    x1 = (int)flx; y1 = (int)fly;
    fx -= flx; fy -= fly;
}

// (w.x v.x, w.x v.y) and (w.y v.x, w.y v.y): a weight held in one half of a register pair times a channel pair, the
// broadcast done by the instruction's operand selects (the compiler splats such a weight with two v_mov_b32 first)
#ifndef FDN_PK_OPSEL
#define FDN_PK_OPSEL 0
#endif
static __device__ __forceinline__ fdn_v2f pk_mul_lo(fdn_v2f w, fdn_v2f v)
{
    fdn_v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(w), "v"(v));
    return r;
}
static __device__ __forceinline__ fdn_v2f pk_mul_hi(fdn_v2f w, fdn_v2f v)
{
    fdn_v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(w), "v"(v));
    return r;
}

// FarnebackUpdateMatrices for one pixel from its operands: (r01, r23, r4c) = R0 at the pixel, g = the
// neighbour expansion's 2 x 2 footprint at (x1, y1) = floor(p + flow), (fx, fy) the fractions (exact f32
// weights, no quantisation), (dx, dy) the flow.  The gather ran at a clamped position; the
// out-of-image case is a select here.  Channel pairs stay packed
// (v_pk_* f32); element-wise the operations and their order are those of OpenCV's scalar loop, so the
// values are bit-identical to the CPU form.  Result: M as (m0, m2), m1, (m3, m4).
static __device__ __forceinline__ void finish_M_p(fdn_v2f r01, fdn_v2f r23, float r4c, const GatherTapsP& g, int H, int W, int x1, int y1,
                                                  float fx, float fy, float dx, float dy, float bxx, float by0, float by1,
                                                  bool damp, fdn_v2f& m02, float& m1, fdn_v2f& m34)
{
    const bool inside = (unsigned)x1 < (unsigned)(W - 1) && (unsigned)y1 < (unsigned)(H - 1);
    // (splatting each weight over a channel pair costs the compiler two moves per weight; writing the eight products as
    // v_pk_mul_f32 with op_sel by hand removes 18 of the fused kernel's 734 instructions per row and changes nothing
    // measurable: 17.1 -> 17.2 ms)
    const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
#if FDN_PK_OPSEL
    const fdn_v2f w0 = {a00, a01}, w1 = {a10, a11};
    const fdn_v2f s01 = pk_mul_lo(w0, g.a0[0]) + pk_mul_hi(w0, g.b0[0]) + pk_mul_lo(w1, g.a1[0]) + pk_mul_hi(w1, g.b1[0]);
    const fdn_v2f s23 = pk_mul_lo(w0, g.a0[1]) + pk_mul_hi(w0, g.b0[1]) + pk_mul_lo(w1, g.a1[1]) + pk_mul_hi(w1, g.b1[1]);
#else
    const fdn_v2f s01 = a00 * g.a0[0] + a01 * g.b0[0] + a10 * g.a1[0] + a11 * g.b1[0];
    const fdn_v2f s23 = a00 * g.a0[1] + a01 * g.b0[1] + a10 * g.a1[1] + a11 * g.b1[1];
#endif
    const float s4 = a00 * g.a0s + a01 * g.b0s + a10 * g.a1s + a11 * g.b1s;
    const fdn_v2f zero = {0.f, 0.f};
    fdn_v2f r23v = inside ? s01 : zero;                       // (r2, r3)
    fdn_v2f r45 = inside ? (r23 + s23) * 0.5f : r23;          // (r4, r5)
    float r6 = inside ? (r4c + s4) * 0.25f : r4c * 0.5f;
    r23v = (r01 - r23v) * 0.5f;
    r23v.x = r23v.x + (r45.x * dy + r6 * dx);
    r23v.y = r23v.y + (r6 * dy + r45.y * dx);
    // ((bx0*bx1)*by0)*by1 as OpenCV, == 1.0f away from the border.  `damp` is OpenCV's own region test
    // ((unsigned)(x-5) >= (unsigned)(W-10) || same for y): for images under 10 pixels it is NOT
    // "within 5 pixels of an edge" (the unsigned difference wraps), and the factors are skipped.
    const float scale = damp ? bxx * by0 * by1 : 1.f;
    r23v *= scale; r45 *= scale; r6 *= scale;
    const float r66 = r6 * r6;
    m02 = r45 * r45 + r66;                                    // (r4 r4 + r6 r6, r5 r5 + r6 r6)
    m1 = (r45.x + r45.y) * r6;
    const fdn_v2f p = r45 * r23v;                             // (r4 r2, r5 r3)
    const fdn_v2f q = r6 * r23v.yx;                           // (r6 r3, r6 r2)
    m34.x = p.x + q.x;                                        // r4 r2 + r6 r3
    m34.y = q.y + p.y;                                        // r6 r2 + r5 r3
}

// FarnebackUpdateMatrices for one pixel (the per-stage kernels): m[5] = M at (x, y) for flow (dx, dy)
static __device__ __forceinline__ void compute_M(const RImage& R0, const RImage& R1, int H, int W, int x, int y,
                                                 float dx, float dy, float m[5])
{
    fdn_v2f r01, r23; float r4;
    load_R(R0, (unsigned)y * (unsigned)W + (unsigned)x, r01, r23, r4);
    int x1, y1; float fx, fy;
    flow_target((float)x, (float)y, dx, dy, x1, y1, fx, fy);
    GatherTapsP g;
    gather_R1_p(R1, H, W, x1, y1, g);
    float by0 = y < 5 ? (y < 2 ? 0.14f : 0.4472f) : 1.f;
    float by1 = y >= H - 5 ? (H - y - 1 < 2 ? 0.14f : 0.4472f) : 1.f;
    fdn_v2f m02, m34;
    finish_M_p(r01, r23, r4, g, H, W, x1, y1, fx, fy, dx, dy, border_factor(x, W), by0, by1,
               border_test(x, W) || border_test(y, H), m02, m[1], m34);
    m[0] = m02.x; m[2] = m02.y; m[3] = m34.x; m[4] = m34.y;
}

// 1.0 / x, correctly rounded, for 2^-500 < |x| < 2^500 (here x = det + 1e-3 of f64 sums of f32 products: about
// [1e-3, 1e80]).  The compiler's IEEE division is v_div_scale x2, v_rcp, four fma, v_mul, v_fma, v_div_fmas,
// v_div_fixup = 11 instructions; in this range neither scaling nor fix-up does anything (the quotient estimate
// 1.0 * y is y itself), which leaves the reciprocal estimate, two Newton steps and the final residual correction:
// the same fma chain, the same bits, 7 instructions.  NaN stays NaN; x = inf gives NaN instead of 0 (such a
// system has NaN flows in the reference as well: inf * 0).
static __device__ __forceinline__ double reciprocal(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(e, y, y);
}

static __device__ __forceinline__ float2 solve_flow(const double a[5], double scale)
{
    double g11 = a[0] * scale, g12 = a[1] * scale, g22 = a[2] * scale, h1 = a[3] * scale, h2 = a[4] * scale;
    double idet = reciprocal(g11 * g22 - g12 * g12 + 1e-3);
    float2 f;
    f.x = (float)((g11 * h2 - g12 * h1) * idet);
    f.y = (float)((g22 * h1 - g12 * h2) * idet);
    return f;
}

// cv::resize INTER_LINEAR of a 2-channel f32 image (HResizeLinear then VResizeLinear, f32
// coefficients) sampled at one destination pixel, times `ps` in f64: calc()'s upsampling of the flow
// between pyramid levels.  scale_x = sw / dw, scale_y = sh / dh.
// a * b + c with one rounding (fused) or two: the "opencv_fma" option (FmaMode, fdn_internal.h).  OpenCV's SIMD filter and
// resize loops use v_muladd, which is a fused multiply-add on AVX2 / FMA3 builds and two operations in their scalar row
// tails; which of the two a given cv2 wheel runs at a given pixel is one of the unknowns of an unpinned cv2.
// i: the element's index in its row of `width` elements (channels interleaved): mode 1 fuses everywhere, mode 2 on the
// vector body -- the first (width / lanes) * lanes elements -- and not on the tail.
static __device__ __forceinline__ bool fma_at(const FmaMode& fm, int i, int width)
{
    return fm.mode == 1 || (fm.mode == 2 && i < width / fm.lanes * fm.lanes);
}
static __device__ __forceinline__ float madf(bool fused, float a, float b, float c)
{
    return fused ? __builtin_fmaf(a, b, c) : a * b + c;
}

struct LinearTap { int s0, s1; float a0, a1; };
static __device__ __forceinline__ LinearTap linear_tap(int d, double scale, int n)
{
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    float fl = floorf(f);
    int s = (int)fl;
    f -= fl;
    if (s < 0) { f = 0; s = 0; }
    if (s >= n - 1) { f = 0; s = n - 1; }
    LinearTap t;
    t.s0 = s; t.s1 = s + 1 < n ? s + 1 : n - 1; t.a1 = f; t.a0 = 1.f - f;
    return t;
}
// dx, dw: the destination column and row length (the "opencv_fma" option goes by the element's place in its row: VResizeLinear)
static __device__ __forceinline__ float2 resize_linear_flow(const float* __restrict__ src, int sw, const LinearTap& tx, const LinearTap& ty, double ps,
                                                            const FmaMode& fm = FmaMode(), int dx = 0, int dw = 0)
{
    const unsigned r0 = (unsigned)ty.s0 * (unsigned)sw, r1 = (unsigned)ty.s1 * (unsigned)sw;
    const float2 p00 = ld_off<float2>(src, (r0 + tx.s0) * 8u), p01 = ld_off<float2>(src, (r0 + tx.s1) * 8u);
    const float2 p10 = ld_off<float2>(src, (r1 + tx.s0) * 8u), p11 = ld_off<float2>(src, (r1 + tx.s1) * 8u);
    float2 v;
    v.x = madf(fma_at(fm, 2 * dx, 2 * dw), p00.x * tx.a0 + p01.x * tx.a1, ty.a0, (p10.x * tx.a0 + p11.x * tx.a1) * ty.a1);
    v.y = madf(fma_at(fm, 2 * dx + 1, 2 * dw), p00.y * tx.a0 + p01.y * tx.a1, ty.a0, (p10.y * tx.a0 + p11.y * tx.a1) * ty.a1);
    v.x = (float)((double)v.x * ps);
    v.y = (float)((double)v.y * ps);
    return v;
}

// remap in two halves so that a kernel can issue the four tap loads one pipeline step before it
// combines them: remap_issue computes the quantised position and loads, remap_finish weights.
// (ax, ay): the 5-bit table indices of the classic path; with the unquantised model (WarpMode::model = 1) the BITS of the two
// float32 fractions instead -- the same registers, so that the Farneback kernels' final stage does not grow)
struct RemapTaps { float v0, v1, v2, v3; int ax, ay; };

// PAIRS: fetch the two taps of a row with one 8-byte load.  Half the gather instructions -- what a memory-bound kernel
// needs (k_sweep_side: 8.6 -> 6.1 ms) -- for four selects more, which the VALU-bound Farneback kernels cannot afford
// (k_farneback_fused: 17.0 -> 17.9 ms): those keep the four dword loads.  Same values either way.
// unq (wave-uniform; the "remap_model" option): the float-map remap that does NOT round the coordinates to 1/32 pixel --
// plain float32 bilinear interpolation at the map position (remap_finish below) -- a model of the reworked linear remap
// newer OpenCV releases are reported to ship; the second known unknown of an unpinned cv2.
template <bool PAIRS = false>
static __device__ __forceinline__ void remap_issue(const float* __restrict__ src, int H, int W, int x, int y, float2 f, RemapTaps& r, bool unq = false)
{
    float mx = (float)((double)f.x + (double)x);
    float my = (float)((double)f.y + (double)y);
    int ix, iy;
    if (unq) {
        const float flx = floorf(mx), fly = floorf(my);
        r.ax = __float_as_int(mx - flx); r.ay = __float_as_int(my - fly);
        ix = (int)fminf(fmaxf(flx, -32768.f), 32767.f); iy = (int)fminf(fmaxf(fly, -32768.f), 32767.f);
    } else {
        // cvRound(v * INTER_TAB_SIZE); bounded so the int conversion is defined for wild flows
        float qx = fminf(fmaxf(rintf(mx * 32.f), -2147483520.f), 2147483520.f);
        float qy = fminf(fmaxf(rintf(my * 32.f), -2147483520.f), 2147483520.f);
        int sx = (int)qx, sy = (int)qy;
        r.ax = sx & 31; r.ay = sy & 31;
        ix = clampi(sx >> 5, -32768, 32767); iy = clampi(sy >> 5, -32768, 32767);
    }
    int xa = clamp0u(ix, W - 1), xb = clamp0u(ix + 1, W - 1);
    int ya = clamp0u(iy, H - 1), yb = clamp0u(iy + 1, H - 1);
    const unsigned oa = __umul24((unsigned)ya, (unsigned)W), ob = __umul24((unsigned)yb, (unsigned)W);   // H, W < 2^24
    if (PAIRS && W >= 2) {
        // the two taps of a row are neighbours except where the clamp folds them onto one pixel: one 8-byte load of the
        // pixel pair at clamp(ix, 0, W-2) serves both (half the gather instructions), the fold is a select
        const int xp = clampi(ix, 0, W - 2);
        const fdn_v2f pa = ld_off_v2a4(src, (oa + xp) * 4u), pb = ld_off_v2a4(src, (ob + xp) * 4u);
        r.v0 = xa == xp ? pa.x : pa.y; r.v1 = xb == xp ? pa.x : pa.y;
        r.v2 = xa == xp ? pb.x : pb.y; r.v3 = xb == xp ? pb.x : pb.y;
    } else {
        r.v0 = ld_off<float>(src, (oa + xa) * 4u); r.v1 = ld_off<float>(src, (oa + xb) * 4u);
        r.v2 = ld_off<float>(src, (ob + xa) * 4u); r.v3 = ld_off<float>(src, (ob + xb) * 4u);
    }
}

static __device__ __forceinline__ float remap_finish(const RemapTaps& r, bool unq = false)
{
    if (unq) {     // two float32 lerps: along x in both rows, then along y
        const float fx = __int_as_float(r.ax), fy = __int_as_float(r.ay);
        const float top = r.v0 * (1.f - fx) + r.v1 * fx, bot = r.v2 * (1.f - fx) + r.v3 * fx;
        return top * (1.f - fy) + bot * fy;
    }
    float tx1 = (float)r.ax * (1.f / 32), tx0 = 1.f - tx1;
    float ty1 = (float)r.ay * (1.f / 32), ty0 = 1.f - ty1;
    float w0 = ty0 * tx0, w1 = ty0 * tx1, w2 = ty1 * tx0, w3 = ty1 * tx1;
    return r.v0 * w0 + r.v1 * w1 + r.v2 * w2 + r.v3 * w3;
}

// cv2.remap of a CV_8U image: remapBilinear<FixedPtCast<int, uchar, 15>, RemapVec_8u, short>: the weights are the float
// table's entries x 2^15 as 16-bit integers -- for INTER_LINEAR the exact products (32 - ax)(32 - ay) 32 ... (the one entry
// that does not fit a short, 32768 at ax = ay = 0, is stored as 32767 with the 1 moved to another tap: an 8-bit result cannot
// tell) -- and the result is (sum + 2^14) >> 15.  Taps are 0..255 held as floats.
static __device__ __forceinline__ float remap_finish_u8(const RemapTaps& r)
{
    const int w0 = (32 - r.ax) * (32 - r.ay) * 32, w1 = r.ax * (32 - r.ay) * 32, w2 = (32 - r.ax) * r.ay * 32, w3 = r.ax * r.ay * 32;
    const int sum = (int)r.v0 * w0 + (int)r.v1 * w1 + (int)r.v2 * w2 + (int)r.v3 * w3;
    const int v = (sum + (1 << 14)) >> 15;
    return (float)(v < 0 ? 0 : v > 255 ? 255 : v);
}

// cv2.remap of a CV_64F image (remapBilinear<Cast<double, double>, ., float>): the float table weights widened, the four
// products and three sums in double, no rounding to float.  pad: the image is the constant `padv` (a mean-pad slice).
static __device__ __forceinline__ double remap_finish_f64(const RemapTaps& r, bool pad, double padv, bool unq = false)
{
    if (unq) {     // the same two lerps on a CV_64F image: values and sums in double, the float32 fractions widened
        const float fx = __int_as_float(r.ax), fy = __int_as_float(r.ay);
        const double v0 = pad ? padv : (double)r.v0, v1 = pad ? padv : (double)r.v1, v2 = pad ? padv : (double)r.v2, v3 = pad ? padv : (double)r.v3;
        const double top = v0 * (double)(1.f - fx) + v1 * (double)fx, bot = v2 * (double)(1.f - fx) + v3 * (double)fx;
        return top * (double)(1.f - fy) + bot * (double)fy;
    }
    float tx1 = (float)r.ax * (1.f / 32), tx0 = 1.f - tx1;
    float ty1 = (float)r.ay * (1.f / 32), ty0 = 1.f - ty1;
    float w0 = ty0 * tx0, w1 = ty0 * tx1, w2 = ty1 * tx0, w3 = ty1 * tx1;
    const double v0 = pad ? padv : (double)r.v0, v1 = pad ? padv : (double)r.v1, v2 = pad ? padv : (double)r.v2, v3 = pad ? padv : (double)r.v3;
    return v0 * (double)w0 + v1 * (double)w1 + v2 * (double)w2 + v3 * (double)w3;
}

// the same remap for a CV_64F image held as doubles (fdn_warp_typed): coordinates and weights as remap_issue / remap_finish_f64
static __device__ __forceinline__ double remap_sample_f64(const double* __restrict__ src, int H, int W, int x, int y, float2 f, bool unq = false)
{
    float mx = (float)((double)f.x + (double)x);
    float my = (float)((double)f.y + (double)y);
    if (unq) {
        const float flx = floorf(mx), fly = floorf(my);
        const float fx = mx - flx, fy = my - fly;
        const int ix = (int)fminf(fmaxf(flx, -32768.f), 32767.f), iy = (int)fminf(fmaxf(fly, -32768.f), 32767.f);
        const int xa = clampi(ix, 0, W - 1), xb = clampi(ix + 1, 0, W - 1), ya = clampi(iy, 0, H - 1), yb = clampi(iy + 1, 0, H - 1);
        const size_t oa = (size_t)ya * W, ob = (size_t)yb * W;
        const double top = src[oa + xa] * (double)(1.f - fx) + src[oa + xb] * (double)fx, bot = src[ob + xa] * (double)(1.f - fx) + src[ob + xb] * (double)fx;
        return top * (double)(1.f - fy) + bot * (double)fy;
    }
    float qx = fminf(fmaxf(rintf(mx * 32.f), -2147483520.f), 2147483520.f);
    float qy = fminf(fmaxf(rintf(my * 32.f), -2147483520.f), 2147483520.f);
    int sx = (int)qx, sy = (int)qy;
    const int ax = sx & 31, ay = sy & 31;
    int ix = clampi(sx >> 5, -32768, 32767), iy = clampi(sy >> 5, -32768, 32767);
    int xa = clampi(ix, 0, W - 1), xb = clampi(ix + 1, 0, W - 1);
    int ya = clampi(iy, 0, H - 1), yb = clampi(iy + 1, 0, H - 1);
    const size_t oa = (size_t)ya * W, ob = (size_t)yb * W;
    float tx1 = (float)ax * (1.f / 32), tx0 = 1.f - tx1;
    float ty1 = (float)ay * (1.f / 32), ty0 = 1.f - ty1;
    float w0 = ty0 * tx0, w1 = ty0 * tx1, w2 = ty1 * tx0, w3 = ty1 * tx1;
    return src[oa + xa] * (double)w0 + src[oa + xb] * (double)w1 + src[ob + xa] * (double)w2 + src[ob + xb] * (double)w3;
}

// acc <- f32(f64(acc) + warp_slice(neighbour, flow) * weight) (seq:106-107) in the semantics the volume's dtype gives the
// reference (fdn_sweep_params.warp_mode; WarpMode of fdn_internal.h):
//   0  float32 volume
//   1  seq on an integer MRC: the padded volume is float64 (seq:88-89): cv2.remap's CV_64F path (table weights widened,
//      products and sums in double, no rounding to float); `pad`: the neighbour is a pad slice, the constant float64 mean
//   2  par on an integer MRC: the neighbour is an integer image: remap rounds half to even and saturates (lo, hi)
//   UNQ (compile time: the "remap_model" option): unquantised float32 bilinear instead of the 1/32-pixel table.  A template
//   parameter and not a flag in a register: as a wave-uniform runtime branch it cost the 3-iteration kernel 4.8 % (15.31
//   against 14.61 ms per launch, same box, round 6) -- one VGPR more and a longer final stage.  The Farneback kernels are built
//   with it for float32 volumes (WM = 4 = 0 + UNQ); the integer semantics under the unquantised model run on the per-stage
//   path, whose k_sweep_side is memory-bound and takes the flag at run time.
template <int WM, bool UNQ = false>
static __device__ __forceinline__ float fold_warped(const float* __restrict__ src, int H, int W, int x, int y, float2 f, float acc_old,
                                                    double weight, bool pad, double pad64, float lo, float hi, bool fixed8 = false)
{
    RemapTaps r;
    // (8-bit images keep their fixed-point table in either remap model: the unquantised model is about float maps on float arithmetic)
    const bool u = UNQ && !(WM == 2 && fixed8);
    remap_issue<false>(src, H, W, x, y, f, r, u);
    if (WM == 1) return (float)((double)acc_old + remap_finish_f64(r, pad, pad64, u) * weight);
    //   2  ... and a uint8 one (fixed8, wave-uniform): cv2.remap's 8-bit fixed point, an integer already
    if (WM == 2) return (float)((double)acc_old + (double)(fixed8 ? remap_finish_u8(r) : fminf(fmaxf(rintf(remap_finish(r, u)), lo), hi)) * weight);
    return (float)((double)acc_old + (double)remap_finish(r, u) * weight);
}

static __device__ __forceinline__ float remap_sample(const float* __restrict__ src, int H, int W, int x, int y, float2 f, bool unq = false)
{
    RemapTaps r;
    remap_issue<false>(src, H, W, x, y, f, r, unq);
    return remap_finish(r, unq);
}

} // namespace fdn
