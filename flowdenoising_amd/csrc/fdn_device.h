// fdn_device.h -- device functions shared by the staged kernels (fdn_kernels.hip) and the fused
// chain-step kernel (fdn_fused.hip).  Compiled with -ffp-contract=off: every multiply and add
// below rounds separately, as in the CPU code the reference runs (see oracle/fdn_oracle.c).
#pragma once
#include "fdn_internal.h"

namespace fdn {

// two adjacent floats at 4-byte alignment (global_load_dwordx2 needs only dword alignment)
struct __attribute__((packed, aligned(4))) float2u { float a, b; };

static __device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static __device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// ---------------------------------------------------------------------------------
// FarnebackUpdateMatrices for one pixel.  r0[5]: R0 at (x,y); R1: planar neighbour
// expansion (gathered bilinearly at (x+dx, y+dy), exact f32 weights, no quantisation).
// Branch-free so that several instances interleave in one basic block: the gather always
// runs at a clamped position and the out-of-image case is a select; the 5-pixel border
// damping is always multiplied in (it is exactly 1.0f in the interior).  Values are
// bit-identical to the branching CPU form.  Needs H >= 2 and W >= 2.
//   bxx = border[x]-factor product for this column, ((x<5 ? b[x] : 1) * (x>=W-5 ? b[W-1-x] : 1))
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

struct GatherTaps { float2u t0[5], t1[5]; };

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
template <> __device__ __forceinline__ float2u ld_off<float2u>(const void* base, unsigned byte_off)   // 4-byte aligned pair
{
    typedef fdn_v2f __attribute__((aligned(4))) v2f_a4;
    const v2f_a4 v = *(const FDN_GLOBAL v2f_a4*)((const FDN_GLOBAL char*)base + byte_off);
    float2u r; r.a = v.x; r.b = v.y;
    return r;
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

// R1p: the five planes' base pointers (wave-uniform)
static __device__ __forceinline__ void gather_R1_planes(const float* const R1p[5], int H, int W, int x1, int y1, GatherTaps& g)
{
    // H, W < 2^24 (fused_supported): the 24-bit multiply is exact and full rate
    const unsigned off = (__umul24((unsigned)clampi(y1, 0, H - 2), (unsigned)W) + (unsigned)clampi(x1, 0, W - 2)) * 4u;
    const unsigned off1 = off + (unsigned)W * 4u;
#pragma unroll
    for (int c = 0; c < 5; c++) {
        g.t0[c] = ld_off<float2u>(R1p[c], off);
        g.t1[c] = ld_off<float2u>(R1p[c], off1);
    }
}

static __device__ __forceinline__ void gather_R1(const float* __restrict__ R1, size_t HW, int H, int W,
                                                 int x1, int y1, GatherTaps& g)
{
    // uniform plane base (SGPRs) + one 32-bit per-lane element offset: the loads take the
    // saddr + voffset form instead of ten 64-bit per-lane address computations
    const unsigned off = ((unsigned)clampi(y1, 0, H - 2) * (unsigned)W + (unsigned)clampi(x1, 0, W - 2)) * 4u;
#pragma unroll
    for (int c = 0; c < 5; c++) {
        const float* plane = R1 + (size_t)c * HW;
        g.t0[c] = ld_off<float2u>(plane, off);
        g.t1[c] = ld_off<float2u>(plane + W, off);
    }
}

static __device__ __forceinline__ void flow_target(float xf, float yf, float dx, float dy, int& x1, int& y1, float& fx, float& fy)
{
    fx = xf + dx; fy = yf + dy;
    float flx = floorf(fx), fly = floorf(fy);
    x1 = (int)flx; y1 = (int)fly;
    fx -= flx; fy -= fly;
}

static __device__ __forceinline__ void finish_M(const float r0[5], const GatherTaps& g, int H, int W, int x1, int y1,
                                                float fx, float fy, float dx, float dy, float bxx, float by0, float by1,
                                                bool damp, float m[5], bool any_damp = true)
{
    const bool inside = (unsigned)x1 < (unsigned)(W - 1) && (unsigned)y1 < (unsigned)(H - 1);
    float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
    float s[5];
#pragma unroll
    for (int c = 0; c < 5; c++) s[c] = a00 * g.t0[c].a + a01 * g.t0[c].b + a10 * g.t1[c].a + a11 * g.t1[c].b;
    float r2 = inside ? s[0] : 0.f;
    float r3 = inside ? s[1] : 0.f;
    float r4 = inside ? (r0[2] + s[2]) * 0.5f : r0[2];
    float r5 = inside ? (r0[3] + s[3]) * 0.5f : r0[3];
    float r6 = inside ? (r0[4] + s[4]) * 0.25f : r0[4] * 0.5f;
    r2 = (r0[0] - r2) * 0.5f;
    r3 = (r0[1] - r3) * 0.5f;
    r2 = r2 + (r4 * dy + r6 * dx);
    r3 = r3 + (r6 * dy + r5 * dx);
    // ((bx0*bx1)*by0)*by1 as OpenCV, == 1.0f away from the border.  `damp` is OpenCV's own region test
    // ((unsigned)(x-5) >= (unsigned)(W-10) || same for y): for images under 10 pixels it is NOT
    // "within 5 pixels of an edge" (the unsigned difference wraps), and the factors are skipped.
    // `any_damp` (wave-uniform; false only when no lane has `damp`): skips the multiplications by 1.0f
    if (any_damp) {
        float scale = damp ? bxx * by0 * by1 : 1.f;
        r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
    }
    m[0] = r4 * r4 + r6 * r6;
    m[1] = (r4 + r5) * r6;
    m[2] = r5 * r5 + r6 * r6;
    m[3] = r4 * r2 + r6 * r3;
    m[4] = r6 * r2 + r5 * r3;
}

static __device__ __forceinline__ void compute_M(const float r0[5], const float* __restrict__ R1, size_t HW,
                                                 int H, int W, int x, int y, float dx, float dy, float m[5])
{
    int x1, y1; float fx, fy;
    flow_target(x, y, dx, dy, x1, y1, fx, fy);
    GatherTaps g;
    gather_R1(R1, HW, H, W, x1, y1, g);
    float by0 = y < 5 ? (y < 2 ? 0.14f : 0.4472f) : 1.f;
    float by1 = y >= H - 5 ? (H - y - 1 < 2 ? 0.14f : 0.4472f) : 1.f;
    finish_M(r0, g, H, W, x1, y1, fx, fy, dx, dy, border_factor(x, W), by0, by1, border_test(x, W) || border_test(y, H), m);
}

static __device__ __forceinline__ float2 solve_flow(const double a[5], double scale)
{
    double g11 = a[0] * scale, g12 = a[1] * scale, g22 = a[2] * scale, h1 = a[3] * scale, h2 = a[4] * scale;
    double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
    float2 f;
    f.x = (float)((g11 * h2 - g12 * h1) * idet);
    f.y = (float)((g22 * h1 - g12 * h2) * idet);
    return f;
}

// cv::resize INTER_LINEAR of a 2-channel f32 image (HResizeLinear then VResizeLinear, f32
// coefficients) sampled at one destination pixel, times `ps` in f64: calc()'s upsampling of the flow
// between pyramid levels.  scale_x = sw / dw, scale_y = sh / dh.
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
static __device__ __forceinline__ float2 resize_linear_flow(const float* __restrict__ src, int sw, const LinearTap& tx, const LinearTap& ty, double ps)
{
    const unsigned r0 = (unsigned)ty.s0 * (unsigned)sw, r1 = (unsigned)ty.s1 * (unsigned)sw;
    const float2 p00 = ld_off<float2>(src, (r0 + tx.s0) * 8u), p01 = ld_off<float2>(src, (r0 + tx.s1) * 8u);
    const float2 p10 = ld_off<float2>(src, (r1 + tx.s0) * 8u), p11 = ld_off<float2>(src, (r1 + tx.s1) * 8u);
    float2 v;
    v.x = (p00.x * tx.a0 + p01.x * tx.a1) * ty.a0 + (p10.x * tx.a0 + p11.x * tx.a1) * ty.a1;
    v.y = (p00.y * tx.a0 + p01.y * tx.a1) * ty.a0 + (p10.y * tx.a0 + p11.y * tx.a1) * ty.a1;
    v.x = (float)((double)v.x * ps);
    v.y = (float)((double)v.y * ps);
    return v;
}

// remap in two halves so that a kernel can issue the four tap loads one pipeline step before it
// combines them: remap_issue computes the quantised position and loads, remap_finish weights.
struct RemapTaps { float v0, v1, v2, v3; int ax, ay; };

static __device__ __forceinline__ void remap_issue(const float* __restrict__ src, int H, int W, int x, int y, float2 f, RemapTaps& r)
{
    float mx = (float)((double)f.x + (double)x);
    float my = (float)((double)f.y + (double)y);
    // cvRound(v * INTER_TAB_SIZE); bounded so the int conversion is defined for wild flows
    float qx = fminf(fmaxf(rintf(mx * 32.f), -2147483520.f), 2147483520.f);
    float qy = fminf(fmaxf(rintf(my * 32.f), -2147483520.f), 2147483520.f);
    int sx = (int)qx, sy = (int)qy;
    r.ax = sx & 31; r.ay = sy & 31;
    int ix = clampi(sx >> 5, -32768, 32767), iy = clampi(sy >> 5, -32768, 32767);
    int xa = clampi(ix, 0, W - 1), xb = clampi(ix + 1, 0, W - 1);
    int ya = clampi(iy, 0, H - 1), yb = clampi(iy + 1, 0, H - 1);
    const unsigned oa = (unsigned)ya * (unsigned)W, ob = (unsigned)yb * (unsigned)W;
    r.v0 = ld_off<float>(src, (oa + xa) * 4u); r.v1 = ld_off<float>(src, (oa + xb) * 4u);
    r.v2 = ld_off<float>(src, (ob + xa) * 4u); r.v3 = ld_off<float>(src, (ob + xb) * 4u);
}

static __device__ __forceinline__ float remap_finish(const RemapTaps& r)
{
    float tx1 = (float)r.ax * (1.f / 32), tx0 = 1.f - tx1;
    float ty1 = (float)r.ay * (1.f / 32), ty0 = 1.f - ty1;
    float w0 = ty0 * tx0, w1 = ty0 * tx1, w2 = ty1 * tx0, w3 = ty1 * tx1;
    return r.v0 * w0 + r.v1 * w1 + r.v2 * w2 + r.v3 * w3;
}

static __device__ __forceinline__ float remap_sample(const float* __restrict__ src, int H, int W, int x, int y, float2 f)
{
    RemapTaps r;
    remap_issue(src, H, W, x, y, f, r);
    return remap_finish(r);
}

} // namespace fdn
