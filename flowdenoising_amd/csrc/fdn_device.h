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
// ---------------------------------------------------------------------------------
static __device__ __forceinline__ void compute_M(const float r0[5], const float* __restrict__ R1, size_t HW,
                                                 int H, int W, int x, int y, float dx, float dy, float m[5])
{
    float fx = (float)x + dx, fy = (float)y + dy;
    float flx = floorf(fx), fly = floorf(fy);
    int x1 = (int)flx, y1 = (int)fly;
    fx -= flx; fy -= fly;
    float r2, r3, r4, r5, r6;
    if ((unsigned)x1 < (unsigned)(W - 1) && (unsigned)y1 < (unsigned)(H - 1)) {
        float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
        // the two taps of a row are adjacent in a plane: one 8-byte (dword-aligned) load each
        const float* p = R1 + (size_t)y1 * W + x1;
        float2u t0, t1;
        t0 = *(const float2u*)p; t1 = *(const float2u*)(p + W); p += HW;
        r2 = a00 * t0.a + a01 * t0.b + a10 * t1.a + a11 * t1.b;
        t0 = *(const float2u*)p; t1 = *(const float2u*)(p + W); p += HW;
        r3 = a00 * t0.a + a01 * t0.b + a10 * t1.a + a11 * t1.b;
        t0 = *(const float2u*)p; t1 = *(const float2u*)(p + W); p += HW;
        r4 = a00 * t0.a + a01 * t0.b + a10 * t1.a + a11 * t1.b;
        t0 = *(const float2u*)p; t1 = *(const float2u*)(p + W); p += HW;
        r5 = a00 * t0.a + a01 * t0.b + a10 * t1.a + a11 * t1.b;
        t0 = *(const float2u*)p; t1 = *(const float2u*)(p + W);
        r6 = a00 * t0.a + a01 * t0.b + a10 * t1.a + a11 * t1.b;
        r4 = (r0[2] + r4) * 0.5f;
        r5 = (r0[3] + r5) * 0.5f;
        r6 = (r0[4] + r6) * 0.25f;
    } else {
        r2 = r3 = 0.f;
        r4 = r0[2];
        r5 = r0[3];
        r6 = r0[4] * 0.5f;
    }
    r2 = (r0[0] - r2) * 0.5f;
    r3 = (r0[1] - r3) * 0.5f;
    r2 = r2 + (r4 * dy + r6 * dx);
    r3 = r3 + (r6 * dy + r5 * dx);
    const int BORDER = 5;
    if ((unsigned)(x - BORDER) >= (unsigned)(W - BORDER * 2) || (unsigned)(y - BORDER) >= (unsigned)(H - BORDER * 2)) {
        // border[] = {0.14, 0.14, 0.4472, 0.4472, 0.4472}
        float bx0 = x < BORDER ? (x < 2 ? 0.14f : 0.4472f) : 1.f;
        float bx1 = x >= W - BORDER ? (W - x - 1 < 2 ? 0.14f : 0.4472f) : 1.f;
        float by0 = y < BORDER ? (y < 2 ? 0.14f : 0.4472f) : 1.f;
        float by1 = y >= H - BORDER ? (H - y - 1 < 2 ? 0.14f : 0.4472f) : 1.f;
        float scale = bx0 * bx1 * by0 * by1;
        r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
    }
    m[0] = r4 * r4 + r6 * r6;
    m[1] = (r4 + r5) * r6;
    m[2] = r5 * r5 + r6 * r6;
    m[3] = r4 * r2 + r6 * r3;
    m[4] = r6 * r2 + r5 * r3;
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

static __device__ __forceinline__ float remap_sample(const float* __restrict__ src, int H, int W, int x, int y, float2 f)
{
    float mx = (float)((double)f.x + (double)x);
    float my = (float)((double)f.y + (double)y);
    // cvRound(v * INTER_TAB_SIZE); bounded so the int conversion is defined for wild flows
    float qx = fminf(fmaxf(rintf(mx * 32.f), -2147483520.f), 2147483520.f);
    float qy = fminf(fmaxf(rintf(my * 32.f), -2147483520.f), 2147483520.f);
    int sx = (int)qx, sy = (int)qy;
    int ax = sx & 31, ay = sy & 31;
    int ix = clampi(sx >> 5, -32768, 32767), iy = clampi(sy >> 5, -32768, 32767);
    float tx1 = (float)ax * (1.f / 32), tx0 = 1.f - tx1;
    float ty1 = (float)ay * (1.f / 32), ty0 = 1.f - ty1;
    float w0 = ty0 * tx0, w1 = ty0 * tx1, w2 = ty1 * tx0, w3 = ty1 * tx1;
    int xa = clampi(ix, 0, W - 1), xb = clampi(ix + 1, 0, W - 1);
    int ya = clampi(iy, 0, H - 1), yb = clampi(iy + 1, 0, H - 1);
    const float* ra = src + (size_t)ya * W;
    const float* rb = src + (size_t)yb * W;
    float v0 = ra[xa], v1 = ra[xb], v2 = rb[xa], v3 = rb[xb];
    return v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3;
}


} // namespace fdn
