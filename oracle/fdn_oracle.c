/*
 * fdn_oracle.c -- CPU restatement of FlowDenoising's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle: it is imported/linked only by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product path
 * (flowdenoising_amd/) never calls it and fails loudly without its HIP library.
 *
 * PARITY STATUS: *unpinned* against real cv2 output.  The reference delegates all
 * arithmetic to opencv-python (unpinned, src/requirements.txt:2) through
 * cv2.calcOpticalFlowFarneback (src/flowdenoising_sequential.py:62) and cv2.remap
 * (src/flowdenoising_sequential.py:56).  Neither cv2 nor OpenCV's sources exist in
 * the build container, so the OpenCV 4.x algorithms (modules/video/src/optflowgf.cpp:
 * FarnebackPrepareGaussian, FarnebackPolyExp, FarnebackUpdateMatrices,
 * FarnebackUpdateFlow_Blur, FarnebackOpticalFlowImpl::calc; modules/imgproc:
 * GaussianBlur/getGaussianKernel, resize, remap) are restated here from their
 * published behaviour.  What IS pinned by reference-generated goldens
 * (tests/golden/): get_gaussian_kernel (seq:30-41) and the no-OF separable sweep
 * (seq:171-192, 290-311, 396-417, 426-431), and -- since round 4 -- the control flow of the
 * sweeps (tests/golden/ref_sweep_*.npz: the reference's own loops run on a cv2 stand-in that
 * forwards to this file; tests/test_ref_sweeps.py); the two OpenCV routines themselves rest on
 * analytic known-answer tests (tests/test_oracle.py): "parity unpinned" against cv2.
 *
 * Citations "seq:N" are lines of /root/reference/src/flowdenoising_sequential.py,
 * "par:N" of /root/reference/src/flowdenoising.py.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 * Float semantics: every expression is written with the operand types OpenCV uses
 * (float vs double); -ffp-contract=off keeps the compiler from fusing them.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <stddef.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define FDO_EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* a-1  get_gaussian_kernel  (seq:30-41 == par:34-45)                        */
/* ------------------------------------------------------------------------ */
/* The reference grows a delta signal until scipy.ndimage.gaussian_filter1d leaves
 * two exact zeros and returns coeffs[1:-1]: that is scipy's own kernel,
 * radius r = int(truncate*sigma + 0.5) with truncate = 4,
 * w[j] = exp(-0.5/sigma^2 * j^2) / sum.  Returns K = 2r+1 (or -needed if cap too small). */
/* numpy's float64 add.reduce (pairwise sum: 8 interleaved accumulators per block of <= 128, halving above) */
static double np_pairwise_sum_f64(const double* a, size_t n)
{
    if (n < 8) { double r = 0; for (size_t i = 0; i < n; i++) r += a[i]; return r; }
    if (n <= 128) {
        double r[8];
        size_t i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8) for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    size_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum_f64(a, n2) + np_pairwise_sum_f64(a + n2, n - n2);
}

FDO_EXPORT int fdo_gaussian_kernel(double sigma, double* out, int cap)
{
    int r = (int)(4.0 * sigma + 0.5);
    int K = 2 * r + 1;
    if (K > cap) return -K;
    double sigma2 = sigma * sigma;
    for (int j = -r; j <= r; j++) out[j + r] = exp(-0.5 / sigma2 * (double)(j * j));
    double s = np_pairwise_sum_f64(out, (size_t)K);   /* phi_x.sum(): numpy's pairwise reduction */
    for (int i = 0; i < K; i++) out[i] = out[i] / s;
    return K;
}

/* ------------------------------------------------------------------------ */
/* OpenCV helpers                                                            */
/* ------------------------------------------------------------------------ */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* cv::borderInterpolate(p, len, BORDER_REFLECT_101) */
static inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

/* cvRound: round-half-to-even (SSE cvtsd2si / lrint under the default rounding mode) */
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }
static inline int cv_floor_d(double v) { int i = (int)v; return i - (i > v); }
static inline int cv_ceil_d(double v) { int i = (int)v; return i + (i < v); }

/* cv::getGaussianKernel(n, sigma, CV_32F) (OpenCV 4.x getGaussianKernelBitExact):
 * fixed small kernels for sigma<=0 and n in {1,3,5,7}; otherwise
 * k[i] = float( exp(-0.5*x^2/sigma^2) * (1/sum) ) evaluated in double. */
static void cv_gaussian_kernel_f32(int n, double sigma, float* k)
{
    if (sigma <= 0) {
        static const float t1[] = {1.f};
        static const float t3[] = {0.25f, 0.5f, 0.25f};
        static const float t5[] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
        static const float t7[] = {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f};
        const float* t = n == 1 ? t1 : n == 3 ? t3 : n == 5 ? t5 : n == 7 ? t7 : NULL;
        if (t) { memcpy(k, t, n * sizeof(float)); return; }
    }
    double sigmaX = sigma > 0 ? sigma : n * 0.15 + 0.35;
    double scale2X = -0.125 / (sigmaX * sigmaX); /* x below is doubled */
    int n2 = (n - 1) / 2;
    double* vals = (double*)malloc((n2 + 1) * sizeof(double));
    double sum = 0;
    for (int i = 0, x = 1 - n; i < n2; i++, x += 2) {
        double t = exp((double)(x * x) * scale2X);
        vals[i] = t;
        sum += t;
    }
    sum *= 2.0;
    sum += 1.0;
    if ((n & 1) == 0) sum += 1.0;
    double mul1 = 1.0 / sum;
    for (int i = 0; i < n2; i++) {
        double t = vals[i] * mul1;
        k[i] = (float)t;
        k[n - 1 - i] = (float)t;
    }
    k[n2] = (float)(1.0 * mul1);
    if ((n & 1) == 0) k[n2 + 1] = (float)(1.0 * mul1);
    free(vals);
}

/* cv::GaussianBlur(src f32, ksize (n,n), sigma, sigma, BORDER_REFLECT_101) through
 * sepFilter2D with an f32 kernel: horizontal pass into an f32 buffer, then vertical.
 * Symmetric small-kernel forms for n==3 / n==5 (SymmRowSmallFilter), left-to-right
 * tap accumulation otherwise (RowFilter); vertical pass is SymmColumnFilter:
 * s = k[c]*S[c] ; s += k[c+j]*(S[c+j] + S[c-j]). */
/* How the multiply-adds of the blur's two passes and of the vertical resize pass round -- NOT a claim about any particular
 * cv2 build: stock x86 wheels run these filters through SIMD code whose v_muladd becomes a fused multiply-add on AVX2 / FMA3
 * machines, and plain C on the row tails.  The same three readings as the product's "opencv_fma" option
 * (flowdenoising_amd/csrc/fdn_internal.h, FmaMode), so that the day a cv2 is at hand the matching one is a switch on both sides:
 *   0  two roundings everywhere (the default: OpenCV's scalar code)
 *   1  fused everywhere
 *   2  fused on the vector body of a row -- its first (width / lanes) * lanes elements (lanes: 8 for AVX2) --, two roundings on
 *      the tail.  `i` is the element's index in its row of `width` elements (channels interleaved).
 * At levels = 0 the only blur taps are powers of two and fusing changes nothing. */
static int g_fma = 0, g_fma_lanes = 8;
FDO_EXPORT void fdo_set_fma(int mode) { g_fma = mode; }
FDO_EXPORT void fdo_set_fma_lanes(int lanes) { g_fma_lanes = lanes > 0 ? lanes : 8; }
static inline float mad_at(int i, int width, float a, float b, float c)
{
    const int fused = g_fma == 1 || (g_fma == 2 && i < width / g_fma_lanes * g_fma_lanes);
    return fused ? fmaf(a, b, c) : a * b + c;
}
/* Which cv2.remap a warp follows (the product's "remap_model" option): 0 = the classic path (coordinates rounded to 1/32 pixel,
 * weights from the 32 x 32 table), 1 = unquantised float32 bilinear interpolation at the map position -- a model of the
 * reworked float-map remap newer OpenCV releases are reported to ship (tests/test_cv2_pin.py classifies a real cv2 between
 * the two).  8-bit images keep their fixed-point table in either model. */
static int g_remap_model = 0;
FDO_EXPORT void fdo_set_remap_model(int model) { g_remap_model = model; }
static inline int floor_index(float v, float* frac)
{
    const float fl = floorf(v);
    *frac = v - fl;
    return (int)fminf(fmaxf(fl, -32768.f), 32767.f);
}

static void cv_gaussian_blur_f32(const float* src, float* dst, int H, int W, int n, double sigma)
{
    float* k = (float*)malloc(n * sizeof(float));
    cv_gaussian_kernel_f32(n, sigma, k);
    int c = n / 2;
    float* tmp = (float*)malloc((size_t)H * W * sizeof(float));
    for (int y = 0; y < H; y++) {
        const float* S = src + (size_t)y * W;
        float* T = tmp + (size_t)y * W;
        for (int x = 0; x < W; x++) {
            float s0;
            if (n == 3) {
                s0 = mad_at(x, W, S[reflect101(x - 1, W)] + S[reflect101(x + 1, W)], k[2], S[x] * k[1]);
            } else if (n == 5) {
                s0 = mad_at(x, W, S[reflect101(x - 1, W)] + S[reflect101(x + 1, W)], k[3], S[x] * k[2]);
                s0 = mad_at(x, W, S[reflect101(x - 2, W)] + S[reflect101(x + 2, W)], k[4], s0);
            } else {
                s0 = k[0] * S[reflect101(x - c, W)];
                for (int j = 1; j < n; j++) s0 = mad_at(x, W, k[j], S[reflect101(x - c + j, W)], s0);
            }
            T[x] = s0;
        }
    }
    for (int y = 0; y < H; y++) {
        float* D = dst + (size_t)y * W;
        const float* Sc = tmp + (size_t)y * W;
        for (int x = 0; x < W; x++) D[x] = k[c] * Sc[x];
        for (int j = 1; j <= c; j++) {
            const float* Sp = tmp + (size_t)reflect101(y + j, H) * W;
            const float* Sm = tmp + (size_t)reflect101(y - j, H) * W;
            float kj = k[c + j];
            for (int x = 0; x < W; x++) D[x] = mad_at(x, W, kj, Sp[x] + Sm[x], D[x]);
        }
    }
    free(tmp);
    free(k);
}

/* cv::resize INTER_LINEAR, f32, cn channels (HResizeLinear + VResizeLinear). */
static void cv_resize_linear_f32(const float* src, int sh, int sw, float* dst, int dh, int dw, int cn)
{
    double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    int* xofs = (int*)malloc(dw * sizeof(int));
    float* xa = (float*)malloc(dw * sizeof(float));
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor_f(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        xa[dx] = fx;
    }
    float* r0 = (float*)malloc((size_t)dw * cn * sizeof(float));
    float* r1 = (float*)malloc((size_t)dw * cn * sizeof(float));
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor_f(fy);
        fy -= sy;
        if (sy < 0) { fy = 0; sy = 0; }
        if (sy >= sh - 1) { fy = 0; sy = sh - 1; }
        int sy1 = sy + 1 < sh ? sy + 1 : sh - 1;
        const float* S0 = src + (size_t)sy * sw * cn;
        const float* S1 = src + (size_t)sy1 * sw * cn;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
            float a1 = xa[dx], a0 = 1.f - a1;
            for (int ch = 0; ch < cn; ch++) {
                r0[dx * cn + ch] = S0[sx * cn + ch] * a0 + S0[sx1 * cn + ch] * a1;
                r1[dx * cn + ch] = S1[sx * cn + ch] * a0 + S1[sx1 * cn + ch] * a1;
            }
        }
        float b1 = fy, b0 = 1.f - fy;
        float* D = dst + (size_t)dy * dw * cn;
        for (int i = 0; i < dw * cn; i++) D[i] = mad_at(i, dw * cn, r0[i], b0, r1[i] * b1);
    }
    free(r0); free(r1); free(xofs); free(xa);
}

typedef struct { int si, di; float alpha; } area_tab_t;

/* cv::computeResizeAreaTab */
static int area_tab(int ssize, int dsize, double scale, area_tab_t* tab)
{
    int k = 0;
    for (int dx = 0; dx < dsize; dx++) {
        double fsx1 = dx * scale;
        double fsx2 = fsx1 + scale;
        double cellWidth = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = cv_ceil_d(fsx1), sx2 = cv_floor_d(fsx2);
        if (sx2 > ssize - 1) sx2 = ssize - 1;
        if (sx1 > sx2) sx1 = sx2;
        if (sx1 - fsx1 > 1e-3) {
            tab[k].di = dx; tab[k].si = sx1 - 1;
            tab[k++].alpha = (float)((sx1 - fsx1) / cellWidth);
        }
        for (int sx = sx1; sx < sx2; sx++) {
            tab[k].di = dx; tab[k].si = sx;
            tab[k++].alpha = (float)(1.0 / cellWidth);
        }
        if (fsx2 - sx2 > 1e-3) {
            double a = fsx2 - sx2; if (a > 1.) a = 1.; if (a > cellWidth) a = cellWidth;
            tab[k].di = dx; tab[k].si = sx2;
            tab[k++].alpha = (float)(a / cellWidth);
        }
    }
    return k;
}

/* cv::resize INTER_AREA for shrinking, f32, cn channels.  Integer ratios use the
 * "fast" block mean (sum in f32 in row-major block order, times 1/area); other
 * ratios use the fractional-coverage tables (ResizeArea_Invoker). */
static void cv_resize_area_f32(const float* src, int sh, int sw, float* dst, int dh, int dw, int cn)
{
    double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    int isx = (int)scale_x, isy = (int)scale_y;
    int fast = fabs(scale_x - isx) < 2.220446049250313e-16 && fabs(scale_y - isy) < 2.220446049250313e-16;
    if (fast) {
        float scale = 1.f / (isx * isy);
        for (int dy = 0; dy < dh; dy++)
            for (int dx = 0; dx < dw; dx++)
                for (int ch = 0; ch < cn; ch++) {
                    float sum = 0;
                    for (int ky = 0; ky < isy; ky++)
                        for (int kx = 0; kx < isx; kx++)
                            sum += src[((size_t)(dy * isy + ky) * sw + dx * isx + kx) * cn + ch];
                    dst[((size_t)dy * dw + dx) * cn + ch] = sum * scale;
                }
        return;
    }
    area_tab_t* xt = (area_tab_t*)malloc(sizeof(area_tab_t) * (size_t)sw * 2);
    area_tab_t* yt = (area_tab_t*)malloc(sizeof(area_tab_t) * (size_t)sh * 2);
    int nx = area_tab(sw, dw, scale_x, xt);
    int ny = area_tab(sh, dh, scale_y, yt);
    float* buf = (float*)malloc((size_t)dw * cn * sizeof(float));
    float* sum = (float*)malloc((size_t)dw * cn * sizeof(float));
    int prev_dy = yt[0].di;
    for (int i = 0; i < dw * cn; i++) sum[i] = 0;
    for (int j = 0; j < ny; j++) {
        float beta = yt[j].alpha;
        int dy = yt[j].di, sy = yt[j].si;
        const float* S = src + (size_t)sy * sw * cn;
        for (int i = 0; i < dw * cn; i++) buf[i] = 0;
        for (int k = 0; k < nx; k++) {
            float a = xt[k].alpha;
            for (int ch = 0; ch < cn; ch++) buf[xt[k].di * cn + ch] += S[xt[k].si * cn + ch] * a;
        }
        if (dy != prev_dy) {
            float* D = dst + (size_t)prev_dy * dw * cn;
            for (int i = 0; i < dw * cn; i++) { D[i] = sum[i]; sum[i] = beta * buf[i]; }
            prev_dy = dy;
        } else {
            for (int i = 0; i < dw * cn; i++) sum[i] += beta * buf[i];
        }
    }
    {
        float* D = dst + (size_t)prev_dy * dw * cn;
        for (int i = 0; i < dw * cn; i++) D[i] = sum[i];
    }
    free(buf); free(sum); free(xt); free(yt);
}

/* cv::resize dispatcher as used by FarnebackOpticalFlowImpl::calc:
 * equal sizes -> copy; INTER_LINEAR with an exact 2x2 shrink is silently turned
 * into INTER_AREA by cv::resize. */
enum { FDO_INTER_LINEAR = 1, FDO_INTER_AREA = 3 };
static void cv_resize_f32(const float* src, int sh, int sw, float* dst, int dh, int dw, int cn, int interp)
{
    if (sh == dh && sw == dw) { memcpy(dst, src, (size_t)sh * sw * cn * sizeof(float)); return; }
    double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    if (interp == FDO_INTER_LINEAR) {
        int isx = (int)scale_x, isy = (int)scale_y;
        int fast = fabs(scale_x - isx) < 2.220446049250313e-16 && fabs(scale_y - isy) < 2.220446049250313e-16;
        if (fast && isx == 2 && isy == 2) interp = FDO_INTER_AREA;
    }
    if (interp == FDO_INTER_AREA && scale_x >= 1 && scale_y >= 1)
        cv_resize_area_f32(src, sh, sw, dst, dh, dw, cn);
    else
        cv_resize_linear_f32(src, sh, sw, dst, dh, dw, cn);
}

/* ------------------------------------------------------------------------ */
/* Farneback (OpenCV optflowgf.cpp), reached from get_flow seq:59-67         */
/* ------------------------------------------------------------------------ */

typedef struct {
    int n;
    float g[32], xg[32], xxg[32]; /* index k + n, k = -n..n */
    double ig11, ig03, ig33, ig55;
} polyexp_consts_t;

/* inverse of a 6x6 SPD matrix the way cv::invert(DECOMP_CHOLESKY) does it for n > 3 (OpenCV
 * hal CholImpl): L L^T factorisation with 1/sqrt on the diagonal, then forward and backward
 * substitution on the identity. */
static int chol_inv6(double A[6][6], double inv[6][6])
{
    const int m = 6;
    double s;
    int i, j, k;
    for (i = 0; i < m; i++)
        for (j = 0; j < m; j++) inv[i][j] = i == j ? 1.0 : 0.0;
    for (i = 0; i < m; i++) {
        for (j = 0; j < i; j++) {
            s = A[i][j];
            for (k = 0; k < j; k++) s -= A[i][k] * A[j][k];
            A[i][j] = s * A[j][j];
        }
        s = A[i][i];
        for (k = 0; k < j; k++) { double t = A[i][k]; s -= t * t; }
        if (s < 2.220446049250313e-16) return 0;
        A[i][i] = 1. / sqrt(s);
    }
    for (i = 0; i < m; i++)
        for (j = 0; j < m; j++) {
            s = inv[i][j];
            for (k = 0; k < i; k++) s -= A[i][k] * inv[k][j];
            inv[i][j] = s * A[i][i];
        }
    for (i = m - 1; i >= 0; i--)
        for (j = 0; j < m; j++) {
            s = inv[i][j];
            for (k = m - 1; k > i; k--) s -= A[k][i] * inv[k][j];
            inv[i][j] = s * A[i][i];
        }
    return 1;
}

/* FarnebackPrepareGaussian */
static void prepare_gaussian(int n, double sigma, polyexp_consts_t* pc)
{
    if (sigma < 1.1920928955078125e-07) sigma = n * 0.3;
    pc->n = n;
    float* g = pc->g + n; float* xg = pc->xg + n; float* xxg = pc->xxg + n;
    double s = 0.;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)exp(-x * x / (2 * sigma * sigma));
        s += g[x];
    }
    s = 1. / s;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)(g[x] * s);
        xg[x] = (float)(x * g[x]);
        xxg[x] = (float)(x * x * g[x]);
    }
    double G[6][6];
    memset(G, 0, sizeof(G));
    for (int y = -n; y <= n; y++)
        for (int x = -n; x <= n; x++) {
            G[0][0] += g[y] * g[x];
            G[1][1] += g[y] * g[x] * x * x;
            G[3][3] += g[y] * g[x] * x * x * x * x;
            G[5][5] += g[y] * g[x] * x * x * y * y;
        }
    G[2][2] = G[0][3] = G[0][4] = G[3][0] = G[4][0] = G[1][1];
    G[4][4] = G[3][3];
    G[3][4] = G[4][3] = G[5][5];
    double invG[6][6];
    chol_inv6(G, invG);
    pc->ig11 = invG[1][1];
    pc->ig03 = invG[0][3];
    pc->ig33 = invG[3][3];
    pc->ig55 = invG[5][5];
}

/* FarnebackPolyExp: src HxW f32 -> dst HxWx5 f32 (interleaved).
 * Channel order: [0]=d/dy, [1]=d/dx, [2]=yy, [3]=xx, [4]=xy. */
static void poly_exp(const float* src, float* dst, int H, int W, const polyexp_consts_t* pc)
{
    int n = pc->n;
    const float* g = pc->g + n; const float* xg = pc->xg + n; const float* xxg = pc->xxg + n;
    float* _row = (float*)malloc((size_t)(W + n * 2) * 3 * sizeof(float));
    float* row = _row + n * 3;
    double ig11 = pc->ig11, ig03 = pc->ig03, ig33 = pc->ig33, ig55 = pc->ig55;
    for (int y = 0; y < H; y++) {
        float g0 = g[0], g1, g2;
        const float* srow0 = src + (size_t)y * W; const float* srow1 = 0;
        float* drow = dst + (size_t)y * W * 5;
        for (int x = 0; x < W; x++) {
            row[x * 3] = srow0[x] * g0;
            row[x * 3 + 1] = row[x * 3 + 2] = 0.f;
        }
        for (int k = 1; k <= n; k++) {
            g0 = g[k]; g1 = xg[k]; g2 = xxg[k];
            srow0 = src + (size_t)(y - k > 0 ? y - k : 0) * W;
            srow1 = src + (size_t)(y + k < H - 1 ? y + k : H - 1) * W;
            for (int x = 0; x < W; x++) {
                float p = srow0[x] + srow1[x];
                float t0 = row[x * 3] + g0 * p;
                float t1 = row[x * 3 + 1] + g1 * (srow1[x] - srow0[x]);
                float t2 = row[x * 3 + 2] + g2 * p;
                row[x * 3] = t0; row[x * 3 + 1] = t1; row[x * 3 + 2] = t2;
            }
        }
        for (int x = 0; x < n * 3; x++) {
            row[-1 - x] = row[2 - x];
            row[W * 3 + x] = row[W * 3 + x - 3];
        }
        for (int x = 0; x < W; x++) {
            g0 = g[0];
            double b1 = row[x * 3] * g0, b2 = 0, b3 = row[x * 3 + 1] * g0,
                   b4 = 0, b5 = row[x * 3 + 2] * g0, b6 = 0;
            for (int k = 1; k <= n; k++) {
                double tg = row[(x + k) * 3] + row[(x - k) * 3];
                g0 = g[k];
                b1 += tg * g0;
                b4 += tg * xxg[k];
                b2 += (row[(x + k) * 3] - row[(x - k) * 3]) * xg[k];
                b3 += (row[(x + k) * 3 + 1] + row[(x - k) * 3 + 1]) * g0;
                b6 += (row[(x + k) * 3 + 1] - row[(x - k) * 3 + 1]) * xg[k];
                b5 += (row[(x + k) * 3 + 2] + row[(x - k) * 3 + 2]) * g0;
            }
            drow[x * 5 + 1] = (float)(b2 * ig11);
            drow[x * 5] = (float)(b3 * ig11);
            drow[x * 5 + 3] = (float)(b1 * ig03 + b4 * ig33);
            drow[x * 5 + 2] = (float)(b1 * ig03 + b5 * ig33);
            drow[x * 5 + 4] = (float)(b6 * ig55);
        }
    }
    free(_row);
}

/* FarnebackUpdateMatrices(R0, R1, flow, M, y0, y1) */
static void update_matrices(const float* R0, const float* R1, const float* flow, float* M,
                            int H, int W, int y0, int y1)
{
    enum { BORDER = 5 };
    static const float border[BORDER] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
    size_t step1 = (size_t)W * 5;
    for (int y = y0; y < y1; y++) {
        const float* fl = flow + (size_t)y * W * 2;
        const float* r0 = R0 + (size_t)y * W * 5;
        float* m = M + (size_t)y * W * 5;
        for (int x = 0; x < W; x++) {
            float dx = fl[x * 2], dy = fl[x * 2 + 1];
            float fx = x + dx, fy = y + dy;
            int x1 = cv_floor_f(fx), yy1 = cv_floor_f(fy);
            float r2, r3, r4, r5, r6;
            fx -= x1; fy -= yy1;
            if ((unsigned)x1 < (unsigned)(W - 1) && (unsigned)yy1 < (unsigned)(H - 1)) {
                const float* ptr = R1 + (size_t)yy1 * step1 + (size_t)x1 * 5;
                float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy),
                      a10 = (1.f - fx) * fy, a11 = fx * fy;
                r2 = a00 * ptr[0] + a01 * ptr[5] + a10 * ptr[step1] + a11 * ptr[step1 + 5];
                r3 = a00 * ptr[1] + a01 * ptr[6] + a10 * ptr[step1 + 1] + a11 * ptr[step1 + 6];
                r4 = a00 * ptr[2] + a01 * ptr[7] + a10 * ptr[step1 + 2] + a11 * ptr[step1 + 7];
                r5 = a00 * ptr[3] + a01 * ptr[8] + a10 * ptr[step1 + 3] + a11 * ptr[step1 + 8];
                r6 = a00 * ptr[4] + a01 * ptr[9] + a10 * ptr[step1 + 4] + a11 * ptr[step1 + 9];
                r4 = (r0[x * 5 + 2] + r4) * 0.5f;
                r5 = (r0[x * 5 + 3] + r5) * 0.5f;
                r6 = (r0[x * 5 + 4] + r6) * 0.25f;
            } else {
                r2 = r3 = 0.f;
                r4 = r0[x * 5 + 2];
                r5 = r0[x * 5 + 3];
                r6 = r0[x * 5 + 4] * 0.5f;
            }
            r2 = (r0[x * 5] - r2) * 0.5f;
            r3 = (r0[x * 5 + 1] - r3) * 0.5f;
            r2 += r4 * dy + r6 * dx;
            r3 += r6 * dy + r5 * dx;
            if ((unsigned)(x - BORDER) >= (unsigned)(W - BORDER * 2) ||
                (unsigned)(y - BORDER) >= (unsigned)(H - BORDER * 2)) {
                float scale = (x < BORDER ? border[x] : 1.f) *
                              (x >= W - BORDER ? border[W - x - 1] : 1.f) *
                              (y < BORDER ? border[y] : 1.f) *
                              (y >= H - BORDER ? border[H - y - 1] : 1.f);
                r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
            }
            m[x * 5] = r4 * r4 + r6 * r6;
            m[x * 5 + 1] = (r4 + r5) * r6;
            m[x * 5 + 2] = r5 * r5 + r6 * r6;
            m[x * 5 + 3] = r4 * r2 + r6 * r3;
            m[x * 5 + 4] = r6 * r2 + r5 * r3;
        }
    }
}

static inline void solve_flow(double g11, double g12, double g22, double h1, double h2,
                              double scale, float* fl)
{
    double g11_ = g11 * scale, g12_ = g12 * scale, g22_ = g22 * scale;
    double h1_ = h1 * scale, h2_ = h2 * scale;
    double idet = 1. / (g11_ * g22_ - g12_ * g12_ + 1e-3);
    fl[0] = (float)((g11_ * h2_ - g12_ * h1_) * idet);
    fl[1] = (float)((g22_ * h1_ - g12_ * h2_) * idet);
}

/* FarnebackUpdateFlow_Blur.  box_mode 0 = OpenCV's running sums (vertical running
 * sum fed by f32 row differences, horizontal running sum in f64); box_mode 1 =
 * direct window sums in f64 (same window, same replicate borders, no running
 * error) -- the form the HIP kernels use; kept here so tests can separate
 * "restatement vs running-sum rounding" from real bugs. */
static void update_flow_blur(const float* R0, const float* R1, float* flow, float* M,
                             int H, int W, int block_size, int update, int box_mode)
{
    int m = block_size / 2;
    double scale = 1. / (block_size * block_size);
    if (box_mode == 1) {
        double* V = (double*)malloc((size_t)W * 5 * sizeof(double));
        for (int y = 0; y < H; y++) {
            for (int i = 0; i < W * 5; i++) {
                double s = 0;
                for (int j = -m; j <= m; j++) s += (double)M[(size_t)clampi(y + j, 0, H - 1) * W * 5 + i];
                V[i] = s;
            }
            float* fl = flow + (size_t)y * W * 2;
            for (int x = 0; x < W; x++) {
                double a[5];
                for (int c = 0; c < 5; c++) {
                    double s = 0;
                    for (int j = -m; j <= m; j++) s += V[clampi(x + j, 0, W - 1) * 5 + c];
                    a[c] = s;
                }
                solve_flow(a[0], a[1], a[2], a[3], a[4], scale, fl + x * 2);
            }
        }
        free(V);
        if (update) update_matrices(R0, R1, flow, M, H, W, 0, H);
        return;
    }
    if (box_mode == 2) {
        /* OpenCV's vertical running sum (the f32-fed recurrence, exactly as below) but the
         * horizontal window summed directly in f64 instead of OpenCV's f64 running sum.  This is
         * the order the HIP kernels use: a running sum along x is a serial chain across the whole
         * row, which a band-parallel kernel cannot reproduce.  The two differ only by f64
         * rounding (~1e-16 relative), which the near-singular 2x2 solve can still amplify. */
        double* vs = (double*)malloc((size_t)W * 5 * sizeof(double));
        const float* r0 = M;
        for (int x = 0; x < W * 5; x++) vs[x] = r0[x] * (m + 2);
        for (int y = 1; y < m; y++) {
            const float* srow = M + (size_t)(y < H - 1 ? y : H - 1) * W * 5;
            for (int x = 0; x < W * 5; x++) vs[x] += srow[x];
        }
        for (int y = 0; y < H; y++) {
            float* fl = flow + (size_t)y * W * 2;
            const float* s0 = M + (size_t)(y - m - 1 > 0 ? y - m - 1 : 0) * W * 5;
            const float* s1 = M + (size_t)(y + m < H - 1 ? y + m : H - 1) * W * 5;
            for (int x = 0; x < W * 5; x++) vs[x] += s1[x] - s0[x];
            for (int x = 0; x < W; x++) {
                double a[5];
                for (int c = 0; c < 5; c++) {   /* the 2m+1 terms left to right, starting FROM the first */
                    double s = vs[clampi(x - m, 0, W - 1) * 5 + c];
                    for (int j = -m + 1; j <= m; j++) s += vs[clampi(x + j, 0, W - 1) * 5 + c];
                    a[c] = s;
                }
                solve_flow(a[0], a[1], a[2], a[3], a[4], scale, fl + x * 2);
            }
        }
        free(vs);
        if (update) update_matrices(R0, R1, flow, M, H, W, 0, H);
        return;
    }
    if (box_mode == 3) {
        /* OpenCV's vertical running sum, the horizontal window summed BY DOUBLING -- the order of the HIP path's
         * one-iteration kernel (fdn_iter.hip, winsize >= 10): T1 = vsum (columns clamped to the image), T2k[x] = Tk[x] +
         * Tk[x + k]; the window [x - m, x + m] is the sum of the Tk of the binary digits of 2m + 1, lowest digit
         * first and rightmost block first: w = 15: T1[x+7] + T2[x+5] + T4[x+1] + T8[x-7].  Exact -- hence equal to
         * every other order -- whenever the f64 sums do not round. */
        int wn = 2 * m + 1, nlev = 0;
        while ((1 << nlev) <= wn) nlev++;
        int ext = W + 4 * m + 4;                        /* columns -m .. W-1+m plus room for the T builds */
        double* T = (double*)malloc((size_t)nlev * ext * 5 * sizeof(double));
        double* vs = (double*)malloc((size_t)W * 5 * sizeof(double));
        for (int x = 0; x < W * 5; x++) vs[x] = M[x] * (m + 2);
        for (int y = 1; y < m; y++) {
            const float* srow = M + (size_t)(y < H - 1 ? y : H - 1) * W * 5;
            for (int x = 0; x < W * 5; x++) vs[x] += srow[x];
        }
        for (int y = 0; y < H; y++) {
            float* fl = flow + (size_t)y * W * 2;
            const float* s0 = M + (size_t)(y - m - 1 > 0 ? y - m - 1 : 0) * W * 5;
            const float* s1 = M + (size_t)(y + m < H - 1 ? y + m : H - 1) * W * 5;
            for (int x = 0; x < W * 5; x++) vs[x] += s1[x] - s0[x];
            /* level arrays over extended columns e = x + m, x = -m .. : T[lev][e] */
            for (int e = 0; e < ext; e++)
                for (int c = 0; c < 5; c++) T[((size_t)0 * ext + e) * 5 + c] = vs[clampi(e - m, 0, W - 1) * 5 + c];
            for (int lev = 1; lev < nlev; lev++) {
                int k = 1 << (lev - 1);
                for (int e = 0; e < ext; e++)
                    for (int c = 0; c < 5; c++) {
                        int e2 = e + k < ext ? e + k : ext - 1;
                        T[((size_t)lev * ext + e) * 5 + c] = T[((size_t)(lev - 1) * ext + e) * 5 + c] + T[((size_t)(lev - 1) * ext + e2) * 5 + c];
                    }
            }
            for (int x = 0; x < W; x++) {
                double a[5];
                int pos = m, first = 1;
                for (int lev = 0; lev < nlev; lev++) {
                    int k = 1 << lev;
                    if (!(wn & k)) continue;
                    pos -= k;
                    for (int c = 0; c < 5; c++) {
                        double term = T[((size_t)lev * ext + (x + pos + 1 + m)) * 5 + c];
                        a[c] = first ? term : a[c] + term;
                    }
                    first = 0;
                }
                solve_flow(a[0], a[1], a[2], a[3], a[4], scale, fl + x * 2);
            }
        }
        free(T); free(vs);
        if (update) update_matrices(R0, R1, flow, M, H, W, 0, H);
        return;
    }
    if (box_mode == 4) {
        /* OpenCV's vertical running sum, the horizontal window summed IN TWO LEVELS OF BLOCKS -- the order of the HIP
         * path's one-iteration kernel (fdn_iter.hip, winsize >= 10): n = 2m + 1, p = max(2, floor(sqrt(n))), q = n / p,
         * rem = n - p q; the window's first p q columns as q blocks of p consecutive columns (columns clamped to the
         * image), each block summed left to right, the blocks added left to right, then the last rem columns one
         * by one.  Exact -- hence equal to every other order -- whenever the f64 sums do not round. */
        int wn = 2 * m + 1, p = 1;
        while ((p + 1) * (p + 1) <= wn) p++;
        if (p < 2) p = 2;
        int q = wn / p, rem = wn - p * q;
        double* vs = (double*)malloc((size_t)W * 5 * sizeof(double));
        for (int x = 0; x < W * 5; x++) vs[x] = M[x] * (m + 2);
        for (int y = 1; y < m; y++) {
            const float* srow = M + (size_t)(y < H - 1 ? y : H - 1) * W * 5;
            for (int x = 0; x < W * 5; x++) vs[x] += srow[x];
        }
        for (int y = 0; y < H; y++) {
            float* fl = flow + (size_t)y * W * 2;
            const float* s0 = M + (size_t)(y - m - 1 > 0 ? y - m - 1 : 0) * W * 5;
            const float* s1 = M + (size_t)(y + m < H - 1 ? y + m : H - 1) * W * 5;
            for (int x = 0; x < W * 5; x++) vs[x] += s1[x] - s0[x];
            for (int x = 0; x < W; x++) {
                double a[5];
                for (int c = 0; c < 5; c++) {
                    double s = 0;
                    for (int b = 0; b < q; b++) {
                        int c0 = x - m + b * p;
                        double blk = vs[clampi(c0, 0, W - 1) * 5 + c];
                        for (int j = 1; j < p; j++) blk += vs[clampi(c0 + j, 0, W - 1) * 5 + c];
                        s = b == 0 ? blk : s + blk;
                    }
                    for (int j = 0; j < rem; j++) s += vs[clampi(x + m - rem + 1 + j, 0, W - 1) * 5 + c];
                    a[c] = s;
                }
                solve_flow(a[0], a[1], a[2], a[3], a[4], scale, fl + x * 2);
            }
        }
        free(vs);
        if (update) update_matrices(R0, R1, flow, M, H, W, 0, H);
        return;
    }
    int y0 = 0, y1;
    int min_update_stripe = (1 << 10) / W > block_size ? (1 << 10) / W : block_size;
    double* _vsum = (double*)malloc((size_t)(W + m * 2 + 2) * 5 * sizeof(double));
    double* vsum = _vsum + (m + 1) * 5;
    const float* srow0 = M;
    for (int x = 0; x < W * 5; x++) vsum[x] = srow0[x] * (m + 2);
    for (int y = 1; y < m; y++) {
        const float* srow = M + (size_t)(y < H - 1 ? y : H - 1) * W * 5;
        for (int x = 0; x < W * 5; x++) vsum[x] += srow[x];
    }
    for (int y = 0; y < H; y++) {
        double g11, g12, g22, h1, h2;
        float* fl = flow + (size_t)y * W * 2;
        const float* s0 = M + (size_t)(y - m - 1 > 0 ? y - m - 1 : 0) * W * 5;
        const float* s1 = M + (size_t)(y + m < H - 1 ? y + m : H - 1) * W * 5;
        for (int x = 0; x < W * 5; x++) vsum[x] += s1[x] - s0[x];
        for (int x = 0; x < (m + 1) * 5; x++) {
            vsum[-1 - x] = vsum[4 - x];
            vsum[W * 5 + x] = vsum[W * 5 + x - 5];
        }
        g11 = vsum[0] * (m + 2); g12 = vsum[1] * (m + 2); g22 = vsum[2] * (m + 2);
        h1 = vsum[3] * (m + 2); h2 = vsum[4] * (m + 2);
        for (int x = 1; x < m; x++) {
            g11 += vsum[x * 5]; g12 += vsum[x * 5 + 1]; g22 += vsum[x * 5 + 2];
            h1 += vsum[x * 5 + 3]; h2 += vsum[x * 5 + 4];
        }
        for (int x = 0; x < W; x++) {
            g11 += vsum[(x + m) * 5] - vsum[(x - m) * 5 - 5];
            g12 += vsum[(x + m) * 5 + 1] - vsum[(x - m) * 5 - 4];
            g22 += vsum[(x + m) * 5 + 2] - vsum[(x - m) * 5 - 3];
            h1 += vsum[(x + m) * 5 + 3] - vsum[(x - m) * 5 - 2];
            h2 += vsum[(x + m) * 5 + 4] - vsum[(x - m) * 5 - 1];
            solve_flow(g11, g12, g22, h1, h2, scale, fl + x * 2);
        }
        y1 = y == H - 1 ? H : y - block_size;
        if (update && (y1 == H || y1 >= y0 + min_update_stripe)) {
            update_matrices(R0, R1, flow, M, H, W, y0, y1);
            y0 = y1;
        }
    }
    free(_vsum);
}

typedef struct {
    int levels, winsize, iters, poly_n;
    double poly_sigma;
    int flags;     /* 4 = OPTFLOW_USE_INITIAL_FLOW; 256 (gaussian window) unsupported */
    int box_mode;  /* see update_flow_blur */
} fdo_fb_params;

/* image (already f32) -> blurred, resized, polynomial expansion at one level */
static void level_poly(const float* img, int H, int W, int h, int w, int smooth_sz, double sigma,
                       const polyexp_consts_t* pc, float* R)
{
    float* f = (float*)malloc((size_t)H * W * sizeof(float));
    cv_gaussian_blur_f32(img, f, H, W, smooth_sz, sigma);
    float* I = (float*)malloc((size_t)h * w * sizeof(float));
    cv_resize_f32(f, H, W, I, h, w, 1, FDO_INTER_LINEAR);
    poly_exp(I, R, h, w, pc);
    free(I); free(f);
}

/* FarnebackOpticalFlowImpl::calc(prev0, next0, flow0); pyr_scale fixed at 0.5 (seq:62). */
FDO_EXPORT void fdo_farneback(const float* prev0, const float* next0, float* flow0, int H, int W,
                              const fdo_fb_params* P)
{
    const int min_size = 32;
    const double pyr_scale = 0.5;
    const float* img[2] = {prev0, next0};
    int k, i, levels = P->levels;
    double scale;
    polyexp_consts_t pc;
    prepare_gaussian(P->poly_n, P->poly_sigma, &pc);

    for (k = 0, scale = 1; k < levels; k++) {
        scale *= pyr_scale;
        if (W * scale < min_size || H * scale < min_size) break;
    }
    levels = k;

    float* prev_flow = NULL; int ph = 0, pw = 0;
    for (k = levels; k >= 0; k--) {
        for (i = 0, scale = 1; i < k; i++) scale *= pyr_scale;
        double sigma = (1. / scale - 1) * 0.5;
        int smooth_sz = cv_round_d(sigma * 5) | 1;
        if (smooth_sz < 3) smooth_sz = 3;
        int width = cv_round_d(W * scale), height = cv_round_d(H * scale);
        float* flow = k > 0 ? (float*)malloc((size_t)height * width * 2 * sizeof(float)) : flow0;
        if (!prev_flow) {
            if (P->flags & 4) {
                if (k > 0) {
                    cv_resize_f32(flow0, H, W, flow, height, width, 2, FDO_INTER_AREA);
                    for (size_t j = 0; j < (size_t)height * width * 2; j++) flow[j] = (float)(flow[j] * scale);
                } /* k == 0: resize onto itself is a copy and scale == 1 */
            } else {
                memset(flow, 0, (size_t)height * width * 2 * sizeof(float));
            }
        } else {
            cv_resize_f32(prev_flow, ph, pw, flow, height, width, 2, FDO_INTER_LINEAR);
            for (size_t j = 0; j < (size_t)height * width * 2; j++) flow[j] = (float)(flow[j] * (1. / pyr_scale));
        }
        float* R[2];
        for (i = 0; i < 2; i++) {
            R[i] = (float*)malloc((size_t)height * width * 5 * sizeof(float));
            level_poly(img[i], H, W, height, width, smooth_sz, sigma, &pc, R[i]);
        }
        float* M = (float*)malloc((size_t)height * width * 5 * sizeof(float));
        update_matrices(R[0], R[1], flow, M, height, width, 0, height);
        for (i = 0; i < P->iters; i++)
            update_flow_blur(R[0], R[1], flow, M, height, width, P->winsize, i < P->iters - 1, P->box_mode);
        free(M); free(R[0]); free(R[1]);
        if (prev_flow) free(prev_flow);
        prev_flow = flow; ph = height; pw = width;
    }
    /* prev_flow == flow0 here */
}

/* Exposed pieces for known-answer tests */
FDO_EXPORT void fdo_poly_exp(const float* src, float* dst, int H, int W, int n, double sigma)
{
    polyexp_consts_t pc;
    prepare_gaussian(n, sigma, &pc);
    poly_exp(src, dst, H, W, &pc);
}
FDO_EXPORT void fdo_polyexp_consts(int n, double sigma, float* g, float* xg, float* xxg, double* ig)
{
    polyexp_consts_t pc;
    prepare_gaussian(n, sigma, &pc);
    memcpy(g, pc.g, (2 * n + 1) * sizeof(float));
    memcpy(xg, pc.xg, (2 * n + 1) * sizeof(float));
    memcpy(xxg, pc.xxg, (2 * n + 1) * sizeof(float));
    ig[0] = pc.ig11; ig[1] = pc.ig03; ig[2] = pc.ig33; ig[3] = pc.ig55;
}
FDO_EXPORT void fdo_gaussian_blur(const float* src, float* dst, int H, int W, int n, double sigma)
{
    cv_gaussian_blur_f32(src, dst, H, W, n, sigma);
}
FDO_EXPORT void fdo_resize(const float* src, int sh, int sw, float* dst, int dh, int dw, int cn, int interp)
{
    cv_resize_f32(src, sh, sw, dst, dh, dw, cn, interp);
}
FDO_EXPORT void fdo_update_matrices(const float* R0, const float* R1, const float* flow, float* M, int H, int W)
{
    update_matrices(R0, R1, flow, M, H, W, 0, H);
}
FDO_EXPORT void fdo_update_flow_blur(const float* R0, const float* R1, float* flow, float* M,
                                     int H, int W, int winsize, int update, int box_mode)
{
    update_flow_blur(R0, R1, flow, M, H, W, winsize, update, box_mode);
}

/* ------------------------------------------------------------------------ */
/* a-4  warp_slice (seq:51-57): map = float32(flow + grid); cv2.remap(INTER_LINEAR,   */
/*      BORDER_REPLICATE) with its 1/32-pixel coordinate quantisation.               */
/* ------------------------------------------------------------------------ */
FDO_EXPORT void fdo_remap_linear_replicate(const float* src, int H, int W, const float* mapxy, float* dst)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const float* mp = mapxy + ((size_t)y * W + x) * 2;
            if (g_remap_model == 1) {       /* unquantised: two float32 lerps, along x in both rows, then along y */
                float fx, fy;
                const int ix = floor_index(mp[0], &fx), iy = floor_index(mp[1], &fy);
                const int x0 = clampi(ix, 0, W - 1), x1 = clampi(ix + 1, 0, W - 1), y0 = clampi(iy, 0, H - 1), y1 = clampi(iy + 1, 0, H - 1);
                const float top = src[(size_t)y0 * W + x0] * (1.f - fx) + src[(size_t)y0 * W + x1] * fx;
                const float bot = src[(size_t)y1 * W + x0] * (1.f - fx) + src[(size_t)y1 * W + x1] * fx;
                dst[(size_t)y * W + x] = top * (1.f - fy) + bot * fy;
                continue;
            }
            int sx = cv_round_f(mp[0] * 32), sy = cv_round_f(mp[1] * 32);
            int ax = sx & 31, ay = sy & 31;
            int ix = clampi(sx >> 5, -32768, 32767), iy = clampi(sy >> 5, -32768, 32767);
            float tx1 = ax * (1.f / 32), tx0 = 1.f - tx1;
            float ty1 = ay * (1.f / 32), ty0 = 1.f - ty1;
            float w0 = ty0 * tx0, w1 = ty0 * tx1, w2 = ty1 * tx0, w3 = ty1 * tx1;
            int x0 = clampi(ix, 0, W - 1), x1 = clampi(ix + 1, 0, W - 1);
            int y0 = clampi(iy, 0, H - 1), y1 = clampi(iy + 1, 0, H - 1);
            float v0 = src[(size_t)y0 * W + x0], v1 = src[(size_t)y0 * W + x1];
            float v2 = src[(size_t)y1 * W + x0], v3 = src[(size_t)y1 * W + x1];
            dst[(size_t)y * W + x] = v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3;
        }
}

/* cv2.remap of a CV_8U image (a slice of a uint8 volume in par, src/flowdenoising.py:312): OpenCV interpolates 8-bit
 * images in FIXED POINT -- remapBilinear<FixedPtCast<int, uchar, INTER_REMAP_COEF_BITS = 15>, RemapVec_8u, short>: the four
 * weights are the float table entries times 32768 as 16-bit integers (for INTER_LINEAR the products (32 - ax)(32 - ay) 32,
 * ... are exact integers that sum to 32768; the one entry that does not fit a short, 32768 at ax = ay = 0, is stored as
 * 32767 with the missing 1 given to another tap by initInterTab2D's fix-up -- which cannot change an 8-bit result: the
 * other tap differs from this one by at most 255 << 16384), the sum is an int and the result (sum + 2^14) >> 15.
 * Same 1/32-pixel coordinates and BORDER_REPLICATE clamping as the float path.  Values 0..255 held in floats. */
static float remap_u8_fixed_point(float v0, float v1, float v2, float v3, int ax, int ay)
{
    const int w0 = (32 - ax) * (32 - ay) * 32, w1 = ax * (32 - ay) * 32, w2 = (32 - ax) * ay * 32, w3 = ax * ay * 32;
    const int sum = (int)v0 * w0 + (int)v1 * w1 + (int)v2 * w2 + (int)v3 * w3;
    int r = (sum + (1 << 14)) >> 15;
    r = r < 0 ? 0 : r > 255 ? 255 : r;      /* saturate_cast<uchar> (a convex combination: never taken) */
    return (float)r;
}
static void remap_u8_image(const float* src, int H, int W, const float* mapxy, float* dst)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const float* mp = mapxy + ((size_t)y * W + x) * 2;
            int sx = cv_round_f(mp[0] * 32), sy = cv_round_f(mp[1] * 32);
            int ax = sx & 31, ay = sy & 31;
            int ix = clampi(sx >> 5, -32768, 32767), iy = clampi(sy >> 5, -32768, 32767);
            int x0 = clampi(ix, 0, W - 1), x1 = clampi(ix + 1, 0, W - 1);
            int y0 = clampi(iy, 0, H - 1), y1 = clampi(iy + 1, 0, H - 1);
            dst[(size_t)y * W + x] = remap_u8_fixed_point(src[(size_t)y0 * W + x0], src[(size_t)y0 * W + x1],
                                                          src[(size_t)y1 * W + x0], src[(size_t)y1 * W + x1], ax, ay);
        }
}
FDO_EXPORT void fdo_remap_linear_replicate_u8(const unsigned char* src, int H, int W, const float* mapxy, unsigned char* dst)
{
    const size_t n = (size_t)H * W;
    float* f = (float*)malloc(n * sizeof(float));
    float* g = (float*)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; i++) f[i] = (float)src[i];
    remap_u8_image(f, H, W, mapxy, g);
    for (size_t i = 0; i < n; i++) dst[i] = (unsigned char)g[i];
    free(f); free(g);
}
static void warp_slice_u8(const float* reference, const float* flow, float* dst, int H, int W)
{
    float* map = (float*)malloc((size_t)H * W * 2 * sizeof(float));
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            size_t i = ((size_t)y * W + x) * 2;
            map[i] = (float)((double)flow[i] + (double)x);
            map[i + 1] = (float)((double)flow[i + 1] + (double)y);
        }
    remap_u8_image(reference, H, W, map, dst);
    free(map);
}

FDO_EXPORT void fdo_warp_slice(const float* reference, const float* flow, float* dst, int H, int W)
{
    float* map = (float*)malloc((size_t)H * W * 2 * sizeof(float));
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            size_t i = ((size_t)y * W + x) * 2;
            /* seq:55: f32 flow + int64 grid -> f64 -> astype(f32) */
            map[i] = (float)((double)flow[i] + (double)x);
            map[i + 1] = (float)((double)flow[i + 1] + (double)y);
        }
    fdo_remap_linear_replicate(reference, H, W, map, dst);
    free(map);
}

/* ------------------------------------------------------------------------ */
/* a-5/6/7/10  axis sweeps (seq:78-130, 235-288, 313-364; par:306-373)       */
/* ------------------------------------------------------------------------ */
static void slice_dims(int Z, int Y, int X, int axis, int* n, int* H, int* W)
{
    if (axis == 0) { *n = Z; *H = Y; *W = X; }
    else if (axis == 1) { *n = Y; *H = Z; *W = X; }
    else { *n = X; *H = Z; *W = Y; }
}
static void get_slice(const float* vol, int Z, int Y, int X, int axis, int s, float* out)
{
    if (axis == 0) memcpy(out, vol + (size_t)s * Y * X, (size_t)Y * X * sizeof(float));
    else if (axis == 1) for (int z = 0; z < Z; z++) memcpy(out + (size_t)z * X, vol + ((size_t)z * Y + s) * X, X * sizeof(float));
    else for (int z = 0; z < Z; z++) for (int y = 0; y < Y; y++) out[(size_t)z * Y + y] = vol[((size_t)z * Y + y) * X + s];
}
static void put_slice(float* vol, int Z, int Y, int X, int axis, int s, const float* in)
{
    if (axis == 0) memcpy(vol + (size_t)s * Y * X, in, (size_t)Y * X * sizeof(float));
    else if (axis == 1) for (int z = 0; z < Z; z++) memcpy(vol + ((size_t)z * Y + s) * X, in + (size_t)z * X, X * sizeof(float));
    else for (int z = 0; z < Z; z++) for (int y = 0; y < Y; y++) vol[((size_t)z * Y + y) * X + s] = in[(size_t)z * Y + y];
}

typedef struct {
    int levels, winsize;
    int border_mode;   /* 0 = mean-pad (seq:88-89), 1 = wrap-around (par:312) */
    int chained;       /* 1 = previous flow seeds the next (seq:97-98), 0 = --recompute_flow (par:89-114) */
    int use_of;        /* 0 = no_OF_filter (seq:171-192) */
    int box_mode;
    int nthreads;      /* slices are split in contiguous chunks like par:181-206 */
} fdo_sweep_params;

/* neighbour slice for tap i of target s: seq pads with `mean` (padded index s+i,
 * data at offset K/2); par wraps (s + i - K/2) mod n. */
static void neighbour(const float* vol, int Z, int Y, int X, int axis, int n, int s, int i, int K,
                      float mean, int border_mode, size_t npx, float* out)
{
    int q = s + i - K / 2;
    if (border_mode == 1) { q %= n; if (q < 0) q += n; get_slice(vol, Z, Y, X, axis, q, out); }
    else if (q < 0 || q >= n) { for (size_t j = 0; j < npx; j++) out[j] = mean; }
    else get_slice(vol, Z, Y, X, axis, q, out);
}

/* Integer-input semantics (seq:513 keeps an integer MRC's dtype, so seq:420's vol.mean() is a float64 and seq:88's
 * np.full makes the PADDED volume float64 -- in all three passes, the mean being computed once): Farneback still
 * converts its images to f32 (pad slices become f32(mean)), but cv2.remap of a CV_64F image weights the four taps in
 * double (f32 table weights widened, products and sums in f64, no rounding to f32) and the pad slices hold the f64
 * mean.  The HIP path implements the same semantics (fdn_sweep_params.warp_mode = FDN_WARP_F64_PADDED, fold_warped<1> in
 * fdn_device.h); tests/test_gpu_integer.py demands bit equality with this restatement. */
static int g_f64_padded = 0;
static double g_mean64 = 0.;
FDO_EXPORT void fdo_set_f64_padded(int on, double mean64) { g_f64_padded = on; g_mean64 = mean64; }

/* par on an integer MRC (par:472 keeps the dtype): the neighbour slices are integer images, so cv2.remap returns that type:
 * remapBilinear<Cast<float, short>> = saturate_cast<short>(float) = cvRound (half to even), clamped to the type's range;
 * and self.filtered_vol = np.zeros_like(vol) truncates every pass's float32 slices toward zero (par:131, par:287-289). */
static int g_int_round = 0;       /* 1: 16-bit images (float interpolation, rounded and saturated); 2: uint8 images (fixed point) */
static float g_int_lo = 0.f, g_int_hi = 0.f;
FDO_EXPORT void fdo_set_int_round(int on, double lo, double hi) { g_int_round = on; g_int_lo = (float)lo; g_int_hi = (float)hi; }

static void warp_slice_f64(const float* reference, int is_pad, const float* flow, double* dst, int H, int W)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            size_t i = ((size_t)y * W + x) * 2;
            float mx = (float)((double)flow[i] + (double)x), my = (float)((double)flow[i + 1] + (double)y);
            if (g_remap_model == 1) {       /* unquantised, on a CV_64F image: values and sums in double, the float32 fractions widened */
                float fx, fy;
                const int ix = floor_index(mx, &fx), iy = floor_index(my, &fy);
                const int x0 = clampi(ix, 0, W - 1), x1 = clampi(ix + 1, 0, W - 1), y0 = clampi(iy, 0, H - 1), y1 = clampi(iy + 1, 0, H - 1);
                double v0, v1, v2, v3;
                if (is_pad) v0 = v1 = v2 = v3 = g_mean64;
                else {
                    v0 = reference[(size_t)y0 * W + x0]; v1 = reference[(size_t)y0 * W + x1];
                    v2 = reference[(size_t)y1 * W + x0]; v3 = reference[(size_t)y1 * W + x1];
                }
                const double top = v0 * (double)(1.f - fx) + v1 * (double)fx, bot = v2 * (double)(1.f - fx) + v3 * (double)fx;
                dst[(size_t)y * W + x] = top * (double)(1.f - fy) + bot * (double)fy;
                continue;
            }
            int sx = cv_round_f(mx * 32), sy = cv_round_f(my * 32);
            int ax = sx & 31, ay = sy & 31;
            int ix = clampi(sx >> 5, -32768, 32767), iy = clampi(sy >> 5, -32768, 32767);
            float tx1 = ax * (1.f / 32), tx0 = 1.f - tx1;
            float ty1 = ay * (1.f / 32), ty0 = 1.f - ty1;
            float w0 = ty0 * tx0, w1 = ty0 * tx1, w2 = ty1 * tx0, w3 = ty1 * tx1;
            int x0 = clampi(ix, 0, W - 1), x1 = clampi(ix + 1, 0, W - 1);
            int y0 = clampi(iy, 0, H - 1), y1 = clampi(iy + 1, 0, H - 1);
            double v0, v1, v2, v3;
            if (is_pad) v0 = v1 = v2 = v3 = g_mean64;
            else {
                v0 = reference[(size_t)y0 * W + x0]; v1 = reference[(size_t)y0 * W + x1];
                v2 = reference[(size_t)y1 * W + x0]; v3 = reference[(size_t)y1 * W + x1];
            }
            dst[(size_t)y * W + x] = v0 * (double)w0 + v1 * (double)w1 + v2 * (double)w2 + v3 * (double)w3;
        }
}

/* cv2.remap(INTER_LINEAR, BORDER_REPLICATE) of a CV_64F image with an explicit map (what the tests' cv2 stand-in needs) */
FDO_EXPORT void fdo_remap_linear_replicate_f64(const double* src, int H, int W, const float* mapxy, double* dst)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const float* mp = mapxy + ((size_t)y * W + x) * 2;
            if (g_remap_model == 1) {
                float fx, fy;
                const int ix = floor_index(mp[0], &fx), iy = floor_index(mp[1], &fy);
                const int x0 = clampi(ix, 0, W - 1), x1 = clampi(ix + 1, 0, W - 1), y0 = clampi(iy, 0, H - 1), y1 = clampi(iy + 1, 0, H - 1);
                const double top = src[(size_t)y0 * W + x0] * (double)(1.f - fx) + src[(size_t)y0 * W + x1] * (double)fx;
                const double bot = src[(size_t)y1 * W + x0] * (double)(1.f - fx) + src[(size_t)y1 * W + x1] * (double)fx;
                dst[(size_t)y * W + x] = top * (double)(1.f - fy) + bot * (double)fy;
                continue;
            }
            int sx = cv_round_f(mp[0] * 32), sy = cv_round_f(mp[1] * 32);
            int ax = sx & 31, ay = sy & 31;
            int ix = clampi(sx >> 5, -32768, 32767), iy = clampi(sy >> 5, -32768, 32767);
            float tx1 = ax * (1.f / 32), tx0 = 1.f - tx1;
            float ty1 = ay * (1.f / 32), ty0 = 1.f - ty1;
            float w0 = ty0 * tx0, w1 = ty0 * tx1, w2 = ty1 * tx0, w3 = ty1 * tx1;
            int x0 = clampi(ix, 0, W - 1), x1 = clampi(ix + 1, 0, W - 1);
            int y0 = clampi(iy, 0, H - 1), y1 = clampi(iy + 1, 0, H - 1);
            dst[(size_t)y * W + x] = src[(size_t)y0 * W + x0] * (double)w0 + src[(size_t)y0 * W + x1] * (double)w1
                                   + src[(size_t)y1 * W + x0] * (double)w2 + src[(size_t)y1 * W + x1] * (double)w3;
        }
}

/* Targets s0 <= s < s1 only (the other output slices are left untouched): used to time a
 * bounded sample of a large volume (bench.py cpu_baseline). */
FDO_EXPORT void fdo_filter_axis_range(const float* vol, float* out, int Z, int Y, int X, int axis,
                                      const double* kernel, int K, float mean, const fdo_sweep_params* sp,
                                      int s0, int s1)
{
    int n, H, W;
    slice_dims(Z, Y, X, axis, &n, &H, &W);
    if (s0 < 0) s0 = 0;
    if (s1 > n) s1 = n;
    size_t npx = (size_t)H * W;
    fdo_fb_params fb = {sp->levels, sp->winsize, 3, 5, 1.2, sp->chained ? 4 : 0, sp->box_mode};
    int nt = sp->nthreads > 0 ? sp->nthreads : 1;
    (void)nt;
#pragma omp parallel num_threads(nt)
    {
        float* target = (float*)malloc(npx * sizeof(float));
        float* ref = (float*)malloc(npx * sizeof(float));
        float* warped = (float*)malloc(npx * sizeof(float));
        float* tmp = (float*)malloc(npx * sizeof(float));
        float* flow = (float*)malloc(npx * 2 * sizeof(float));
        double* warped64 = g_f64_padded ? (double*)malloc(npx * sizeof(double)) : NULL;
#pragma omp for schedule(static)
        for (int s = s0; s < s1; s++) {
            get_slice(vol, Z, Y, X, axis, s, target);
            for (size_t j = 0; j < npx; j++) tmp[j] = 0.f;
            if (!sp->use_of) { /* seq:184-185: taps in index order 0..K-1 */
                for (int i = 0; i < K; i++) {
                    neighbour(vol, Z, Y, X, axis, n, s, i, K, mean, sp->border_mode, npx, ref);
                    int q = s + i - K / 2;
                    if (g_f64_padded && sp->border_mode == 0 && (q < 0 || q >= n)) {
                        for (size_t j = 0; j < npx; j++) tmp[j] = (float)((double)tmp[j] + g_mean64 * kernel[i]);
                        continue;
                    }
                    for (size_t j = 0; j < npx; j++)
                        tmp[j] = (float)((double)tmp[j] + (double)ref[j] * kernel[i]);
                }
                if (g_int_round) for (size_t j = 0; j < npx; j++) { float v = truncf(tmp[j]); tmp[j] = v < g_int_lo ? g_int_lo : v > g_int_hi ? g_int_hi : v; }
                put_slice(out, Z, Y, X, axis, s, tmp);
                continue;
            }
            for (int side = 0; side < 2; side++) {
                if (side == 1) /* centre tap between the two chains (seq:108) */
                    for (size_t j = 0; j < npx; j++)
                        tmp[j] = (float)((double)tmp[j] + (double)target[j] * kernel[K / 2]);
                memset(flow, 0, npx * 2 * sizeof(float)); /* seq:94, seq:109 */
                for (int step = 0; step < K / 2; step++) {
                    int i = side == 0 ? K / 2 - 1 - step : K / 2 + 1 + step; /* seq:95, seq:110 */
                    neighbour(vol, Z, Y, X, axis, n, s, i, K, mean, sp->border_mode, npx, ref);
                    if (!sp->chained) memset(flow, 0, npx * 2 * sizeof(float));
                    fdo_farneback(target, ref, flow, H, W, &fb);   /* prev=target, next=reference (seq:62) */
                    if (g_f64_padded) {
                        int q = s + i - K / 2;
                        warp_slice_f64(ref, sp->border_mode == 0 && (q < 0 || q >= n), flow, warped64, H, W);
                        for (size_t j = 0; j < npx; j++) tmp[j] = (float)((double)tmp[j] + warped64[j] * kernel[i]);
                        continue;
                    }
                    if (g_int_round == 2) warp_slice_u8(ref, flow, warped, H, W);   /* a uint8 neighbour image: fixed-point remap, already integral */
                    else fdo_warp_slice(ref, flow, warped, H, W);       /* seq:106 */
                    if (g_int_round == 1)
                        for (size_t j = 0; j < npx; j++) {
                            float v = rintf(warped[j]);
                            warped[j] = v < g_int_lo ? g_int_lo : v > g_int_hi ? g_int_hi : v;
                        }
                    /* seq:107: f32 array * f64 scalar is f64 under numpy>=2; += stores f32 */
                    for (size_t j = 0; j < npx; j++)
                        tmp[j] = (float)((double)tmp[j] + (double)warped[j] * kernel[i]);
                }
            }
            if (g_int_round) for (size_t j = 0; j < npx; j++) { float v = truncf(tmp[j]); tmp[j] = v < g_int_lo ? g_int_lo : v > g_int_hi ? g_int_hi : v; }
            put_slice(out, Z, Y, X, axis, s, tmp);
        }
        free(target); free(ref); free(warped); free(tmp); free(flow); free(warped64);
    }
}

FDO_EXPORT void fdo_filter_axis(const float* vol, float* out, int Z, int Y, int X, int axis,
                                const double* kernel, int K, float mean, const fdo_sweep_params* sp)
{
    int n, H, W;
    slice_dims(Z, Y, X, axis, &n, &H, &W);
    fdo_filter_axis_range(vol, out, Z, Y, X, axis, kernel, K, mean, sp, 0, n);
}

/* a-8 OF_filter (seq:419-424) / a-9 no_OF_filter (seq:426-431): one mean, Z then Y then X.
 * K[a] == 0 skips axis a (used for the "Z only" configuration). */
FDO_EXPORT void fdo_filter_3d(const float* vol, float* out, int Z, int Y, int X,
                              const double* kz, int Kz, const double* ky, int Ky, const double* kx, int Kx,
                              float mean, const fdo_sweep_params* sp)
{
    size_t nvox = (size_t)Z * Y * X;
    float* a = (float*)malloc(nvox * sizeof(float));
    float* b = (float*)malloc(nvox * sizeof(float));
    memcpy(a, vol, nvox * sizeof(float));
    const double* ks[3] = {kz, ky, kx};
    int Ks[3] = {Kz, Ky, Kx};
    for (int axis = 0; axis < 3; axis++) {
        if (Ks[axis] <= 0) continue;
        fdo_filter_axis(a, b, Z, Y, X, axis, ks[axis], Ks[axis], mean, sp);
        float* t = a; a = b; b = t;
    }
    memcpy(out, a, nvox * sizeof(float));
    free(a); free(b);
}

FDO_EXPORT int fdo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
