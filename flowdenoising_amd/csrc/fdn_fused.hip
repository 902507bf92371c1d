// fdn_fused.hip -- the fast path: one kernel launch = one chain step of the sweep for EVERY
// target slice of the batch: the whole cv2.calcOpticalFlowFarneback(prev=target,
// next=neighbour, flow=previous flow, levels=0) (src/flowdenoising_sequential.py:62), the
// warp of the neighbour (seq:51-57) and the weighted accumulate (seq:107), fused.
//
// Decomposition (wave64, no block-level synchronisation at all):
//   one wave = one band of 64 image columns of one (target, neighbour) pair, marching down the
//   rows.  The ITERS flow iterations run as a software pipeline staggered by MH+1 rows:
//     stage A   row t            : M0 = UpdateMatrices(R0, R1, flow_in)
//     stage k   row t - k(MH+1)  : vsum_k += f32(M_{k-1}[y+MH] - M_{k-1}[y-MH-1])   (OpenCV's running sum,
//                                  carried in registers from row 0, hence bit-faithful)
//                                  box sum across lanes (wave shuffles, f64), 2x2 solve -> flow_k
//                                  k < ITERS: M_k = UpdateMatrices(R0, R1, flow_k)
//                                  k = ITERS: store flow, warp the neighbour, accumulate
//   The 2MH+2 most recent rows of each M_k stay on chip: M0 in a VGPR shift register, the others
//   in LDS rings private to the wave ([slot][channel][lane] -> conflict-free).  No M ever
//   reaches HBM; per pair the kernel reads R0, R1 (L2-served re-reads), the chain flow, the
//   neighbour image and the accumulator once.
//   Each iteration loses MH columns of validity either side, so a band yields
//   64 - 2*MH*ITERS output columns (52 for winsize 5); windows that reach outside the image
//   read the lane of the clamped column, which is BORDER_REPLICATE of the running sums.
//   In the steady state (all stages active) the step body is one branch-free basic block so
//   that the three stages' independent memory latencies overlap; the loads that do not depend
//   on this step's flows (stage A's operands, the R0 rows) are issued before the solves.
#include "fdn_internal.h"
#include "fdn_device.h"

namespace fdn {

template <int MH, int ITERS, bool HAS_FIN>
__global__ __launch_bounds__(256, 2) void k_farneback_fused(const float* __restrict__ Rstack, const float* __restrict__ stack,
                                                            const float* __restrict__ flow_in_base, float* __restrict__ flow_out_base,
                                                            float* __restrict__ acc_base, PairBatch pb, int H, int W,
                                                            double scale, double weight, int nbands)
{
    constexpr int RS = 2 * MH + 2;       // rows of M_k a consumer can still need
    constexpr int HALO = MH * ITERS;
    constexpr int BW = 64 - 2 * HALO;
    constexpr int NL = ITERS > 1 ? ITERS - 1 : 1;
    __shared__ float ringL[4][NL][RS][5][64];

    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const long gw = (long)blockIdx.x * 4 + wv;
    if (gw >= (long)nbands * pb.npairs) return;   // whole wave leaves; nothing below synchronises across waves
    const int b = (int)(gw / nbands);
    const int band = (int)(gw - (long)b * nbands);
    const int x = band * BW - HALO + lane;
    const int xc = clampi(x, 0, W - 1);
    const bool owner = lane >= HALO && lane < 64 - HALO && x < W;
    const size_t HW = (size_t)H * W;
    const float* R0 = Rstack + (size_t)(pb.t0 + b) * 5 * HW;
    const float* R1 = Rstack + (size_t)(pb.t0 + b + pb.d) * 5 * HW;
    const float* img1 = stack + (size_t)(pb.t0 + b + pb.d) * HW;
    const float2* flow_in = HAS_FIN ? (const float2*)flow_in_base + (size_t)b * HW : nullptr;
    float2* flow_out = flow_out_base ? (float2*)flow_out_base + (size_t)b * HW : nullptr;
    float* acc = acc_base + (size_t)b * HW;
    float (*ring)[RS][5][64] = ringL[wv];
    const float bxx = border_factor(xc, W);
    const bool xdamp = border_test(xc, W);

    // Lanes the horizontal window of this lane reads: the lane that owns column clamp(x+j).
    // (Replica lanes outside the image are never read: a replica's own window is shifted, so
    // from the second iteration on it would no longer equal the border column it stands for.)
    int src[2 * MH + 1];
#pragma unroll
    for (int j = -MH; j <= MH; j++) src[j + MH] = clampi(clampi(x + j, 0, W - 1) - (x - lane), 0, 63);

    float ring0[RS][5];       // M0 rows: ring0[j] = row clamp(newest - j)
    double vs[ITERS][5];
#pragma unroll
    for (int j = 0; j < RS; j++)
#pragma unroll
        for (int c = 0; c < 5; c++) ring0[j][c] = 0.f;
#pragma unroll
    for (int k = 0; k < ITERS; k++)
#pragma unroll
        for (int c = 0; c < 5; c++) vs[k][c] = 0.;

    auto row_factor = [&](int y, float& by0, float& by1) {
        by0 = y < 5 ? (y < 2 ? 0.14f : 0.4472f) : 1.f;
        by1 = y >= H - 5 ? (H - y - 1 < 2 ? 0.14f : 0.4472f) : 1.f;
    };

    // ---- general step: any stage may be inactive (pipeline fill / drain) ------------------
    auto general_step = [&](int t) __attribute__((always_inline)) {
        // consumers first (highest stage first): every ring is read before this step overwrites it
#pragma unroll
        for (int k = ITERS; k >= 1; k--) {
            const int y = t - k * (MH + 1);
            if (y < 0 || y >= H) continue;            // wave-uniform
            float mnew[5], mold[5];
            if (k == 1) {
                if (y == 0) { // vsum before row 0: f32(M[0]*(m+2)) + rows 1..m-1   (ring0[j] = row MH - j here)
#pragma unroll
                    for (int c = 0; c < 5; c++) {
                        double v = (double)(ring0[MH][c] * (float)(MH + 2));
#pragma unroll
                        for (int yy = 1; yy < MH; yy++) v += (double)ring0[MH - yy][c];
                        vs[0][c] = v;
                    }
                }
#pragma unroll
                for (int c = 0; c < 5; c++) { mnew[c] = ring0[0][c]; mold[c] = ring0[RS - 1][c]; }
            } else {
                float (*rg)[5][64] = ring[k - 2];
                if (y == 0) {
#pragma unroll
                    for (int c = 0; c < 5; c++) {
                        double v = (double)(rg[0][c][lane] * (float)(MH + 2));
#pragma unroll
                        for (int yy = 1; yy < MH; yy++) v += (double)rg[(yy < H - 1 ? yy : H - 1) % RS][c][lane];
                        vs[k - 1][c] = v;
                    }
                }
                const int rn = (y + MH < H - 1 ? y + MH : H - 1) % RS;
                const int ro = (y - MH - 1 > 0 ? y - MH - 1 : 0) % RS;
#pragma unroll
                for (int c = 0; c < 5; c++) { mnew[c] = rg[rn][c][lane]; mold[c] = rg[ro][c][lane]; }
            }
            double a[5];
#pragma unroll
            for (int c = 0; c < 5; c++) {
                vs[k - 1][c] += (double)(mnew[c] - mold[c]);
                double s = 0;
#pragma unroll
                for (int j = 0; j <= 2 * MH; j++) s += j == MH ? vs[k - 1][c] : __shfl(vs[k - 1][c], src[j], 64);
                a[c] = s;
            }
            const float2 f = solve_flow(a, scale);
            const size_t o = (size_t)y * W + xc;
            if (k < ITERS) {
                float r0[5], mm[5];
#pragma unroll
                for (int c = 0; c < 5; c++) r0[c] = R0[c * HW + o];
                compute_M(r0, R1, HW, H, W, xc, y, f.x, f.y, mm);
#pragma unroll
                for (int c = 0; c < 5; c++) ring[k - 1][y % RS][c][lane] = mm[c];
            } else if (owner) {
                if (flow_out) flow_out[o] = f;
                float v = remap_sample(img1, H, W, x, y, f);
                acc[o] = (float)((double)acc[o] + (double)v * weight);
            }
        }
        // stage A: push M0 row t (below the image: a replica of the last row, so that
        // ring0[j] always holds row clamp(newest - j))
#pragma unroll
        for (int j = RS - 1; j >= 1; j--)
#pragma unroll
            for (int c = 0; c < 5; c++) ring0[j][c] = ring0[j - 1][c];
        if (t < H) {
            const size_t o = (size_t)t * W + xc;
            float2 f = HAS_FIN ? flow_in[o] : make_float2(0.f, 0.f);
            float r0[5], mm[5];
#pragma unroll
            for (int c = 0; c < 5; c++) r0[c] = R0[c * HW + o];
            compute_M(r0, R1, HW, H, W, xc, t, f.x, f.y, mm);
#pragma unroll
            for (int c = 0; c < 5; c++) ring0[0][c] = mm[c];
            if (t == 0) { // rows above the image replicate row 0
#pragma unroll
                for (int j = 1; j < RS; j++)
#pragma unroll
                    for (int c = 0; c < 5; c++) ring0[j][c] = mm[c];
            }
        }
    };

    const int T = H + ITERS * (MH + 1);
    // steady state: every stage active, no row clamps: (ITERS+1)(MH+1) <= t <= H-1
    const int ts0 = (ITERS + 1) * (MH + 1);
    const int ts1 = H; // exclusive
    int t = 0;
    for (; t < T && (t < ts0 || ts0 >= ts1); t++) general_step(t);

    if (t < ts1) {
        int slot[ITERS + 1]; // slot[k] = y_k % RS (k >= 1)
#pragma unroll
        for (int k = 1; k <= ITERS; k++) slot[k] = (t - k * (MH + 1)) % RS;

        for (; t < ts1; t++) {
            // ---- loads that depend on nothing computed in this step go first: stage A's
            //      operands (the only ones that miss to HBM) and the R0 rows of the other stages
            const size_t oA = (size_t)t * W + xc;
            const float2 fA = HAS_FIN ? flow_in[oA] : make_float2(0.f, 0.f);
            float r0A[5];
#pragma unroll
            for (int c = 0; c < 5; c++) r0A[c] = R0[c * HW + oA];
            int x1A, y1A; float fxA, fyA;
            flow_target(xc, t, fA.x, fA.y, x1A, y1A, fxA, fyA);
            GatherTaps gA;
            gather_R1(R1, HW, H, W, x1A, y1A, gA);
            float r0k[ITERS][5];
#pragma unroll
            for (int k = 1; k < ITERS; k++) {
                const size_t o = (size_t)(t - k * (MH + 1)) * W + xc;
#pragma unroll
                for (int c = 0; c < 5; c++) r0k[k][c] = R0[c * HW + o];
            }
            const int yl = t - ITERS * (MH + 1);
            const size_t ol = (size_t)yl * W + xc;
            const float acc_old = acc[ol];
            // ---- running sums, box sums, solves for all stages --------------------------
            float2 f[ITERS + 1];
#pragma unroll
            for (int k = 1; k <= ITERS; k++) {
                float mnew[5], mold[5];
                if (k == 1) {
#pragma unroll
                    for (int c = 0; c < 5; c++) { mnew[c] = ring0[0][c]; mold[c] = ring0[RS - 1][c]; }
                } else {
                    // rows y+MH and y-MH-1 of M_{k-1}: slots (y+MH)%RS and (y+MH+1)%RS
                    int rn = slot[k] + MH; rn = rn >= RS ? rn - RS : rn;
                    int ro = rn + 1 == RS ? 0 : rn + 1;
#pragma unroll
                    for (int c = 0; c < 5; c++) { mnew[c] = ring[k - 2][rn][c][lane]; mold[c] = ring[k - 2][ro][c][lane]; }
                }
                double a[5];
#pragma unroll
                for (int c = 0; c < 5; c++) {
                    vs[k - 1][c] += (double)(mnew[c] - mold[c]);
                    double s = 0;
#pragma unroll
                    for (int j = 0; j <= 2 * MH; j++) s += j == MH ? vs[k - 1][c] : __shfl(vs[k - 1][c], src[j], 64);
                    a[c] = s;
                }
                f[k] = solve_flow(a, scale);
            }
            // ---- loads that depend on the new flows ---------------------------------------
            GatherTaps gk[ITERS];
            int x1k[ITERS], y1k[ITERS]; float fxk[ITERS], fyk[ITERS];
#pragma unroll
            for (int k = 1; k < ITERS; k++) {
                flow_target(xc, t - k * (MH + 1), f[k].x, f[k].y, x1k[k], y1k[k], fxk[k], fyk[k]);
                gather_R1(R1, HW, H, W, x1k[k], y1k[k], gk[k]);
            }
            const float warped = remap_sample(img1, H, W, xc, yl, f[ITERS]);
            // ---- matrices ------------------------------------------------------------------
            float by0, by1, mm[5];
            row_factor(t, by0, by1);
            finish_M(r0A, gA, H, W, x1A, y1A, fxA, fyA, fA.x, fA.y, bxx, by0, by1, xdamp || border_test(t, H), mm);
#pragma unroll
            for (int j = RS - 1; j >= 1; j--)
#pragma unroll
                for (int c = 0; c < 5; c++) ring0[j][c] = ring0[j - 1][c];
#pragma unroll
            for (int c = 0; c < 5; c++) ring0[0][c] = mm[c];
#pragma unroll
            for (int k = 1; k < ITERS; k++) {
                row_factor(t - k * (MH + 1), by0, by1);
                finish_M(r0k[k], gk[k], H, W, x1k[k], y1k[k], fxk[k], fyk[k], f[k].x, f[k].y, bxx, by0, by1,
                         xdamp || border_test(t - k * (MH + 1), H), mm);
#pragma unroll
                for (int c = 0; c < 5; c++) ring[k - 1][slot[k]][c][lane] = mm[c];
            }
            // ---- outputs (computed outside the predicate so the loads are not sunk into it) ---
            const float acc_new = (float)((double)acc_old + (double)warped * weight);
            if (owner) {
                if (flow_out) flow_out[ol] = f[ITERS];
                acc[ol] = acc_new;
            }
#pragma unroll
            for (int k = 1; k <= ITERS; k++) slot[k] = slot[k] + 1 == RS ? 0 : slot[k] + 1;
        }
    }
    for (; t < T; t++) general_step(t);
}

bool fused_supported(int winsize, int iters, int H, int W)
{
    return winsize / 2 == 2 && iters == 3 && H >= 2 && W >= 2;
}

void launch_farneback_fused(const float* Rstack, const float* stack, const float* flow_in, float* flow_out, float* acc,
                            PairBatch pb, int H, int W, int winsize, int iters, double weight, hipStream_t st)
{
    if (pb.npairs <= 0) return;
    (void)iters;
    constexpr int MH = 2, ITERS = 3;
    const int BW = 64 - 2 * MH * ITERS;
    int nbands = (W + BW - 1) / BW;
    long waves = (long)nbands * pb.npairs;
    double scale = 1. / ((double)winsize * winsize);
    dim3 grid((unsigned)((waves + 3) / 4));
    if (flow_in)
        hipLaunchKernelGGL((k_farneback_fused<MH, ITERS, true>), grid, dim3(256), 0, st,
                           Rstack, stack, flow_in, flow_out, acc, pb, H, W, scale, weight, nbands);
    else
        hipLaunchKernelGGL((k_farneback_fused<MH, ITERS, false>), grid, dim3(256), 0, st,
                           Rstack, stack, flow_in, flow_out, acc, pb, H, W, scale, weight, nbands);
}

} // namespace fdn
