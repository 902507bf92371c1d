// fdn_iter.hip -- one Farneback ITERATION per launch, for any window size: the wide-window path
// (winsize >= 10; BASELINE configs[4] runs -l 3 -w 15, src/flowdenoising.py:48 defaults to 3 levels).
//
// cv2.calcOpticalFlowFarneback (src/flowdenoising_sequential.py:62) per level: M = UpdateMatrices(flow);
// 3 x { flow = solve(box_w(M)); M = UpdateMatrices(flow) }.  The 3-iteration kernel of fdn_fused.hip keeps all
// of that on chip, but every fused iteration costs w/2 columns of validity either side of a 64-column band
// (22 of 64 lanes left at winsize 15) and 2 (w/2) + 2 rows of matrices per stage.  Here ONE launch is
//     M = UpdateMatrices(R0, R1, flow_in)   ->   flow_out = solve(box_w(M))   [last iteration: + warp, accumulate]
// so a band loses w/2 columns per side once (50 of 64 lanes at winsize 15), the matrices still never reach
// HBM, and what an iteration costs in HBM is its operands: R0 20 + R1 20 + flow in 8 + flow out 8 bytes per
// pixel (the staged kernels move 308 per chain step, three of these launches 180).
//
// Workgroup = 128 threads (192 on the warping launch) = one 64-column band of one (target, neighbour) pair, marching
// down the rows:
//   wave 0 (producer)  row t         : M row from flow_in, R0 and the bilinear gather of R1 -> LDS ring slot t % RS
//   wave 1 (consumer)  row t - MH - 1: OpenCV's vertical running sum vsum += f32(M[y+MH] - M[y-MH-1]) (carried in
//                                      registers from row 0: the f32-fed recurrence is what makes results bit-faithful,
//                                      DESIGN.md 3.2), horizontal window across lanes in two levels of blocks: the
//                                      first (<= 3 consecutive columns) by DPP lane shifts, the second through an
//                                      LDS row; halo lanes idle from there (EXEC mask); 2 x 2 solve, store
//   wave 2 (warper)    row t - MH - 2: last iteration of level 0 only: the 1/32-px remap of the neighbour at p + flow
//                                      and acc = f32(f64(acc) + f64(v) w) (seq:106-107); the flow comes from the
//                                      consumer through two LDS slots
//   one s_barrier per row step; the ring holds rows t - 2 MH - 1 .. t (RS = 2 MH + 2 rows x 1280 B): the slot the
//   producer writes in step t held row t - 2 MH - 2, the consumer's trailing row of this step -- which it has read
//   one step ahead, into registers.
// LDS per workgroup at winsize 15: 20.0 KB ring + 5.3 KB window rows = 25.3 KB -> 6 workgroups = 12 waves per CU.
//
// (Round 5 also ran BOTH sides of a chain step in one launch, mirror pairs in one workgroup: bit-equal, half the HBM bytes
// of the warping launch and no faster -- this kernel is paced by the CU's L1 path, not by HBM.  Removed in round 6:
// profiles/history/r06_two_sided_removed.patch, measurements in profiles/history/NOTES_r05.md.)
#include "fdn_internal.h"
#include "fdn_device.h"
#include <algorithm>
#include <utility>
#include <vector>

#ifndef FDN_ITER_MASK
#define FDN_ITER_MASK 1
#endif

namespace fdn {

// Block length of the two-level window sum of n columns (see the consumer): floor(sqrt(n)), at least 2.
// oracle/fdn_oracle.c box_mode 4 sums in the same order.
static __host__ __device__ constexpr int window_block(int n)
{
    int p = 1;
    while ((p + 1) * (p + 1) <= n) p++;
    return p < 2 ? 2 : p;
}

#ifndef FDN_ITER_DPP_BLOCKS
#define FDN_ITER_DPP_BLOCKS 1
#endif
// whole-wave lane shifts of an f64 (DPP wave_shr:1 / wave_shl:1 on the two halves; 0 comes in at the ends), as in fdn_fused.hip
static __device__ __forceinline__ double lane_shr1(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ double lane_shl1(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

static __device__ __forceinline__ void lds_barrier_iter()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// MHT: window half-width when known at compile time (0: runtime mh).  FIN 0: zero initial flow, 1: flow_in has the
// image's size, 2: flow_in is the next coarser level's (fs.h x fs.w) result, resized INTER_LINEAR and doubled on the fly
// (calc()'s upsampling).  ACC: also warp the neighbour with the new flow and accumulate.
template <int MHT, int FIN, bool ACC, int WM = 0>
__global__ __launch_bounds__(ACC ? 192 : 128) void k_farneback_iter(const float* __restrict__ Rstack, const float* __restrict__ stack,
                                                        const float* __restrict__ flow_in_base, float* __restrict__ flow_out_base,
                                                        float* __restrict__ acc_base, PairBatch pb, int H, int W, int mh_rt,
                                                        double scale, double weight, int nbands, FlowSource fs, WarpMode wm)
{
    const int MH = MHT ? MHT : mh_rt;
    const int RS = 2 * MH + 2;
    const int BW = 64 - 2 * MH;
    const int WP = window_block(2 * MH + 1), WQ = (2 * MH + 1) / WP, WREM = 2 * MH + 1 - WP * WQ, WE = MH % WP;
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave 0: producer, 1: consumer, 2: warper
    float* lds = lds_all;
    // ring row: [ (m0, m2) x 64 ][ (m3, m4) x 64 ][ m1 x 64 ] floats; then the consumer's window row: 5 x (64 + 2 MH) doubles
    float* ring = lds;
    // the window rows (see the consumer): [2][XR] doubles, channel c's lane L at MH + 64 c + L.  Reads run up to MH
    // entries before and MH + 1 after a channel's 64: into the neighbouring channel or the padding at the row's ends --
    // values that only reach lanes whose results are never used (the band's halo lanes)
    double* xch = (double*)(lds + (size_t)RS * 320);
    const int XR = 5 * 64 + 2 * MH + 2;
    // warping launches: the flow of a row goes from the consumer to the third wave through two slots of 64 float2
    float2* fho = (float2*)(xch + 2 * XR);

    const int lane = threadIdx.x & 63;
    // XCD-aware order as in k_farneback_fused: XCD j gets the j-th contiguous eighth of the (pair, band) list
    const long nwg = gridDim.x, q8 = nwg >> 3, rem8 = nwg & 7;
    const long xcd = blockIdx.x & 7;
    const long gw = xcd * q8 + (xcd < rem8 ? xcd : rem8) + (blockIdx.x >> 3);
    const int bw = (int)(gw / nbands);              // position in the walk over the pairs (pair_walk: chains of stride |d|)
    const int band = (int)(gw - (long)bw * nbands);
    const int b = pair_walk(bw, pb.npairs, pb.d);
    const int d = pb.d;
    const int xb = band * BW - MH;
    const int x = xb + lane;
    const int xc = clampi(x, 0, W - 1);       // lanes outside the image replicate the border column (BORDER_REPLICATE of vsum)
    const size_t HW = (size_t)H * W;
    const int T = H + MH + 1 + (ACC ? 1 : 0); // row steps = barriers every wave executes

    if (role == 0) {
        // ===== producer ==================================================================================
        const float* R0 = Rstack + (size_t)(pb.t0 + b) * 5 * HW;
        const float* R1 = Rstack + (size_t)(pb.t0 + b + d) * 5 * HW;
        const RImage R0i = {uniform_ptr(R0), uniform_ptr(R0 + 2 * HW), uniform_ptr(R0 + 4 * HW)};
        const RImage R1i = {uniform_ptr(R1), uniform_ptr(R1 + 2 * HW), uniform_ptr(R1 + 4 * HW)};
        const float* flow_in = FIN == 1 ? uniform_ptr(flow_in_base + (size_t)b * HW * 2)
                             : FIN == 2 ? uniform_ptr(flow_in_base + (size_t)b * fs.h * fs.w * 2) : nullptr;
        const LinearTap ftx = FIN == 2 ? linear_tap(xc, fs.sx, fs.w) : LinearTap{};
        auto load_flow = [&](int row) __attribute__((always_inline)) -> float2 {
            if (FIN == 1) return ld_off<float2>(flow_in, ((unsigned)row * (unsigned)W + (unsigned)xc) * 8u);
            if (FIN == 2) return resize_linear_flow(flow_in, fs.w, ftx, linear_tap(row, fs.sy, fs.h), 2.0, fs.fm, xc, W);
            return make_float2(0.f, 0.f);
        };
        const float bxx = border_factor(xc, W);
        const bool xdamp = border_test(xc, W);
        const float xf = (float)xc;
        // Operands of a row arrive in two dependent hops: its flow and R0, then -- at the position the flow points to --
        // the bilinear footprint of R1.  Three register sets rotate (the loop is unrolled by three) so that neither
        // hop is waited for in the step that issues it: step t loads flow/R0 of row t + 2, issues the gather of row
        // t + 1 (whose flow was loaded a step earlier) and turns row t into M.  (A register copy at the end of a step
        // would make the wave wait for the loads right away.  Streaming (nt) loads of flow / R0 and nt stores of the new
        // flow: 9 % slower -- the columns neighbouring bands share come out of L2.
        // Deeper pipelines measured: two steps for the flow / R0
        // loads, four sets: the same time; two steps per hop, five sets: 164-196 VGPRs, two waves per SIMD, 10 % slower.)
        struct RowOps { float2 f; fdn_v2f r01, r23; float r4; int x1, y1; float fx, fy; GatherTapsP g; };
        auto load_ops = [&](int row, RowOps& o) __attribute__((always_inline)) {
            row = row < H ? row : H - 1;
            o.f = load_flow(row);
            load_R(R0i, (unsigned)row * (unsigned)W + (unsigned)xc, o.r01, o.r23, o.r4);
        };
        auto gather_ops = [&](int row, RowOps& o) __attribute__((always_inline)) {
            row = row < H ? row : H - 1;
            flow_target(xf, (float)row, o.f.x, o.f.y, o.x1, o.y1, o.fx, o.fy);
            gather_R1_p(R1i, H, W, o.x1, o.y1, o.g);
        };
        int slot = 0;
        auto step = [&](int t, const RowOps& cur, RowOps& mid, RowOps& far) __attribute__((always_inline)) {
            if (t < H) {
                load_ops(t + 2, far);
                gather_ops(t + 1, mid);
                const float by0 = t < 5 ? (t < 2 ? 0.14f : 0.4472f) : 1.f;
                const float by1 = t >= H - 5 ? (H - t - 1 < 2 ? 0.14f : 0.4472f) : 1.f;
                fdn_v2f m02, m34; float m1;
                finish_M_p(cur.r01, cur.r23, cur.r4, cur.g, H, W, cur.x1, cur.y1, cur.fx, cur.fy, cur.f.x, cur.f.y, bxx, by0, by1,
                           xdamp || border_test(t, H), m02, m1, m34);
                float* row = ring + (size_t)slot * 320;
                *(fdn_v2f*)(row + 2 * lane) = m02;
                *(fdn_v2f*)(row + 128 + 2 * lane) = m34;
                row[256 + lane] = m1;
                slot = slot + 1 == RS ? 0 : slot + 1;
            }
            lds_barrier_iter();
        };
        RowOps P0, P1, P2;
        load_ops(0, P0);
        load_ops(1, P1);
        gather_ops(0, P0);
        for (int t = 0; t < T; t += 3) {
            step(t, P0, P1, P2);
            if (t + 1 < T) step(t + 1, P1, P2, P0);
            if (t + 2 < T) step(t + 2, P2, P0, P1);
        }
        return;
    }

    const bool owner = lane >= MH && lane < 64 - MH && x < W;
    const float* img1 = ACC ? uniform_ptr(stack + (size_t)(pb.t0 + b + d) * HW) : nullptr;
    float2* flow_out = flow_out_base ? uniform_ptr((float2*)flow_out_base + (size_t)b * HW) : nullptr;
    float* acc = ACC ? uniform_ptr(acc_base + (size_t)b * HW) : nullptr;

    if (ACC && role == 2) {
        // ===== warper (warping launches only): row t - MH - 2, one step behind the consumer ==============
        // acc = f32(f64(acc) + f64(remap(neighbour, p + flow)) w) (seq:106-107).  The remap's four taps sit at p + flow: a
        // dependent global gather whose latency would otherwise sit in the consumer's step (12.0 against 9.2 ms per launch).
        for (int t = 0; t < T; t++) {
            const int y = t - MH - 2;
            if (y >= 0 && y < H && (!FDN_ITER_MASK || owner)) {       // halo lanes idle (EXEC-masked): the launch runs at its power cap
                const float2 f = fho[(y & 1) * 64 + lane];
                const unsigned o = (unsigned)y * (unsigned)W + (unsigned)xc;
                const int q = pb.t0 + b + d;           // (WM: integer-volume semantics, warped_value in fdn_device.h)
                const bool pad = (WM & 3) == 1 && (q < wm.pad_lo || q >= wm.pad_hi);
                const float acc_old = ld_off<float>(acc, o * 4u);
                const float acc_new = fold_warped<(WM & 3), (WM & 4) != 0>(img1, H, W, xc, y, f, acc_old, weight, pad, wm.pad64, wm.lo, wm.hi, wm.fixed8 != 0);
                if (owner) st_off(acc, o * 4u, acc_new);
            }
            lds_barrier_iter();
        }
        return;
    }

    // ===== consumer ======================================================================================
    auto ring_row = [&](int r, float m[5]) __attribute__((always_inline)) {       // channels in OpenCV's order m0..m4
        const float* row = ring + (size_t)(r % RS) * 320;
        const fdn_v2f p = *(const fdn_v2f*)(row + 2 * lane), q = *(const fdn_v2f*)(row + 128 + 2 * lane);
        m[0] = p.x; m[2] = p.y; m[3] = q.x; m[4] = q.y; m[1] = row[256 + lane];
    };
    double vs[5] = {0., 0., 0., 0., 0.};
    float trail[5] = {0.f, 0.f, 0.f, 0.f, 0.f};     // row y - MH - 1 of M, read one step ahead (its slot is rewritten in step t)
    for (int t = 0; t < T; t++) {
        if (t == MH) {   // rows 0 .. MH-1 are in the ring: vsum before row 0 = f32(M[0] (MH + 2)) + rows 1 .. MH-1 (clamped)
            float m[5];
            ring_row(0, m);
#pragma unroll
            for (int c = 0; c < 5; c++) vs[c] = (double)(m[c] * (float)(MH + 2));
            for (int yy = 1; yy < MH; yy++) {
                ring_row(yy < H - 1 ? yy : H - 1, m);
#pragma unroll
                for (int c = 0; c < 5; c++) vs[c] += (double)m[c];
            }
            ring_row(0, trail);          // row 0's trailing row is row 0 (clamped)
        } else if (t > MH && t - MH - 1 < H) {
            const int y = t - MH - 1;
            const unsigned o = (unsigned)y * (unsigned)W + (unsigned)xc;
            float lead[5];
            ring_row(y + MH < H - 1 ? y + MH : H - 1, lead);
#pragma unroll
            for (int c = 0; c < 5; c++) vs[c] += (double)(lead[c] - trail[c]);
            ring_row(y - MH > 0 ? y - MH : 0, trail);           // the next row's trailing row, while its slot still holds it
            // Horizontal window of n = 2 MH + 1 columns in two levels of blocks, through two LDS rows: with p =
            // floor(sqrt(n)) (window_block), q = n / p, rem = n - p q,
            //     B[L] = v[L-e] + ... + v[L-e+p-1]                (p consecutive columns, left to right; e = MH % p)
            //     window[L] = B[L-MH+e] + B[L-MH+e+p] + ... (q blocks, left to right) + v[L+MH-rem+1] + ... + v[L+MH]
            // e places one of the q blocks at offset 0 and the lane's own column inside B: both come from registers.
            // winsize 15: p = 3, q = 5: 2 + 4 LDS reads, 2 writes and 6 additions per channel in two write -> read round
            // trips.  (The consumer's row step is a chain of dependent LDS round trips, and that chain -- not the
            // operation count -- is what the step waits for: summing the window by doubling, T2k[L] = Tk[L] + Tk[L+k],
            // takes 7 reads, 4 writes, 6 additions but FOUR round trips: 16 % slower, 1.84 against 1.55 s per -l 3 -w 15
            // volume; radix 4 then singles: 1.73 s; the plain 14-read window in one round trip: 2.2 s.)
            // A wave's LDS operations execute in order, so a row may be rewritten once the reads of its previous
            // content have been issued; the fences only keep the compiler from reordering across them.
            // (Measured and rejected, all bit-identical: splitting the row into pipeline stages over consecutive steps
            // -- window sums / solve + tap issue / weighting -- 6 % slower; only deferring the remap taps' weighting to
            // the next step, the loop unrolled by two for the alternating tap registers: 6 % slower; two neighbouring
            // bands per workgroup so that their shared columns come from HBM once: 5 % slower, four waves per barrier.
            // With the two-level window: reading the leading row one step ahead as well (the consumer one more row
            // behind): no change -- the step no longer waits for the consumer's chain.  Timing-only builds, same
            // volume: producer alone 1.26 s, consumer alone 1.04 s, both 1.55 s.)
            double a[5], blk[5];
            {
                double* row0 = xch;
                double* row1 = xch + XR;
                // the single columns go to LDS only when the window's tail (rem) reads them; the blocks of p <= 3
                // consecutive columns are summed by lane shifts (same terms, same order; a shift brings 0 in at the
                // wave's ends where the LDS row holds the neighbouring channel: halo lanes only, either way) --
                // one write -> read round trip per row instead of two (FDN_ITER_DPP_BLOCKS = 0: both levels through LDS)
                const bool dpp_blocks = FDN_ITER_DPP_BLOCKS && MHT != 0 && WP <= 3;
                if (!dpp_blocks || WREM > 0) {
#pragma unroll
                    for (int c = 0; c < 5; c++) row0[c * 64 + lane + MH] = vs[c];
                }
                if (!dpp_blocks) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
#pragma unroll
                for (int c = 0; c < 5; c++) {
                    double s;
                    if (dpp_blocks) {       // B[L] = v[L-e] + ... + v[L-e+p-1], left to right
                        const double l1 = lane_shr1(vs[c]), r1 = lane_shl1(vs[c]);
                        if (WP == 2) s = WE == 1 ? l1 + vs[c] : vs[c] + r1;
                        else if (WE == 0) s = (vs[c] + r1) + lane_shl1(r1);
                        else if (WE == 1) s = (l1 + vs[c]) + r1;
                        else s = (lane_shr1(l1) + l1) + vs[c];
                    } else {
                        const double* r = row0 + c * 64 + lane + MH;
                        s = WE == 0 ? vs[c] : r[-WE];
                        for (int j = 1; j < WP; j++) s += j == WE ? vs[c] : r[j - WE];
                    }
                    blk[c] = s;
                    row1[c * 64 + lane + MH] = s;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (!FDN_ITER_MASK || owner) {      // the band's halo lanes stop here: their window sums feed nobody
#pragma unroll
                    for (int c = 0; c < 5; c++) {
                        const double* r1 = row1 + c * 64 + lane + MH;
                        const double* r0 = row0 + c * 64 + lane + MH;
                        double s = 0.;
                        for (int bk = 0; bk < WQ; bk++) {
                            const int off = -MH + WE + WP * bk;
                            const double term = off == 0 ? blk[c] : r1[off];
                            s = bk == 0 ? term : s + term;
                        }
                        for (int j = 0; j < WREM; j++) s += r0[MH - WREM + 1 + j];
                        a[c] = s;
                    }
                    const float2 f = solve_flow(a, scale);
                    if (ACC) fho[(y & 1) * 64 + lane] = f;         // to the warper; this slot was last read two steps ago
                    if (owner && flow_out) st_off(flow_out, o * 8u, f);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        lds_barrier_iter();
    }
}

bool iter_supported(int winsize, int H, int W)
{
    const int mh = winsize / 2;
    // the window row needs 2 mh < 64 columns left for outputs; pixels are addressed by 32-bit byte offsets
    return mh >= 1 && mh <= 24 && H >= 2 && W >= 2 && H < (1 << 24) && W < (1 << 24) && (size_t)H * W < ((size_t)1 << 29);
}

size_t iter_lds_bytes(int mh, bool acc)
{
    return (size_t)(2 * mh + 2) * 320 * sizeof(float) + (size_t)2 * (5 * 64 + 2 * mh + 2) * sizeof(double) + (acc ? 2 * 64 * sizeof(float2) : 0);
}

template <int MHT>
static int launch_iter_t(const float* Rstack, const float* stack, const float* flow_in, float* flow_out, float* acc, PairBatch pb,
                         int H, int W, int mh, double scale, double weight, FlowSource fs, hipStream_t st, const WarpMode& wm)
{
    const int BW = 64 - 2 * mh;
    const int nbands = (W + BW - 1) / BW;
    dim3 grid((unsigned)((long)nbands * pb.npairs));
    const size_t lds = iter_lds_bytes(mh, acc != nullptr);
    const int fin = !flow_in ? 0 : fs.h > 0 ? 2 : 1;
    auto launch = [&](auto kern) -> int {
        if (lds > 48 * 1024) {      // a kernel must be told (per device and host thread) that it may take that much dynamic LDS:
            struct Told { const void* k; int dev; size_t bytes; };      // once per size it has not been granted yet
            static thread_local std::vector<Told> told;
            int dev = 0;
            (void)hipGetDevice(&dev);
            auto it = std::find_if(told.begin(), told.end(), [&](const Told& t) { return t.k == (const void*)kern && t.dev == dev; });
            if (it == told.end() || it->bytes < lds) {
                if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); return -1; }
                if (it == told.end()) told.push_back(Told{(const void*)kern, dev, lds}); else it->bytes = lds;
            }
        }
        hipLaunchKernelGGL(kern, grid, dim3(acc ? 192 : 128), lds, st, Rstack, stack, flow_in, flow_out, acc, pb, H, W, mh, scale, weight,
                           nbands, fs, wm);
        return hipGetLastError() == hipSuccess ? 0 : -1;      // a bad launch configuration is this launch's error, not the next check's
    };
    if (acc && wm.kind == 1) {
        if (fin == 2) return launch(k_farneback_iter<MHT, 2, true, 1>);
        if (fin == 1) return launch(k_farneback_iter<MHT, 1, true, 1>);
        return launch(k_farneback_iter<MHT, 0, true, 1>);
    }
    if (acc && wm.kind == 2) {
        if (fin == 2) return launch(k_farneback_iter<MHT, 2, true, 2>);
        if (fin == 1) return launch(k_farneback_iter<MHT, 1, true, 2>);
        return launch(k_farneback_iter<MHT, 0, true, 2>);
    }
    if (acc && wm.model == 1) {         // float32 volume, unquantised remap (the "remap_model" option): WM = 4
        if (fin == 2) return launch(k_farneback_iter<MHT, 2, true, 4>);
        if (fin == 1) return launch(k_farneback_iter<MHT, 1, true, 4>);
        return launch(k_farneback_iter<MHT, 0, true, 4>);
    }
    if (acc) {
        if (fin == 2) return launch(k_farneback_iter<MHT, 2, true, 0>);
        if (fin == 1) return launch(k_farneback_iter<MHT, 1, true, 0>);
        return launch(k_farneback_iter<MHT, 0, true, 0>);
    }
    if (fin == 2) return launch(k_farneback_iter<MHT, 2, false, 0>);
    if (fin == 1) return launch(k_farneback_iter<MHT, 1, false, 0>);
    return launch(k_farneback_iter<MHT, 0, false, 0>);
}

// One iteration for every pair of the batch.  flow_in: nullptr = zero flow; coarse_h, coarse_w > 0: flow_in is the next
// coarser level's flow of that size.  acc != nullptr: this is the last iteration of the finest level: warp + accumulate
// (flow_out may then be nullptr when nobody needs the flow).  flow_in and flow_out must be different buffers.
int launch_farneback_iter(const float* Rstack, const float* stack, const float* flow_in, float* flow_out, float* acc,
                          PairBatch pb, int H, int W, int winsize, double weight, hipStream_t st, int coarse_h, int coarse_w, const WarpMode& wm, const FmaMode& fm)
{
    if (pb.npairs <= 0) return 0;
    FlowSource fs{coarse_h, coarse_w, coarse_h > 0 ? (double)coarse_w / W : 1.0, coarse_h > 0 ? (double)coarse_h / H : 1.0, fm};
    const double scale = 1. / ((double)winsize * winsize);
    const int mh = winsize / 2;
    switch (mh) {   // compile-time windows for the usual sizes; anything else takes the runtime-width build
    case 2: return launch_iter_t<2>(Rstack, stack, flow_in, flow_out, acc, pb, H, W, mh, scale, weight, fs, st, wm);
    case 5: return launch_iter_t<5>(Rstack, stack, flow_in, flow_out, acc, pb, H, W, mh, scale, weight, fs, st, wm);
    case 7: return launch_iter_t<7>(Rstack, stack, flow_in, flow_out, acc, pb, H, W, mh, scale, weight, fs, st, wm);
    default: return launch_iter_t<0>(Rstack, stack, flow_in, flow_out, acc, pb, H, W, mh, scale, weight, fs, st, wm);
    }
}

} // namespace fdn
