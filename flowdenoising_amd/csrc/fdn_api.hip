// fdn_api.hip -- C ABI (include/flowdn.h) over the gfx950 kernels: handle, scratch
// memory, the sweep driver that batches the Farneback chain over all target slices, and
// the host-pointer convenience entry points.
#include "../../include/flowdn.h"
#include "fdn_internal.h"

#include <fcntl.h>
#include <math.h>
#include <stdarg.h>
#include <unistd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <array>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace fdn {

static thread_local std::string g_err;

static int fail(const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
}

#define FDN_HIP(expr)                                                                   \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) return fail("%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// ---- host-side constants ----------------------------------------------------------
// FarnebackPrepareGaussian (OpenCV optflowgf.cpp), reached from seq:62 with poly_n = 5,
// poly_sigma = 1.2: f32 taps g, x*g, x^2*g and four entries of the inverse of the 6x6 Gram matrix
// of (1, x, y, x^2, y^2, xy).  The inverse follows cv::invert(DECOMP_CHOLESKY) operation by
// operation: the last bit of ig03/ig33 decides the f32 rounding of a handful of expansion
// coefficients per volume, and the chained flow solve amplifies even that.
static void spd6_inverse_cholesky(double G[6][6], double X[6][6])
{
    for (int r = 0; r < 6; r++)
        for (int c = 0; c < 6; c++) X[r][c] = r == c ? 1.0 : 0.0;
    for (int r = 0; r < 6; r++) {                       // G <- L, with 1/L_rr on the diagonal
        for (int c = 0; c < r; c++) {
            double acc = G[r][c];
            for (int q = 0; q < c; q++) acc -= G[r][q] * G[c][q];
            G[r][c] = acc * G[c][c];
        }
        double acc = G[r][r];
        for (int q = 0; q < r; q++) { double t = G[r][q]; acc -= t * t; }
        G[r][r] = 1. / sqrt(acc);
    }
    for (int r = 0; r < 6; r++)                         // L Y = I
        for (int c = 0; c < 6; c++) {
            double acc = X[r][c];
            for (int q = 0; q < r; q++) acc -= G[r][q] * X[q][c];
            X[r][c] = acc * G[r][r];
        }
    for (int r = 5; r >= 0; r--)                        // L^T X = Y
        for (int c = 0; c < 6; c++) {
            double acc = X[r][c];
            for (int q = 5; q > r; q--) acc -= G[q][r] * X[q][c];
            X[r][c] = acc * G[r][r];
        }
}

void prepare_poly_consts(int n, double sigma, PolyConsts* pc)
{
    if (sigma < 1.1920928955078125e-07) sigma = n * 0.3;
    std::vector<float> gb(2 * n + 1), xgb(2 * n + 1), xxgb(2 * n + 1);
    float* g = gb.data() + n; float* xg = xgb.data() + n; float* xxg = xxgb.data() + n;
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
    double G[6][6] = {};
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
    double X[6][6];
    spd6_inverse_cholesky(G, X);
    pc->n = n;
    pc->ig11 = X[1][1];
    pc->ig03 = X[0][3];
    pc->ig33 = X[3][3];
    pc->ig55 = X[5][5];
    for (int k = 0; k <= n; k++) { pc->g[k] = g[k]; pc->xg[k] = xg[k]; pc->xxg[k] = xxg[k]; }
}

// cv::getGaussianKernel(n, sigma, CV_32F)
void prepare_blur_taps(int n, double sigma, BlurTaps* bt)
{
    bt->n = n;
    if (sigma <= 0 && (n == 1 || n == 3 || n == 5 || n == 7)) {
        static const float t1[] = {1.f};
        static const float t3[] = {0.25f, 0.5f, 0.25f};
        static const float t5[] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
        static const float t7[] = {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f};
        const float* t = n == 1 ? t1 : n == 3 ? t3 : n == 5 ? t5 : t7;
        for (int i = 0; i < n; i++) bt->k[i] = t[i];
        return;
    }
    double sigmaX = sigma > 0 ? sigma : n * 0.15 + 0.35;
    double scale2X = -0.125 / (sigmaX * sigmaX);
    int n2 = (n - 1) / 2;
    std::vector<double> vals(n2 + 1);
    double sum = 0;
    for (int i = 0, x = 1 - n; i < n2; i++, x += 2) {
        vals[i] = exp((double)(x * x) * scale2X);
        sum += vals[i];
    }
    sum *= 2.0;
    sum += 1.0;
    if ((n & 1) == 0) sum += 1.0;
    double mul1 = 1.0 / sum;
    for (int i = 0; i < n2; i++) bt->k[i] = bt->k[n - 1 - i] = (float)(vals[i] * mul1);
    bt->k[n2] = (float)mul1;
    if ((n & 1) == 0) bt->k[n2 + 1] = (float)mul1;
}

} // namespace fdn

using namespace fdn;

// Device-wide operations of the runtime -- allocating, freeing (hipFree waits for the whole device), creating and destroying
// streams and events -- are taken one at a time across the handles of a process.  Handles are per host thread (par calls its
// pair operators from P threads, par:187-193; the out-of-core mode has worker threads with handles of their own), and in round
// 5 long sessions that had created and destroyed handles in several threads at once saw the runtime abort some calls later
// (profiles/history/NOTES_r05.md, section 5).  These calls are rare and slow; the lock costs nothing measurable.
static std::recursive_mutex g_device_wide;
#define FDN_DEVICE_WIDE std::lock_guard<std::recursive_mutex> device_wide_guard_(g_device_wide)

// ---- handle -----------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct fdn_ctx {
    // Every entry point that takes a handle holds this for the whole call (FDN_ENTER): par calls its pair operators from P
    // pool threads at once (src/flowdenoising.py:187-193, 299-327) and all of them may share one handle -- its stream, its
    // pinned staging buffer and its device scratch are then used by one call at a time.  Recursive: some entry points are
    // written in terms of others (fdn_farneback -> fdn_farneback_strided, the sharded mean -> fdn_np_chunk_sums_dev).
    std::recursive_mutex mu;
    bool reserve_only = false;   // fdn_reserve_3d: size and allocate every buffer of a call, launch nothing
    size_t reserve_extern = 0;   // ... leaving this much free: the caller's own input / output volumes, not allocated yet
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // Second stream of the handle: a pass over a SMALL grid runs its target slices as two independent sub-batches, one per
    // stream (sweep_stack), so that the tail of one launch is filled by the other sub-batch's work instead of idling.
    // Created on first use; forked from / joined into `stream` by the two events.
    static constexpr int MAX_SUB = 4;
    hipStream_t aux_stream[MAX_SUB - 1] = {};
    hipEvent_t ev_fork = nullptr, ev_join[MAX_SUB - 1] = {};
    int last_sub_batches = 1;    // what the last sweep ran with (fdn_get_option "last_sub_batches")
    size_t ws_limit = 0;
    Tuning tn;
    DevBuf R, M0, M1, flow, stack, sweep_out, vol_a, vol_b, partials, pair, vol_in, vol_out;
    void* pinned = nullptr;          // host staging of the pair-level entry points (hipHostMalloc)
    size_t pinned_cap = 0;
    void* bounce = nullptr;          // page-locked bounce buffer of copy_host / copy_host_2d (hipHostMalloc): two halves
    size_t bounce_cap = 0;
    hipEvent_t bounce_ev[2] = {};    // ... and when the copy engine is done with each half
    DevBuf Rpyr, flow_pyr, pyr_tmp, area_tab;   // pyramid levels >= 1
    DevBuf sh_send, sh_recv, sh_stack, sh_out[2], sh_tmp;   // fdn_filter_3d_sharded: staging, stack and pass outputs
    struct AreaKey { int sh, sw, dh, dw; } area_key = {0, 0, 0, 0};
    struct AreaPtrs { const int *x_si, *x_start, *y_si, *y_start; const float *x_alpha, *y_alpha; } area = {};
    // timers: event pairs are recorded asynchronously and resolved in fdn_get_timers
    bool timers = false;
    double tms[FDN_TIMER_COUNT] = {};
    long long tcount[FDN_TIMER_COUNT] = {};
    struct Stamp { hipEvent_t a, b; int which; };
    std::vector<Stamp> stamps;        // recorded, not yet resolved
    std::vector<hipEvent_t> ev_pool;  // free events
    int trace_fd = -1;                // FDN_LAUNCH_TRACE=<file>: every stage of the per-stage / pyramid paths is named there before it
                                      // is launched and waited for afterwards -- a GPU fault then has a last line (diagnostic only)
};

// Diagnostic: name the stage that is about to run (and wait for everything before it), so that an abort of the runtime --
// a memory fault of a kernel is reported asynchronously, without the kernel's name -- can be attributed.
static void trace_point(fdn_ctx* h, const char* fmt, ...)
{
    if (h->trace_fd < 0) return;
    (void)hipStreamSynchronize(h->stream);
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    int n = vsnprintf(buf, sizeof buf - 1, fmt, ap);
    va_end(ap);
    if (n < 0) return;
    if (n > (int)sizeof buf - 2) n = (int)sizeof buf - 2;
    buf[n++] = '\n';
    (void)!write(h->trace_fd, buf, (size_t)n);
}

static hipEvent_t get_event(fdn_ctx* h)
{
    if (!h->ev_pool.empty()) { hipEvent_t e = h->ev_pool.back(); h->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    FDN_DEVICE_WIDE;
    (void)hipEventCreate(&e);
    return e;
}

static void resolve_stamps(fdn_ctx* h)
{
    if (h->stamps.empty()) return;
    (void)hipStreamSynchronize(h->stream);
    for (hipStream_t s : h->aux_stream) if (s) (void)hipStreamSynchronize(s);     // stamps of a sub-batch were recorded there
    for (auto& s : h->stamps) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { h->tms[s.which] += ms; h->tcount[s.which]++; }
        h->ev_pool.push_back(s.a);
        h->ev_pool.push_back(s.b);
    }
    h->stamps.clear();
}

// ---- FDN_GUARD_ALLOC=1: a debugging allocator (never the default) ---------------------------------------------------------
// Every device buffer the library allocates -- its own workspaces and what fdn_malloc hands out -- is placed through HIP's
// virtual-memory API so that the bytes right after its end (and the pages before its mapping) are NOT mapped: a kernel that
// reads or writes past the end of a buffer then faults at once and reproducibly, instead of silently touching a neighbour
// (there is no GPU AddressSanitizer on this pool).  A clean run of the parity tests under it is evidence that no kernel
// addresses beyond its operands (profiles/history/NOTES_r06.md, section 8).  Buffers end 16-byte aligned at the mapping's end.
// FDN_GUARD_ALLOC=2 / 3 additionally fill a fresh block with 0xFF bytes (NaNs) / zeros: a result that depends on the fill is a
// read of memory nobody wrote.  FDN_GUARD_ONLY=<names> guards only the named buffers.
struct GuardBlock { void* va; size_t va_bytes; size_t map_bytes; hipMemGenericAllocationHandle_t handle; };
static std::mutex g_guard_mu;
static std::map<void*, GuardBlock> g_guard;
static bool guard_mode()
{
    static const bool on = [] { const char* e = getenv("FDN_GUARD_ALLOC"); return e && atoi(e) != 0; }();
    return on;
}
static hipError_t guard_malloc(void** out, size_t bytes, int device)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
    if (e != hipSuccess) return e;
    if (!gran) gran = 4096;
    const size_t need = (std::max<size_t>(bytes, 1) + 15) & ~(size_t)15;
    GuardBlock g;
    g.map_bytes = (need + gran - 1) / gran * gran;
    g.va_bytes = g.map_bytes + 2 * gran;
    if ((e = hipMemAddressReserve(&g.va, g.va_bytes, gran, nullptr, 0)) != hipSuccess) return e;
    if ((e = hipMemCreate(&g.handle, g.map_bytes, &prop, 0)) != hipSuccess) { (void)hipMemAddressFree(g.va, g.va_bytes); return e; }
    char* base = (char*)g.va + gran;                       // one unmapped granule before, one after
    if ((e = hipMemMap(base, g.map_bytes, 0, g.handle, 0)) != hipSuccess) { (void)hipMemRelease(g.handle); (void)hipMemAddressFree(g.va, g.va_bytes); return e; }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if ((e = hipMemSetAccess(base, g.map_bytes, &acc, 1)) != hipSuccess) { (void)hipMemUnmap(base, g.map_bytes); (void)hipMemRelease(g.handle); (void)hipMemAddressFree(g.va, g.va_bytes); return e; }
    // FDN_GUARD_ALLOC=2: the fresh mapping filled with 0xFF bytes (float NaNs), 3: with zeros -- a result that depends on the
    // fill is a read of memory nobody wrote
    {
        const char* ev = getenv("FDN_GUARD_ALLOC");
        const int mode = ev ? atoi(ev) : 0;
        if (mode == 2 || mode == 3) { (void)hipMemset(base, mode == 2 ? 0xFF : 0, g.map_bytes); (void)hipDeviceSynchronize(); }
    }
    *out = base + (g.map_bytes - need);                    // the buffer ENDS where the mapping ends
    std::lock_guard<std::mutex> lk(g_guard_mu);
    g_guard[*out] = g;
    return hipSuccess;
}
// FDN_GUARD_ONLY=<name>[,<name>...]: guard only these buffers (R, flow, stack, sweep_out, vol_a, vol_b, partials, pair, vol_in,
// vol_out, Rpyr, flow_pyr, pyr_tmp, area_tab, M0, M1, sh, user = fdn_malloc): to find which placement a failure depends on
static bool guard_wanted(const char* name)
{
    const char* only = getenv("FDN_GUARD_ONLY");
    if (!only || !*only) return true;
    const size_t n = strlen(name);
    for (const char* p = only; (p = strstr(p, name)) != nullptr; p += n)
        if ((p == only || p[-1] == ',') && (p[n] == 0 || p[n] == ',')) return true;
    return false;
}
static hipError_t dev_malloc(void** out, size_t bytes, int device, const char* name = "user")
{
    return guard_mode() && guard_wanted(name) ? guard_malloc(out, bytes, device) : hipMalloc(out, bytes);
}
static hipError_t dev_free(void* p)
{
    if (!guard_mode() || !p) return hipFree(p);
    GuardBlock g;
    {
        std::lock_guard<std::mutex> lk(g_guard_mu);
        auto it = g_guard.find(p);
        if (it == g_guard.end()) return hipFree(p);
        g = it->second;
        g_guard.erase(it);
    }
    (void)hipDeviceSynchronize();
    // A guarded block is never unmapped (its memory is leaked; guard mode is for small test volumes): on this stack, unmapping and
    // releasing a mapping and then mapping new memory at the address range the runtime hands out next leaves the GPU with the OLD
    // translation -- every result after the first re-allocation of a workspace was garbage, and none with this early return
    // (round 6, tools/scratch/guard_case.py; FDN_GUARD_FREE=1 restores the unmapping to show it).
    { const char* e = getenv("FDN_GUARD_FREE"); if (!e || !atoi(e)) return hipSuccess; }
    size_t gran = (g.va_bytes - g.map_bytes) / 2;
    (void)hipMemUnmap((char*)g.va + gran, g.map_bytes);
    (void)hipMemRelease(g.handle);
    return hipMemAddressFree(g.va, g.va_bytes);
}

static int ensure(fdn_ctx* h, DevBuf& b, size_t bytes)
{
    // under a workspace limit a buffer is also given back when it is more than a quarter (and more than 16 MB, or
    // 1/256 of a smaller limit) too large, so that what an earlier, differently shaped call left behind does not count against the limit for ever.
    // (The 16 MB keep the small buffers that one job asks for at alternating sizes -- reduction partials, the pair
    // operators' scratch -- from being freed and reallocated, with a stream synchronisation, on every call.)
    // Without a limit a buffer is given back only when it is grossly too large (more than four times and more than 1 GiB over):
    // a handle that once filtered a 100 GB job must not sit on that memory while it filters small volumes -- other handles,
    // the runtime's own allocations (and, in tests, PyTorch) live on the same device.
    const bool oversized = h->ws_limit ? b.cap > bytes + bytes / 4 + std::min<size_t>((size_t)16 << 20, h->ws_limit >> 8)
                                       : (b.cap / 4 > bytes && b.cap - bytes > ((size_t)1 << 30));
    if (b.cap >= bytes && !oversized) return 0;
    if (b.p) FDN_HIP(hipStreamSynchronize(h->stream));
    FDN_DEVICE_WIDE;
    if (b.p) {
        FDN_HIP(dev_free(b.p));
        b.p = nullptr; b.cap = 0;
    }
    const char* name = &b == &h->R ? "R" : &b == &h->flow ? "flow" : &b == &h->stack ? "stack" : &b == &h->sweep_out ? "sweep_out" : &b == &h->vol_a ? "vol_a"
                     : &b == &h->vol_b ? "vol_b" : &b == &h->partials ? "partials" : &b == &h->pair ? "pair" : &b == &h->vol_in ? "vol_in" : &b == &h->vol_out ? "vol_out"
                     : &b == &h->Rpyr ? "Rpyr" : &b == &h->flow_pyr ? "flow_pyr" : &b == &h->pyr_tmp ? "pyr_tmp" : &b == &h->area_tab ? "area_tab" : &b == &h->M0 ? "M0"
                     : &b == &h->M1 ? "M1" : "sh";
    hipError_t e = dev_malloc(&b.p, bytes, h->device, name);
    if (e != hipSuccess) { b.p = nullptr; return fail("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e)); }
    b.cap = bytes;
    return 0;
}

static void release(DevBuf& b)
{
    FDN_DEVICE_WIDE;
    if (b.p) (void)dev_free(b.p);
    b.p = nullptr; b.cap = 0;
}

static int ensure_pinned(fdn_ctx* h, size_t bytes)
{
    if (h->pinned_cap >= bytes) return 0;
    FDN_HIP(hipStreamSynchronize(h->stream));
    FDN_DEVICE_WIDE;
    if (h->pinned) { FDN_HIP(hipHostFree(h->pinned)); h->pinned = nullptr; h->pinned_cap = 0; }
    hipError_t e = hipHostMalloc(&h->pinned, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { h->pinned = nullptr; return fail("hipHostMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e)); }
    h->pinned_cap = bytes;
    return 0;
}

// ---- copies between CALLER-OWNED host memory and the device -----------------------------------------------------------
// The library never hands pageable host memory of its callers (or of its own short-lived std::vectors) to the copy engines.
// For such a copy above about 1 MB the HIP runtime page-locks the pages ON THE FLY (hsa_amd_memory_lock_to_pool on the
// page-rounded range, the engine then reads or writes the user's memory directly) and keeps that registration after the
// call; it is not told when the application frees or trims that memory (numpy arrays, the heap's top, thread arenas), and a
// later copy that meets a registration whose pages are gone ends in "Memory access fault by GPU ... Reason: Unknown" and an
// abort of the process -- the abort that ended 6 of 19 long test sessions in round 5 and was caught with this message in
// round 6, in a plain 1.09 MB fdn_memcpy_d2h (profiles/history/NOTES_r06.md, section 2).  So a copy here is one of three things:
//   * the memory is page-locked already (hipHostRegister by the caller -- the CLI, the out-of-core mode --, hipHostMalloc):
//     one DMA, as before;
//   * 8 MB and more: page-locked by the library for the duration of the call (hipHostRegister / hipHostUnregister: an
//     explicit registration is removed when the call returns; 20 ms per 2 GiB), one DMA at PCIe speed;
//   * otherwise, or when the registration is refused: through the handle's own page-locked bounce buffer in pieces of 4 MB
//     (one host memcpy more: 0.1 ms per MB).
// All of them are complete when the function returns (the callers of these entry points wait anyway).
static constexpr size_t BOUNCE_BYTES = (size_t)4 << 20;

// What has been page-locked THROUGH the library (fdn_host_register): start -> bytes.  (Asking the runtime instead --
// hipPointerGetAttributes -- works too, but it logs an error for every pageable pointer it is asked about.)  Memory a caller
// has page-locked by other means is simply treated as pageable: bounced, which is always safe.
static std::mutex g_locked_mu;
static std::map<uintptr_t, size_t> g_locked;

static bool host_is_locked(const void* p, size_t bytes)
{
    if (!bytes) return true;
    std::lock_guard<std::mutex> g(g_locked_mu);
    auto it = g_locked.upper_bound((uintptr_t)p);
    if (it == g_locked.begin()) return false;
    --it;
    return (uintptr_t)p + bytes <= it->first + it->second;
}

static int ensure_bounce(fdn_ctx* h, size_t bytes = BOUNCE_BYTES)
{
    if (h->bounce_cap >= bytes) return 0;
    FDN_DEVICE_WIDE;
    if (h->bounce) { FDN_HIP(hipStreamSynchronize(h->stream)); FDN_HIP(hipHostFree(h->bounce)); h->bounce = nullptr; h->bounce_cap = 0; }
    hipError_t e = hipHostMalloc(&h->bounce, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { h->bounce = nullptr; return fail("hipHostMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e)); }
    h->bounce_cap = bytes;
    for (hipEvent_t& ev : h->bounce_ev) if (!ev) FDN_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    return 0;
}

// a host-side copy on up to four threads (one core moves about 10 GB/s, a PCIe DMA 57)
static void host_copy(void* dst, const void* src, size_t n)
{
    const unsigned nt = n >= ((size_t)8 << 20) ? std::min(4u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
    if (nt <= 1) { memcpy(dst, src, n); return; }
    std::vector<std::thread> pool;
    const size_t per = (n / nt + 4095) & ~(size_t)4095;
    for (unsigned t = 0; t < nt; t++) {
        const size_t o = std::min(n, (size_t)t * per), e = std::min(n, o + per);
        if (o < e) pool.emplace_back([=] { memcpy((char*)dst + o, (const char*)src + o, e - o); });
    }
    for (auto& th : pool) th.join();
}

// host <-> device, contiguous; returns after the copy has completed
static int copy_host(fdn_ctx* h, void* dst, const void* src, size_t bytes, bool to_device)
{
    if (!bytes) return 0;
    hipStream_t st = h->stream;
    void* host = to_device ? const_cast<void*>(src) : dst;
    const hipMemcpyKind kind = to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
    bool locked_here = false;
    if (!host_is_locked(host, bytes) && bytes >= ((size_t)8 << 20)) {
        if (hipHostRegister(host, bytes, hipHostRegisterDefault) == hipSuccess) locked_here = true;
        else (void)hipGetLastError();          // refused (a read-only mapping, the locked-memory limit): the bounce buffer below
    }
    if (locked_here || host_is_locked(host, bytes)) {
        hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (locked_here) (void)hipHostUnregister(host);
        if (e != hipSuccess) return fail("hipMemcpyAsync (%zu bytes, %s): %s", bytes, to_device ? "host to device" : "device to host", hipGetErrorString(e));
        return 0;
    }
    // Two halves of the bounce buffer in flight: the host-side copy of one piece overlaps the DMA of the other (a large
    // transfer takes 32 MB pieces and copies each on four threads: a 2 GiB volume that cannot be page-locked -- a read-only
    // mapping of a file, say -- moves at the host copy's 30-40 GB/s instead of one core's 10).
    const size_t piece = bytes >= ((size_t)64 << 20) ? (size_t)32 << 20 : BOUNCE_BYTES;
    if (ensure_bounce(h, 2 * piece)) return -1;
    const size_t np = (bytes + piece - 1) / piece;
    auto buf = [&](size_t i) { return (char*)h->bounce + (i & 1) * piece; };
    auto len = [&](size_t i) { return std::min(piece, bytes - i * piece); };
    if (to_device) {
        for (size_t i = 0; i < np; i++) {
            if (i >= 2) FDN_HIP(hipEventSynchronize(h->bounce_ev[i & 1]));      // the engine has read this half
            host_copy(buf(i), (const char*)src + i * piece, len(i));
            FDN_HIP(hipMemcpyAsync((char*)dst + i * piece, buf(i), len(i), hipMemcpyHostToDevice, st));
            FDN_HIP(hipEventRecord(h->bounce_ev[i & 1], st));
        }
        FDN_HIP(hipStreamSynchronize(st));
    } else {
        for (size_t i = 0; i <= np; i++) {
            if (i < np) {
                FDN_HIP(hipMemcpyAsync(buf(i), (const char*)src + i * piece, len(i), hipMemcpyDeviceToHost, st));
                FDN_HIP(hipEventRecord(h->bounce_ev[i & 1], st));
            }
            if (i >= 1) {                                                        // the previous piece has arrived: hand it over
                FDN_HIP(hipEventSynchronize(h->bounce_ev[(i - 1) & 1]));
                host_copy((char*)dst + (i - 1) * piece, buf(i - 1), len(i - 1));
            }
        }
    }
    return 0;
}

// the strided form: `height` rows of `width` bytes; the host side's pitch is hpitch, the device side's dpitch
static int copy_host_2d(fdn_ctx* h, void* dev, size_t dpitch, void* host, size_t hpitch, size_t width, size_t height, bool to_device)
{
    if (!width || !height) return 0;
    hipStream_t st = h->stream;
    const size_t span = (height - 1) * hpitch + width;
    if (host_is_locked(host, span)) {
        if (to_device) FDN_HIP(hipMemcpy2DAsync(dev, dpitch, host, hpitch, width, height, hipMemcpyHostToDevice, st));
        else FDN_HIP(hipMemcpy2DAsync(host, hpitch, dev, dpitch, width, height, hipMemcpyDeviceToHost, st));
        FDN_HIP(hipStreamSynchronize(st));
        return 0;
    }
    if (ensure_bounce(h)) return -1;
    if (width > BOUNCE_BYTES) {                 // rows longer than the buffer: row by row, each as a contiguous copy
        for (size_t r = 0; r < height; r++) {
            if (to_device ? copy_host(h, (char*)dev + r * dpitch, (const char*)host + r * hpitch, width, true)
                          : copy_host(h, (char*)host + r * hpitch, (const char*)dev + r * dpitch, width, false)) return -1;
        }
        return 0;
    }
    const size_t rows = std::max<size_t>(1, BOUNCE_BYTES / width);      // rows per piece, packed in the buffer
    for (size_t r0 = 0; r0 < height; r0 += rows) {
        const size_t nr = std::min(rows, height - r0);
        if (to_device) {
            for (size_t r = 0; r < nr; r++) memcpy((char*)h->bounce + r * width, (const char*)host + (r0 + r) * hpitch, width);
            FDN_HIP(hipMemcpy2DAsync((char*)dev + r0 * dpitch, dpitch, h->bounce, width, width, nr, hipMemcpyHostToDevice, st));
            FDN_HIP(hipStreamSynchronize(st));
        } else {
            FDN_HIP(hipMemcpy2DAsync(h->bounce, width, (const char*)dev + r0 * dpitch, dpitch, width, nr, hipMemcpyDeviceToHost, st));
            FDN_HIP(hipStreamSynchronize(st));
            for (size_t r = 0; r < nr; r++) memcpy((char*)host + (r0 + r) * hpitch, (const char*)h->bounce + r * width, width);
        }
    }
    return 0;
}

// dst (contiguous H x W) <- the view src[r * rs + c * cs] of host memory
static void gather_host(float* dst, const float* src, ptrdiff_t rs, ptrdiff_t cs, int H, int W)
{
    for (int r = 0; r < H; r++) {
        const float* s = src + (ptrdiff_t)r * rs;
        float* d = dst + (size_t)r * W;
        if (cs == 1) memcpy(d, s, (size_t)W * sizeof(float));
        else for (int c = 0; c < W; c++) d[c] = s[(ptrdiff_t)c * cs];
    }
}

// the same from an image of another depth, converted like cv::Mat::convertTo(CV_32F) (float64 -> float32 rounds to nearest)
template <typename T, typename D = float> static void gather_host_as(D* dst, const void* src_, ptrdiff_t rs, ptrdiff_t cs, int H, int W)
{
    const T* src = (const T*)src_;
    for (int r = 0; r < H; r++) {
        const T* s = src + (ptrdiff_t)r * rs;
        D* d = dst + (size_t)r * W;
        for (int c = 0; c < W; c++) d[c] = (D)s[(ptrdiff_t)c * cs];
    }
}
static int gather_host_depth(float* dst, const void* src, int depth, ptrdiff_t rs, ptrdiff_t cs, int H, int W)
{
    switch (depth) {
    case FDN_DEPTH_F32: gather_host(dst, (const float*)src, rs, cs, H, W); return 0;
    case FDN_DEPTH_F64: gather_host_as<double>(dst, src, rs, cs, H, W); return 0;
    case FDN_DEPTH_I16: gather_host_as<int16_t>(dst, src, rs, cs, H, W); return 0;
    case FDN_DEPTH_U16: gather_host_as<uint16_t>(dst, src, rs, cs, H, W); return 0;
    case FDN_DEPTH_I8: gather_host_as<int8_t>(dst, src, rs, cs, H, W); return 0;
    case FDN_DEPTH_U8: gather_host_as<uint8_t>(dst, src, rs, cs, H, W); return 0;
    }
    return fail("unknown image depth %d (FDN_DEPTH_*)", depth);
}

static size_t owned_bytes(const fdn_ctx* h)
{
    const DevBuf* bufs[] = {&h->R, &h->M0, &h->M1, &h->flow, &h->stack, &h->sweep_out, &h->vol_a, &h->vol_b, &h->partials, &h->pair,
                            &h->vol_in, &h->vol_out, &h->Rpyr, &h->flow_pyr, &h->pyr_tmp, &h->area_tab,
                            &h->sh_send, &h->sh_recv, &h->sh_stack, &h->sh_out[0], &h->sh_out[1], &h->sh_tmp};
    size_t n = 0;
    for (const DevBuf* b : bufs) n += b->cap;
    return n;
}

static void free_all(fdn_ctx* h)
{
    DevBuf* bufs[] = {&h->R, &h->M0, &h->M1, &h->flow, &h->stack, &h->sweep_out, &h->vol_a, &h->vol_b, &h->partials, &h->pair,
                      &h->vol_in, &h->vol_out, &h->Rpyr, &h->flow_pyr, &h->pyr_tmp, &h->area_tab,
                      &h->sh_send, &h->sh_recv, &h->sh_stack, &h->sh_out[0], &h->sh_out[1], &h->sh_tmp};
    FDN_DEVICE_WIDE;
    for (DevBuf* b : bufs) { if (b->p) (void)dev_free(b->p); b->p = nullptr; b->cap = 0; }
    h->area_key = {0, 0, 0, 0};
}

struct ScopedTimer {
    fdn_ctx* h; int which; hipEvent_t a = nullptr;
    ScopedTimer(fdn_ctx* h_, int w) : h(h_), which(w)
    {
        if (!h->timers) return;
        a = get_event(h);
        (void)hipEventRecord(a, h->stream);
    }
    ~ScopedTimer()
    {
        if (!a) return;
        hipEvent_t b = get_event(h);
        (void)hipEventRecord(b, h->stream);
        h->stamps.push_back({a, b, which});
        if (h->stamps.size() > 4096) resolve_stamps(h);
    }
};

static int check_params(const fdn_sweep_params* p, int K)
{
    if (!p) return fail("params is NULL");
    if (K < 1 || (K & 1) == 0) return fail("kernel.size must be odd (seq:93), got %d", K);
    if (p->use_of) {
        if (p->winsize < 1 || p->winsize > 49) return fail("winsize must be in 1..49, got %d", p->winsize);
        if (p->poly_n < 1 || p->poly_n > 7) return fail("poly_n must be in 1..7, got %d", p->poly_n);
        if (p->iters < 0) return fail("iters must be >= 0");
        if (p->levels < 0) return fail("levels must be >= 0");
    }
    if (p->warp_mode < 0 || p->warp_mode > 3) return fail("warp_mode must be FDN_WARP_F32, _F64_PADDED, _ROUND_INT or _FIXED_U8, got %d", p->warp_mode);
    if (p->warp_mode == FDN_WARP_F64_PADDED && p->border_mode != FDN_BORDER_MEAN_PAD)
        return fail("FDN_WARP_F64_PADDED belongs to the mean-padded volume of seq:88-89 (border_mode FDN_BORDER_MEAN_PAD)");
    if (p->warp_mode == FDN_WARP_F64_PADDED && (p->pad_lo < 0 || p->pad_hi < 0)) return fail("pad_lo / pad_hi must be >= 0");
    if (p->warp_mode == FDN_WARP_ROUND_INT && !(p->round_lo < p->round_hi)) return fail("FDN_WARP_ROUND_INT needs round_lo < round_hi");
    return 0;
}

// fdn_sweep_params -> what the folding kernels need; pad slices in the coordinates of a stack of S + 2r slices
static WarpMode warp_mode_of(const fdn_ctx* h, const fdn_sweep_params* p, int S, int r)
{
    WarpMode wm;
    wm.model = h->tn.remap_model;
    wm.kind = p->warp_mode;
    if (wm.kind == FDN_WARP_FIXED_U8) { wm.kind = FDN_WARP_ROUND_INT; wm.fixed8 = 1; wm.lo = 0.f; wm.hi = 255.f; return wm; }   // the ROUND_INT kernels, their remap in 8-bit fixed point
    if (wm.kind == FDN_WARP_F64_PADDED) { wm.pad_lo = p->pad_lo; wm.pad_hi = S + 2 * r - p->pad_hi; wm.pad64 = p->pad64; }
    if (wm.kind == FDN_WARP_ROUND_INT) { wm.lo = (float)p->round_lo; wm.hi = (float)p->round_hi; }
    return wm;
}

// number of pyramid levels OpenCV actually uses (calc(): min_size = 32)
static int effective_levels(int levels, int H, int W)
{
    int k = 0;
    double scale = 1;
    for (; k < levels; k++) {
        scale *= 0.5;
        if (W * scale < 32 || H * scale < 32) break;
    }
    return k;
}

// ---- pyramid (levels > 0): FarnebackOpticalFlowImpl::calc's per-level geometry ----------
struct PyrLevel { int h, w, smooth_sz; double sigma, scale; size_t r_off; size_t f_off; };

static std::vector<PyrLevel> pyramid_levels(int levels, int H, int W)
{
    int L = effective_levels(levels, H, W);
    std::vector<PyrLevel> v(L + 1);
    for (int k = 0; k <= L; k++) {
        double scale = 1;
        for (int i = 0; i < k; i++) scale *= 0.5;
        double sigma = (1. / scale - 1) * 0.5;
        int smooth = (int)lrint(sigma * 5) | 1;
        v[k] = PyrLevel{(int)lrint(H * scale), (int)lrint(W * scale), smooth < 3 ? 3 : smooth, sigma, scale, 0, 0};
    }
    return v;
}

// cv::computeResizeAreaTab in CSR form (entries of output index d are [start[d], start[d+1]))
static void area_table(int ssize, int dsize, std::vector<int>& si, std::vector<float>& alpha, std::vector<int>& start)
{
    double scale = (double)ssize / dsize;
    si.clear(); alpha.clear(); start.assign(1, 0);
    for (int d = 0; d < dsize; d++) {
        double f1 = d * scale, f2 = f1 + scale;
        double cell = std::min(scale, ssize - f1);
        int s1 = (int)ceil(f1), s2 = (int)floor(f2);
        s2 = std::min(s2, ssize - 1);
        s1 = std::min(s1, s2);
        if (s1 - f1 > 1e-3) { si.push_back(s1 - 1); alpha.push_back((float)((s1 - f1) / cell)); }
        for (int q = s1; q < s2; q++) { si.push_back(q); alpha.push_back((float)(1.0 / cell)); }
        if (f2 - s2 > 1e-3) { si.push_back(s2); alpha.push_back((float)(std::min(std::min(f2 - s2, 1.), cell) / cell)); }
        start.push_back((int)si.size());
    }
}

static int ensure_area_tables(fdn_ctx* h, int sh, int sw, int dh, int dw)
{
    if (h->area_key.sh == sh && h->area_key.sw == sw && h->area_key.dh == dh && h->area_key.dw == dw) return 0;
    std::vector<int> xsi, xst, ysi, yst;
    std::vector<float> xa, ya;
    area_table(sw, dw, xsi, xa, xst);
    area_table(sh, dh, ysi, ya, yst);
    size_t n = xsi.size() + xa.size() + xst.size() + ysi.size() + ya.size() + yst.size();
    if (ensure(h, h->area_tab, n * 4)) return -1;
    FDN_HIP(hipStreamSynchronize(h->stream));
    char* d = (char*)h->area_tab.p;
    auto put = [&](const void* src, size_t cnt) -> const void* {
        const void* at = d;
        (void)copy_host(h, d, src, cnt * 4, true);
        d += cnt * 4;
        return at;
    };
    h->area.x_si = (const int*)put(xsi.data(), xsi.size());
    h->area.x_alpha = (const float*)put(xa.data(), xa.size());
    h->area.x_start = (const int*)put(xst.data(), xst.size());
    h->area.y_si = (const int*)put(ysi.data(), ysi.size());
    h->area.y_alpha = (const float*)put(ya.data(), ya.size());
    h->area.y_start = (const int*)put(yst.data(), yst.size());
    h->area_key = {sh, sw, dh, dw};
    return 0;
}

// cv::resize as calc() uses it, on nimg device images of cn interleaved channels
static int resize_dev(fdn_ctx* h, const float* in, int sh, int sw, float* out, int dh, int dw, int cn, int nimg,
                      int interp, bool apply_ps, double ps)
{
    if (resize_needs_tables(sh, sw, dh, dw, interp)) {
        if (ensure_area_tables(h, sh, sw, dh, dw)) return -1;
        resize_area_tab(in, sh, sw, out, dh, dw, cn, nimg, h->area.x_si, h->area.x_alpha, h->area.x_start,
                        h->area.y_si, h->area.y_alpha, h->area.y_start, apply_ps, ps, h->stream);
    } else {
        resize_images(in, sh, sw, out, dh, dw, cn, nimg, interp, apply_ps, ps, h->stream, h->tn.fma);
    }
    return 0;
}

// R_k[s] = polyexp(resize(GaussianBlur(img[s], smooth_k, sigma_k), (w_k, h_k), INTER_LINEAR)) for levels
// k >= 1 of every image; lv[k].r_off (floats) locates level k inside h->Rpyr.
static int build_R_pyramid(fdn_ctx* h, const float* imgs, int nimg, int H, int W, std::vector<PyrLevel>& lv, const PolyConsts& pc)
{
    const size_t HW = (size_t)H * W;
    size_t total = 0;
    for (size_t k = 1; k < lv.size(); k++) { lv[k].r_off = total; total += (size_t)nimg * 5 * lv[k].h * lv[k].w; }
    if (ensure(h, h->Rpyr, total * sizeof(float))) return -1;
    const size_t tmp_cap = h->ws_limit ? (size_t)1 << 28 : (size_t)1 << 30;
    const int chunk = std::max(1, std::min(nimg, (int)(tmp_cap / (HW * 12))));
    if (ensure(h, h->pyr_tmp, (size_t)chunk * HW * 3 * sizeof(float))) return -1;
    if (h->reserve_only) return 0;
    float* tmp = (float*)h->pyr_tmp.p;
    float* blurred = tmp + (size_t)chunk * HW;
    float* small = blurred + (size_t)chunk * HW;
    ScopedTimer t(h, FDN_TIMER_POLYEXP);
    for (size_t k = 1; k < lv.size(); k++) {
        trace_point(h, "build_R_pyramid level %zu: %d images %dx%d -> %dx%d, blur %d taps", k, nimg, W, H, lv[k].w, lv[k].h, lv[k].smooth_sz);
        if (lv[k].smooth_sz > FDN_MAX_BLUR_TAPS) return fail("pyramid level %zu needs a %d-tap blur (max %d)", k, lv[k].smooth_sz, FDN_MAX_BLUR_TAPS);
        BlurTaps bt;
        prepare_blur_taps(lv[k].smooth_sz, lv[k].sigma, &bt);
        for (int s0 = 0; s0 < nimg; s0 += chunk) {
            int n = std::min(chunk, nimg - s0);
            if (lv[k].h < H && lv[k].w < W) {     // always, for a pyramid level: blur only where the resize reads
                launch_blur_resize(imgs + (size_t)s0 * HW, tmp, small, n, H, W, lv[k].h, lv[k].w, bt, h->stream, h->tn.fma);
            } else {
                launch_gaussian_blur(imgs + (size_t)s0 * HW, tmp, blurred, n, H, W, bt, h->stream, h->tn.fma);
                if (resize_dev(h, blurred, H, W, small, lv[k].h, lv[k].w, 1, n, 1, false, 1.0)) return -1;
            }
            launch_polyexp(small, (float*)h->Rpyr.p + lv[k].r_off + (size_t)s0 * 5 * lv[k].h * lv[k].w, n, lv[k].h, lv[k].w, pc, h->stream);
        }
    }
    return 0;
}

// Farneback level-0 iterations for a batch of pairs whose R planes are in Rstack and whose
// flows (initial -> final) are in `flow`; M0/M1 are ping-pong scratch for npairs.
// flow_in: the initial flow (nullptr = zero); flow: receives the result; the two may be the same buffer (cv2 updates its
// flow in place, seq:98) or different ones (the sweeps keep every step's flow for the one-launch sweep of a side).
static int run_iterations(fdn_ctx* h, const float* Rstack, const float* flow_in, float* flow, float* M0, float* M1, PairBatch pb,
                          int H, int W, int winsize, int iters)
{
    if (h->tn.strict_order && !strict_order_supported(W, winsize))
        return fail("strict order was requested, but a row of %d columns (winsize %d) does not fit the 160 KB of LDS "
                    "its serial running sum needs; strict mode never falls back silently", W, winsize);
    trace_point(h, "run_iterations %dx%d npairs %d t0 %d d %d winsize %d: update_matrices", W, H, pb.npairs, pb.t0, pb.d, winsize);
    {
        ScopedTimer t(h, FDN_TIMER_UPDATE_MATRICES);
        launch_update_matrices(Rstack, flow_in, M0, pb, H, W, h->stream);
    }
    if (iters <= 0 && flow_in != flow) {       // no iteration: the flow is its initial value
        const size_t bytes = (size_t)pb.npairs * H * W * 2 * sizeof(float);
        if (flow_in) FDN_HIP(hipMemcpyAsync(flow, flow_in, bytes, hipMemcpyDeviceToDevice, h->stream));
        else launch_fill(flow, 0.f, (size_t)pb.npairs * H * W * 2, h->stream);
    }
    float* cur = M0; float* nxt = M1;
    for (int it = 0; it < iters; it++) {
        bool update = it < iters - 1;
        trace_point(h, "  update_flow it %d (%s)", it, h->tn.strict_order ? "strict" : "scan");
        ScopedTimer t(h, FDN_TIMER_UPDATE_FLOW);
        if (h->tn.strict_order) {
            if (!launch_update_flow_strict(Rstack, cur, update ? nxt : nullptr, flow, pb, H, W, winsize, h->stream))
                return fail("strict order: the serial running-sum kernel could not be launched for rows of %d columns", W);
        } else
            launch_update_flow(Rstack, cur, update ? nxt : nullptr, flow, pb, H, W, winsize, h->stream);
        std::swap(cur, nxt);
    }
    return 0;
}

// All levels of calc() for a batch of pairs: coarsest flow = INTER_AREA shrink of the initial flow
// times the level scale (zeros without one), iterate, INTER_LINEAR to the next finer level times 2.
// flow_init (n x H x W x 2, nullptr = none) is the initial flow, flow_full receives the result (they may be one buffer).
static int pyramid_batch(fdn_ctx* h, const std::vector<PyrLevel>& lv, const float* R0, const float* flow_init, float* flow_full,
                         float* M0, float* M1, PairBatch pb, int H, int W, int winsize, int iters)
{
    const bool has_initial = flow_init != nullptr;
    const int L = (int)lv.size() - 1;
    float* fp = (float*)h->flow_pyr.p;
    const int n = pb.npairs;
    float* fl = fp + lv[L].f_off;
    trace_point(h, "pyramid_batch %dx%d levels %d n %d initial %d", W, H, L, n, (int)has_initial);
    if (has_initial) {
        if (resize_dev(h, flow_init, H, W, fl, lv[L].h, lv[L].w, 2, n, 3, true, lv[L].scale)) return -1;
    } else {
        launch_fill(fl, 0.f, (size_t)n * lv[L].h * lv[L].w * 2, h->stream);
    }
    for (int k = L; k >= 0; k--) {
        float* cur = k == 0 ? flow_full : fp + lv[k].f_off;
        trace_point(h, " level %d: %dx%d%s", k, lv[k].w, lv[k].h, k < L ? ", upsample the coarser flow" : "");
        if (k < L)
            if (resize_dev(h, fp + lv[k + 1].f_off, lv[k + 1].h, lv[k + 1].w, cur, lv[k].h, lv[k].w, 2, n, 1, true, 2.0)) return -1;
        const float* Rk = k == 0 ? R0 : (const float*)h->Rpyr.p + lv[k].r_off;
        if (run_iterations(h, Rk, cur, cur, M0, M1, pb, lv[k].h, lv[k].w, winsize, iters)) return -1;
    }
    return 0;
}

// `copies` flow buffers per level k >= 1 (the fused kernel cannot update a flow in place: the bands of
// one pair run at their own pace and read each other's halo columns)
// lays `copies` flow buffers of n images per level k >= 1 out from float offset `base` on; returns the end offset
static size_t flow_pyramid_layout(std::vector<PyrLevel>& lv, int n, int copies, size_t base)
{
    size_t total = base;
    for (size_t k = 1; k < lv.size(); k++) { lv[k].f_off = total; total += (size_t)copies * n * lv[k].h * lv[k].w * 2; }
    return total;
}
static int ensure_flow_pyramid(fdn_ctx* h, std::vector<PyrLevel>& lv, int n, int copies = 1)
{
    const size_t total = flow_pyramid_layout(lv, n, copies, 0);
    return ensure(h, h->flow_pyr, std::max<size_t>(total, 1) * sizeof(float));
}

// One chain step of a pyramid sweep on the fused kernel: calc()'s levels, coarsest first, each
// one launch; the finest level also warps the neighbour and accumulates.  Every level but the
// coarsest reads the next coarser level's flow and upsamples it on the fly (INTER_LINEAR x 2).
// `prev` (n x H x W x 2) is the previous step's flow or nullptr; out0 receives this step's (or nullptr).
static int pyramid_step_fused(fdn_ctx* h, const std::vector<PyrLevel>& lv, const float* R0, const float* stack, const float* prev,
                              float* out0, float* acc, PairBatch pb, int H, int W, int winsize, int iters, double weight,
                              const WarpMode& wm = WarpMode())
{
    const int L = (int)lv.size() - 1;
    float* fp = (float*)h->flow_pyr.p;
    const int n = pb.npairs;
    const float* fin = nullptr;      // flow handed to the next launch
    int ch = 0, cw = 0;              // its size when it is a coarser level's
    if (prev) {
        float* a = fp + lv[L].f_off;
        ScopedTimer t(h, FDN_TIMER_PERMUTE);
        if (resize_dev(h, prev, H, W, a, lv[L].h, lv[L].w, 2, n, 3, true, lv[L].scale)) return -1;
        fin = a;
    }
    for (int k = L; k >= 1; k--) {
        float* b = fp + lv[k].f_off + (size_t)n * lv[k].h * lv[k].w * 2;
        ScopedTimer t(h, FDN_TIMER_FUSED);
        launch_farneback_fused((const float*)h->Rpyr.p + lv[k].r_off, nullptr, fin, b, nullptr, pb, lv[k].h, lv[k].w,
                               winsize, iters, 0.0, h->stream, h->tn, ch, cw);
        fin = b; ch = lv[k].h; cw = lv[k].w;
    }
    ScopedTimer t(h, FDN_TIMER_FUSED);
    launch_farneback_fused(R0, stack, fin, out0, acc, pb, H, W, winsize, iters, weight, h->stream, h->tn, ch, cw, wm);
    return 0;
}

// Which kernels a sweep runs on: 0 = the 3-iteration fused kernel, 1 = the one-iteration kernel, 2 = one kernel per stage
static int sweep_path(const fdn_ctx* h, const fdn_sweep_params* p, const std::vector<PyrLevel>& lv, int H, int W)
{
    // the unquantised remap model on an integer volume's own semantics: the per-stage kernels (k_sweep_side takes the model at
    // run time; the Farneback kernels are built with it for float32 volumes only -- fdn_device.h, fold_warped)
    if (h->tn.remap_model == 1 && p->warp_mode != FDN_WARP_F32) return 2;
    bool fused = fused_supported(p->winsize, p->iters, H, W) && h->tn.path == 0 && !h->tn.strict_order;
    for (size_t k = 1; k < lv.size(); k++) fused = fused && fused_supported(p->winsize, p->iters, lv[k].h, lv[k].w);
    if (fused) return 0;
    // windows the 3-iteration kernel does not cover (winsize >= 10): one launch per iteration, matrices in LDS
    bool iter = p->iters >= 1 && h->tn.path != 1 && !h->tn.strict_order && iter_supported(p->winsize, H, W);
    for (size_t k = 1; k < lv.size(); k++) iter = iter && iter_supported(p->winsize, lv[k].h, lv[k].w);
    return iter ? 1 : 2;
}
// bytes of flows (and matrices) per pixel of a target slice on that path: fused / iter: two buffers (16 B), with a pyramid
// two more per coarser level (22); per-stage: one flow per step of a side (8 r B) + two M sets (40 B)
static size_t sweep_flow_px(int path, int r, bool pyramid)
{
    return path < 2 ? (pyramid ? 22 : 16) : (size_t)8 * std::max(r, 1) + 40 + (pyramid ? 4 : 0);
}

// One chain step on the one-iteration kernels (fdn_iter.hip): calc()'s levels, coarsest first, `iters` launches each;
// the last launch of level 0 also warps the neighbour and accumulates.  bufs: two full-resolution flow buffers; `prev`
// (one of them, or nullptr) holds the previous step's flow.  *result = where this step's flow is (when keep).
static int chain_step_iter(fdn_ctx* h, const std::vector<PyrLevel>& lv, const float* R0, const float* stack, const float* prev,
                           float* const bufs[2], float* acc, PairBatch pb, int H, int W, int winsize, int iters, double weight,
                           bool keep, float** result, const WarpMode& wm = WarpMode())
{
    const int L = (int)lv.size() - 1;
    float* fp = (float*)h->flow_pyr.p;
    const int n = pb.npairs;
    const float* fin = prev;         // flow handed to the next launch
    int ch = 0, cw = 0;              // its size when it is a coarser level's
    if (L > 0 && prev) {             // coarsest level starts from the INTER_AREA shrink of the previous flow, times its scale
        float* a = fp + lv[L].f_off;
        ScopedTimer t(h, FDN_TIMER_PERMUTE);
        if (resize_dev(h, prev, H, W, a, lv[L].h, lv[L].w, 2, n, 3, true, lv[L].scale)) return -1;
        fin = a;
    }
    for (int k = L; k >= 0; k--) {
        float* A = k ? fp + lv[k].f_off : bufs[0];
        float* B = k ? A + (size_t)n * lv[k].h * lv[k].w * 2 : bufs[1];
        const float* Rk = k ? (const float*)h->Rpyr.p + lv[k].r_off : R0;
        for (int it = 0; it < iters; it++) {
            const bool last = k == 0 && it == iters - 1;
            float* fout = fin == A ? B : A;
            ScopedTimer t(h, FDN_TIMER_ITER);
            if (launch_farneback_iter(Rk, stack, fin, last && !keep ? nullptr : fout, last ? acc : nullptr, pb, lv[k].h, lv[k].w,
                                      winsize, weight, h->stream, ch, cw, wm, h->tn.fma))
                return fail("k_farneback_iter could not be launched (winsize %d needs %zu bytes of LDS)", winsize, iter_lds_bytes(winsize / 2, true));
            fin = fout; ch = cw = 0;
        }
        ch = lv[k].h; cw = lv[k].w;   // the next (finer) level upsamples this one's result
    }
    *result = const_cast<float*>(fin);
    return 0;
}

static int ensure_aux_streams(fdn_ctx* h, int n_aux)
{
    FDN_DEVICE_WIDE;
    if (!h->ev_fork) FDN_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < n_aux; i++) {
        if (!h->aux_stream[i]) FDN_HIP(hipStreamCreateWithFlags(&h->aux_stream[i], hipStreamNonBlocking));
        if (!h->ev_join[i]) FDN_HIP(hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
    }
    return 0;
}

// While a batch runs as sub-batches h->stream points at the stream of the sub-batch being enqueued (every helper launches
// on h->stream).  Leaving the scope -- normally or through an error return -- puts the handle's stream back, and if the
// second stream has not been joined (an error on the way) waits for it: nothing of the handle may be freed under it.
struct SubBatchScope {
    fdn_ctx* h; hipStream_t main; int n_aux; bool joined = false;
    SubBatchScope(fdn_ctx* h_, int n_aux_) : h(h_), main(h_->stream), n_aux(n_aux_) {}
    int fork()
    {
        FDN_HIP(hipEventRecord(h->ev_fork, main));
        for (int i = 0; i < n_aux; i++) FDN_HIP(hipStreamWaitEvent(h->aux_stream[i], h->ev_fork, 0));
        return 0;
    }
    int join()
    {
        h->stream = main;
        h->tn.occ_blocks = 0;
        for (int i = 0; i < n_aux; i++) {
            FDN_HIP(hipEventRecord(h->ev_join[i], h->aux_stream[i]));
            FDN_HIP(hipStreamWaitEvent(main, h->ev_join[i], 0));
        }
        joined = true;
        return 0;
    }
    ~SubBatchScope()
    {
        h->stream = main;
        h->tn.occ_blocks = 0;
        if (n_aux > 0 && !joined) for (int i = 0; i < n_aux; i++) (void)hipStreamSynchronize(h->aux_stream[i]);
    }
};

// Sub-batches of a batch of n target slices: 1 or 2.  Automatic (tn.sub_batches == 0): two when some launch of a chain step
// -- at any pyramid level -- would fill the GPU's workgroup slots fewer than four times over (the tail of a launch is about
// half a round: under four rounds it is more than a tenth of the launch).  Slots: 4 workgroups per CU for the 3-iteration
// kernel, 6 for the one-iteration kernel (their LDS footprints at the usual windows).  The 10-round launches of BASELINE
// configs[2] (10 240 workgroups) stay on one stream.
static int sweep_sub_batches(const fdn_ctx* h, const fdn_sweep_params* p, const std::vector<PyrLevel>& lv, int path, int n, int H, int W)
{
    if (n < 2 || path > 1 || h->tn.sub_batches == 1) return 1;
    if (h->tn.sub_batches >= 2) return std::min(n, std::min(h->tn.sub_batches, (int)fdn_ctx::MAX_SUB));
    const int mh = p->winsize / 2;
    const int bw = path == 0 ? 64 - 6 * mh : 64 - 2 * mh;
    const long slots = (long)h->tn.cus * (path == 0 ? 4 : 6);
    // (the finest level decides: it is four fifths of a pyramid's work, and on a grid of many rounds two streams cost more --
    //  two launches interleaved on every XCD share its L2 -- than the coarse levels' short launches gain: round 6, -l 3 -w 15
    //  bench volume 1 548 ms on one stream, 1 567 on two)
    const long blocks = (long)((W + bw - 1) / bw) * n;
    (void)H; (void)lv;
    return blocks < 4 * slots ? 2 : 1;
}

static int sweep_stack(fdn_ctx* h, const float* stack, float* out, int S, int H, int W, const double* kernel, int K,
                       const fdn_sweep_params* p)
{
    if (check_params(p, K)) return -1;
    if (S <= 0 || H <= 0 || W <= 0) return fail("bad stack dims S=%d H=%d W=%d", S, H, W);
    const int r = K / 2;
    const size_t HW = (size_t)H * W;
    hipStream_t st = h->stream;
    if (p->warp_mode == FDN_WARP_F64_PADDED && (long)p->pad_lo + p->pad_hi > (long)S + 2 * r)
        return fail("pad_lo + pad_hi = %d + %d exceeds the stack's %d slices", p->pad_lo, p->pad_hi, S + 2 * r);
    const WarpMode wm_all = warp_mode_of(h, p, S, r);   // pad slices in the coordinates of the whole stack
    // (an integer volume's accumulate -- float64 padded volume / integer images -- is part of the Farneback kernels' final
    //  stage too: fold_warped<WM>; only the per-stage path folds with k_sweep_side)

    if (!p->use_of) { // seq:184-185: taps in index order
        if (h->reserve_only) return 0;
        ScopedTimer t(h, FDN_TIMER_WARP);
        launch_fill(out, 0.f, (size_t)S * HW, st);
        for (int i = 0; i < K; i++) launch_axpy_slices(stack, out, PairBatch{S, r, i - r}, H, W, kernel[i], st, wm_all);
        if (wm_all.kind == FDN_WARP_ROUND_INT) launch_trunc_clamp(out, (size_t)S * HW, wm_all.lo, wm_all.hi, st);
        FDN_HIP(hipGetLastError());
        return 0;
    }
    if (H < 2 || W < 2) return fail("optical flow needs images of at least 2x2 pixels, got %dx%d", W, H);
    // the kernels address the pixels of one image by 32-bit byte offsets (8 bytes per pixel pair plane)
    if (H >= (1 << 24) || W >= (1 << 24) || (size_t)H * W >= ((size_t)1 << 29))
        return fail("images of %dx%d pixels are not supported (limit: 2^29 pixels per image)", W, H);
    std::vector<PyrLevel> lv = pyramid_levels(p->levels, H, W);
    const bool pyramid = lv.size() > 1;

    PolyConsts pc;
    prepare_poly_consts(p->poly_n, p->poly_sigma, &pc);
    const int path = sweep_path(h, p, lv, H, W);
    const bool fused = path == 0, iter = path == 1;
    if (h->tn.path == 2 && !iter) return fail("path 2 (one-iteration kernels) cannot run winsize %d, iters %d here", p->winsize, p->iters);
    // Scratch of a batch of C target slices, per pixel: the flows (sweep_flow_px) and, when a workspace limit is set, the
    // polynomial expansions too (20 B per slice, 26.7 with a pyramid): R is then rebuilt per batch for its C + 2r slices.
    const size_t flow_px = sweep_flow_px(path, r, pyramid);
    const size_t r_px = pyramid ? 27 : 20;
    const bool limited = h->ws_limit != 0;
    bool rebuild_r = false;
    int C;
    if (limited) {
        const size_t other = owned_bytes(h) - (h->R.cap + h->Rpyr.cap + h->pyr_tmp.cap + h->flow.cap + h->M0.cap + h->M1.cap + h->flow_pyr.cap);
        const size_t fixed = (size_t)2 * r * HW * r_px + (pyramid ? std::min<size_t>((size_t)1 << 28, HW * 12 * (size_t)(S + 2 * r)) : 0);   // halo slices' R + blur scratch
        const size_t per_target = HW * (flow_px + r_px);
        if (h->ws_limit < other + fixed + per_target)
            return fail("workspace limit of %zu bytes is too small: a pass over %d x %d images with K = %d needs at least %zu "
                        "(%zu held by volumes and stacks)", h->ws_limit, W, H, K, other + fixed + per_target, other);
        C = (int)std::min<size_t>((size_t)S, (h->ws_limit - other - fixed) / per_target);
    } else {
        // No limit: the expansions of the whole pass are the larger buffers and do not depend on C: they come first,
        // and the flow batch is then sized from what is left after them (and after the pyramid's buffers, which
        // build_R_pyramid allocates further down).  If not even the expansions fit next to one target's flows, R is
        // rebuilt per batch as under a limit instead of failing in hipMalloc.
        size_t fre = 0, tot = 0;
        FDN_HIP(hipMemGetInfo(&fre, &tot));
        if (h->reserve_only) fre -= std::min(fre, h->reserve_extern);     // the caller's two whole-volume buffers come after the reservation
        auto grow = [](const DevBuf& b, size_t want) { return want > b.cap ? want : (size_t)0; };     // bytes ensure() will newly claim (the old block is freed first)
        const size_t r_all = (size_t)(S + 2 * r) * 5 * HW * sizeof(float);
        const size_t pyr_all = pyramid ? (size_t)(S + 2 * r) * HW * 7 + std::min<size_t>((size_t)1 << 30, HW * 12 * (size_t)(S + 2 * r)) : 0;
        const size_t pyr_have = h->Rpyr.cap + h->pyr_tmp.cap;
        const size_t have = h->flow.cap + h->M0.cap + h->M1.cap + h->flow_pyr.cap;
        // what this call may take: what is free plus what the handle would reuse, less a reserve that stays free whatever
        // happens -- 1/16 of the device and at least 6 GiB: the runtime's own allocations, other handles of the process
        // (the CLI's upload handle, a writer's page-locked mapping), a caller's volumes that come later
        const size_t keep_free = std::max<size_t>(tot / 16, (size_t)6 << 30);
        const size_t gross = fre + have + h->R.cap + std::min(pyr_have, pyr_all);
        const size_t avail = gross > keep_free ? gross - keep_free : 0;
        const size_t need_r = r_all + pyr_all;
        if (avail / 10 * 9 > need_r + HW * flow_px) {
            C = (int)std::min<size_t>((size_t)S, std::max<size_t>(1, (avail - need_r) / 10 * 8 / (HW * flow_px)));
            (void)grow;
        } else {       // the expansions of the whole stack do not fit: batches of C targets with their own R
            const size_t fixed = (size_t)2 * r * HW * r_px + (pyramid ? std::min<size_t>((size_t)1 << 30, HW * 12 * (size_t)(S + 2 * r)) : 0);
            const size_t per_target = HW * (flow_px + r_px);
            if (avail / 10 * 9 < fixed + per_target)
                return fail("not enough device memory for a pass over %d x %d images with K = %d: %zu bytes usable (%zu kept in reserve), %zu needed", W, H, K, avail, keep_free, fixed + per_target);
            C = (int)std::min<size_t>((size_t)S, (avail / 10 * 9 - fixed) / per_target);
            rebuild_r = true;
        }
    }
    const int CR = limited || rebuild_r ? C : S;              // target slices per rebuild of R
    if (ensure(h, h->R, (size_t)(CR + 2 * r) * 5 * HW * sizeof(float))) return -1;
    float* R = (float*)h->R.p;
    if (fused || iter) {
        if (ensure(h, h->flow, (size_t)C * HW * 16)) return -1;
        if (pyramid && ensure_flow_pyramid(h, lv, C, 2)) return -1;
    } else {
        if (ensure(h, h->flow, (size_t)C * HW * 8 * std::max(r, 1))) return -1;
        if (ensure(h, h->M0, (size_t)C * HW * 20)) return -1;
        if (ensure(h, h->M1, (size_t)C * HW * 20)) return -1;
        if (pyramid && ensure_flow_pyramid(h, lv, C)) return -1;
    }
    if (h->reserve_only) {           // the pyramid's buffers as well, then done: nothing is launched
        if (pyramid && build_R_pyramid(h, stack, std::min(CR, S) + 2 * r, H, W, lv, pc)) return -1;
        return 0;
    }
    float* flow = (float*)h->flow.p;
    float* flowB = flow + (size_t)C * HW * 2; // fused / iter only
    float* M0 = (float*)h->M0.p; float* M1 = (float*)h->M1.p;
    const float* const stack_all = stack;
    float* const out_all = out;

    for (int cr0 = 0; cr0 < S; cr0 += CR) {
    const int nr = std::min(CR, S - cr0);
    stack = stack_all + (size_t)cr0 * HW;        // the batch's own stack: target q is its slice q + r
    out = out_all + (size_t)cr0 * HW;
    WarpMode wm = wm_all;
    wm.pad_lo = std::max(0, wm_all.pad_lo - cr0); wm.pad_hi = wm_all.pad_hi - cr0;
    trace_point(h, "sweep_stack S %d %dx%d K %d path %d levels %zu: polyexp of %d slices", S, W, H, K, path, lv.size() - 1, nr + 2 * r);
    {
        ScopedTimer t(h, FDN_TIMER_POLYEXP);
        launch_blur3_polyexp(stack, R, nr + 2 * r, H, W, pc, st);
    }
    if (pyramid && build_R_pyramid(h, stack, nr + 2 * r, H, W, lv, pc)) return -1;
    for (int c0 = 0; c0 < nr; c0 += C) {
        int n = std::min(C, nr - c0);
        float* acc = out + (size_t)c0 * HW;
        launch_fill(acc, 0.f, (size_t)n * HW, st);
        if (fused || iter) {
            // The target slices of a batch are independent (seq:92: one output slice per loop trip): with sub-batches the
            // batch's targets run as two halves, each the complete chain of both sides on a stream of its own.  A launch
            // that fills the GPU's workgroup slots only a few times over ends in a tail of idle slots -- 1 280 workgroups
            // (the 64-slice Z-slab of an 8-GPU rank) are 1.25 rounds of 1 024 slots and cost two; the other half's launches
            // fill that tail (par's remainder round, src/flowdenoising.py:194-206, exists for the same reason: keep the
            // workers busy).  Same buffers, same kernels, disjoint slices of every buffer: the bits cannot change.
            const int G = sweep_sub_batches(h, p, lv, path, n, H, W);
            h->last_sub_batches = G;
            if (G > 1 && ensure_aux_streams(h, G - 1)) return -1;
            SubBatchScope scope(h, G - 1);
            ScopedTimer span(h, FDN_TIMER_CHAINS);                         // on the main stream: fork ... join
            if (G > 1) {
                if (scope.fork()) return -1;                               // R, the pyramid and the zeroed accumulators are ready
                long blocks = 0;                                           // the occupancy build is chosen for the whole batch's grid
                if (fused) blocks = (long)((W + (64 - 6 * (p->winsize / 2)) - 1) / (64 - 6 * (p->winsize / 2))) * n;
                h->tn.occ_blocks = blocks;
            }
            size_t pyr_base = 0;
            for (int g = 0; g < G; g++) {
                const int off = (int)((long)n * g / G);                    // near-equal contiguous parts (par:181-206 splits its chunks so)
                const int ng = (int)((long)n * (g + 1) / G) - off;
                h->stream = g == 0 ? scope.main : h->aux_stream[g - 1];
                std::vector<PyrLevel> lvg = lv;                            // this sub-batch's own part of the flow pyramid
                if (pyramid) pyr_base = flow_pyramid_layout(lvg, ng, 2, pyr_base);
                float* const accg = acc + (size_t)off * HW;
                float* const fA = flow + (size_t)off * HW * 2;
                float* const fB = flowB + (size_t)off * HW * 2;
                const int t0 = r + c0 + off;
                for (int side = 0; side < 2; side++) {
                    if (side == 1) launch_axpy_slices(stack, accg, PairBatch{ng, t0, 0}, H, W, kernel[r], h->stream); // seq:108
                    if (fused) {
                        const float* fin = nullptr;          // seq:94,109: the chain restarts from zero flow
                        float* fout = fA;
                        for (int step = 0; step < r; step++) {
                            int d = side == 0 ? -(step + 1) : (step + 1); // nearest neighbour first (seq:95,110)
                            bool keep = p->chained && step + 1 < r;       // the next step is seeded with this flow (seq:98)
                            if (pyramid) {
                                if (pyramid_step_fused(h, lvg, R, stack, fin, keep ? fout : nullptr, accg, PairBatch{ng, t0, d}, H, W,
                                                       p->winsize, p->iters, kernel[r + d], wm)) return -1;
                                if (keep) { fin = fout; fout = fout == fA ? fB : fA; }
                                continue;
                            }
                            ScopedTimer t(h, FDN_TIMER_FUSED);
                            launch_farneback_fused(R, stack, fin, keep ? fout : nullptr, accg, PairBatch{ng, t0, d}, H, W,
                                                   p->winsize, p->iters, kernel[r + d], h->stream, h->tn, 0, 0, wm);     // wm: an integer volume's own accumulate
                            if (keep) { fin = fout; fout = fout == fA ? fB : fA; }
                        }
                        continue;
                    }
                    float* const bufs[2] = {fA, fB};
                    const float* prev = nullptr;         // seq:94,109: the chain restarts from zero flow
                    for (int step = 0; step < r; step++) {
                        int d = side == 0 ? -(step + 1) : (step + 1);
                        bool keep = p->chained && step + 1 < r;
                        float* res = nullptr;
                        if (chain_step_iter(h, lvg, R, stack, prev, bufs, accg, PairBatch{ng, t0, d}, H, W, p->winsize, p->iters,
                                            kernel[r + d], keep, &res, wm)) return -1;
                        prev = keep ? res : nullptr;
                    }
                }
            }
            if (G > 1 && scope.join()) return -1;
            continue;
        }
        for (int side = 0; side < 2; side++) {
            if (side == 1) launch_axpy_slices(stack, acc, PairBatch{n, r + c0, 0}, H, W, kernel[r], st); // seq:108
            // per-stage kernels: the flow of every step of the side is kept ([r][n][HW][2]) and the side's warped-
            // Gaussian sweep runs as ONE launch afterwards, the accumulator in a register (12 B per pixel and pair)
            for (int step = 0; step < r; step++) {
                int d = side == 0 ? -(step + 1) : (step + 1);
                PairBatch pb{n, r + c0, d};
                float* fcur = flow + (size_t)step * C * HW * 2;
                const float* fprev = p->chained && step > 0 ? flow + (size_t)(step - 1) * C * HW * 2 : nullptr;   // seq:94,109: zero at the chain's start
                if (pyramid) {
                    if (pyramid_batch(h, lv, R, fprev, fcur, M0, M1, pb, H, W, p->winsize, p->iters)) return -1;
                } else if (run_iterations(h, R, fprev, fcur, M0, M1, pb, H, W, p->winsize, p->iters)) return -1;
            }
            trace_point(h, "sweep_side %d", side);
            {
                ScopedTimer t(h, FDN_TIMER_WARP);
                std::vector<double> wts(r);
                for (int step = 0; step < r; step++) wts[step] = kernel[side == 0 ? r - 1 - step : r + 1 + step];
                // steps of a batch smaller than C are still C * HW * 2 floats apart: one launch per step stride needs
                // npairs == C, so a short last batch folds step by step
                if (n == C) launch_sweep_side(stack, flow, acc, PairBatch{n, r + c0, side == 0 ? -1 : 1}, r, 0, H, W, wts.data(), st, wm);
                else
                    for (int step = 0; step < r; step++)
                        launch_sweep_side(stack, flow + (size_t)step * C * HW * 2, acc, PairBatch{n, r + c0, side == 0 ? -1 : 1}, 1, step, H, W, &wts[step], st, wm);
            }
        }
    }
    }   // batches of R
    if (wm_all.kind == FDN_WARP_ROUND_INT) launch_trunc_clamp(out_all, (size_t)S * HW, wm_all.lo, wm_all.hi, st);   // par:131, 287: the integer volume takes the pass
    trace_point(h, "sweep_stack done");
    FDN_HIP(hipGetLastError());
    return 0;
}

// stack[r + s] = slice s of `in` along `axis`, re-oriented so that slices are outermost and
// images keep the reference's (rows, cols): Z: (Y,X); Y: (Z,X) (seq:255); X: (Z,Y) (seq:333).
static void axis_dims(int Z, int Y, int X, int axis, int* S, int* H, int* W)
{
    if (axis == 0) { *S = Z; *H = Y; *W = X; }
    else if (axis == 1) { *S = Y; *H = Z; *W = X; }
    else { *S = X; *H = Z; *W = Y; }
}

// slices [g0, g0 + cnt) of `d_in` along `axis` (all inside the volume) -> cnt consecutive images at dst
static int load_slices(fdn_ctx* h, const float* d_in, float* dst, int g0, int cnt, int Z, int Y, int X, int axis)
{
    if (cnt <= 0) return 0;
    hipStream_t st = h->stream;
    if (axis == 0) FDN_HIP(hipMemcpyAsync(dst, d_in + (size_t)g0 * Y * X, (size_t)cnt * Y * X * sizeof(float), hipMemcpyDeviceToDevice, st));
    else if (axis == 1) launch_permute(d_in + (size_t)g0 * X, dst, cnt, Z, X, X, (int64_t)Y * X, 1, st);          // dst[y][z][x]
    else launch_permute(d_in + g0, dst, cnt, Z, Y, 1, (int64_t)Y * X, X, st);                                      // dst[x][z][y]
    return 0;
}

static int filter_axis_dev(fdn_ctx* h, const float* d_in, float* d_out, int Z, int Y, int X, int axis,
                           const double* kernel, int K, float pad_value, const fdn_sweep_params* p)
{
    if (check_params(p, K)) return -1;
    if (axis < 0 || axis > 2) return fail("axis must be 0, 1 or 2");
    if (Z <= 0 || Y <= 0 || X <= 0) return fail("bad volume dims");
    if (d_in == d_out) return fail("in and out must not alias");
    if (p->warp_mode == FDN_WARP_F64_PADDED) pad_value = (float)p->pad64;     // what Farneback's convertTo(CV_32F) makes of the float64 pad slices
    int S, H, W;
    axis_dims(Z, Y, X, axis, &S, &H, &W);
    const int r = K / 2;
    const size_t HW = (size_t)H * W;
    hipStream_t st = h->stream;
    // Targets per chunk of the pass.  Without a workspace limit: the whole pass at once.  With one: what fits next to
    // the buffers that must stay (intermediate volumes, staging) -- per target slice the stack (4 B/px), the re-oriented
    // output (4, Y and X passes) and the sweep's own expansions and flows; the K-1 halo slices come on top per chunk.
    int NP = S;
    if (h->ws_limit) {
        const size_t keep = h->vol_a.cap + h->vol_b.cap + h->vol_in.cap + h->vol_out.cap + h->pair.cap + h->partials.cap + h->area_tab.cap;
        const std::vector<PyrLevel> lv = pyramid_levels(p->use_of ? p->levels : 0, H, W);
        const bool pyr = lv.size() > 1;
        const int spath = sweep_path(h, p, lv, H, W);
        const size_t sweep_px = !p->use_of ? 0 : (pyr ? 27 : 20) + sweep_flow_px(spath, r, pyr);
        const size_t per_target = HW * (4 + (axis ? 4 : 0) + sweep_px);
        const size_t fixed = (size_t)2 * r * HW * (4 + (p->use_of ? (pyr ? 27 : 20) : 0)) + (pyr ? std::min<size_t>((size_t)1 << 28, HW * 12 * (size_t)(S + 2 * r)) : 0);
        if (h->ws_limit < keep + fixed + per_target)
            return fail("workspace limit of %zu bytes is too small for a pass over %d x %d images with K = %d: needs at least %zu",
                        h->ws_limit, W, H, K, keep + fixed + per_target);
        NP = (int)std::min<size_t>((size_t)S, (h->ws_limit - keep - fixed) / per_target);
    }
    if (ensure(h, h->stack, (size_t)(NP + 2 * r) * HW * sizeof(float))) return -1;
    float* stack = (float*)h->stack.p;
    if (axis != 0 && ensure(h, h->sweep_out, (size_t)NP * HW * sizeof(float))) return -1;
    if (h->reserve_only) {
        fdn_sweep_params pc = *p;
        pc.pad_lo = pc.pad_hi = 0;
        return sweep_stack(h, stack, stack, std::min(NP, S), H, W, kernel, K, &pc);
    }
    for (int s0 = 0; s0 < S; s0 += NP) {
        const int np = std::min(NP, S - s0);
        {   // stack position q holds slice s0 - r + q: from the volume, wrapped (par:312), or the pad value (seq:88-89)
            ScopedTimer t(h, FDN_TIMER_PERMUTE);
            const int nq = np + 2 * r;
            int q = 0;
            while (q < nq) {
                int g = s0 - r + q;
                if (p->border_mode == FDN_BORDER_WRAP) g = ((g % S) + S) % S;
                if (g < 0 || g >= S) {                       // a run of padding
                    int cnt = g < 0 ? std::min(-g, nq - q) : nq - q;
                    launch_fill(stack + (size_t)q * HW, pad_value, (size_t)cnt * HW, st);
                    q += cnt;
                } else {                                     // a run of consecutive slices inside the volume
                    int cnt = std::min(S - g, nq - q);
                    if (load_slices(h, d_in, stack + (size_t)q * HW, g, cnt, Z, Y, X, axis)) return -1;
                    q += cnt;
                }
            }
        }
        float* sw_out = axis == 0 ? d_out + (size_t)s0 * HW : (float*)h->sweep_out.p;
        fdn_sweep_params pc = *p;         // this chunk's pad slices (seq:88-89): before slice 0, after slice S - 1
        pc.pad_lo = p->border_mode == FDN_BORDER_MEAN_PAD ? std::min(std::max(0, r - s0), np + 2 * r) : 0;
        pc.pad_hi = p->border_mode == FDN_BORDER_MEAN_PAD ? std::min(std::max(0, s0 + np + r - S), np + 2 * r) : 0;
        if (sweep_stack(h, stack, sw_out, np, H, W, kernel, K, &pc)) return -1;
        if (axis != 0) {
            ScopedTimer t(h, FDN_TIMER_PERMUTE);
            if (axis == 1)      // out[z][s0 + yy][x] = t[yy][z][x]
                launch_permute(sw_out, d_out + (size_t)s0 * X, Z, np, X, X, (int64_t)Z * X, 1, st, (int64_t)Y * X, X);
            else                // out[z][y][s0 + xx] = t[xx][z][y]
                launch_permute(sw_out, d_out + s0, Z, Y, np, Y, 1, (int64_t)Z * Y, st, (int64_t)Y * X, X);
        }
    }
    FDN_HIP(hipGetLastError());
    return 0;
}

static int filter_3d_dev(fdn_ctx* h, const float* d_in, float* d_out, int Z, int Y, int X,
                         const double* const kernels[3], const int K[3], float pad_value, const fdn_sweep_params* p)
{
    if (!kernels || !K) return fail("kernels/K is NULL");
    int axes[3], na = 0;
    for (int a = 0; a < 3; a++) if (kernels[a] && K[a] > 0) axes[na++] = a;
    const size_t bytes = (size_t)Z * Y * X * sizeof(float);
    if (na == 0) {
        if (d_in != d_out) FDN_HIP(hipMemcpyAsync(d_out, d_in, bytes, hipMemcpyDeviceToDevice, h->stream));
        return 0;
    }
    if (!h->ws_limit) {
        // every whole-volume buffer of the call before the first pass sizes its flow batch from the free memory: the
        // intermediate volumes, the stack of the longest pass and the re-oriented output of the Y / X passes
        size_t stack_max = 0;
        bool reorient = false;
        for (int i = 0; i < na; i++) {
            int S, Hh, Ww;
            axis_dims(Z, Y, X, axes[i], &S, &Hh, &Ww);
            stack_max = std::max(stack_max, (size_t)(S + 2 * (K[axes[i]] / 2)) * Hh * Ww * sizeof(float));
            reorient = reorient || axes[i] != 0;
        }
        if (na > 1 && ensure(h, h->vol_a, bytes)) return -1;
        if (na > 2 && ensure(h, h->vol_b, bytes)) return -1;
        if (ensure(h, h->stack, stack_max)) return -1;
        if (reorient && ensure(h, h->sweep_out, bytes)) return -1;
    }
    // ping-pong so that the last pass lands in d_out and the input is never written
    const float* src = d_in;
    for (int i = 0; i < na; i++) {
        float* dst;
        if (i == na - 1) dst = d_out;
        else {
            DevBuf& b = (i & 1) ? h->vol_b : h->vol_a;
            if (ensure(h, b, bytes)) return -1;
            dst = (float*)b.p;
        }
        if (src == dst) return fail("in and out must not alias");
        int a = axes[i];
        if (filter_axis_dev(h, src, dst, Z, Y, X, a, kernels[a], K[a], pad_value, p)) return -1;
        src = dst;
    }
    return 0;
}

// numpy's add.reduce over a contiguous array (loops.c.src @TYPE@_pairwise_sum): blocks of <= 128 elements
// are summed with 8 interleaved accumulators, larger ranges split at n/2 rounded down to a multiple of 8.
// float: vol.mean() (seq:420) = f32(sum) / f32(n); double: the normalisation of the Gaussian taps (seq:37,
// scipy's phi_x.sum()).
template <typename T> static T np_pairwise_sum(const T* a, size_t n)
{
    if (n < 8) {
        T res = 0;
        for (size_t i = 0; i < n; i++) res += a[i];
        return res;
    }
    if (n <= 128) {
        T r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        size_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    size_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}
static float np_pairwise_sum_f32(const float* a, size_t n) { return np_pairwise_sum<float>(a, n); }


// =====================================================================================================================
// The sharded filter below the ABI (include/flowdn.h, fdn_filter_3d_sharded): the schedule of
// flowdenoising_amd/distributed.py (SlabPlan / SlabEngine) in C++, the transport behind two callbacks.
// =====================================================================================================================
namespace {

struct Range { int lo, hi; };
struct Block { int p0; Range rng[3]; };            // stack position of its first B-slice + global index ranges
static const int ORIENT[3][3] = {{0, 1, 2}, {1, 0, 2}, {2, 0, 1}};   // stack dims (slices, H, W) of a pass as global axes

struct ShardPlan {
    int N[3], world, rank;
    Range part(int axis, int r) const
    {
        const int base = N[axis] / world, rem = N[axis] % world;
        const int s = r * base + std::min(r, rem);
        return {s, s + base + (r < rem ? 1 : 0)};
    }
    // rank j's stack along `axis` holds positions p = 0 .. len_j + 2r - 1 = global slices s_j - r + p: maximal runs
    // (p0, g0, count) of consecutive positions with consecutive global slices inside the volume (wrap: modulo)
    void stack_runs(int axis, int j, int r, bool wrap, std::vector<std::array<int, 3>>& runs) const
    {
        runs.clear();
        const int n = N[axis];
        const Range pj = part(axis, j);
        for (int p = 0; p < pj.hi - pj.lo + 2 * r; p++) {
            int g = pj.lo - r + p;
            if (wrap) g = ((g % n) + n) % n;
            else if (g < 0 || g >= n) continue;
            if (!runs.empty() && runs.back()[0] + runs.back()[2] == p && runs.back()[1] + runs.back()[2] == g) runs.back()[2]++;
            else runs.push_back({p, g, 1});
        }
    }
    // what rank i (partition along A) sends rank j for the pass along B
    void blocks(int A, int B, int r, bool wrap, int i, int j, std::vector<Block>& out) const
    {
        out.clear();
        const Range pi = part(A, i);
        std::vector<std::array<int, 3>> runs;
        stack_runs(B, j, r, wrap, runs);
        for (auto& run : runs) {
            Block b;
            for (int ax = 0; ax < 3; ax++) b.rng[ax] = {0, N[ax]};
            if (A == B) {
                const int lo = std::max(run[1], pi.lo), hi = std::min(run[1] + run[2], pi.hi);
                if (lo >= hi) continue;
                b.rng[B] = {lo, hi};
                b.p0 = run[0] + lo - run[1];
            } else {
                b.rng[B] = {run[1], run[1] + run[2]};
                b.rng[A] = pi;
                b.p0 = run[0];
            }
            out.push_back(b);
        }
    }
    static size_t numel(const Block& b) { size_t n = 1; for (int ax = 0; ax < 3; ax++) n *= (size_t)(b.rng[ax].hi - b.rng[ax].lo); return n; }
};

} // namespace

// cur: this rank's slab of the partition along A, laid out in A's orientation (own A-range outermost).  Fills `stack`
// (orientation of B, n_loc + 2 r slices) from the slabs of all ranks: pack into the receiver's orientation, one batched
// group of messages, unpack with row-contiguous strided copies.
static int shard_exchange(fdn_ctx* h, const ShardPlan& pl, const fdn_comm* comm, const float* cur, int A, int B, int r, bool wrap,
                          float* stack, int Hs, int Ws)
{
    const int me = pl.rank, world = pl.world;
    // loopback (a test switch, fdn_set_option "shard_loopback"): the block this rank keeps goes through the transport as a
    // send to self inside the group, so that one rank on one GPU issues exactly the calls of an N > 1 run
    const bool loop = h->tn.shard_loopback != 0;
    const int* oa = ORIENT[A];
    const int* ob = ORIENT[B];
    const Range mineA = pl.part(A, me);
    int64_t stride_cur[3];                                   // element strides of `cur` per GLOBAL axis
    stride_cur[oa[0]] = (int64_t)pl.N[oa[1]] * pl.N[oa[2]];
    stride_cur[oa[1]] = pl.N[oa[2]];
    stride_cur[oa[2]] = 1;
    std::vector<std::vector<Block>> sends(world), recvs(world);
    size_t n_send = 0, n_recv = 0;
    for (int j = 0; j < world; j++) {
        pl.blocks(A, B, r, wrap, me, j, sends[j]);
        pl.blocks(A, B, r, wrap, j, me, recvs[j]);
        for (auto& b : sends[j]) n_send += ShardPlan::numel(b);
        if (j != me || loop) for (auto& b : recvs[j]) n_recv += ShardPlan::numel(b);
    }
    if (ensure(h, h->sh_send, std::max<size_t>(n_send, 1) * sizeof(float)) || ensure(h, h->sh_recv, std::max<size_t>(n_recv, 1) * sizeof(float))) return -1;
    float* sendbuf = (float*)h->sh_send.p;
    float* recvbuf = (float*)h->sh_recv.p;
    std::vector<size_t> send_off(world + 1, 0), recv_off(world + 1, 0);
    {   // 1. pack every block into the receiver's orientation
        ScopedTimer t(h, FDN_TIMER_PERMUTE);
        size_t off = 0;
        for (int j = 0; j < world; j++) {
            send_off[j] = off;
            for (auto& b : sends[j]) {
                int64_t base = 0;
                for (int ax = 0; ax < 3; ax++) base += (int64_t)(b.rng[ax].lo - (ax == A ? mineA.lo : 0)) * stride_cur[ax];
                const int d0 = b.rng[ob[0]].hi - b.rng[ob[0]].lo, d1 = b.rng[ob[1]].hi - b.rng[ob[1]].lo, d2 = b.rng[ob[2]].hi - b.rng[ob[2]].lo;
                launch_permute(cur + base, sendbuf + off, d0, d1, d2, stride_cur[ob[0]], stride_cur[ob[1]], stride_cur[ob[2]], h->stream);
                off += (size_t)d0 * d1 * d2;
            }
        }
        send_off[world] = off;
        FDN_HIP(hipGetLastError());
    }
    {   // 2. one batched group of point-to-point messages: every pair at once
        size_t off = 0;
        std::vector<fdn_msg> msgs;
        for (int i = 0; i < world; i++) {
            recv_off[i] = off;
            if (i == me && !loop) continue;
            size_t n = 0;
            for (auto& b : recvs[i]) n += ShardPlan::numel(b);
            if (n) msgs.push_back({recvbuf + off, n * sizeof(float), i, 0});
            off += n;
        }
        recv_off[world] = off;
        for (int j = 0; j < world; j++)
            if ((j != me || loop) && send_off[j + 1] > send_off[j]) msgs.push_back({sendbuf + send_off[j], (send_off[j + 1] - send_off[j]) * sizeof(float), j, 1});
        if (world > 1 || loop) {
            ScopedTimer t(h, FDN_TIMER_COLLECTIVE);
            if (comm->exchange(comm->ctx, (int)msgs.size(), msgs.data(), (void*)h->stream)) return fail("fdn_comm.exchange failed (pass along axis %d)", B);
        }
    }
    {   // 3. unpack into the stack (the block this rank keeps comes straight from its send buffer)
        ScopedTimer t(h, FDN_TIMER_PERMUTE);
        for (int i = 0; i < world; i++) {
            const float* buf = (i == me && !loop) ? sendbuf + send_off[me] : recvbuf + recv_off[i];
            size_t off = 0;
            for (auto& b : recvs[i]) {
                const int n0 = b.rng[ob[0]].hi - b.rng[ob[0]].lo, n1 = b.rng[ob[1]].hi - b.rng[ob[1]].lo, n2 = b.rng[ob[2]].hi - b.rng[ob[2]].lo;
                float* dst = stack + ((size_t)b.p0 * Hs + b.rng[ob[1]].lo) * Ws + b.rng[ob[2]].lo;
                launch_permute(buf + off, dst, n0, n1, n2, (int64_t)n1 * n2, n2, 1, h->stream, (int64_t)Hs * Ws, Ws);
                off += (size_t)n0 * n1 * n2;
            }
        }
        FDN_HIP(hipGetLastError());
    }
    return 0;
}

// numpy's float32 mean of the WHOLE volume (seq:420) from Z-slabs, bit for bit: every 8192-element chunk of the flattened
// volume is summed (numpy's pairwise order) by the rank that holds its first element; a chunk that straddles a slab
// boundary gets its missing elements from the following rank; the chunk sums are gathered and accumulated in order.
static int shard_mean(fdn_ctx* h, const ShardPlan& pl, const fdn_comm* comm, const float* slab, float* mean_out)
{
    const int me = pl.rank, world = pl.world;
    const size_t YX = (size_t)pl.N[1] * pl.N[2], ntot = YX * pl.N[0];
    std::vector<size_t> starts(world + 1);
    size_t minlen = ntot;
    for (int k = 0; k < world; k++) { starts[k] = (size_t)pl.part(0, k).lo * YX; }
    starts[world] = ntot;
    for (int k = 0; k < world; k++) minlen = std::min(minlen, starts[k + 1] - starts[k]);
    const size_t mylen = starts[me + 1] - starts[me];
    if (world == 1 && !h->tn.shard_loopback) return fdn_mean_dev(h, slab, ntot, mean_out);
    if (minlen < 8192) {        // tiny volume: a chunk may span several slabs; gather the whole thing (it is small)
        size_t m = 0;
        for (int k = 0; k < world; k++) m = std::max(m, starts[k + 1] - starts[k]);
        std::vector<float> mine(m, 0.f), all(m * world);
        if (copy_host(h, mine.data(), slab, mylen * sizeof(float), false)) return -1;
        if (comm->allgather_host(comm->ctx, mine.data(), all.data(), m * sizeof(float))) return fail("fdn_comm.allgather_host failed");
        std::vector<float> whole(ntot);
        for (int k = 0; k < world; k++) memcpy(whole.data() + starts[k], all.data() + (size_t)k * m, (starts[k + 1] - starts[k]) * sizeof(float));
        return fdn_mean_host(whole.data(), ntot, mean_out);
    }
    auto up = [](size_t v) { return (v + 8191) / 8192 * 8192; };
    std::vector<size_t> first(world + 1);
    for (int k = 0; k < world; k++) first[k] = std::min(up(starts[k]), ntot);
    first[world] = ntot;
    const size_t head = first[me] - starts[me];              // my leading elements belong to the previous rank's last chunk
    const size_t tail = first[me + 1] - starts[me + 1];      // elements of my last chunk held by the next rank
    if (ensure(h, h->sh_tmp, 2 * 8192 * sizeof(float))) return -1;
    float* tmp = (float*)h->sh_tmp.p;                        // [0, 8192): my last chunk assembled; [8192, ...): the tail received
    {
        std::vector<fdn_msg> msgs;
        if (tail > 0) msgs.push_back({tmp + 8192, tail * sizeof(float), me + 1, 0});
        if (head > 0) msgs.push_back({const_cast<float*>(slab), head * sizeof(float), me - 1, 1});
        if (comm->exchange(comm->ctx, (int)msgs.size(), msgs.data(), (void*)h->stream)) return fail("fdn_comm.exchange failed (mean)");
    }
    const float* own = slab + head;
    const size_t nown = mylen - head;
    std::vector<size_t> per(world);
    size_t maxper = 1;
    for (int k = 0; k < world; k++) { per[k] = (first[k + 1] - first[k] + 8191) / 8192; maxper = std::max(maxper, per[k]); }
    std::vector<float> mine(maxper, 0.f), all(maxper * world);
    size_t got = 0;
    if (tail > 0) {
        const size_t nfull = nown / 8192 * 8192;
        if (nfull) { if (fdn_np_chunk_sums_dev(h, own, nfull, mine.data())) return -1; got = nfull / 8192; }
        FDN_HIP(hipMemcpyAsync(tmp, own + nfull, (nown - nfull) * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
        FDN_HIP(hipMemcpyAsync(tmp + (nown - nfull), tmp + 8192, tail * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
        if (fdn_np_chunk_sums_dev(h, tmp, nown - nfull + tail, mine.data() + got)) return -1;
        got += (nown - nfull + tail + 8191) / 8192;
    } else if (nown) {
        if (fdn_np_chunk_sums_dev(h, own, nown, mine.data())) return -1;
        got = (nown + 8191) / 8192;
    }
    if (got != per[me]) return fail("internal: %zu chunk sums, expected %zu", got, per[me]);
    if (comm->allgather_host(comm->ctx, mine.data(), all.data(), maxper * sizeof(float))) return fail("fdn_comm.allgather_host failed");
    float tot = 0.f;
    for (int k = 0; k < world; k++)
        for (size_t c = 0; c < per[k]; c++) tot += all[(size_t)k * maxper + c];      // numpy accumulates its chunks left to right in float32
    *mean_out = tot / (float)ntot;
    return 0;
}

static int filter_3d_sharded(fdn_ctx* h, const float* slab_in, float* slab_out, int Z, int Y, int X, const double* const kernels[3],
                             const int K[3], const fdn_sweep_params* p, const fdn_comm* comm)
{
    ShardPlan pl;
    pl.N[0] = Z; pl.N[1] = Y; pl.N[2] = X; pl.world = comm->world; pl.rank = comm->rank;
    if (pl.world < 1 || pl.rank < 0 || pl.rank >= pl.world) return fail("bad communicator: rank %d of %d", pl.rank, pl.world);
    if (pl.world > std::min(Z, std::min(Y, X))) return fail("%d ranks need every axis >= %d", pl.world, pl.world);
    const bool wrap = p->border_mode == FDN_BORDER_WRAP;
    float mean = 0.f;
    if (p->warp_mode == FDN_WARP_F64_PADDED) mean = (float)p->pad64;
    else if (!wrap) {
        ScopedTimer t(h, FDN_TIMER_MEAN);
        if (shard_mean(h, pl, comm, slab_in, &mean)) return -1;
    }
    const float* cur = slab_in;
    int cur_axis = 0, flip = 0;
    for (int axis = 0; axis < 3; axis++) {
        if (!kernels[axis] || K[axis] <= 0) continue;
        if (check_params(p, K[axis])) return -1;
        const int r = K[axis] / 2;
        const Range mine = pl.part(axis, pl.rank);
        const int n_loc = mine.hi - mine.lo, Hs = pl.N[ORIENT[axis][1]], Ws = pl.N[ORIENT[axis][2]];
        const size_t HW = (size_t)Hs * Ws;
        if (ensure(h, h->sh_stack, (size_t)(n_loc + 2 * r) * HW * sizeof(float))) return -1;
        float* stack = (float*)h->sh_stack.p;
        if (shard_exchange(h, pl, comm, cur, cur_axis, axis, r, wrap, stack, Hs, Ws)) return -1;
        fdn_sweep_params pp = *p;
        pp.pad_lo = pp.pad_hi = 0;
        if (!wrap) {             // stack positions whose slice lies outside the volume (seq:88-89)
            const int lo = std::max(0, r - mine.lo);
            const int hi = std::min(n_loc + 2 * r, pl.N[axis] - mine.lo + r);
            if (lo > 0) launch_fill(stack, mean, (size_t)lo * HW, h->stream);
            if (hi < n_loc + 2 * r) launch_fill(stack + (size_t)hi * HW, mean, (size_t)(n_loc + 2 * r - hi) * HW, h->stream);
            pp.pad_lo = lo; pp.pad_hi = n_loc + 2 * r - hi;
        }
        DevBuf& ob = h->sh_out[flip];
        if (ensure(h, ob, (size_t)n_loc * HW * sizeof(float))) return -1;
        if (sweep_stack(h, stack, (float*)ob.p, n_loc, Hs, Ws, kernels[axis], K[axis], &pp)) return -1;
        cur = (const float*)ob.p; cur_axis = axis; flip ^= 1;
    }
    const Range mz = pl.part(0, pl.rank);
    if (cur_axis != 0) {         // back to Z-slabs: the same exchange with no halo
        if (shard_exchange(h, pl, comm, cur, cur_axis, 0, 0, false, slab_out, Y, X)) return -1;
    } else if (cur != slab_out) {
        FDN_HIP(hipMemcpyAsync(slab_out, cur, (size_t)(mz.hi - mz.lo) * Y * X * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    }
    FDN_HIP(hipGetLastError());
    return 0;
}

// ---- exported C ABI ---------------------------------------------------------------
extern "C" {

#define FDN_API __attribute__((visibility("default")))

FDN_API const char* fdn_last_error(void) { return g_err.c_str(); }
FDN_API const char* fdn_version(void) { return "flowdn 0.1 gfx950"; }

FDN_API int fdn_create(int device, fdn_handle* out)
{
    if (!out) return fail("out is NULL");
    int ndev = 0;
    FDN_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail("device %d out of range (have %d)", device, ndev);
    FDN_HIP(hipSetDevice(device));
    fdn_ctx* h = new fdn_ctx();
    h->device = device;
    FDN_DEVICE_WIDE;
    hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return fail("hipStreamCreate failed: %s", hipGetErrorString(e)); }
    h->stream = h->own_stream;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) h->tn.cus = prop.multiProcessorCount;
    // the environment is read here, once per handle; fdn_set_option changes a live handle
    auto env_int = [](const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; };
    h->tn.strict_order = env_int("FDN_STRICT_ORDER") != 0;
    h->tn.path = env_int("FDN_FORCE_STAGED") ? 1 : env_int("FDN_PATH");
    h->tn.fused_occ = env_int("FDN_FUSED_OCC");
    h->tn.lds_pad = (unsigned)env_int("FDN_LDS_PAD");
    h->tn.sub_batches = std::min((int)fdn_ctx::MAX_SUB, std::max(0, env_int("FDN_SUB_BATCHES")));
    h->tn.fma.mode = std::min(2, std::max(0, env_int("FDN_OPENCV_FMA")));
    if (getenv("FDN_OPENCV_FMA_LANES")) h->tn.fma.lanes = std::min(64, std::max(1, env_int("FDN_OPENCV_FMA_LANES")));
    h->tn.remap_model = env_int("FDN_REMAP_MODEL") == 1 ? 1 : 0;
    if (const char* tp = getenv("FDN_LAUNCH_TRACE")) h->trace_fd = open(tp, O_CREAT | O_WRONLY | O_APPEND, 0644);
    *out = h;
    return 0;
}

FDN_API int fdn_destroy(fdn_handle h)
{
    if (!h) return 0;
    {   // a call still running on another thread finishes first (the caller must not START one after this: the handle is gone)
        std::lock_guard<std::recursive_mutex> enter_guard_(h->mu);
        (void)hipSetDevice(h->device);
        (void)hipStreamSynchronize(h->stream);
        FDN_DEVICE_WIDE;
        free_all(h);
        if (h->pinned) (void)hipHostFree(h->pinned);
        if (h->bounce) (void)hipHostFree(h->bounce);
        for (hipEvent_t ev : h->bounce_ev) if (ev) (void)hipEventDestroy(ev);
        resolve_stamps(h);
        for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
        for (hipStream_t s : h->aux_stream) if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
        if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
        for (hipEvent_t e : h->ev_join) if (e) (void)hipEventDestroy(e);
        if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
        if (h->trace_fd >= 0) close(h->trace_fd);
    }
    delete h;
    return 0;
}

#define FDN_ENTER(h)                                                  \
    if (!(h)) return fail("handle is NULL");                          \
    std::lock_guard<std::recursive_mutex> enter_guard_((h)->mu);      \
    FDN_HIP(hipSetDevice((h)->device))

FDN_API int fdn_set_stream(fdn_handle h, void* s)
{
    FDN_ENTER(h);
    FDN_HIP(hipStreamSynchronize(h->stream));
    h->stream = (hipStream_t)s; // NULL is a valid choice: the legacy default stream (what torch uses by default)
    return 0;
}
FDN_API int fdn_reset_stream(fdn_handle h)
{
    FDN_ENTER(h);
    FDN_HIP(hipStreamSynchronize(h->stream));
    h->stream = h->own_stream;
    return 0;
}
FDN_API int fdn_synchronize(fdn_handle h)
{
    FDN_ENTER(h);
    FDN_HIP(hipStreamSynchronize(h->stream));
    return 0;
}
FDN_API int fdn_set_workspace_limit(fdn_handle h, size_t bytes)
{
    FDN_ENTER(h);
    FDN_HIP(hipStreamSynchronize(h->stream));
    free_all(h);              // buffers only ever grow: start over so that the new limit holds from the next call on
    h->ws_limit = bytes;
    return 0;
}
FDN_API int fdn_mem_info(fdn_handle h, size_t* free_out, size_t* total_out)
{
    FDN_ENTER(h);
    size_t fre = 0, tot = 0;
    FDN_HIP(hipMemGetInfo(&fre, &tot));
    if (free_out) *free_out = fre;
    if (total_out) *total_out = tot;
    return 0;
}
FDN_API int fdn_workspace_bytes(fdn_handle h, size_t* bytes_out)
{
    FDN_ENTER(h);
    if (!bytes_out) return fail("bytes_out is NULL");
    *bytes_out = owned_bytes(h);
    return 0;
}
FDN_API int fdn_set_option(fdn_handle h, const char* name, long value)
{
    FDN_ENTER(h);
    if (!name) return fail("option name is NULL");
    FDN_HIP(hipStreamSynchronize(h->stream));
    if (!strcmp(name, "strict_order")) h->tn.strict_order = value != 0;
    else if (!strcmp(name, "path")) { if (value < 0 || value > 2) return fail("path must be 0 (auto), 1 (staged) or 2 (per-iteration kernels)"); h->tn.path = (int)value; }
    else if (!strcmp(name, "fused_occ")) { if (value && (value < 3 || value > 5) && value != 8) return fail("fused_occ must be 0, 3, 4, 5 or 8"); h->tn.fused_occ = (int)value; }
    else if (!strcmp(name, "lds_pad")) { if (value < 0 || value > 160 * 1024) return fail("lds_pad out of range"); h->tn.lds_pad = (unsigned)value; }
    else if (!strcmp(name, "shard_loopback")) h->tn.shard_loopback = value != 0;
    else if (!strcmp(name, "opencv_fma")) { if (value < 0 || value > 2) return fail("opencv_fma must be 0 (two roundings), 1 (fused) or 2 (fused on the vector body of a row)"); h->tn.fma.mode = (int)value; }
    else if (!strcmp(name, "opencv_fma_lanes")) { if (value < 1 || value > 64) return fail("opencv_fma_lanes must be 1 .. 64"); h->tn.fma.lanes = (int)value; }
    else if (!strcmp(name, "remap_model")) { if (value < 0 || value > 1) return fail("remap_model must be 0 (1/32-pixel table) or 1 (unquantised float32 bilinear)"); h->tn.remap_model = (int)value; }
    else if (!strcmp(name, "sub_batches")) { if (value < 0 || value > fdn_ctx::MAX_SUB) return fail("sub_batches must be 0 (automatic) or 1 .. %d", (int)fdn_ctx::MAX_SUB); h->tn.sub_batches = (int)value; }
    else return fail("unknown option '%s' (strict_order, path, fused_occ, lds_pad, shard_loopback, sub_batches, opencv_fma, opencv_fma_lanes, remap_model)", name);
    return 0;
}
FDN_API int fdn_get_option(fdn_handle h, const char* name, long* value_out)
{
    FDN_ENTER(h);
    if (!name || !value_out) return fail("NULL pointer");
    if (!strcmp(name, "strict_order")) *value_out = h->tn.strict_order;
    else if (!strcmp(name, "path")) *value_out = h->tn.path;
    else if (!strcmp(name, "fused_occ")) *value_out = h->tn.fused_occ;
    else if (!strcmp(name, "lds_pad")) *value_out = (long)h->tn.lds_pad;
    else if (!strcmp(name, "shard_loopback")) *value_out = h->tn.shard_loopback;
    else if (!strcmp(name, "sub_batches")) *value_out = h->tn.sub_batches;
    else if (!strcmp(name, "opencv_fma")) *value_out = h->tn.fma.mode;
    else if (!strcmp(name, "opencv_fma_lanes")) *value_out = h->tn.fma.lanes;
    else if (!strcmp(name, "remap_model")) *value_out = h->tn.remap_model;
    else if (!strcmp(name, "last_sub_batches")) *value_out = h->last_sub_batches;     // read-only: what the last sweep ran with
    else if (!strcmp(name, "compute_units")) *value_out = h->tn.cus;                   // read-only
    else return fail("unknown option '%s'", name);
    return 0;
}
FDN_API int fdn_malloc(fdn_handle h, size_t bytes, void** dptr)
{
    FDN_ENTER(h);
    if (!dptr) return fail("dptr is NULL");
    FDN_DEVICE_WIDE;
    FDN_HIP(dev_malloc(dptr, bytes ? bytes : 1, h->device));
    return 0;
}
FDN_API int fdn_free(fdn_handle h, void* dptr)
{
    FDN_ENTER(h);
    FDN_HIP(hipStreamSynchronize(h->stream));
    FDN_DEVICE_WIDE;
    if (dptr) FDN_HIP(dev_free(dptr));
    return 0;
}
FDN_API int fdn_memcpy_h2d(fdn_handle h, void* dst, const void* src, size_t bytes)
{
    FDN_ENTER(h);
    if (!dst || !src) return bytes ? fail("NULL pointer") : 0;
    ScopedTimer t(h, FDN_TIMER_TRANSFER);
    return copy_host(h, dst, src, bytes, true);
}
FDN_API int fdn_memcpy_d2h(fdn_handle h, void* dst, const void* src, size_t bytes)
{
    FDN_ENTER(h);
    if (!dst || !src) return bytes ? fail("NULL pointer") : 0;
    ScopedTimer t(h, FDN_TIMER_TRANSFER);
    return copy_host(h, dst, src, bytes, false);
}
// strided host <-> device copies (a slab of a host volume that is not contiguous: volume[:, y0:y1, :] or volume[:, :, x0:x1])
FDN_API int fdn_memcpy2d_h2d(fdn_handle h, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes, size_t height)
{
    FDN_ENTER(h);
    if (!dst || !src) return fail("NULL pointer");
    if (!width_bytes || !height) return 0;
    ScopedTimer t(h, FDN_TIMER_TRANSFER);
    return copy_host_2d(h, dst, dpitch, const_cast<void*>(src), spitch, width_bytes, height, true);
}
FDN_API int fdn_memcpy2d_d2h(fdn_handle h, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes, size_t height)
{
    FDN_ENTER(h);
    if (!dst || !src) return fail("NULL pointer");
    if (!width_bytes || !height) return 0;
    ScopedTimer t(h, FDN_TIMER_TRANSFER);
    return copy_host_2d(h, const_cast<void*>(src), spitch, dst, dpitch, width_bytes, height, false);
}
// page-lock / release a caller's host buffer so that copies from and to it run as DMA at PCIe speed
FDN_API int fdn_host_register(fdn_handle h, void* ptr, size_t bytes)
{
    FDN_ENTER(h);
    if (!ptr || !bytes) return fail("NULL pointer");
    const hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) {      // a refusal is an answer, not a failure of the handle: callers fall back to pageable copies,
        (void)hipGetLastError();   // so the runtime's sticky last-error must not surface in their next launch check
        return fail("hipHostRegister(%zu bytes): %s", bytes, hipGetErrorString(e));
    }
    std::lock_guard<std::mutex> g(g_locked_mu);
    g_locked[(uintptr_t)ptr] = bytes;
    return 0;
}
FDN_API int fdn_host_unregister(fdn_handle h, void* ptr)
{
    // A registration belongs to the process, not to the handle that made it: it can be released through any handle, and
    // through none (h == NULL) -- a buffer must never stay registered because the handle it was registered through is gone.
    std::unique_lock<std::recursive_mutex> enter_guard_;
    if (h) {
        enter_guard_ = std::unique_lock<std::recursive_mutex>(h->mu);
        FDN_HIP(hipSetDevice(h->device));
    }
    if (!ptr) return fail("NULL pointer");
    {
        std::lock_guard<std::mutex> g(g_locked_mu);
        g_locked.erase((uintptr_t)ptr);
    }
    const hipError_t e = hipHostUnregister(ptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail("hipHostUnregister: %s", hipGetErrorString(e));
    }
    return 0;
}
FDN_API int fdn_memset_f32(fdn_handle h, float* dst, float value, size_t count)
{
    FDN_ENTER(h);
    launch_fill(dst, value, count, h->stream);
    FDN_HIP(hipGetLastError());
    return 0;
}

// a-1 (seq:30-41): scipy.ndimage.gaussian_filter1d's kernel, truncate = 4
FDN_API int fdn_gaussian_kernel(double sigma, double* out, int cap)
{
    if (!(sigma > 0)) return fail("sigma must be > 0");
    int r = (int)(4.0 * sigma + 0.5);
    int K = 2 * r + 1;
    if (!out || K > cap) return -K;
    const double sigma2 = sigma * sigma;
    for (int j = -r; j <= r; j++) out[j + r] = exp(-0.5 / sigma2 * (double)(j * j));
    const double s = np_pairwise_sum(out, (size_t)K);     // phi_x.sum() of scipy's _gaussian_kernel1d
    for (int i = 0; i < K; i++) out[i] /= s;
    return K;
}

// The pair-level Farneback on device data: img = [prev | next] contiguous H x W each, flow (H x W x 2) holds the
// initial flow (or anything, without USE_INITIAL_FLOW) and receives the result; R, M0, M1 are scratch.
static int farneback_pair_dev(fdn_ctx* h, float* img, float* R, float* flow, float* M0, float* M1, int H, int W,
                              int levels, int winsize, int iters, int poly_n, double poly_sigma, int flags)
{
    std::vector<PyrLevel> lv = pyramid_levels(levels, H, W);
    hipStream_t st = h->stream;
    if (!(flags & FDN_USE_INITIAL_FLOW)) launch_fill(flow, 0.f, (size_t)H * W * 2, st);
    PolyConsts pc;
    prepare_poly_consts(poly_n, poly_sigma, &pc);
    {
        ScopedTimer t(h, FDN_TIMER_POLYEXP);
        launch_blur3_polyexp(img, R, 2, H, W, pc, st);
    }
    if (lv.size() > 1) {
        if (build_R_pyramid(h, img, 2, H, W, lv, pc)) return -1;
        if (ensure_flow_pyramid(h, lv, 1)) return -1;
        if (pyramid_batch(h, lv, R, (flags & FDN_USE_INITIAL_FLOW) ? flow : nullptr, flow, M0, M1, PairBatch{1, 0, 1}, H, W, winsize, iters)) return -1;
    } else if (run_iterations(h, R, flow, flow, M0, M1, PairBatch{1, 0, 1}, H, W, winsize, iters)) return -1;
    FDN_HIP(hipGetLastError());
    return 0;
}

static int check_pair_args(int H, int W, int levels, int winsize, int iters, int poly_n, double poly_sigma, int flags)
{
    if (H < 2 || W < 2) return fail("optical flow needs images of at least 2x2 pixels, got %dx%d", W, H);
    // the kernels address the pixels of one image by 32-bit byte offsets (8 bytes per pixel pair plane)
    if (H >= (1 << 24) || W >= (1 << 24) || (size_t)H * W >= ((size_t)1 << 29))
        return fail("images of %dx%d pixels are not supported (limit: 2^29 pixels per image)", W, H);
    if (flags & ~FDN_USE_INITIAL_FLOW) return fail("unsupported flags 0x%x (only OPTFLOW_USE_INITIAL_FLOW)", flags);
    fdn_sweep_params p{levels, winsize, iters, poly_n, poly_sigma, 0, 1, 1};
    return check_params(&p, 1);
}

// pair scratch: [prev, next] images | R x2 | flow | M0 | M1
struct PairScratch { float *img, *R, *flow, *M0, *M1; };
static int pair_scratch(fdn_ctx* h, int H, int W, PairScratch* ps)
{
    const size_t HW = (size_t)H * W;
    if (ensure(h, h->pair, HW * 4 * (2 + 10 + 2 + 5 + 5))) return -1;
    ps->img = (float*)h->pair.p;
    ps->R = ps->img + 2 * HW;
    ps->flow = ps->R + 10 * HW;
    ps->M0 = ps->flow + 2 * HW;
    ps->M1 = ps->M0 + 5 * HW;
    return 0;
}

FDN_API int fdn_farneback_strided(fdn_handle h, const float* prev, ptrdiff_t prev_rs, ptrdiff_t prev_cs,
                                  const float* next, ptrdiff_t next_rs, ptrdiff_t next_cs, float* flow_io, int H, int W,
                                  int levels, int winsize, int iters, int poly_n, double poly_sigma, int flags)
{
    FDN_ENTER(h);
    if (!prev || !next || !flow_io) return fail("NULL image/flow pointer");
    if (check_pair_args(H, W, levels, winsize, iters, poly_n, poly_sigma, flags)) return -1;
    const size_t HW = (size_t)H * W;
    hipStream_t st = h->stream;
    PairScratch ps;
    if (pair_scratch(h, H, W, &ps)) return -1;
    // host staging (pinned, owned by the handle): the two views gathered contiguously | the flow
    if (ensure_pinned(h, HW * 4 * 4)) return -1;
    float* stage = (float*)h->pinned;
    {
        ScopedTimer t(h, FDN_TIMER_TRANSFER);
        gather_host(stage, prev, prev_rs, prev_cs, H, W);
        gather_host(stage + HW, next, next_rs, next_cs, H, W);
        FDN_HIP(hipMemcpyAsync(ps.img, stage, HW * 8, hipMemcpyHostToDevice, st));
        if (flags & FDN_USE_INITIAL_FLOW) {
            memcpy(stage + 2 * HW, flow_io, HW * 8);
            FDN_HIP(hipMemcpyAsync(ps.flow, stage + 2 * HW, HW * 8, hipMemcpyHostToDevice, st));
        }
    }
    if (farneback_pair_dev(h, ps.img, ps.R, ps.flow, ps.M0, ps.M1, H, W, levels, winsize, iters, poly_n, poly_sigma, flags)) return -1;
    {
        ScopedTimer t(h, FDN_TIMER_TRANSFER);
        FDN_HIP(hipMemcpyAsync(stage + 2 * HW, ps.flow, HW * 8, hipMemcpyDeviceToHost, st));
        FDN_HIP(hipStreamSynchronize(st));
        memcpy(flow_io, stage + 2 * HW, HW * 8);
    }
    return 0;
}

FDN_API int fdn_farneback(fdn_handle h, const float* prev, const float* next, float* flow_io, int H, int W,
                          int levels, int winsize, int iters, int poly_n, double poly_sigma, int flags)
{
    return fdn_farneback_strided(h, prev, W, 1, next, W, 1, flow_io, H, W, levels, winsize, iters, poly_n, poly_sigma, flags);
}

FDN_API int fdn_farneback_dev(fdn_handle h, const float* d_prev, ptrdiff_t prev_rs, ptrdiff_t prev_cs,
                              const float* d_next, ptrdiff_t next_rs, ptrdiff_t next_cs, float* d_flow_io, int H, int W,
                              int levels, int winsize, int iters, int poly_n, double poly_sigma, int flags)
{
    FDN_ENTER(h);
    if (!d_prev || !d_next || !d_flow_io) return fail("NULL image/flow pointer");
    if (check_pair_args(H, W, levels, winsize, iters, poly_n, poly_sigma, flags)) return -1;
    PairScratch ps;
    if (pair_scratch(h, H, W, &ps)) return -1;
    const size_t HW = (size_t)H * W;
    launch_copy_strided(d_prev, prev_rs, prev_cs, ps.img, H, W, h->stream);
    launch_copy_strided(d_next, next_rs, next_cs, ps.img + HW, H, W, h->stream);
    // the caller's flow buffer is used in place (cv2 updates `flow` in place too, seq:98)
    return farneback_pair_dev(h, ps.img, ps.R, d_flow_io, ps.M0, ps.M1, H, W, levels, winsize, iters, poly_n, poly_sigma, flags);
}

static int check_warp_args(const void* a, const void* b, const void* c, int H, int W)
{
    if (!a || !b || !c) return fail("NULL pointer");
    if (H <= 0 || W <= 0) return fail("bad image dims");
    if (H >= (1 << 24) || W >= (1 << 24) || (size_t)H * W >= ((size_t)1 << 29))
        return fail("images of %dx%d pixels are not supported (limit: 2^29 pixels per image)", W, H);
    return 0;
}

FDN_API int fdn_warp_strided(fdn_handle h, const float* reference, ptrdiff_t rs, ptrdiff_t cs, const float* flow, float* dst, int H, int W)
{
    FDN_ENTER(h);
    if (check_warp_args(reference, flow, dst, H, W)) return -1;
    const size_t HW = (size_t)H * W;
    hipStream_t st = h->stream;
    if (ensure(h, h->pair, HW * 4 * 4)) return -1;
    if (ensure_pinned(h, HW * 4 * 4)) return -1;
    float* d_src = (float*)h->pair.p;
    float* d_flow = d_src + HW;
    float* d_dst = d_flow + 2 * HW;
    float* stage = (float*)h->pinned;
    {
        ScopedTimer t(h, FDN_TIMER_TRANSFER);
        gather_host(stage, reference, rs, cs, H, W);
        memcpy(stage + HW, flow, HW * 8);
        FDN_HIP(hipMemcpyAsync(d_src, stage, HW * 12, hipMemcpyHostToDevice, st));
    }
    launch_warp(d_src, d_flow, d_dst, H, W, st, h->tn.remap_model);
    FDN_HIP(hipGetLastError());
    ScopedTimer t(h, FDN_TIMER_TRANSFER);
    FDN_HIP(hipMemcpyAsync(stage + 3 * HW, d_dst, HW * 4, hipMemcpyDeviceToHost, st));
    FDN_HIP(hipStreamSynchronize(st));
    memcpy(dst, stage + 3 * HW, HW * 4);
    return 0;
}

FDN_API int fdn_warp(fdn_handle h, const float* reference, const float* flow, float* dst, int H, int W)
{
    return fdn_warp_strided(h, reference, W, 1, flow, dst, H, W);
}

FDN_API int fdn_farneback_typed(fdn_handle h, const void* prev, int prev_depth, ptrdiff_t prev_rs, ptrdiff_t prev_cs,
                                const void* next, int next_depth, ptrdiff_t next_rs, ptrdiff_t next_cs, float* flow_io, int H, int W,
                                int levels, int winsize, int iters, int poly_n, double poly_sigma, int flags)
{
    FDN_ENTER(h);
    if (!prev || !next || !flow_io) return fail("NULL image/flow pointer");
    if (check_pair_args(H, W, levels, winsize, iters, poly_n, poly_sigma, flags)) return -1;
    const size_t HW = (size_t)H * W;
    hipStream_t st = h->stream;
    PairScratch ps;
    if (pair_scratch(h, H, W, &ps)) return -1;
    if (ensure_pinned(h, HW * 4 * 4)) return -1;
    float* stage = (float*)h->pinned;
    {
        ScopedTimer t(h, FDN_TIMER_TRANSFER);
        // calc(): img->convertTo(fimg, CV_32F) -- whatever the two depths are
        if (gather_host_depth(stage, prev, prev_depth, prev_rs, prev_cs, H, W)) return -1;
        if (gather_host_depth(stage + HW, next, next_depth, next_rs, next_cs, H, W)) return -1;
        FDN_HIP(hipMemcpyAsync(ps.img, stage, HW * 8, hipMemcpyHostToDevice, st));
        if (flags & FDN_USE_INITIAL_FLOW) {
            memcpy(stage + 2 * HW, flow_io, HW * 8);
            FDN_HIP(hipMemcpyAsync(ps.flow, stage + 2 * HW, HW * 8, hipMemcpyHostToDevice, st));
        }
    }
    if (farneback_pair_dev(h, ps.img, ps.R, ps.flow, ps.M0, ps.M1, H, W, levels, winsize, iters, poly_n, poly_sigma, flags)) return -1;
    {
        ScopedTimer t(h, FDN_TIMER_TRANSFER);
        FDN_HIP(hipMemcpyAsync(stage + 2 * HW, ps.flow, HW * 8, hipMemcpyDeviceToHost, st));
        FDN_HIP(hipStreamSynchronize(st));
        memcpy(flow_io, stage + 2 * HW, HW * 8);
    }
    return 0;
}

FDN_API int fdn_warp_typed(fdn_handle h, const void* reference, int depth, ptrdiff_t rs, ptrdiff_t cs, const float* flow, void* dst, int H, int W)
{
    if (depth == FDN_DEPTH_F32) return fdn_warp_strided(h, (const float*)reference, rs, cs, flow, (float*)dst, H, W);
    FDN_ENTER(h);
    if (check_warp_args(reference, flow, dst, H, W)) return -1;
    if (depth == FDN_DEPTH_I8) return fail("cv2.remap does not support 8-bit signed images (the reference raises there)");
    if (depth != FDN_DEPTH_F64 && depth != FDN_DEPTH_I16 && depth != FDN_DEPTH_U16 && depth != FDN_DEPTH_U8) return fail("unknown image depth %d (FDN_DEPTH_*)", depth);
    const size_t HW = (size_t)H * W;
    hipStream_t st = h->stream;
    if (depth == FDN_DEPTH_F64) {      // remapBilinear<Cast<double, double>, ., float>: doubles in, doubles out
        if (ensure(h, h->pair, HW * 8 * 3)) return -1;
        if (ensure_pinned(h, HW * 8 * 3)) return -1;
        double* d_src = (double*)h->pair.p;
        double* d_dst = d_src + HW;
        float* d_flow = (float*)(d_dst + HW);
        double* stage = (double*)h->pinned;
        {
            ScopedTimer t(h, FDN_TIMER_TRANSFER);
            gather_host_as<double, double>(stage, reference, rs, cs, H, W);
            memcpy(stage + 2 * HW, flow, HW * 8);
            FDN_HIP(hipMemcpyAsync(d_src, stage, HW * 8, hipMemcpyHostToDevice, st));
            FDN_HIP(hipMemcpyAsync(d_flow, stage + 2 * HW, HW * 8, hipMemcpyHostToDevice, st));
        }
        launch_warp_f64(d_src, d_flow, d_dst, H, W, st, h->tn.remap_model);
        FDN_HIP(hipGetLastError());
        ScopedTimer t(h, FDN_TIMER_TRANSFER);
        FDN_HIP(hipMemcpyAsync(stage + HW, d_dst, HW * 8, hipMemcpyDeviceToHost, st));
        FDN_HIP(hipStreamSynchronize(st));
        memcpy(dst, stage + HW, HW * 8);
        return 0;
    }
    // 16-bit integers: remapBilinear<Cast<float, T>>: float arithmetic on the (exactly converted) values, then
    // saturate_cast<T>(float) = cvRound (half to even), clamped to the type's range; uint8: fixed point (remap_finish_u8)
    if (ensure(h, h->pair, HW * 4 * 4)) return -1;
    if (ensure_pinned(h, HW * 4 * 4)) return -1;
    float* d_src = (float*)h->pair.p;
    float* d_flow = d_src + HW;
    float* d_dst = d_flow + 2 * HW;
    float* stage = (float*)h->pinned;
    {
        ScopedTimer t(h, FDN_TIMER_TRANSFER);
        if (gather_host_depth(stage, reference, depth, rs, cs, H, W)) return -1;
        memcpy(stage + HW, flow, HW * 8);
        FDN_HIP(hipMemcpyAsync(d_src, stage, HW * 12, hipMemcpyHostToDevice, st));
    }
    if (depth == FDN_DEPTH_U8) launch_warp_u8(d_src, d_flow, d_dst, H, W, st);     // 8-bit fixed-point interpolation: integers already
    else launch_warp(d_src, d_flow, d_dst, H, W, st, h->tn.remap_model);
    FDN_HIP(hipGetLastError());
    ScopedTimer t(h, FDN_TIMER_TRANSFER);
    FDN_HIP(hipMemcpyAsync(stage + 3 * HW, d_dst, HW * 4, hipMemcpyDeviceToHost, st));
    FDN_HIP(hipStreamSynchronize(st));
    const float* res = stage + 3 * HW;
    if (depth == FDN_DEPTH_U8) {
        uint8_t* o = (uint8_t*)dst;
        for (size_t i = 0; i < HW; i++) o[i] = (uint8_t)res[i];
    } else if (depth == FDN_DEPTH_I16) {
        int16_t* o = (int16_t*)dst;
        for (size_t i = 0; i < HW; i++) o[i] = (int16_t)fminf(fmaxf(rintf(res[i]), -32768.f), 32767.f);
    } else {
        uint16_t* o = (uint16_t*)dst;
        for (size_t i = 0; i < HW; i++) o[i] = (uint16_t)fminf(fmaxf(rintf(res[i]), 0.f), 65535.f);
    }
    return 0;
}

FDN_API int fdn_warp_dev(fdn_handle h, const float* d_reference, ptrdiff_t rs, ptrdiff_t cs, const float* d_flow, float* d_dst, int H, int W)
{
    FDN_ENTER(h);
    if (check_warp_args(d_reference, d_flow, d_dst, H, W)) return -1;
    const float* src = d_reference;
    if (rs != W || cs != 1) {
        if (ensure(h, h->pair, (size_t)H * W * 4)) return -1;
        launch_copy_strided(d_reference, rs, cs, (float*)h->pair.p, H, W, h->stream);
        src = (const float*)h->pair.p;
    }
    if (src == d_dst) return fail("reference and dst must not alias");
    launch_warp(src, d_flow, d_dst, H, W, h->stream, h->tn.remap_model);
    FDN_HIP(hipGetLastError());
    return 0;
}

FDN_API int fdn_filter_axis_dev(fdn_handle h, const float* d_in, float* d_out, int Z, int Y, int X, int axis,
                                const double* kernel, int K, float pad_value, const fdn_sweep_params* p)
{
    FDN_ENTER(h);
    if (!d_in || !d_out || !kernel) return fail("NULL pointer");
    return filter_axis_dev(h, d_in, d_out, Z, Y, X, axis, kernel, K, pad_value, p);
}

FDN_API int fdn_filter_3d_dev(fdn_handle h, const float* d_in, float* d_out, int Z, int Y, int X,
                              const double* const kernels[3], const int K[3], float pad_value, const fdn_sweep_params* p)
{
    FDN_ENTER(h);
    if (!d_in || !d_out) return fail("NULL pointer");
    return filter_3d_dev(h, d_in, d_out, Z, Y, X, kernels, K, pad_value, p);
}

static int host_roundtrip(fdn_ctx* h, const float* in, float* out, size_t count, float** d_in, float** d_out)
{
    if (ensure(h, h->vol_b, count * sizeof(float) * 2)) return -1;
    *d_in = (float*)h->vol_b.p;
    *d_out = *d_in + count;
    ScopedTimer t(h, FDN_TIMER_TRANSFER);
    (void)out;
    return copy_host(h, *d_in, in, count * sizeof(float), true);
}

FDN_API int fdn_filter_axis(fdn_handle h, const float* in, float* out, int Z, int Y, int X, int axis,
                            const double* kernel, int K, float pad_value, const fdn_sweep_params* p)
{
    FDN_ENTER(h);
    if (!in || !out || !kernel) return fail("NULL pointer");
    if (Z <= 0 || Y <= 0 || X <= 0) return fail("bad volume dims");
    size_t count = (size_t)Z * Y * X;
    // vol_b holds [in | out]; filter_3d_dev's ping-pong never uses vol_b for a single axis
    float *d_in, *d_out;
    if (host_roundtrip(h, in, out, count, &d_in, &d_out)) return -1;
    if (filter_axis_dev(h, d_in, d_out, Z, Y, X, axis, kernel, K, pad_value, p)) return -1;
    ScopedTimer t(h, FDN_TIMER_TRANSFER);
    return copy_host(h, out, d_out, count * sizeof(float), false);
}

FDN_API int fdn_filter_3d(fdn_handle h, const float* in, float* out, int Z, int Y, int X,
                          const double* const kernels[3], const int K[3], float pad_value, const fdn_sweep_params* p)
{
    FDN_ENTER(h);
    if (!in || !out) return fail("NULL pointer");
    if (Z <= 0 || Y <= 0 || X <= 0) return fail("bad volume dims");
    const size_t bytes = (size_t)Z * Y * X * sizeof(float);
    // device copies of the volume live in the handle and are reused by the next call
    if (ensure(h, h->vol_in, bytes) || ensure(h, h->vol_out, bytes)) return -1;
    {   // (copy_host: page-locked for the call when the volume is 8 MB or more -- one DMA at PCIe speed --, bounced otherwise)
        ScopedTimer t(h, FDN_TIMER_TRANSFER);
        if (copy_host(h, h->vol_in.p, in, bytes, true)) return -1;
    }
    if (filter_3d_dev(h, (const float*)h->vol_in.p, (float*)h->vol_out.p, Z, Y, X, kernels, K, pad_value, p)) {
        (void)hipStreamSynchronize(h->stream);
        return -1;
    }
    {
        ScopedTimer t(h, FDN_TIMER_TRANSFER);
        if (copy_host(h, out, h->vol_out.p, bytes, false)) return -1;
    }
    // the two whole-volume device copies of a host-pointer call go back when they are large (2 x the volume on top of
    // the pass's own buffers would otherwise stay with a process-wide handle: other handles, torch, ranks sharing the
    // GPU need that memory); small volumes keep them, so that repeated calls do not allocate
    if (bytes >= ((size_t)256 << 20)) { release(h->vol_in); release(h->vol_out); }
    return 0;
}

FDN_API int fdn_sum_dev(fdn_handle h, const float* d_in, size_t count, double* sum_out)
{
    FDN_ENTER(h);
    if (!d_in || !sum_out) return fail("NULL pointer");
    const int MAXB = 4096;
    if (ensure(h, h->partials, MAXB * sizeof(double))) return -1;
    int nb = launch_sum_partials(d_in, count, (double*)h->partials.p, MAXB, h->stream);
    FDN_HIP(hipGetLastError());
    std::vector<double> host(nb);
    if (copy_host(h, host.data(), h->partials.p, nb * sizeof(double), false)) return -1;
    double s = 0;
    for (int i = 0; i < nb; i++) s += host[i];
    *sum_out = s;
    return 0;
}

FDN_API int fdn_reserve_3d(fdn_handle h, int Z, int Y, int X, const int K[3], const fdn_sweep_params* p)
{
    FDN_ENTER(h);
    if (!K) return fail("K is NULL");
    if (Z <= 0 || Y <= 0 || X <= 0) return fail("bad volume dims");
    if (K[0] <= 0 && K[1] <= 0 && K[2] <= 0) return 0;
    static const double one = 1.0;
    const double* kern[3] = {K[0] > 0 ? &one : nullptr, K[1] > 0 ? &one : nullptr, K[2] > 0 ? &one : nullptr};   // never read
    float* const fake_in = (float*)(uintptr_t)16;    // never dereferenced: the sizing logic only compares them
    float* const fake_out = (float*)(uintptr_t)32;
    // The flow batch of a pass is sized from the free device memory (sweep_stack), and a reservation runs BEFORE the caller
    // allocates its own input and output volumes (the CLI reserves while it reads the file): without their 8 bytes per voxel
    // set aside here a volume whose flows do not all fit would get a batch that leaves no room for them.
    h->reserve_only = true;
    h->reserve_extern = (size_t)2 * Z * Y * X * sizeof(float);
    const int rc = filter_3d_dev(h, fake_in, fake_out, Z, Y, X, kern, K, 0.f, p);
    h->reserve_only = false;
    h->reserve_extern = 0;
    return rc;
}

FDN_API int fdn_reserve_stack(fdn_handle h, int S, int H, int W, int K, const fdn_sweep_params* p)
{
    FDN_ENTER(h);
    if (S <= 0 || H <= 0 || W <= 0 || K <= 0) return fail("bad stack dims S=%d H=%d W=%d K=%d", S, H, W, K);
    std::vector<double> taps((size_t)K, 1.0);          // never read: nothing is launched
    float* const fake = (float*)(uintptr_t)16;
    h->reserve_only = true;
    const int rc = sweep_stack(h, fake, fake, S, H, W, taps.data(), K, p);
    h->reserve_only = false;
    return rc;
}

FDN_API int fdn_stats_dev(fdn_handle h, const float* d_in, size_t count, double* out4)
{
    FDN_ENTER(h);
    if (!d_in || !out4) return fail("NULL pointer");
    if (!count) return fail("empty volume");
    const int MAXB = 4096;
    if (ensure(h, h->partials, (size_t)MAXB * 4 * sizeof(double))) return -1;
    std::vector<double> host((size_t)MAXB * 4);
    double mn = 0, mx = 0, sum = 0, mean = 0, sq = 0;
    for (int pass = 0; pass < 2; pass++) {          // pass 0: min, max, sum; pass 1: squared deviations from the mean
        const int nb = launch_stats_partials(d_in, count, mean, (double*)h->partials.p, MAXB, h->stream);
        FDN_HIP(hipGetLastError());
        if (copy_host(h, host.data(), h->partials.p, (size_t)nb * 4 * sizeof(double), false)) return -1;
        mn = host[0]; mx = host[1]; sum = 0; sq = 0;
        for (int i = 0; i < nb; i++) {
            const double a = host[4 * i], c = host[4 * i + 1];
            mn = (a < mn || a != a) ? a : mn; mx = (c > mx || c != c) ? c : mx;      // a NaN extreme propagates, as numpy's does
            sum += host[4 * i + 2]; sq += host[4 * i + 3];
        }
        mean = sum / (double)count;
    }
    out4[0] = mn; out4[1] = mx; out4[2] = mean; out4[3] = sqrt(sq / (double)count);
    return 0;
}

FDN_API int fdn_stats_slices_dev(fdn_handle h, const float* d_in, int nslices, size_t slice_elems, double centre, double* out)
{
    FDN_ENTER(h);
    if (!d_in || !out) return fail("NULL pointer");
    if (nslices <= 0 || !slice_elems) return fail("empty volume");
    const int B = FDN_STATS_BLOCKS_PER_SLICE;
    const int step = 4096;                         // slices per launch (the grid's y extent is limited to 65535)
    if (ensure(h, h->partials, (size_t)std::min(nslices, step) * B * 4 * sizeof(double))) return -1;
    std::vector<double> host((size_t)std::min(nslices, step) * B * 4);
    for (int s0 = 0; s0 < nslices; s0 += step) {
        const int n = std::min(step, nslices - s0);
        launch_stats_slices(d_in + (size_t)s0 * slice_elems, n, slice_elems, centre, (double*)h->partials.p, h->stream);
        FDN_HIP(hipGetLastError());
        if (copy_host(h, host.data(), h->partials.p, (size_t)n * B * 4 * sizeof(double), false)) return -1;
        for (int s = 0; s < n; s++) {
            const double* p = host.data() + (size_t)s * B * 4;
            double mn = p[0], mx = p[1], sum = 0, sq = 0;
            for (int b = 0; b < B; b++) {
                const double a = p[4 * b], c = p[4 * b + 1];
                mn = (a < mn || a != a) ? a : mn; mx = (c > mx || c != c) ? c : mx;
                sum += p[4 * b + 2]; sq += p[4 * b + 3];
            }
            double* o = out + (size_t)(s0 + s) * 4;
            o[0] = mn; o[1] = mx; o[2] = sum; o[3] = sq;
        }
    }
    return 0;
}

FDN_API int fdn_device_count(int* count_out)
{
    if (!count_out) return fail("count_out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count_out = n;
    return 0;
}

FDN_API int fdn_device_pci_id(int device, char* buf, int cap)
{
    if (!buf || cap < 1) return fail("buf is NULL");
    buf[0] = 0;
    hipError_t e = hipDeviceGetPCIBusId(buf, cap, device);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail("hipDeviceGetPCIBusId(%d): %s", device, hipGetErrorString(e)); }
    return 0;
}

FDN_API int fdn_convert_dev(fdn_handle h, const void* d_src, int depth, float* d_dst, size_t count)
{
    FDN_ENTER(h);
    if (!d_src || !d_dst) return fail("NULL pointer");
    if (depth != FDN_DEPTH_I16 && depth != FDN_DEPTH_U16 && depth != FDN_DEPTH_I8 && depth != FDN_DEPTH_U8)
        return fail("fdn_convert_dev converts 8- and 16-bit integer volumes (depth %d)", depth);
    launch_convert_f32(d_src, depth, d_dst, count, h->stream);
    FDN_HIP(hipGetLastError());
    return 0;
}

FDN_API int fdn_truncate_dev(fdn_handle h, const float* d_src, int depth, void* d_dst, size_t count)
{
    FDN_ENTER(h);
    if (!d_src || !d_dst) return fail("NULL pointer");
    if (depth != FDN_DEPTH_U16 && depth != FDN_DEPTH_U8) return fail("fdn_truncate_dev writes uint8 or uint16 (depth %d)", depth);
    launch_truncate_from_f32(d_src, depth, d_dst, count, h->stream);
    FDN_HIP(hipGetLastError());
    return 0;
}

FDN_API int fdn_mean_host(const float* in, size_t count, float* mean_out)
{
    if (!in || !mean_out) return fail("NULL pointer");
    if (!count) return fail("empty volume");
    // the reduction runs through numpy's buffered iterator: pairwise sums of 8192-element chunks,
    // accumulated left to right in float32
    // The chunk sums are independent of one another: large arrays compute them on several threads (a 2 GiB volume costs
    // one core 0.25 s), the left-to-right float32 accumulation of the sums stays serial -- the same bits either way.
    const size_t nchunks = (count + 8191) / 8192;
    std::vector<float> sums(nchunks);
    unsigned nthr = count >= ((size_t)1 << 22) ? std::min(16u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
    if (const char* e = getenv("FDN_HOST_THREADS")) nthr = (unsigned)std::max(1, atoi(e));
    auto work = [&](size_t c0, size_t c1) {
        for (size_t c = c0; c < c1; c++) sums[c] = np_pairwise_sum_f32(in + c * 8192, std::min<size_t>(8192, count - c * 8192));
    };
    if (nthr <= 1) work(0, nchunks);
    else {
        std::vector<std::thread> pool;
        const size_t per = (nchunks + nthr - 1) / nthr;
        for (unsigned t = 0; t < nthr; t++) {
            const size_t c0 = std::min(nchunks, (size_t)t * per), c1 = std::min(nchunks, c0 + per);
            if (c0 < c1) pool.emplace_back(work, c0, c1);
        }
        for (auto& th : pool) th.join();
    }
    float tot = 0.f;
    for (size_t c = 0; c < nchunks; c++) tot += sums[c];
    *mean_out = tot / (float)count;
    return 0;
}

// The per-chunk sums of that reduction for a DEVICE array: sums_out[c] (host, ceil(count / 8192) values)
// = numpy's pairwise sum of chunk c; the last chunk may be partial.
FDN_API int fdn_np_chunk_sums_dev(fdn_handle h, const float* d_in, size_t count, float* sums_out)
{
    FDN_ENTER(h);
    if (!d_in || !sums_out) return fail("NULL pointer");
    if (!count) return fail("empty volume");
    size_t full = count / 8192, rest = count % 8192;
    if (full) {
        if (ensure(h, h->partials, full * sizeof(float))) return -1;
        launch_np_chunk_sums(d_in, full, (float*)h->partials.p, h->stream);
        FDN_HIP(hipGetLastError());
        if (copy_host(h, sums_out, h->partials.p, full * sizeof(float), false)) return -1;
    }
    std::vector<float> tail(rest);
    if (rest && copy_host(h, tail.data(), d_in + full * 8192, rest * sizeof(float), false)) return -1;
    if (rest) sums_out[full] = np_pairwise_sum_f32(tail.data(), rest);
    return 0;
}

FDN_API int fdn_mean_dev(fdn_handle h, const float* d_in, size_t count, float* mean_out)
{
    if (!mean_out) return fail("NULL pointer");
    if (!count) return fail("empty volume");
    std::vector<float> sums((count + 8191) / 8192);
    if (fdn_np_chunk_sums_dev(h, d_in, count, sums.data())) return -1;
    float tot = 0.f;
    for (float v : sums) tot += v;     // numpy accumulates its buffered chunks left to right in float32
    *mean_out = tot / (float)count;
    return 0;
}

FDN_API int fdn_filter_3d_sharded(fdn_handle h, const float* d_slab_in, float* d_slab_out, int Z, int Y, int X,
                                  const double* const kernels[3], const int K[3], const fdn_sweep_params* p, const fdn_comm* comm)
{
    FDN_ENTER(h);
    if (!d_slab_in || !d_slab_out || !kernels || !K || !p || !comm) return fail("NULL pointer");
    if (!comm->exchange || !comm->allgather_host) return fail("fdn_comm needs both callbacks");
    if (Z <= 0 || Y <= 0 || X <= 0) return fail("bad volume dims");
    if (d_slab_in == d_slab_out) return fail("in and out must not alias");
    return filter_3d_sharded(h, d_slab_in, d_slab_out, Z, Y, X, kernels, K, p, comm);
}

FDN_API int fdn_sweep_stack_dev(fdn_handle h, const float* d_stack, float* d_out, int S, int H, int W,
                                const double* kernel, int K, const fdn_sweep_params* p)
{
    FDN_ENTER(h);
    if (!d_stack || !d_out || !kernel) return fail("NULL pointer");
    return sweep_stack(h, d_stack, d_out, S, H, W, kernel, K, p);
}

FDN_API int fdn_permute_dev(fdn_handle h, const float* d_in, float* d_out, int A, int B, int C,
                            int64_t sa, int64_t sb, int64_t sc)
{
    FDN_ENTER(h);
    if (!d_in || !d_out) return fail("NULL pointer");
    if (A <= 0 || B <= 0 || C <= 0) return fail("bad dims");
    if (sa != 1 && sb != 1 && sc != 1) return fail("one input stride must be 1");
    ScopedTimer t(h, FDN_TIMER_PERMUTE);
    launch_permute(d_in, d_out, A, B, C, sa, sb, sc, h->stream);
    FDN_HIP(hipGetLastError());
    return 0;
}

FDN_API int fdn_enable_timers(fdn_handle h, int on)
{
    FDN_ENTER(h);
    h->timers = on != 0;
    return 0;
}
FDN_API int fdn_add_timer(fdn_handle h, int which, double ms, long long count)
{
    FDN_ENTER(h);
    if (which < 0 || which >= FDN_TIMER_COUNT) return fail("timer category %d out of range", which);
    h->tms[which] += ms;
    h->tcount[which] += count;
    return 0;
}
FDN_API int fdn_get_timers(fdn_handle h, double* ms_out, long long* count_out, int reset)
{
    FDN_ENTER(h);
    resolve_stamps(h);
    if (ms_out) for (int i = 0; i < FDN_TIMER_COUNT; i++) ms_out[i] = h->tms[i];
    if (count_out) for (int i = 0; i < FDN_TIMER_COUNT; i++) count_out[i] = h->tcount[i];
    if (reset) for (int i = 0; i < FDN_TIMER_COUNT; i++) { h->tms[i] = 0; h->tcount[i] = 0; }
    return 0;
}

} // extern "C"
