// fdn_rccl.hip -- libflowdn_rccl.so (include/flowdn_rccl.h): the transports behind fdn_filter_3d_sharded's two callbacks.
//   RCCL: ncclSend / ncclRecv inside one group per exchange, on the caller's stream, nothing waits on the host;
//   SHM:  N ranks sharing one GPU (rehearsal), staged through shared-memory files;
//   NULL: moves nothing (per-rank overhead emulation).
// Host code only; built with hipcc for the HIP runtime API and linked against librccl.
#include "../../include/flowdn_rccl.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <future>
#include <memory>
#include <thread>
#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

namespace {

thread_local std::string g_err;

int fail(const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
}

#define T_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail("%s: %s", #call, hipGetErrorString(e_)); } while (0)
#define T_NCCL(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return fail("%s: %s", #call, ncclGetErrorString(r_)); } while (0)

double now_s()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double rdv_timeout()
{
    const char* e = getenv("FDN_RDV_TIMEOUT");
    const double v = e ? atof(e) : 0.;
    return v > 0. ? v : 600.;
}

void nap(int spins)
{
    if (spins < 200) sched_yield();
    else { timespec ts = {0, spins < 2000 ? 50 * 1000 : 1000 * 1000}; nanosleep(&ts, nullptr); }
}

// ---- shared-memory pieces of the SHM transport --------------------------------------------------------------------
struct Ctl {                                   // <dir>/ctl, one page, zero-filled by ftruncate
    std::atomic<uint32_t> arrived;
    std::atomic<uint32_t> generation;
    std::atomic<uint32_t> failed;              // a rank that gives up says so: the others stop waiting
};

struct OutHeader {                             // start of <dir>/out.<rank>: where this rank's bytes for each peer begin
    uint64_t off[64];                          // world <= 64
    uint64_t bytes[64];
};
constexpr size_t kHeaderBytes = 4096;
static_assert(sizeof(OutHeader) <= kHeaderBytes, "header");

struct Mapping {
    int fd = -1;
    char* p = nullptr;
    size_t len = 0;
    void close_()
    {
        if (p) munmap(p, len);
        if (fd >= 0) close(fd);
        p = nullptr; fd = -1; len = 0;
    }
};

} // namespace

struct fdn_transport {
    int kind = FDN_TRANSPORT_NULL, rank = 0, world = 1, device = 0;
    fdn_comm comm{};
    std::string what;
    // RCCL
    ncclComm_t nccl = nullptr;
    hipStream_t side = nullptr;                // stream of the host all-gather
    hipEvent_t order = nullptr;                // ... ordered behind the last exchange: recorded on the exchange's stream right after its
    bool have_order = false;                   // ncclGroupEnd (the stream itself is the caller's and may be gone by the next gather)
    bool failed = false;                       // the communicator was aborted (a failed init, fdn_transport_abort): destroy waits for nothing
    char* stage = nullptr;                     // device staging of the host all-gather
    char* hstage = nullptr;                    // ... and its page-locked host side (the caller's buffers are never handed to the copy engines)
    size_t stage_cap = 0;
    int count = 0;                             // ranks the communicator itself reports (ncclCommCount; SHM: ranks met at the first barrier)
    std::string device_id;                     // PCI bus id of this rank's device ("host": no device)
    // SHM
    std::string dir;
    Ctl* ctl = nullptr;
    Mapping out;                               // my outbox (read-write)
    std::vector<Mapping> in;                   // peers' outboxes (read-only)
    uint32_t my_gen = 0;
    void* bounce = nullptr;                    // page-locked bounce buffer (hipHostMalloc) between the device and the mappings
};

namespace {

int wait_for_file(const std::string& path, size_t min_size)
{
    const double t0 = now_s(), limit = rdv_timeout();
    for (int spins = 0;; spins++) {
        struct stat st;
        if (stat(path.c_str(), &st) == 0 && (size_t)st.st_size >= min_size) return 0;
        if (now_s() - t0 > limit) return fail("rendezvous: %s did not appear within %.0f s (a rank died or was never started?)", path.c_str(), limit);
        nap(spins + 200);
    }
}

int write_file_atomically(const std::string& path, const void* data, size_t bytes)
{
    const std::string tmp = path + ".tmp";
    int fd = open(tmp.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0600);
    if (fd < 0) return fail("cannot create %s: %s", tmp.c_str(), strerror(errno));
    const ssize_t w = write(fd, data, bytes);
    close(fd);
    if (w != (ssize_t)bytes) return fail("short write to %s", tmp.c_str());
    if (rename(tmp.c_str(), path.c_str())) return fail("rename %s: %s", path.c_str(), strerror(errno));
    return 0;
}

// ---- RCCL ---------------------------------------------------------------------------------------------------------
// wait for what is enqueued on `st`, at most `limit` seconds: a peer that never posts its half of an exchange must end in
// an error here, not in a process that hangs until somebody's outer timeout
int wait_stream(hipStream_t st, double limit, const char* what)
{
    const double t0 = now_s();
    for (int spins = 0;; spins++) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) return fail("%s: %s", what, hipGetErrorString(e));
        if (now_s() - t0 > limit) return fail("%s did not complete within %.0f s (a peer is missing or the links are down)", what, limit);
        nap(spins);
    }
}

double sanity_timeout()
{
    const char* e = getenv("FDN_RCCL_SANITY_TIMEOUT");
    const double v = e ? atof(e) : 0.;
    return v > 0. ? v : 120.;
}

int rccl_init(fdn_transport* t, const char* rendezvous)
{
    static std::atomic<int> seq{0};            // several communicators of one job: all ranks create them in the same order
    const int my_seq = seq.fetch_add(1);
    ncclUniqueId id;
    memset(&id, 0, sizeof id);
    if (t->world > 1 && (!rendezvous || !*rendezvous)) return fail("FDN_TRANSPORT_RCCL with %d ranks needs a rendezvous directory", t->world);
    const std::string path = std::string(rendezvous ? rendezvous : "") + "/rccl_uid." + std::to_string(my_seq);
    if (t->rank == 0) {
        T_NCCL(ncclGetUniqueId(&id));
        if (t->world > 1 && write_file_atomically(path, &id, sizeof id)) return -1;
    } else {
        if (wait_for_file(path, sizeof id)) return -1;
        FILE* f = fopen(path.c_str(), "rb");
        if (!f || fread(&id, 1, sizeof id, f) != sizeof id) { if (f) fclose(f); return fail("cannot read %s", path.c_str()); }
        fclose(f);
    }
    if (t->world == 1) T_NCCL(ncclCommInitRank(&t->nccl, 1, id, 0));
    else {
        // ncclCommInitRank blocks until every rank has joined and has no timeout of its own: it runs in a helper thread that
        // is given rdv_timeout() seconds (a rank that died, IPC handles the driver refuses ...).  On a timeout the thread is
        // left behind -- the caller is expected to end the process (flowdenoising_amd/launch.py does, with os._exit).
        struct Shared { ncclComm_t comm = nullptr; ncclResult_t res = ncclSuccess; hipError_t dev = hipSuccess; };
        auto sh = std::make_shared<Shared>();
        std::promise<void> done;
        std::future<void> fut = done.get_future();
        const int world = t->world, rank = t->rank, device = t->device;
        std::thread th([sh, id, world, rank, device](std::promise<void> p) {
            sh->dev = hipSetDevice(device);
            if (sh->dev == hipSuccess) sh->res = ncclCommInitRank(&sh->comm, world, id, rank);
            p.set_value();
        }, std::move(done));
        const double limit = rdv_timeout();
        if (fut.wait_for(std::chrono::duration<double>(limit)) != std::future_status::ready) {
            th.detach();
            return fail("ncclCommInitRank (rank %d of %d, device %d) did not return within %.0f s: a rank is missing, or RCCL cannot "
                        "share memory between the ranks' processes (HSA_ENABLE_IPC_MODE_LEGACY=0 must be set on this host driver)",
                        rank, world, device, limit);
        }
        th.join();
        if (sh->dev != hipSuccess) return fail("hipSetDevice(%d): %s", device, hipGetErrorString(sh->dev));
        if (sh->res != ncclSuccess) return fail("ncclCommInitRank: %s", ncclGetErrorString(sh->res));
        t->nccl = sh->comm;
    }
    // From here on a communicator exists.  Whatever fails below leaves it ABORTED (ncclCommAbort stops its kernels and makes
    // the pending operations of the first exchange return), marked `failed`: fdn_transport_destroy then neither waits for the
    // side stream nor calls ncclCommDestroy on a communicator with work in flight -- either could hang for ever, which is what
    // the deadlines here exist to prevent.
    struct AbortOnFailure {
        fdn_transport* t; bool armed = true;
        ~AbortOnFailure() { if (armed && t->nccl) { (void)ncclCommAbort(t->nccl); t->nccl = nullptr; t->failed = true; } }
    } guard{t};
    T_HIP(hipStreamCreateWithFlags(&t->side, hipStreamNonBlocking));
    T_HIP(hipEventCreateWithFlags(&t->order, hipEventDisableTiming));
    T_NCCL(ncclCommCount(t->nccl, &t->count));
    if (t->count != t->world) return fail("the communicator reports %d ranks, %d expected", t->count, t->world);
    if (t->world > 1) {
        // a first, small exchange under a deadline: every rank sends its rank number round a ring -- the calls of a real
        // exchange (grouped ncclSend / ncclRecv); links or IPC that do not work show here, in seconds, with a message
        int* buf = nullptr;
        T_HIP(hipMalloc((void**)&buf, 512));
        const int mine = t->rank, next = (t->rank + 1) % t->world, prev = (t->rank + t->world - 1) % t->world;
        int got = -1;
        T_HIP(hipMemcpy(buf, &mine, sizeof mine, hipMemcpyHostToDevice));
        T_NCCL(ncclGroupStart());
        const ncclResult_t r1 = ncclSend(buf, 64, ncclInt8, next, t->nccl, t->side);
        const ncclResult_t r2 = ncclRecv((char*)buf + 256, 64, ncclInt8, prev, t->nccl, t->side);
        const ncclResult_t r3 = ncclGroupEnd();
        if (r1 != ncclSuccess || r2 != ncclSuccess || r3 != ncclSuccess)
            return fail("first exchange: %s", ncclGetErrorString(r1 != ncclSuccess ? r1 : r2 != ncclSuccess ? r2 : r3));
        if (wait_stream(t->side, sanity_timeout(), "the first ncclSend / ncclRecv exchange between the ranks")) return -1;
        T_HIP(hipMemcpy(&got, (char*)buf + 256, sizeof got, hipMemcpyDeviceToHost));
        (void)hipFree(buf);
        if (got != prev) return fail("first exchange: rank %d received %d from rank %d", t->rank, got, prev);
    }
    int ver = 0;
    (void)ncclGetVersion(&ver);
    char buf[160];
    snprintf(buf, sizeof buf, "rccl %d.%d.%d, rank %d of %d, device %d", ver / 10000, (ver / 100) % 100, ver % 100, t->rank, t->world, t->device);
    t->what = buf;
    guard.armed = false;
    return 0;
}

int rccl_exchange(fdn_transport* t, int n, const fdn_msg* msgs, hipStream_t st)
{
    if (n <= 0) return 0;
    for (int i = 0; i < n; i++)
        if (msgs[i].peer < 0 || msgs[i].peer >= t->world) return fail("message %d: peer %d outside 0..%d", i, msgs[i].peer, t->world - 1);
    T_NCCL(ncclGroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int i = 0; i < n && bad == ncclSuccess; i++) {
        if (!msgs[i].bytes) continue;
        bad = msgs[i].is_send ? ncclSend(msgs[i].d_buf, msgs[i].bytes, ncclInt8, msgs[i].peer, t->nccl, st)
                              : ncclRecv(msgs[i].d_buf, msgs[i].bytes, ncclInt8, msgs[i].peer, t->nccl, st);
    }
    const ncclResult_t end = ncclGroupEnd();
    if (bad != ncclSuccess) return fail("ncclSend/ncclRecv: %s", ncclGetErrorString(bad));
    if (end != ncclSuccess) return fail("ncclGroupEnd: %s", ncclGetErrorString(end));
    // what a later host all-gather has to wait for: recorded here, on the stream the exchange was enqueued on, while that
    // stream certainly exists (the caller may destroy it -- close its fdn handle -- before the next gather or barrier)
    T_HIP(hipEventRecord(t->order, st));
    t->have_order = true;
    return 0;
}

// One communicator, two streams (the caller's for the exchanges, `side` for this host-level gather): RCCL wants the operations
// of a communicator issued in one order on every rank, so the gather is ordered behind whatever the last exchange left in
// flight on the caller's stream (an event there, waited for on `side`) -- and it has completed before this returns.
int rccl_allgather_host(fdn_transport* t, const void* send, void* recv, size_t bytes)
{
    if (!bytes) return 0;
    const size_t slot = (bytes + 255) & ~(size_t)255;           // `all` starts on a 256-byte boundary whatever `bytes` is
    const size_t need = slot + bytes * (size_t)t->world;
    if (t->stage_cap < need) {
        if (t->stage) { T_HIP(hipStreamSynchronize(t->side)); T_HIP(hipFree(t->stage)); T_HIP(hipHostFree(t->hstage)); t->stage = t->hstage = nullptr; t->stage_cap = 0; }
        T_HIP(hipMalloc((void**)&t->stage, need));
        T_HIP(hipHostMalloc((void**)&t->hstage, need, hipHostMallocDefault));
        t->stage_cap = need;
    }
    char* mine = t->stage;
    char* all = t->stage + slot;
    if (t->have_order) T_HIP(hipStreamWaitEvent(t->side, t->order, 0));
    memcpy(t->hstage, send, bytes);
    T_HIP(hipMemcpyAsync(mine, t->hstage, bytes, hipMemcpyHostToDevice, t->side));
    T_NCCL(ncclAllGather(mine, all, bytes, ncclInt8, t->nccl, t->side));
    T_HIP(hipMemcpyAsync(t->hstage + slot, all, bytes * (size_t)t->world, hipMemcpyDeviceToHost, t->side));
    T_HIP(hipStreamSynchronize(t->side));
    memcpy(recv, t->hstage + slot, bytes * (size_t)t->world);
    return 0;
}

// ---- SHM ----------------------------------------------------------------------------------------------------------
int shm_barrier(fdn_transport* t)
{
    Ctl* c = t->ctl;
    const uint32_t gen = c->generation.load(std::memory_order_acquire);
    if (c->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)t->world) {
        c->arrived.store(0, std::memory_order_relaxed);
        c->generation.store(gen + 1, std::memory_order_release);
        return 0;
    }
    const double t0 = now_s(), limit = rdv_timeout();
    for (int spins = 0; c->generation.load(std::memory_order_acquire) == gen; spins++) {
        if (c->failed.load(std::memory_order_relaxed)) return fail("shm transport: another rank failed");
        if ((spins & 1023) == 1023 && now_s() - t0 > limit) {
            c->failed.store(1);
            return fail("shm transport: rank %d waited %.0f s at a barrier (a rank died?)", t->rank, limit);
        }
        nap(spins);
    }
    return 0;
}

int shm_map(Mapping& m, const std::string& path, size_t len, bool writable)
{
    if (m.fd < 0) {
        m.fd = open(path.c_str(), writable ? (O_CREAT | O_RDWR) : O_RDONLY, 0600);
        if (m.fd < 0) return fail("cannot open %s: %s", path.c_str(), strerror(errno));
    }
    if (m.p && m.len >= len) return 0;
    if (m.p) { munmap(m.p, m.len); m.p = nullptr; m.len = 0; }
    if (writable) {
        len = std::max(len + len / 2, (size_t)1 << 20);
        if (ftruncate(m.fd, (off_t)len)) return fail("ftruncate %s to %zu: %s", path.c_str(), len, strerror(errno));
    } else {
        struct stat st;
        if (fstat(m.fd, &st)) return fail("fstat %s: %s", path.c_str(), strerror(errno));
        if ((size_t)st.st_size < len) return fail("%s holds %zu bytes, %zu expected", path.c_str(), (size_t)st.st_size, len);
        len = (size_t)st.st_size;
    }
    void* p = mmap(nullptr, len, writable ? (PROT_READ | PROT_WRITE) : PROT_READ, MAP_SHARED, m.fd, 0);
    if (p == MAP_FAILED) return fail("mmap %s (%zu bytes): %s", path.c_str(), len, strerror(errno));
    m.p = (char*)p;
    m.len = len;
    return 0;
}

std::string out_path(const fdn_transport* t, int r) { return t->dir + "/out." + std::to_string(r); }

// Device <-> a shared-memory mapping, through a page-locked bounce buffer of the transport's own.  The mappings are pageable
// memory that is unmapped and mapped again as the outboxes grow: handed to hipMemcpy directly, pieces above about 1 MB would
// be page-locked by the runtime on the fly, and such a registration outlives the mapping it was made for (the GPU memory
// fault of round 6, flowdenoising_amd/csrc/fdn_api.hip: copy_host).
constexpr size_t kBounceBytes = (size_t)4 << 20;
int shm_copy(fdn_transport* t, void* dst, const void* src, size_t bytes, bool to_device)
{
    if (!t->bounce) T_HIP(hipHostMalloc(&t->bounce, kBounceBytes, hipHostMallocDefault));
    for (size_t off = 0; off < bytes; off += kBounceBytes) {
        const size_t n = std::min(kBounceBytes, bytes - off);
        if (to_device) {
            memcpy(t->bounce, (const char*)src + off, n);
            T_HIP(hipMemcpy((char*)dst + off, t->bounce, n, hipMemcpyHostToDevice));
        } else {
            T_HIP(hipMemcpy(t->bounce, (const char*)src + off, n, hipMemcpyDeviceToHost));
            memcpy((char*)dst + off, t->bounce, n);
        }
    }
    return 0;
}

int shm_init(fdn_transport* t, const char* rendezvous)
{
    if (!rendezvous || !*rendezvous) return fail("FDN_TRANSPORT_SHM needs a rendezvous directory");
    if (t->world > 64) return fail("shm transport: at most 64 ranks");
    t->dir = rendezvous;
    const std::string ctl = t->dir + "/ctl";
    if (t->rank == 0) {
        const std::string tmp = ctl + ".tmp";
        int fd = open(tmp.c_str(), O_CREAT | O_TRUNC | O_RDWR, 0600);
        if (fd < 0) return fail("cannot create %s: %s", tmp.c_str(), strerror(errno));
        if (ftruncate(fd, 4096)) { close(fd); return fail("ftruncate %s: %s", tmp.c_str(), strerror(errno)); }
        close(fd);
        if (rename(tmp.c_str(), ctl.c_str())) return fail("rename %s: %s", ctl.c_str(), strerror(errno));
    } else if (wait_for_file(ctl, 4096)) return -1;
    int fd = open(ctl.c_str(), O_RDWR);
    if (fd < 0) return fail("cannot open %s: %s", ctl.c_str(), strerror(errno));
    void* p = mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return fail("mmap %s: %s", ctl.c_str(), strerror(errno));
    t->ctl = (Ctl*)p;
    t->in.resize(t->world);
    if (shm_map(t->out, out_path(t, t->rank), kHeaderBytes, true)) return -1;
    if (shm_barrier(t)) return -1;             // every outbox exists
    t->count = t->world;                       // (fdn_transport_create counts the distinct ranks it actually hears from)
    char buf[160];
    snprintf(buf, sizeof buf, "shm (host-staged, ranks share devices), rank %d of %d, device %d", t->rank, t->world, t->device);
    t->what = buf;
    return 0;
}

// publish `n` outgoing pieces (host or device memory), grouped by destination in call order; then every rank has published
int shm_publish(fdn_transport* t, int n, const void* const* src, const size_t* bytes, const int* peer, bool device_src)
{
    OutHeader hd;
    memset(&hd, 0, sizeof hd);
    for (int i = 0; i < n; i++) hd.bytes[peer[i]] += bytes[i];
    size_t off = kHeaderBytes;
    for (int j = 0; j < t->world; j++) { hd.off[j] = off; off += (hd.bytes[j] + 63) & ~(size_t)63; }
    if (shm_map(t->out, out_path(t, t->rank), off, true)) return -1;
    std::vector<size_t> cur(t->world);
    for (int j = 0; j < t->world; j++) cur[j] = hd.off[j];
    for (int i = 0; i < n; i++) {
        if (!bytes[i]) continue;
        if (device_src && t->device >= 0) { if (shm_copy(t, t->out.p + cur[peer[i]], src[i], bytes[i], false)) return -1; }
        else memcpy(t->out.p + cur[peer[i]], src[i], bytes[i]);
        cur[peer[i]] += bytes[i];
    }
    memcpy(t->out.p, &hd, sizeof hd);
    std::atomic_thread_fence(std::memory_order_release);
    return shm_barrier(t);
}

int shm_exchange(fdn_transport* t, int n, const fdn_msg* msgs, hipStream_t st)
{
    for (int i = 0; i < n; i++)
        if (msgs[i].peer < 0 || msgs[i].peer >= t->world) return fail("message %d: peer %d outside 0..%d", i, msgs[i].peer, t->world - 1);
    if (t->device >= 0) T_HIP(hipStreamSynchronize(st));          // what is sent was produced on the caller's stream
    std::vector<const void*> src;
    std::vector<size_t> nb;
    std::vector<int> peer;
    for (int i = 0; i < n; i++)
        if (msgs[i].is_send) { src.push_back(msgs[i].d_buf); nb.push_back(msgs[i].bytes); peer.push_back(msgs[i].peer); }
    if (shm_publish(t, (int)src.size(), src.data(), nb.data(), peer.data(), true)) return -1;
    std::vector<size_t> taken(t->world, 0);
    for (int i = 0; i < n; i++) {
        if (msgs[i].is_send || !msgs[i].bytes) continue;
        const int p = msgs[i].peer;
        Mapping& m = p == t->rank ? t->out : t->in[p];
        if (p != t->rank && shm_map(m, out_path(t, p), kHeaderBytes, false)) return -1;
        OutHeader hd;
        memcpy(&hd, m.p, sizeof hd);
        if (taken[p] + msgs[i].bytes > hd.bytes[t->rank])
            return fail("shm transport: rank %d expects %zu more bytes from rank %d than it sent (%zu)", t->rank, msgs[i].bytes, p, (size_t)hd.bytes[t->rank]);
        if (p != t->rank && shm_map(m, out_path(t, p), hd.off[t->rank] + hd.bytes[t->rank], false)) return -1;
        if (t->device >= 0) { if (shm_copy(t, msgs[i].d_buf, m.p + hd.off[t->rank] + taken[p], msgs[i].bytes, true)) return -1; }
        else memcpy(msgs[i].d_buf, m.p + hd.off[t->rank] + taken[p], msgs[i].bytes);
        taken[p] += msgs[i].bytes;
    }
    return shm_barrier(t);                     // every rank has taken its bytes: the outboxes may be rewritten
}

int shm_allgather_host(fdn_transport* t, const void* send, void* recv, size_t bytes)
{
    std::vector<const void*> src(t->world, send);
    std::vector<size_t> nb(t->world, bytes);
    std::vector<int> peer(t->world);
    for (int j = 0; j < t->world; j++) peer[j] = j;
    if (shm_publish(t, t->world, src.data(), nb.data(), peer.data(), false)) return -1;
    for (int p = 0; p < t->world; p++) {
        Mapping& m = p == t->rank ? t->out : t->in[p];
        if (p != t->rank && shm_map(m, out_path(t, p), kHeaderBytes, false)) return -1;
        OutHeader hd;
        memcpy(&hd, m.p, sizeof hd);
        if (hd.bytes[t->rank] != bytes) return fail("shm all-gather: rank %d contributed %zu bytes, %zu expected", p, (size_t)hd.bytes[t->rank], bytes);
        if (p != t->rank && shm_map(m, out_path(t, p), hd.off[t->rank] + bytes, false)) return -1;
        memcpy((char*)recv + (size_t)p * bytes, m.p + hd.off[t->rank], bytes);
    }
    return shm_barrier(t);
}

// ---- the fdn_comm callbacks -----------------------------------------------------------------------------------------
int cb_exchange(void* ctx, int n, const fdn_msg* msgs, void* stream) { return fdn_transport_exchange((fdn_transport*)ctx, n, msgs, stream); }
int cb_allgather(void* ctx, const void* send, void* recv, size_t bytes) { return fdn_transport_allgather_host((fdn_transport*)ctx, send, recv, bytes); }

} // namespace

extern "C" {

#define FDN_API __attribute__((visibility("default")))

FDN_API const char* fdn_transport_last_error(void) { return g_err.c_str(); }

FDN_API int fdn_transport_create(int kind, int rank, int world, int device, const char* rendezvous, fdn_transport_t* out)
{
    if (!out) return fail("out is NULL");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail("bad rank %d of %d", rank, world);
    fdn_transport* t = new fdn_transport();
    t->kind = kind; t->rank = rank; t->world = world; t->device = device;
    t->comm.ctx = t; t->comm.rank = rank; t->comm.world = world;
    t->comm.exchange = cb_exchange; t->comm.allgather_host = cb_allgather;
    int rc = 0;
    if (kind == FDN_TRANSPORT_NULL) {
        char buf[96];
        snprintf(buf, sizeof buf, "null (moves nothing), rank %d of %d", rank, world);
        t->what = buf;
    } else {
        // device -1 (SHM only): the buffers are HOST memory and no GPU is touched -- the CPU tests of the message matching
        hipError_t e = (kind == FDN_TRANSPORT_SHM && device < 0) ? hipSuccess : hipSetDevice(device);
        if (e != hipSuccess) rc = fail("hipSetDevice(%d): %s", device, hipGetErrorString(e));
        else if (kind == FDN_TRANSPORT_RCCL) rc = rccl_init(t, rendezvous);
        else if (kind == FDN_TRANSPORT_SHM) rc = shm_init(t, rendezvous);
        else rc = fail("unknown transport kind %d", kind);
    }
    if (!rc && kind == FDN_TRANSPORT_NULL) t->count = world;
    if (!rc && kind == FDN_TRANSPORT_SHM) {    // who is there: every rank's number gathered once, distinct ones counted
        std::vector<int> all(world, -1);
        const int mine = rank;
        rc = shm_allgather_host(t, &mine, all.data(), sizeof mine);
        std::sort(all.begin(), all.end());
        t->count = (int)(std::unique(all.begin(), all.end()) - all.begin());
        if (!rc && t->count != world) rc = fail("shm transport: %d distinct ranks answered, %d expected", t->count, world);
    }
    if (!rc) {
        char id[64] = "host";
        if (device >= 0 && hipDeviceGetPCIBusId(id, (int)sizeof id, device) != hipSuccess) { (void)hipGetLastError(); snprintf(id, sizeof id, "device %d", device); }
        t->device_id = id;
    }
    if (rc) { const std::string keep = g_err; fdn_transport_destroy(t); g_err = keep; return -1; }
    *out = t;
    return 0;
}

FDN_API int fdn_transport_destroy(fdn_transport_t t)
{
    if (!t) return 0;
    if (t->kind == FDN_TRANSPORT_RCCL && !t->failed) {
        (void)hipSetDevice(t->device);
        if (t->side) { (void)hipStreamSynchronize(t->side); }
        if (t->nccl) (void)ncclCommDestroy(t->nccl);
        if (t->stage) (void)hipFree(t->stage);
        if (t->hstage) (void)hipHostFree(t->hstage);
        if (t->order) (void)hipEventDestroy(t->order);
        if (t->side) (void)hipStreamDestroy(t->side);
    }
    // (an aborted communicator -- a failed init, fdn_transport_abort: its side stream may still hold operations of peers that
    //  never answered; waiting for it, freeing device memory (hipFree waits for the device) or ncclCommDestroy could hang.
    //  The stream, the event and the 512-byte staging block are left to the process's end, which is where such a rank is headed.)
    if (t->kind == FDN_TRANSPORT_SHM) {
        if (t->bounce) { (void)hipSetDevice(t->device); (void)hipHostFree(t->bounce); }
        t->out.close_();
        for (auto& m : t->in) m.close_();
        if (t->ctl) munmap(t->ctl, 4096);
    }
    delete t;
    return 0;
}

FDN_API const char* fdn_transport_describe(fdn_transport_t t) { return t ? t->what.c_str() : ""; }

FDN_API int fdn_transport_count(fdn_transport_t t, int* out)
{
    if (!t || !out) return fail("transport / out is NULL");
    *out = t->count;
    if (t->kind == FDN_TRANSPORT_RCCL) T_NCCL(ncclCommCount(t->nccl, out));      // asked again: the live communicator's own answer
    return 0;
}

FDN_API int fdn_transport_device_id(fdn_transport_t t, char* buf, int cap)
{
    if (!t || !buf || cap < 1) return fail("transport / buf is NULL");
    snprintf(buf, (size_t)cap, "%s", t->device_id.c_str());
    return 0;
}

FDN_API int fdn_transport_abort(fdn_transport_t t)
{
    if (!t) return 0;
    if (t->kind == FDN_TRANSPORT_SHM && t->ctl) t->ctl->failed.store(1);
    if (t->kind == FDN_TRANSPORT_RCCL && t->nccl) { (void)ncclCommAbort(t->nccl); t->nccl = nullptr; t->failed = true; }
    return 0;
}

FDN_API const fdn_comm* fdn_transport_comm(fdn_transport_t t)
{
    if (!t) { fail("transport is NULL"); return nullptr; }
    return &t->comm;
}

FDN_API int fdn_transport_exchange(fdn_transport_t t, int n, const fdn_msg* msgs, void* stream)
{
    if (!t) return fail("transport is NULL");
    if (n > 0 && !msgs) return fail("msgs is NULL");
    switch (t->kind) {
    case FDN_TRANSPORT_RCCL: return rccl_exchange(t, n, msgs, (hipStream_t)stream);
    case FDN_TRANSPORT_SHM: return shm_exchange(t, n, msgs, (hipStream_t)stream);
    default:
        // NULL: nothing travels.  What would have been received is zero-filled on the stream, so that the passes of an
        // overhead emulation run on finite numbers (uninitialised staging memory may hold NaNs) -- device >= 0 only
        if (t->device >= 0)
            for (int i = 0; i < n; i++)
                if (!msgs[i].is_send && msgs[i].bytes) T_HIP(hipMemsetAsync(msgs[i].d_buf, 0, msgs[i].bytes, (hipStream_t)stream));
        return 0;
    }
}

FDN_API int fdn_transport_allgather_host(fdn_transport_t t, const void* send, void* recv, size_t bytes)
{
    if (!t) return fail("transport is NULL");
    if (bytes && (!send || !recv)) return fail("send / recv is NULL");
    switch (t->kind) {
    case FDN_TRANSPORT_RCCL: return rccl_allgather_host(t, send, recv, bytes);
    case FDN_TRANSPORT_SHM: return shm_allgather_host(t, send, recv, bytes);
    default:
        for (int r = 0; r < t->world; r++) memcpy((char*)recv + (size_t)r * bytes, send, bytes);
        return 0;
    }
}

FDN_API int fdn_transport_barrier(fdn_transport_t t)
{
    if (!t) return fail("transport is NULL");
    if (t->kind == FDN_TRANSPORT_SHM) return shm_barrier(t);
    if (t->kind == FDN_TRANSPORT_RCCL) {
        char one = 1;
        std::vector<char> all(t->world);
        return rccl_allgather_host(t, &one, all.data(), 1);
    }
    return 0;
}

} // extern "C"
