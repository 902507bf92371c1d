/*
 * flowdn_rccl.h -- C ABI of libflowdn_rccl.so: the transports behind fdn_filter_3d_sharded (include/flowdn.h).
 *
 * The reference splits a pass over P worker processes that share ONE host volume: contiguous chunks of dim // P target
 * slices per worker plus a remainder round (src/flowdenoising.py:181-206, "par"), the volume itself in
 * multiprocessing shared memory.  With one process per GPU there is no shared volume: every rank holds a slab, and what
 * the shared memory did implicitly -- a worker reading the K//2 slices either side of its chunk, and the next pass
 * reading the previous one's output in another orientation -- becomes one explicit exchange per pass
 * (fdn_filter_3d_sharded: schedule, packing and passes in libflowdn.so).  This library supplies the two callbacks of
 * `fdn_comm` that move the bytes, so that a multi-GPU caller needs neither Python nor PyTorch:
 *
 *   FDN_TRANSPORT_RCCL   one GPU per rank.  exchange = ncclGroupStart(); ncclSend / ncclRecv per message on the
 *                        caller's stream; ncclGroupEnd() -- point-to-point over xGMI, every pair at once, nothing
 *                        synchronises the host.  allgather_host = ncclAllGather on a device staging buffer, on a stream of
 *                        its own that is ordered (event) behind the stream of the last exchange.  The ncclUniqueId
 *                        travels from rank 0 to the others through a file in the rendezvous directory; ncclCommInitRank
 *                        is given FDN_RDV_TIMEOUT seconds, and a first ring exchange FDN_RCCL_SANITY_TIMEOUT (120):
 *                        IPC or links that do not work end in an error message, not in a hang.
 *   FDN_TRANSPORT_SHM    ranks that share a GPU (a rehearsal of N ranks on a one-GPU box; RCCL refuses two ranks on
 *                        one device): messages are staged through POSIX shared memory files in the rendezvous
 *                        directory.  Same schedule, same kernels; not a measurement of anything.
 *   FDN_TRANSPORT_NULL   moves nothing (exchange returns at once, allgather_host repeats the caller's own
 *                        contribution): the per-rank overhead of rank r of an N-rank plan -- packing, unpacking, the
 *                        mean -- timed on one GPU with the shapes of the real run (tools/rank_emulation.py).
 *                        Results are garbage by construction.
 *
 * Conventions as in flowdn.h: int status (0 ok, negative error), fdn_transport_last_error() thread-local, no
 * exceptions across the boundary.  All ranks are processes of ONE node.
 */
#ifndef FLOWDN_RCCL_H
#define FLOWDN_RCCL_H

#include "flowdn.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fdn_transport* fdn_transport_t;

#define FDN_TRANSPORT_RCCL 0
#define FDN_TRANSPORT_SHM 1
#define FDN_TRANSPORT_NULL 2

/* Collective over the `world` ranks of a job (NULL: local).  `device`: the HIP device of this rank (set current by the
 * call; -1 with FDN_TRANSPORT_SHM: "device" buffers are host memory and no GPU is touched -- CPU tests).  `rendezvous`: a directory every rank of the job can reach and nobody else uses (the launcher makes it, e.g.
 * under /dev/shm; flowdenoising_amd/launch.py); ignored by FDN_TRANSPORT_NULL.  Ranks wait for each other up to
 * FDN_RDV_TIMEOUT seconds (environment, default 600) and fail -- never hang -- after that.
 * Replaces the worker pool of par:187-193 (PoolExecutor(max_workers = P)) as the thing that makes P workers one job. */
int fdn_transport_create(int kind, int rank, int world, int device, const char* rendezvous, fdn_transport_t* out);
int fdn_transport_destroy(fdn_transport_t t);
const char* fdn_transport_last_error(void);
/* "rccl 2.x.y, 8 ranks, device 3" / "shm, 2 ranks ..." */
const char* fdn_transport_describe(fdn_transport_t t);

/* Who is there, as the communicator itself says it -- not as the launcher's environment claims: RCCL: ncclCommCount of
 * the live communicator; SHM: the distinct ranks that answered the first all-gather; NULL: `world`.  bench.py prints it
 * as `n_ranks_seen`. */
int fdn_transport_count(fdn_transport_t t, int* out);
/* The PCI bus id of this rank's device (hipDeviceGetPCIBusId, e.g. "0000:05:00.0"; "host" for device -1): all-gathered,
 * N distinct strings show N distinct GPUs -- a run whose ranks share devices is a rehearsal and says so. */
int fdn_transport_device_id(fdn_transport_t t, char* buf, int cap);
/* A rank that cannot go on (a reader or writer failed, an exception above the ABI) gives its transport up before it leaves.
 * SHM sets the job's failed flag: the other ranks see it in their next wait and return an error.  RCCL: ncclCommAbort --
 * which is LOCAL: it stops this rank's own pending operations so that its teardown cannot hang; peers blocked in an
 * exchange with this rank are not woken by it (RCCL has no timeout after initialisation).  They are ended by whoever
 * supervises the job: the rank leaves error.<rank> in the rendezvous directory (launch.report_failure) and the parent
 * (launch.spawn) or the per-rank supervisors (launch.supervise_rank) terminate the remaining rank processes.
 * Afterwards the transport can only be destroyed; fdn_transport_destroy then waits for nothing (the aborted communicator's
 * stream may hold operations that never complete). */
int fdn_transport_abort(fdn_transport_t t);

/* The communicator to hand to fdn_filter_3d_sharded (valid until fdn_transport_destroy). */
const fdn_comm* fdn_transport_comm(fdn_transport_t t);

/* The callbacks themselves, for host programs that move slabs of their own (gathering a result, a barrier around a
 * timed region): one batched group of point-to-point messages on DEVICE buffers, enqueued on `stream` (RCCL: returns
 * without waiting; SHM: completes before it returns) ... */
int fdn_transport_exchange(fdn_transport_t t, int n, const fdn_msg* msgs, void* stream);
/* ... every rank's `bytes` from `send` into `recv` (world * bytes) in rank order, HOST memory, complete on return ... */
int fdn_transport_allgather_host(fdn_transport_t t, const void* send, void* recv, size_t bytes);
/* ... and a barrier of the ranks (host side; RCCL: a one-byte all-gather). */
int fdn_transport_barrier(fdn_transport_t t);

#ifdef __cplusplus
}
#endif
#endif /* FLOWDN_RCCL_H */
