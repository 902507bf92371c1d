"""Starting and joining the ranks of a multi-GPU job on one node -- without PyTorch.

The reference makes P workers one job with a process pool over a shared-memory volume
(src/flowdenoising.py:187-193, PoolExecutor(max_workers = P)); here a job is N processes, one per GPU, each holding
a Z-slab, that meet in a rendezvous directory (include/flowdn_rccl.h):

    spawn(argv, world)      the parent: N plain subprocess.Popen children BEFORE anything touches a GPU (never an exec),
                            FDN_RANK / FDN_WORLD / FDN_RDV in their environment; returns the first non-zero exit code
    job()                   a child: (rank, world, local_rank, rendezvous dir) -- from spawn()'s variables, or from the
                            RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT a torch.distributed.run parent sets (how the
                            round driver starts bench.py); None when this process is not a rank of a job
    make_transport(...)     the fdn_comm of this rank: RCCL when every rank has a GPU of its own, else the shared-memory
                            rehearsal transport (ranks share GPUs)

Nothing here imports torch.
"""
import os
import shutil
import subprocess
import sys
import tempfile
import threading


def _rendezvous_root():
    for d in ("/dev/shm", tempfile.gettempdir()):
        if os.path.isdir(d) and os.access(d, os.W_OK):
            return d
    return tempfile.gettempdir()


def spawn(argv, world, env=None, relay=None):
    """Run `argv` (a full command line, e.g. [sys.executable, script, ...]) as `world` rank processes and wait for them.
    relay: a function called with every stdout line of rank 0 (default: print it); the other ranks' stdout is dropped,
    stderr goes straight through.  A rank that fails takes the others down (they would wait for it otherwise)."""
    rdv = tempfile.mkdtemp(prefix="fdn_rdv_", dir=_rendezvous_root())
    base = dict(os.environ if env is None else env)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this host driver
    base.setdefault("OMP_NUM_THREADS", "2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):         # a job of our own, whatever started the parent
        base.pop(k, None)
    procs = []
    try:
        for r in range(world):
            e = dict(base, FDN_RANK=str(r), FDN_WORLD=str(world), FDN_RDV=rdv)
            procs.append(subprocess.Popen(argv, env=e, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
        def pump():                                # rank 0's stdout, line by line, while the main thread watches the ranks
            for line in procs[0].stdout:
                if relay is not None:
                    relay(line)
                else:
                    sys.stdout.write(line)
                    sys.stdout.flush()

        reader = threading.Thread(target=pump, daemon=True)
        reader.start()
        code = 0
        pending = list(procs)
        while pending:
            for p in list(pending):
                try:
                    rc = p.wait(timeout=0.1)
                except subprocess.TimeoutExpired:
                    continue
                pending.remove(p)
                if rc != 0 and code == 0:
                    code = rc
                    for q in pending:              # a failed rank takes the others down -- the exact processes started above
                        q.terminate()
        reader.join(timeout=10)
        return code
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(rdv, ignore_errors=True)


def _proc_start_time(pid):
    try:
        with open(f"/proc/{pid}/stat") as f:
            return f.read().rsplit(")", 1)[1].split()[19]
    except (OSError, IndexError):
        return "0"


def job():
    """(rank, world, local_rank, rendezvous_dir) of this process, or None when it is not a rank of a multi-rank job."""
    if "FDN_RANK" in os.environ and int(os.environ.get("FDN_WORLD", "1")) > 1:
        r = int(os.environ["FDN_RANK"])
        return r, int(os.environ["FDN_WORLD"]), r, os.environ["FDN_RDV"]
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # started by torch.distributed.run (the round driver's bench launch) or by hand: the ranks are siblings, so their
        # parent's pid and start time plus the job's port name a directory nobody else derives
        r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        ppid = os.getppid()
        name = f"fdn_rdv_{os.getuid()}_{os.environ.get('MASTER_PORT', '0')}_{ppid}_{_proc_start_time(ppid)}"
        rdv = os.environ.get("FDN_RDV") or os.path.join(_rendezvous_root(), name)
        os.makedirs(rdv, exist_ok=True)
        return r, w, int(os.environ.get("LOCAL_RANK", r)), rdv
    return None


_seq = [0]


def make_transport(rank, world, local_rank, rdv, kind=None):
    """(transport, device): RCCL with one GPU per rank when the node has that many, otherwise the shared-memory rehearsal
    transport with the ranks dealt round-robin over the GPUs there are.  Collective: every rank calls it, in the same order."""
    from . import _lib
    ngpu = _lib.device_count()
    if ngpu < 1:
        raise _lib.FlowdnError("no HIP device visible to this rank")
    if kind is None:
        kind = "rccl" if ngpu >= world else "shm"
    device = local_rank if kind == "rccl" else local_rank % ngpu
    sub = os.path.join(rdv, f"t{_seq[0]}")
    _seq[0] += 1
    os.makedirs(sub, exist_ok=True)
    return _lib.Transport(kind, rank, world, device, sub), device
