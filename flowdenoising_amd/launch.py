"""Starting and joining the ranks of a multi-GPU job on one node -- without PyTorch.

The reference makes P workers one job with a process pool over a shared-memory volume
(src/flowdenoising.py:187-193, PoolExecutor(max_workers = P)); here a job is N processes, one per GPU, each holding
a Z-slab, that meet in a rendezvous directory (include/flowdn_rccl.h):

    spawn(argv, world)      the parent: N plain subprocess.Popen children BEFORE anything touches a GPU (never an exec),
                            FDN_RANK / FDN_WORLD / FDN_RDV in their environment; returns the first non-zero exit code
    supervise_rank(argv, job)   a rank that torch.distributed.run started (it has not touched a GPU) runs the native job
                            in ONE child of its own and watches it: a failed or overdue native rank anywhere in the job
                            ends all of them, and every rank process is still fresh for another engine
    report_failure(...)     a rank that cannot go on leaves its reason in the rendezvous directory for the others
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
import time


def _rendezvous_root():
    for d in ("/dev/shm", tempfile.gettempdir()):
        if os.path.isdir(d) and os.access(d, os.W_OK):
            return d
    return tempfile.gettempdir()


def spawn(argv, world, env=None, relay=None, errors=None, deadline=None):
    """Run `argv` (a full command line, e.g. [sys.executable, script, ...]) as `world` rank processes and wait for them.
    relay: a function called with every stdout line of rank 0 (default: print it); the other ranks' stdout is dropped,
    stderr goes straight through.  A rank that fails takes the others down (they would wait for it otherwise) -- and so
    does a rank that has left error.<rank> in the rendezvous directory but cannot leave (report_failure), and the passing
    of `deadline` seconds (FDN_NATIVE_DEADLINE; default: none for the CLI, whose jobs are as long as their volumes).
    errors: a list that receives what failed ranks left behind with report_failure()."""
    if deadline is None and os.environ.get("FDN_NATIVE_DEADLINE"):
        deadline = float(os.environ["FDN_NATIVE_DEADLINE"])
    rdv = tempfile.mkdtemp(prefix="fdn_rdv_", dir=_rendezvous_root())      # mode 0700, ours alone
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
        t0 = time.monotonic()
        told = False
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
            # a rank that reported a failure but is stuck on its way out (a thread inside RCCL), or an overdue job
            overdue = deadline is not None and time.monotonic() - t0 > deadline
            if pending and not told and (overdue or failure_reports(rdv)):
                told = True
                if overdue:
                    report_failure(rdv, "parent", f"the job did not finish within {deadline:.0f} s")
                if code == 0:
                    code = 1
                grace = time.monotonic() + 10
                for q in pending:
                    q.terminate()
                while any(q.poll() is None for q in pending) and time.monotonic() < grace:
                    time.sleep(0.05)
                for q in pending:
                    if q.poll() is None:
                        q.kill()
        reader.join(timeout=10)
        if errors is not None:
            errors.extend(failure_reports(rdv))
        return code
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(rdv, ignore_errors=True)


def report_failure(rdv, rank, message):
    """Leave `message` in the job's rendezvous directory (error.<rank>): the supervisor -- spawn()'s parent, or the
    torch.distributed.run rank that watches this one -- quotes it, and its presence tells the other ranks' supervisors
    that the native job is over."""
    try:
        tmp = os.path.join(rdv, f".error.{rank}.tmp")
        with open(tmp, "w") as f:
            f.write(str(message)[:2000])
        os.replace(tmp, os.path.join(rdv, f"error.{rank}"))
    except OSError:
        pass


def failure_reports(rdv):
    out = []
    try:
        names = sorted(n for n in os.listdir(rdv) if n.startswith("error."))
    except OSError:
        return out
    for n in names:
        try:
            with open(os.path.join(rdv, n)) as f:
                out.append(f"rank {n[6:]}: {f.read().strip()}")
        except OSError:
            pass
    return out


def started_by_torchrun():
    """This process is a rank that torch.distributed.run (or a launcher with its conventions) started -- not one of ours."""
    return "FDN_RANK" not in os.environ and "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1


def supervise_rank(argv, job, relay=None, deadline=None):
    """Under torch.distributed.run every rank process is a sibling started by somebody else: nobody here is the parent of
    the whole job.  So each rank -- before it has touched a GPU -- runs the native job's rank in ONE child process of its
    own (FDN_RANK / FDN_WORLD / FDN_RDV as spawn() would set them) and watches it.  A native rank that fails leaves
    error.<rank> in the shared rendezvous directory; every supervisor that sees one ends its own child (the exact process
    it started) -- as it does when `deadline` seconds pass (FDN_NATIVE_DEADLINE, default 900).
    The supervisors then AGREE before any of them returns: each leaves done.<rank> ("ok" or "failed") once its child has
    gone, waits until every rank's is there, and all of them return the same verdict -- 0 only if every child ended well and
    nobody left an error.<rank> (a child writes its report before it exits, so the reports are complete by then).  Rank 0's
    stdout is held back until that verdict and dropped if it is a failure: a job never prints a result line and then another
    one from the fallback engine.  Returns (exit code, reasons): 0 = the native job ran to its end on every rank; otherwise
    the caller is a process that has never initialised the GPU and may run another engine in itself -- as all its peers do."""
    rank, world, local, rdv = job
    if deadline is None:
        deadline = float(os.environ.get("FDN_NATIVE_DEADLINE", "900"))
    env = dict(os.environ, FDN_RANK=str(rank), FDN_WORLD=str(world), FDN_RDV=rdv, FDN_LOCAL_RANK=str(local))
    env.setdefault("FDN_RDV_TIMEOUT", "300")      # ranks reach their rendezvous a first `import torch` apart (minutes on a fresh box)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK"):
        env.pop(k, None)
    child = subprocess.Popen(argv, env=env, stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, text=(rank == 0))
    reader = None
    held = []                                     # rank 0's stdout, until the supervisors have agreed
    if rank == 0:
        def pump():
            for line in child.stdout:
                held.append(line)
        reader = threading.Thread(target=pump, daemon=True)
        reader.start()
    t0 = time.monotonic()
    rc = None
    try:
        while rc is None:
            try:
                rc = child.wait(timeout=0.2)
            except subprocess.TimeoutExpired:
                overdue = time.monotonic() - t0 > deadline
                if overdue:
                    report_failure(rdv, rank, f"the native job did not finish within {deadline:.0f} s")
                if overdue or failure_reports(rdv):
                    child.terminate()
                    try:
                        child.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        child.kill()
                        child.wait()
                    rc = child.returncode or 1
        if rc != 0 and not failure_reports(rdv):
            report_failure(rdv, rank, f"native rank exited with code {rc}")
    finally:
        if child.poll() is None:
            child.kill()
    if reader is not None:
        reader.join(timeout=10)
    # the agreement: every supervisor's done.<rank>, then one verdict for all
    _leave_note(rdv, f"done.{rank}", "ok" if rc == 0 else "failed")
    limit = time.monotonic() + max(30.0, deadline - (time.monotonic() - t0) + 30.0)
    verdicts = None
    while verdicts is None and time.monotonic() < limit:
        verdicts = _read_notes(rdv, "done.", world)
        if verdicts is None:
            time.sleep(0.05)
    reasons = failure_reports(rdv)
    if verdicts is None:
        reasons = reasons + [f"rank {rank}: not every rank's supervisor reported within the deadline"]
    ok = verdicts is not None and all(v == "ok" for v in verdicts) and not reasons
    _leave_note(rdv, f"ack.{rank}", "1")          # (rank 0 removes the directory: not before everybody has read it)
    if rank == 0:
        until = time.monotonic() + 10.0
        while _read_notes(rdv, "ack.", world) is None and time.monotonic() < until:
            time.sleep(0.05)
        if ok:
            for line in held:
                if relay is not None:
                    relay(line)
                else:
                    sys.stdout.write(line)
                    sys.stdout.flush()
        elif held:
            sys.stderr.write("flowdenoising_amd.launch: the native job's output is dropped: not every rank ended well\n")
    if ok:
        return 0, []
    return (rc or 1), (reasons or [f"rank {rank}: another rank's native process failed"])


def _leave_note(rdv, name, text):
    try:
        tmp = os.path.join(rdv, "." + name + ".tmp")
        with open(tmp, "w") as f:
            f.write(text)
        os.replace(tmp, os.path.join(rdv, name))
    except OSError:
        pass


def _read_notes(rdv, prefix, world):
    """The `world` notes <prefix><rank> in rank order, or None while one is missing."""
    out = []
    for r in range(world):
        try:
            with open(os.path.join(rdv, f"{prefix}{r}")) as f:
                out.append(f.read().strip())
        except OSError:
            return None
    return out


def _proc_start_time(pid):
    try:
        with open(f"/proc/{pid}/stat") as f:
            return f.read().rsplit(")", 1)[1].split()[19]
    except (OSError, IndexError):
        return "0"


def _private_dir(path):
    """A directory of ours alone: created with mode 0700, and refused when somebody else owns it (its name is derivable)."""
    try:
        os.makedirs(path, mode=0o700)
    except FileExistsError:
        pass
    st = os.stat(path)
    if st.st_uid != os.getuid():
        raise PermissionError(f"rendezvous directory {path} belongs to uid {st.st_uid}, not to this user")
    return path


def job():
    """(rank, world, local_rank, rendezvous_dir) of this process, or None when it is not a rank of a multi-rank job."""
    # dmabuf IPC: RCCL across processes needs it on this host driver -- whoever started the ranks (spawn() sets it too)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "FDN_RANK" in os.environ and int(os.environ.get("FDN_WORLD", "1")) > 1:
        r = int(os.environ["FDN_RANK"])
        return r, int(os.environ["FDN_WORLD"]), int(os.environ.get("FDN_LOCAL_RANK", r)), os.environ["FDN_RDV"]
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # started by torch.distributed.run (the round driver's bench launch) or by hand: the ranks are siblings, so their
        # parent's pid and start time plus the job's port (and, under torchelastic, the run id and restart count: a
        # restarted worker group must not find the previous attempt's files) name a directory nobody else derives
        r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        os.environ.setdefault("OMP_NUM_THREADS", "2")
        ppid = os.getppid()
        tag = "".join(c if c.isalnum() else "-" for c in os.environ.get("TORCHELASTIC_RUN_ID", ""))[:24]
        name = (f"fdn_rdv_{os.getuid()}_{os.environ.get('MASTER_PORT', '0')}_{ppid}_{_proc_start_time(ppid)}"
                f"_{tag}_{os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')}")
        # FDN_RDV names a directory of the user's choice: the job still gets a subdirectory of its own there, so that what an
        # earlier job left behind (error.*, dev.*, done.*) is never read as this job's
        base = os.environ.get("FDN_RDV")
        if base:
            _private_dir(base)
        rdv = os.path.join(base or _rendezvous_root(), name)
        _derived.add(rdv)
        return r, w, int(os.environ.get("LOCAL_RANK", r)), _private_dir(rdv)
    return None


_derived = set()


def remove_derived_rendezvous(rdv):
    """A directory job() derived under torch.distributed.run is ours to remove (spawn() removes its own; a rank process that
    was handed its directory through FDN_RDV leaves it to whoever made it)."""
    if rdv in _derived:
        shutil.rmtree(rdv, ignore_errors=True)


_seq = [0]


def make_transport(rank, world, local_rank, rdv, kind=None):
    """(transport, device): RCCL with one GPU per rank when the node has that many, otherwise the shared-memory rehearsal
    transport with the ranks dealt round-robin over the GPUs there are.  Collective: every rank calls it, in the same order."""
    from . import _lib
    ngpu = _lib.device_count()
    if ngpu < 1:
        raise _lib.FlowdnError("no HIP device visible to this rank")
    sub = os.path.join(rdv, f"t{_seq[0]}")
    _seq[0] += 1
    os.makedirs(sub, mode=0o700, exist_ok=True)
    device = local_rank % ngpu
    if kind is None:
        kind = "rccl" if ngpu >= world else "shm"
        if ngpu < world:
            # fewer visible GPUs than ranks: either the ranks share GPUs (a rehearsal: shared memory), or the launcher
            # gave every rank a GPU of its own through HIP_VISIBLE_DEVICES -- the PCI bus ids tell the two apart
            ids = _exchange_lines(sub, rank, world, _lib.device_pci_id(device))
            if len(set(ids)) == world:
                kind = "rccl"
    try:
        return _lib.Transport(kind, rank, world, device, sub), device
    except _lib.FlowdnError as e:
        # The transport could not be created.  After an RCCL initialisation that ran into its deadline a helper thread is
        # still inside ncclCommInitRank (include/flowdn_rccl.h): the interpreter's orderly teardown -- atexit handlers, the
        # HIP runtime's own -- could wait for it for ever.  Say why, for the other ranks' supervisors, and leave at once.
        report_failure(rdv, rank, f"{type(e).__name__}: {e}")
        sys.stderr.write(f"flowdenoising_amd.launch: rank {rank}: {e}\n")
        sys.stderr.flush()
        sys.stdout.flush()
        os._exit(1)


def _exchange_lines(sub, rank, world, line, timeout=None):
    """Every rank's `line`, in rank order, through files in the job's directory (before any transport exists)."""
    if timeout is None:
        timeout = float(os.environ.get("FDN_RDV_TIMEOUT", "600"))
    tmp = os.path.join(sub, f".dev.{rank}.tmp")
    with open(tmp, "w") as f:
        f.write(line)
    os.replace(tmp, os.path.join(sub, f"dev.{rank}"))
    out, t0 = [], time.monotonic()
    for r in range(world):
        path = os.path.join(sub, f"dev.{r}")
        while not os.path.exists(path):
            if time.monotonic() - t0 > timeout:
                raise TimeoutError(f"rank {r} of {world} did not reach the rendezvous directory within {timeout:.0f} s")
            time.sleep(0.01)
        with open(path) as f:
            out.append(f.read())
    return out
