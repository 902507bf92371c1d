"""No-GPU checks of libflowdn_rccl.so (include/flowdn_rccl.h): it loads, exports every declared symbol, links librccl, and
its shared-memory and null transports move (or do not move) bytes between real processes."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fdn_[a-z0-9_]+)\s*\(", text)))


def test_transport_library_exports_every_declared_symbol(fdn):
    lib = ctypes.CDLL(fdn._lib.RCCL_LIB_PATH)
    declared = _declared("flowdn_rccl.h")
    assert len(declared) >= 8
    for name in declared:
        assert hasattr(lib, name), f"libflowdn_rccl.so does not export {name}"
    assert sorted(fdn._lib.RCCL_EXPORTS) == declared


def test_transport_library_links_rccl(fdn):
    out = subprocess.run(["ldd", fdn._lib.RCCL_LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl" in out, out


def test_null_transport_moves_nothing(fdn):
    t = fdn._lib.Transport("null", 5, 8)
    assert "rank 5 of 8" in t.describe()
    a = np.arange(4, dtype=np.float32)
    t.exchange([(a.ctypes.data, a.nbytes, 2, False), (a.ctypes.data, a.nbytes, 6, True)], 0)
    assert a.tolist() == [0, 1, 2, 3]
    assert t.allgather_host(b"xy") == b"xy" * 8
    t.barrier()
    t.close()
    with pytest.raises(fdn._lib.FlowdnError):
        fdn._lib.Transport("null", 8, 8)


@pytest.mark.parametrize("world", [2, 4])
def test_shm_transport_between_processes(fdn, world):
    from flowdenoising_amd import launch
    lines = []
    rc = launch.spawn([sys.executable, os.path.join(ROOT, "tests", "_transport_worker.py")], world, relay=lines.append)
    assert rc == 0
    assert "".join(lines).strip() == "rank 0 ok"


def test_spawn_reports_a_failed_rank_and_stops_the_others(fdn):
    from flowdenoising_amd import launch
    prog = "import os, sys, time; r = int(os.environ['FDN_RANK']); sys.exit(7) if r == 1 else time.sleep(60)"
    import time
    t0 = time.perf_counter()
    assert launch.spawn([sys.executable, "-c", prog], 3) == 7
    assert time.perf_counter() - t0 < 30


def test_shm_transport_gives_up_when_a_rank_never_arrives(fdn, tmp_path):
    env = dict(os.environ, FDN_RDV_TIMEOUT="2")
    prog = ("import sys; sys.path.insert(0, %r)\n"
            "from flowdenoising_amd import _lib\n"
            "try:\n    _lib.Transport('shm', 0, 2, -1, %r)\nexcept _lib.FlowdnError as e:\n    print('gave up:', e)\n") % (ROOT, str(tmp_path))
    r = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=120)
    assert "gave up" in r.stdout and "barrier" in r.stdout, r.stdout + r.stderr


def test_shm_transport_with_eight_rank_threads_in_one_process(fdn, tmp_path):
    """The ranks of the shared-memory transport as THREADS of one process (how the GPU tests rehearse world = 7 and 8 on
    a box that allows six GPU processes): the control block's barrier, the outboxes and the error strings hold up; the
    communicator's own rank count and the all-gathered device ids come back; a rank that aborts stops the others."""
    import threading
    world, errors, seen = 8, [], {}

    def rank_main(r):
        try:
            t = fdn._lib.Transport("shm", r, world, -1, str(tmp_path))
            for rnd in range(3):
                sends = {j: np.full(500 + 13 * r + 7 * j, 100.0 * r + j + rnd, dtype=np.float32) for j in range(world) if j != r}
                recvs = {i: np.zeros(500 + 13 * i + 7 * r, dtype=np.float32) for i in range(world) if i != r}
                t.exchange([(a.ctypes.data, a.nbytes, i, False) for i, a in recvs.items()] + [(a.ctypes.data, a.nbytes, j, True) for j, a in sends.items()], 0)
                for i, a in recvs.items():
                    assert np.all(a == np.float32(100.0 * i + r + rnd)), (rnd, r, i)
            assert t.count() == world
            assert t.devices() == ["host"] * world
            assert t.allgather_host(bytes([r]) * 3) == b"".join(bytes([q]) * 3 for q in range(world))      # 3 bytes: the staging stays aligned
            t.barrier()
            seen[r] = t.describe()
            t.close()
        except BaseException as e:      # noqa: BLE001
            errors.append((r, e))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
    assert not errors, errors
    assert len(seen) == world and "rank 0 of 8" in seen[0]


def test_an_aborting_rank_stops_the_waiting_ones(fdn, tmp_path):
    import threading
    import time
    out = {}

    def waiter():
        t = fdn._lib.Transport("shm", 0, 2, -1, str(tmp_path))
        try:
            t.barrier()
            out["waiter"] = "passed"
        except fdn._lib.FlowdnError as e:
            out["waiter"] = str(e)
        t.close()

    def quitter():
        t = fdn._lib.Transport("shm", 1, 2, -1, str(tmp_path))
        time.sleep(0.5)
        t.abort()
        t.close()

    ths = [threading.Thread(target=waiter), threading.Thread(target=quitter)]
    t0 = time.perf_counter()
    for th in ths:
        th.start()
    for th in ths:
        th.join(timeout=60)
    assert "another rank failed" in out["waiter"] and time.perf_counter() - t0 < 30


def _supervised(tmp_path, child_code, world=2, extra_env=None, port=29876):
    """`world` processes with torch.distributed.run's environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT, a common
    parent), each supervising one child that runs `child_code`; returns their (exit code, stdout) pairs."""
    sup = ("import sys, json; sys.path.insert(0, %r)\n"
           "from flowdenoising_amd import launch\n"
           "job = launch.job()\n"
           "assert launch.started_by_torchrun()\n"
           "rc, why = launch.supervise_rank([sys.executable, '-c', %r], job, relay=lambda ln: sys.stdout.write('child: ' + ln))\n"
           "print(json.dumps({'rank': job[0], 'rc': rc, 'why': why}))\n"
           "if job[0] == 0: launch.remove_derived_rendezvous(job[3])\n") % (ROOT, child_code)
    procs = []
    for r in range(world):
        env = {k: v for k, v in os.environ.items() if k not in ("FDN_RANK", "FDN_WORLD", "FDN_RDV")}
        env.update(RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_PORT=str(port), FDN_RDV_TIMEOUT="30")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, "-c", sup], env=env, stdout=subprocess.PIPE, text=True))
    return [(p.wait(timeout=120), p.stdout.read()) for p in procs]


def test_supervised_native_ranks_under_a_torchrun_like_parent(fdn, tmp_path):
    """launch.supervise_rank: a rank that torch.distributed.run started runs its native rank in ONE child (FDN_RANK / FDN_WORLD /
    FDN_RDV set, the torchrun variables gone) and relays rank 0's output; all children meet in the directory every supervisor
    derives from the common parent."""
    import json
    child = ("import os, sys; sys.path.insert(0, %r)\n"
             "from flowdenoising_amd import _lib, launch\n"
             "assert 'RANK' not in os.environ and not launch.started_by_torchrun()\n"
             "r, w, l, rdv = launch.job()\n"
             "t = _lib.Transport('shm', r, w, -1, rdv)\n"
             "assert t.count() == w\n"
             "t.barrier(); t.close()\n"
             "print('rank', r, 'of', w, 'ok')\n") % ROOT
    res = _supervised(tmp_path, child)
    recs = [json.loads(out.strip().splitlines()[-1]) for _, out in res]
    assert [r["rc"] for r in recs] == [0, 0] and all(code == 0 for code, _ in res)
    assert "child: rank 0 of 2 ok" in res[0][1] and "child:" not in res[1][1]


def test_a_failed_native_rank_ends_every_supervised_rank_with_the_reason(fdn, tmp_path):
    """Rank 1's native child gives up (report_failure + exit code 3) while rank 0's waits at a barrier with a long timeout:
    rank 0's supervisor sees error.1 in the shared directory, ends its own child, and both supervisors return the reason --
    promptly, and still fresh for another engine."""
    import json
    import time
    child = ("import os, sys, time; sys.path.insert(0, %r)\n"
             "from flowdenoising_amd import _lib, launch\n"
             "r, w, l, rdv = launch.job()\n"
             "if r == 1:\n"
             "    launch.report_failure(rdv, r, 'RuntimeError: no links today'); sys.exit(3)\n"
             "time.sleep(300)\n") % ROOT
    t0 = time.perf_counter()
    res = _supervised(tmp_path, child, port=29877)
    assert time.perf_counter() - t0 < 60
    recs = [json.loads(out.strip().splitlines()[-1]) for _, out in res]
    assert all(r["rc"] != 0 for r in recs)
    assert all(any("no links today" in w for w in r["why"]) for r in recs), recs


def test_supervisors_agree_when_a_rank_fails_after_another_has_finished(fdn, tmp_path):
    """Rank 0's native child prints its result and exits 0 at once; rank 1's fails a second later (in its teardown, say).
    The supervisors agree before either returns (done.<rank> notes): BOTH report the failure -- so both go on to the same
    fallback together instead of one waiting for peers that are gone -- and rank 0's already printed line is dropped, so
    the job cannot emit a native result and then a second line from the fallback."""
    import json
    child = ("import os, sys, time; sys.path.insert(0, %r)\n"
             "from flowdenoising_amd import launch\n"
             "r, w, l, rdv = launch.job()\n"
             "if r == 0:\n"
             "    print('{\"value\": 1}'); sys.exit(0)\n"
             "time.sleep(1.0)\n"
             "launch.report_failure(rdv, r, 'RuntimeError: teardown failed'); sys.exit(3)\n") % ROOT
    res = _supervised(tmp_path, child, port=29878)
    recs = [json.loads(out.strip().splitlines()[-1]) for _, out in res]
    assert all(r["rc"] != 0 for r in recs), recs
    assert all(any("teardown failed" in w for w in r["why"]) for r in recs), recs
    assert "child:" not in res[0][1]


def test_a_user_supplied_rendezvous_directory_gets_a_subdirectory_per_job(fdn, tmp_path):
    """FDN_RDV under a torchrun-like parent names a directory of the user's choice; stale error.* files of an earlier job in
    it must not end this job's children at once: the job works in a subdirectory of its own."""
    import json
    (tmp_path / "rdv").mkdir()
    (tmp_path / "rdv" / "error.1").write_text("stale: from an earlier run")
    child = ("import os, sys; sys.path.insert(0, %r)\n"
             "from flowdenoising_amd import _lib, launch\n"
             "r, w, l, rdv = launch.job()\n"
             "t = _lib.Transport('shm', r, w, -1, rdv)\n"
             "t.barrier(); t.close()\n"
             "print('rank', r, 'ok')\n") % ROOT
    res = _supervised(tmp_path, child, extra_env={"FDN_RDV": str(tmp_path / "rdv")}, port=29879)
    recs = [json.loads(out.strip().splitlines()[-1]) for _, out in res]
    assert [r["rc"] for r in recs] == [0, 0], recs
    assert "child: rank 0 ok" in res[0][1]
    assert (tmp_path / "rdv" / "error.1").exists()                 # the user's directory itself is left alone ...
    assert [n for n in os.listdir(tmp_path / "rdv") if n.startswith("fdn_rdv_")] == []   # ... and the job's own part is gone


def test_spawn_ends_a_job_that_is_overdue_or_whose_failed_rank_cannot_leave(fdn):
    """launch.spawn has a deadline of its own (FDN_NATIVE_DEADLINE / `deadline`), and a rank that has REPORTED a failure but is
    stuck on its way out (a helper thread still inside RCCL after an initialisation that ran into its own deadline) does not keep
    the parent waiting: the report in the rendezvous directory ends the job."""
    import time
    from flowdenoising_amd import launch
    forever = "import time; time.sleep(300)"
    t0 = time.perf_counter()
    errors = []
    assert launch.spawn([sys.executable, "-c", forever], 2, deadline=2.0, errors=errors) != 0
    assert time.perf_counter() - t0 < 40 and any("did not finish within" in e for e in errors), errors
    stuck = ("import os, sys, time; sys.path.insert(0, %r)\n"
             "from flowdenoising_amd import launch\n"
             "r, w, l, rdv = launch.job()\n"
             "if r == 1:\n"
             "    launch.report_failure(rdv, r, 'RuntimeError: the links are down')\n"
             "time.sleep(300)\n") % ROOT
    t0 = time.perf_counter()
    errors = []
    assert launch.spawn([sys.executable, "-c", stuck], 2, errors=errors) != 0
    assert time.perf_counter() - t0 < 40 and any("the links are down" in e for e in errors), errors


def test_make_transport_failure_reports_and_ends_the_rank_at_once(fdn, tmp_path):
    """A transport that cannot be created ends its rank process right there -- exit code 1, the reason on stderr and in error.<rank>
    for the other ranks' supervisors -- instead of unwinding through an interpreter teardown that may wait for a thread still
    inside RCCL (launch.make_transport)."""
    prog = ("import sys; sys.path.insert(0, %r)\n"
            "from flowdenoising_amd import launch, _lib\n"
            "_lib.device_count = lambda: 2\n"                              # (no GPU needed: the shared-memory kind with no peer)
            "launch.make_transport(0, 2, 0, %r, kind='shm')\n"
            "print('still here')\n") % (ROOT, str(tmp_path))
    r = subprocess.run([sys.executable, "-c", prog], env=dict(os.environ, FDN_RDV_TIMEOUT="2"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "still here" not in r.stdout, (r.returncode, r.stdout, r.stderr[-500:])
    assert "rank 0" in r.stderr
    # (without a GPU the reason is hipSetDevice's; on a GPU box the peer that never arrives)
    assert (tmp_path / "error.0").read_text().startswith("FlowdnError:")
