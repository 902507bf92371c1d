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
