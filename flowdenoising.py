#!/usr/bin/env python3
"""flowdenoising.py -- drop-in command line (see flowdenoising_amd/cli.py for the option list)."""
import os
import sys
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def _warm_gpu(device):
    """Load the HIP runtime and libflowdn.so and create the GPU context WHILE numpy and the package are being imported
    (0.2 s of interpreter work on one side, 0.15-0.3 s of driver work on the other).  Single-GPU runs only: the parent of
    `--gpus N` never touches a GPU, and its rank processes each have a device of their own to open."""
    try:
        import ctypes
        lib = ctypes.CDLL(os.path.join(HERE, "flowdenoising_amd", "libflowdn.so"))
        h = ctypes.c_void_p()
        if lib.fdn_create(ctypes.c_int(device), ctypes.byref(h)) == 0:
            lib.fdn_destroy(h)
    except Exception:       # an optimisation only: the real calls report real problems
        pass


WARM_THREAD_NAME = "fdn-warm-gpu"        # flowdenoising_amd.operators.handle() joins a thread of this name before it makes the first handle


def _early_start(argv):
    # `--gpus` in any spelling argparse accepts (it takes unambiguous prefixes: --g, --gp, --gpu; nothing else begins with --g):
    # the parent of a multi-GPU run must never open a GPU context
    def names(a):
        return a.split("=", 1)[0]
    if any(a in ("-h", "--help") or (len(names(a)) >= 3 and "--gpus".startswith(names(a))) for a in argv) or "FDN_RANK" in os.environ or "RANK" in os.environ:
        return
    device = 0
    for i, a in enumerate(argv):
        n = names(a)
        if len(n) >= 3 and "--device".startswith(n):
            v = a.split("=", 1)[1] if "=" in a else (argv[i + 1] if i + 1 < len(argv) else "")
            if v.isdigit():
                device = int(v)
    # this process never imports torch: /opt/rocm's HIP runtime, whichever thread loads the library first (flowdenoising_amd/_lib.py)
    os.environ.setdefault("FDN_SYSTEM_ROCM", "1")
    t = threading.Thread(target=_warm_gpu, args=(device,), daemon=True, name=WARM_THREAD_NAME)
    t.start()
    # a fast exit -- an option the parser refuses, a missing input file -- must not run the interpreter's and the HIP runtime's
    # teardown while this thread is still inside the driver's initialisation: atexit handlers run first, and this one waits
    import atexit
    atexit.register(t.join, 30.0)


if __name__ == "__main__":
    _early_start(sys.argv[1:])

from flowdenoising_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
