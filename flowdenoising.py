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


def _early_start(argv):
    if any(a in ("-h", "--help", "--gpus") or a.startswith("--gpus=") for a in argv) or "FDN_RANK" in os.environ or "RANK" in os.environ:
        return
    device = 0
    for i, a in enumerate(argv):
        if a == "--device" and i + 1 < len(argv) and argv[i + 1].isdigit():
            device = int(argv[i + 1])
        elif a.startswith("--device=") and a[9:].isdigit():
            device = int(a[9:])
    # this process never imports torch: /opt/rocm's HIP runtime, whichever thread loads the library first (flowdenoising_amd/_lib.py)
    os.environ.setdefault("FDN_SYSTEM_ROCM", "1")
    threading.Thread(target=_warm_gpu, args=(device,), daemon=True).start()


if __name__ == "__main__":
    _early_start(sys.argv[1:])

from flowdenoising_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
