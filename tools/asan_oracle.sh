#!/bin/bash
# CPU only: rebuild the oracle with AddressSanitizer + UBSan and run the oracle-facing CPU tests against it, then restore
# the normal build.  (The GPU pool has no sanitizer support; the oracle shares the test process with the HIP library, so
# a heap error in it would surface anywhere.)
set -e
cd "$(dirname "$0")/.."
ASAN=$(gcc -print-file-name=libasan.so)
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -fno-fast-math -fopenmp -fPIC -fvisibility=hidden \
    -std=gnu11 -shared -o oracle/libfdn_oracle.so oracle/fdn_oracle.c -lm
trap 'make -C oracle -B > /dev/null' EXIT
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_oracle.py tests/test_cv2_pin.py tests/test_distributed_cpu.py -x -q -m "not gpu"
