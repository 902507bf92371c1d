#!/bin/bash
# usage: build_variant.sh NAME "EXTRA HIPCC FLAGS"  -> build_variants/lib_NAME.so (a full libflowdn.so built with the extra
# -D switches; tools/sweep_variants.sh benches every library in build_variants/ inside one gpurun call)
set -e
name=$1; extra=$2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/flowdenoising_amd/csrc
obj=/tmp/fdn_variant_$name
mkdir -p $obj $root/build_variants
FLAGS="-O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off --offload-arch=${ARCH:-gfx950} -Wall -Wno-unused-function -I$src -I$root/include $extra"
for f in fdn_api fdn_kernels fdn_iter; do
  # only the fused / iter kernels take experiment switches: reuse the product objects for the rest when they are current
  if [ "$f" != fdn_iter ] && [ -f $src/obj/$f.o ] && [ $src/obj/$f.o -nt $src/$f.hip ]; then cp $src/obj/$f.o $obj/$f.o; else /opt/rocm/bin/hipcc $FLAGS -c -o $obj/$f.o $src/$f.hip & fi
done
/opt/rocm/bin/hipcc $FLAGS -mllvm -amdgpu-sched-strategy=max-ilp -c -o $obj/fdn_fused.o $src/fdn_fused.hip &
wait
/opt/rocm/bin/hipcc $FLAGS -shared -o $root/build_variants/lib_$name.so $obj/fdn_api.o $obj/fdn_kernels.o $obj/fdn_fused.o $obj/fdn_iter.o
ls -la $root/build_variants/lib_$name.so
