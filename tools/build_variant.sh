#!/bin/bash
# tools/build_variant.sh NAME "EXTRA FLAGS" -- development build of deform.hip with extra flags (e.g. -DFDGS_PROFILE_D4), linked
# with the regular objects of the other translation units into tools/_variants/libfdgs_NAME.so; run with FDGS_LIB=<that path>.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/tools/_variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS $2 -c $R/4dgaussians_amd/csrc/deform.hip -o $R/tools/_variants/deform_$1.o
OBJS=$(ls $R/4dgaussians_amd/build/*.o | grep -v "/deform.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_variants/libfdgs_$1.so $OBJS $R/tools/_variants/deform_$1.o
echo $R/tools/_variants/libfdgs_$1.so
