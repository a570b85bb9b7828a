#!/bin/bash
# tools/build_variant_tu.sh NAME FILE.hip "EXTRA FLAGS" -- like build_variant.sh for any one translation unit of csrc/
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/tools/_variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function"
B=$(basename $2 .hip)
/opt/rocm/bin/hipcc $FLAGS $3 -c $R/4dgaussians_amd/csrc/$2 -o $R/tools/_variants/${B}_$1.o
OBJS=$(ls $R/4dgaussians_amd/build/*.o | grep -v "/$B.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_variants/libfdgs_$1.so $OBJS $R/tools/_variants/${B}_$1.o
echo $R/tools/_variants/libfdgs_$1.so
