#!/bin/bash
# A/B builds of libcd_amd.so with one translation unit compiled differently (kernel experiments measured in ONE gpurun call):
#   tools/exp/build_variants.sh <name> <source.hip> [extra hipcc flags...]   ->  tools/exp/variants/libcd_amd_<name>.so
# Every other object comes from consistent_depth_amd/csrc/build (run consistent_depth_amd.build_native first).
# Load with CD_AMD_LIB=tools/exp/variants/libcd_amd_<name>.so (consistent_depth_amd/_native.py).
set -e
REPO=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; SRC=$2; shift 2
OUT=$REPO/tools/exp/variants
mkdir -p $OUT
OBJ=$OUT/$(basename $SRC).$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I $REPO/include -I $REPO/consistent_depth_amd/csrc "$@" -c $SRC -o $OBJ
OBJS=$(ls $REPO/consistent_depth_amd/csrc/build/*.o | grep -v "/$(basename $SRC).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libcd_amd_$NAME.so $OBJS $OBJ -ldl
echo $OUT/libcd_amd_$NAME.so
