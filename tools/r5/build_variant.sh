#!/bin/bash
# A build of the library with extra -D flags on ONE source file: tools/r5/build_variant.sh <file.hip> <name> <flags...> -> tools/r5/ab/lib_<name>.so
set -e
file=$1; name=$2; shift 2
R=$(cd $(dirname $0)/../.. && pwd)
mkdir -p $R/tools/r5/ab /tmp/abv_$name
extra=""
grep -q SE3_EXACT_FP $R/se3et_amd/csrc/$file && extra="-ffp-contract=off"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra "$@" -c $R/se3et_amd/csrc/$file -o /tmp/abv_$name/${file%.hip}.o
objs=$(ls $R/se3et_amd/csrc/build/*.o | grep -v "/${file%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/r5/ab/lib_$name.so $objs /tmp/abv_$name/${file%.hip}.o
