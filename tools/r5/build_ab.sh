#!/bin/bash
# A second build of the library in which ONE source file is taken from a git revision: tools/r5/build_ab.sh <rev> <file.hip> <name>
# -> tools/r5/ab/lib_<name>.so (git-ignored, travels to the GPU box); run with SE3_LIB=tools/r5/ab/lib_<name>.so
set -e
rev=$1; file=$2; name=$3
R=$(cd $(dirname $0)/../.. && pwd)
mkdir -p $R/tools/r5/ab /tmp/ab_$name
git -C $R show $rev:se3et_amd/csrc/$file > /tmp/ab_$name/$file
cp $R/se3et_amd/csrc/*.h /tmp/ab_$name/; sed -i "s|\.\./\.\./include/se3et_hip.h|$R/include/se3et_hip.h|" /tmp/ab_$name/common.h
extra=""
grep -q SE3_EXACT_FP /tmp/ab_$name/$file && extra="-ffp-contract=off"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include $extra -c /tmp/ab_$name/$file -o /tmp/ab_$name/${file%.hip}.o
objs=$(ls $R/se3et_amd/csrc/build/*.o | grep -v "/${file%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/r5/ab/lib_$name.so $objs /tmp/ab_$name/${file%.hip}.o
ls -la $R/tools/r5/ab/lib_$name.so
