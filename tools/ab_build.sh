#!/bin/bash
# Development A/B builds: tools/ab_build.sh NAME "-DFLAG=.. ..." -> tools/ab_libs/lib_NAME.so (igemm.hip rebuilt with the flags, the other
# objects of the product build linked as they are).  Use with GWD_LIB=tools/ab_libs/lib_NAME.so on the micro-benchmarks.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/ab_libs
name=$1; flags=$2; src=${3:-igemm}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Wno-unused-result $flags -c gw_depth_amd/csrc/$src.hip -o tools/ab_libs/${src}_$name.o
objs=$(ls gw_depth_amd/csrc/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/ab_libs/${src}_$name.o -o tools/ab_libs/lib_$name.so
echo built tools/ab_libs/lib_$name.so
