#!/bin/bash
# Build a second copy of libgill_amd.so with extra compiler flags, for same-box A/B runs (tools/ab_bench.sh):
#   tools/build_variant.sh <name> "<extra flags>"     ->  tools/_lib_<name>.so   (git-ignored; travels to the GPU box)
set -e
name=$1; flags=$2
d=/tmp/gill_variant_$name
rm -rf $d && mkdir -p $d/gill_amd/csrc $d/include && cp gill_amd/csrc/*.hip gill_amd/csrc/*.h gill_amd/csrc/Makefile $d/gill_amd/csrc/ && cp include/*.h $d/include/
sed -i "s#^TARGET  = .*#TARGET  = $PWD/tools/_lib_$name.so#" $d/gill_amd/csrc/Makefile
make -C $d/gill_amd/csrc -j16 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -Wno-c++20-extensions -ffp-contract=fast $flags" 2>&1 | grep -E "error|Error" || true
ls -la tools/_lib_$name.so
