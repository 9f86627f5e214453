#!/bin/bash
# usage: tools/ab_build_wt.sh <name> [extra compiler flags]  -> vcfgl_amd/lib_ab/<name>.so = the library built from the WORKING TREE's
# vcfgl_amd/csrc with the extra flags (experiments: -DVGL_EXP_...; A/B timing with tools/ab_time.sh)
set -eu
name=$1; shift || true
tmp=$(mktemp -d)
mkdir -p "$tmp/vcfgl_amd" "$tmp/build"
cp -r vcfgl_amd/csrc "$tmp/vcfgl_amd/csrc"; cp -r include "$tmp/include"
mkdir -p vcfgl_amd/lib_ab
( cd "$tmp/vcfgl_amd/csrc" && make -s -j8 ../lib/libvcfgl_hip.so CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -fvisibility=hidden -mllvm -amdgpu-atomic-optimizer-strategy=None $*" >/dev/null 2>&1 )
cp "$tmp/vcfgl_amd/lib/libvcfgl_hip.so" "vcfgl_amd/lib_ab/$name.so"
rm -rf "$tmp"
ls -la "vcfgl_amd/lib_ab/$name.so"
