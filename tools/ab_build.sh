#!/bin/bash
# usage: tools/ab_build.sh <git rev> <name>   -> vcfgl_amd/lib_ab/<name>.so = the library built from vcfgl_amd/csrc at <rev>
# (A/B timing on one GPU box: VGL_LIB=vcfgl_amd/lib_ab/<name>.so python bench.py ...)
set -eu
rev=$1; name=$2
tmp=$(mktemp -d)
git archive "$rev" vcfgl_amd/csrc include | tar -x -C "$tmp"
mkdir -p vcfgl_amd/lib_ab
( cd "$tmp/vcfgl_amd/csrc" && make -s ../lib/libvcfgl_hip.so >/dev/null 2>&1 )
cp "$tmp/vcfgl_amd/lib/libvcfgl_hip.so" "vcfgl_amd/lib_ab/$name.so"
rm -rf "$tmp"
ls -la "vcfgl_amd/lib_ab/$name.so"
