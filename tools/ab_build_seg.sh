#!/bin/bash
# usage: tools/ab_build_seg.sh <name> [SEGFLAGS ...]  -> vcfgl_amd/lib_ab/<name>.so = the library built from the WORKING TREE with the given options for
# vgl_sample_seg.hip alone (the k_sample_seg kernels; Makefile: SEGFLAGS); A/B timing with tools/ab_time.sh
set -eu
name=$1; shift || true
tmp=$(mktemp -d)
mkdir -p "$tmp/vcfgl_amd" "$tmp/build"
cp -r vcfgl_amd/csrc "$tmp/vcfgl_amd/csrc"; cp -r include "$tmp/include"
mkdir -p vcfgl_amd/lib_ab
( cd "$tmp/vcfgl_amd/csrc" && make -s -j8 ../lib/libvcfgl_hip.so SEGFLAGS="$*" >/dev/null 2>"$tmp/err.txt" ) || { tail -5 "$tmp/err.txt"; exit 1; }
cp "$tmp/vcfgl_amd/lib/libvcfgl_hip.so" "vcfgl_amd/lib_ab/$name.so"
rm -rf "$tmp"
ls -la "vcfgl_amd/lib_ab/$name.so"
