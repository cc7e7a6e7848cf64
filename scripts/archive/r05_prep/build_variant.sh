#!/bin/bash
# Builds a patched copy of the library next to the product one (CPU only, hipcc cross-compiles):
#   scripts/r05_prep/build_variant.sh lazy_inverse  ->  sdk_amd/variants/libspiral_hip_lazy_inverse.so
# (in-tree, git-ignored like every .so, so it travels to the GPU box with the snapshot; SPIRAL_HIP_LIB selects it)
set -eu
name=$1
R=$(cd "$(dirname "$0")/../.." && pwd)
T=$(mktemp -d)
mkdir -p "$T/sdk_amd" "$R/sdk_amd/variants"
cp -r "$R/sdk_amd/csrc" "$T/sdk_amd/csrc"
cp -r "$R/include" "$T/include"
rm -rf "$T/sdk_amd/csrc/build"
( cd "$T" && patch -p1 -s < "$R/scripts/r05_prep/$name.patch" )
make -C "$T/sdk_amd/csrc" -s -j8
cp "$T/sdk_amd/libspiral_hip.so" "$R/sdk_amd/variants/libspiral_hip_$name.so"
rm -rf "$T"
echo "$R/sdk_amd/variants/libspiral_hip_$name.so"
