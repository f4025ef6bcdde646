#!/bin/bash
# tools/ab_build.sh <name> <python patch script>: an experimental build of the library beside the product, for same-box A/B runs on the GPU (TERRA_LIB=tools/_ab/<name>/libterra_hip.so).
# The patch script gets the directory of a COPY of 3dworld_amd/csrc as argv[1] and edits it; the product sources are not touched.  tools/_ab/ is git-ignored.
set -e
NAME=$1; PATCH=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=$ROOT/tools/_ab/$NAME
rm -rf "$D"; mkdir -p "$D"
cp -r "$ROOT/3dworld_amd/csrc" "$D/csrc"
mkdir -p "$D/include"; cp "$ROOT/include/terra.h" "$D/include/"
[ -n "$PATCH" ] && python3 "$PATCH" "$D/csrc"
# (csrc includes "../../include/terra.h": keep the relative layout)
mkdir -p "$D/x"; mv "$D/csrc" "$D/x/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-gpu-rdc -Wno-unused-function -Wno-unknown-pragmas -Wno-unused-result "$D/x/csrc/terra_hip.hip" "$D/x/csrc/terra_fz.hip" -o "$D/libterra_hip.so" -lz
echo "built $D/libterra_hip.so"
