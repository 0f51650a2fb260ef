#!/bin/bash
# Build an experimental variant of libdfusion_hip.so into build_variants/<name>.so
# usage: tools/build_variant.sh <name> [-DMACRO ...]     (used with tools/kbench*.py --lib)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_variants/obj_$name
objs=""
for s in dynamicfusion_body_amd/csrc/*.hip; do
  o=build_variants/obj_$name/$(basename ${s%.hip}).o
  # only dfh_fuse_volume / integrate see the macros; other objects are reused from the main build when present
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude "$@" -c $s -o $o &
  objs="$objs $o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o build_variants/$name.so $objs
echo build_variants/$name.so
