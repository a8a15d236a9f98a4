#!/bin/bash
# Builds the library as of a git revision (default HEAD) into ab/libA.so — the "before" side of scripts/ab_bench.sh;
# the working tree's build (toy-example-of-ilqr_amd/libcilqr_amd.so) is copied to ab/libB.so.
#   scripts/build_ref.sh [rev]
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
mkdir -p "$TMP/pkg/csrc" "$TMP/include" "$ROOT/ab"
for f in cilqr_amd.hip cilqr_device.hpp detmath.h scenario.cpp; do
  git -C "$ROOT" show "$REV:toy-example-of-ilqr_amd/csrc/$f" > "$TMP/pkg/csrc/$f"
done
git -C "$ROOT" show "$REV:include/cilqr_amd.h" > "$TMP/include/cilqr_amd.h"
(cd "$TMP/pkg/csrc" && /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -fPIC -shared \
   -Wno-unused-result cilqr_amd.hip scenario.cpp -o "$ROOT/ab/libA.so")
cp "$ROOT/toy-example-of-ilqr_amd/libcilqr_amd.so" "$ROOT/ab/libB.so"
rm -rf "$TMP"
ls -la "$ROOT/ab"
