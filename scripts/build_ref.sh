#!/bin/bash
# Builds the production library as of a git revision (default HEAD) into ab/libA.so — the "before" side of
# scripts/ab_bench.sh — with that revision's own build.py; the working tree's build is copied to ab/libB.so.
#   scripts/build_ref.sh [rev]
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
mkdir -p "$ROOT/ab"
git -C "$ROOT" archive "$REV" toy-example-of-ilqr_amd include | tar -x -C "$TMP"
(cd "$TMP/toy-example-of-ilqr_amd" && python -c "
import build
print(build.build_library(force=True, out='$ROOT/ab/libA.so'))")
cp "$ROOT/toy-example-of-ilqr_amd/libcilqr_amd.so" "$ROOT/ab/libB.so"
rm -rf "$TMP"
ls -la "$ROOT/ab"
