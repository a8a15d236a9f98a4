#!/bin/bash
# Build container: stamp what is about to be measured, send the collection to the GPU box, summarise into profiles/.
#   scripts/r05_collect.sh TAG [quick]
TAG=${1:-r05_v1}; QUICK=${2:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
mkdir -p gpurun_out/$TAG
python - "$TAG" <<'PY'
import json, subprocess, sys
sys.path.insert(0, ".")
import bench
rev = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "toy-example-of-ilqr_amd/csrc", "include"], capture_output=True, text=True).stdout.strip())
json.dump({"git": rev, "csrc_dirty_at_collection": dirty, "csrc_sha16": bench.csrc_fingerprint()}, open(f"gpurun_out/{sys.argv[1]}/_collected.json", "w"))
print("stamped", rev, dirty)
PY
/usr/local/graft/bin/gpurun --timeout 5400 -- "bash scripts/collect_profiles.sh $TAG $QUICK" > /tmp/gpurun_collect_$TAG.log 2>&1
tail -12 /tmp/gpurun_collect_$TAG.log | cut -c1-300
