#!/bin/bash
# GPU box: phase accounting + block timeline of the long grouped layout on configs[3]'s shard, next to k_solve's build
TAG=${1:-r05h}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
python scripts/phase_profile.py --config 4 --group 2 > "$OUT/phase_c4_grouped.json" 2>"$OUT/err.log"
python scripts/phase_profile.py --config 4 > "$OUT/phase_c4_lone.json" 2>>"$OUT/err.log"
python scripts/block_timeline.py 4 > "$OUT/timeline_c4_grouped.json" 2>>"$OUT/err.log"
GROUP_MODE=0 python scripts/block_timeline.py 4 > "$OUT/timeline_c4_lone.json" 2>>"$OUT/err.log"
TAG=$TAG python - <<'PY'
import json, os
for f in ("phase_c4_grouped", "phase_c4_lone"):
    try:
        p = json.load(open("gpurun_out/%s/%s.json" % (os.environ["TAG"], f)))
    except Exception as e:
        print(f, e); continue
    print(f, p['kernel_ms'], {k: round(v) for k, v in p['cycles_per_iteration'].items()}, round(p['cycles_per_trial_cost']),
          p.get('grouped_extra_cycles_per_iteration'), p['rollout_passes'])
for f in ("timeline_c4_grouped", "timeline_c4_lone"):
    try:
        print(f, open("gpurun_out/%s/%s.json" % (os.environ["TAG"], f)).read()[:1500])
    except Exception as e:
        print(f, e)
PY
