#!/bin/bash
# GPU box: cycle accounting of the long grouped layout on configs[3]'s shard (sliced solves off: a resumed solve's accounting restarts)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
CILQR_TUNE=group_slice_long=0,group_slice=0 python scripts/phase_profile.py --config ${1:-4} --group 2 > gpurun_out/phase_c${1:-4}_grp.json 2>gpurun_out/phase_err.log
python - gpurun_out/phase_c${1:-4}_grp.json <<'PY'
import json, sys
p = json.load(open(sys.argv[1]))
print(p["kernel_ms"], {k: round(v) for k, v in p["cycles_per_iteration"].items()}, round(p["cycles_per_trial_cost"]),
      p.get("grouped_extra_cycles_per_iteration"), p["rollout_passes"], p.get("rollout_steps_in_the_small_angle_form_frac"))
PY
