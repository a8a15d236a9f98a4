#!/bin/bash
# Runs ON THE GPU BOX: the known-byte kernels of scripts/probes/traffic_calibration.hip through the same rocprofv3
# counter passes as the solve kernel (one counter per pass, --kernel-trace only), then scripts/calibrate_traffic.py
# turns the counter files into profiles-ready JSON.   scripts/calibrate_traffic.sh TAG
TAG=${1:-r03_cal}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
BIN=$ROOT/scripts/probes/traffic_calibration
[ -x "$BIN" ] || /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 "$ROOT/scripts/probes/traffic_calibration.hip" -o "$BIN"
cd /tmp && export TMPDIR=/tmp
"$BIN" > "$OUT/known_bytes.jsonl"
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    name=$(echo $c | tr ' ' '+')
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_$name" -- "$BIN" > "$OUT/pmc_$name.log" 2>&1
done
python "$ROOT/scripts/calibrate_traffic.py" "$OUT" > "$OUT/traffic_calibration.json"
tail -c 3000 "$OUT/traffic_calibration.json"
