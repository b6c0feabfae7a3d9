#!/bin/bash
# A/B of HIP runtime settings on the main bench leg: tools/env_ab.sh "NAME=VALUE ..." ...  (one quoted group per run)
set -u
cd "$(dirname "$0")/.."
i=0
for group in "$@"; do
    i=$((i + 1))
    out=gpurun_out/env_ab_$i.json
    env $group timeout -k 10 200 python -u bench.py --no-extra-legs --no-cpu-baseline --steps 3 > $out 2> gpurun_out/env_ab_$i.err || { echo "run $i ($group) failed"; tail -5 gpurun_out/env_ab_$i.err; exit 1; }
    python - "$out" "$group" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ds = d["kernel_breakdown"].get("decode_step", {})
print("%-60s value %9.1f  ms/step %8.1f  attn %5.2f us  decode launch ms/step %s  step frac %s" % (
    sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], ds.get("launch_ms_per_step"), ds.get("frac_of_hbm_peak")))
PY
done
