#!/bin/bash
# L2 hit / miss counts of every kernel of one attention step under given tuning knobs:
#   bash tools/pmc_l2.sh <outdir> [tune_sweep args...]     (one rocprofv3 --pmc pass, under a timeout)
OUT=${1:-gpurun_out/pmc_l2}; shift || true
mkdir -p "$OUT"
HERE=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$HERE"
timeout -k 5 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT" -o l2 -- python tools/tune_sweep.py --steps 1 "$@" > "$OUT/l2.log" 2>&1
echo "[l2] rc=$?"
D=$(dirname "$(find "$OUT" -name '*_counter_collection.csv' | head -1)")
python tools/pmc_summary.py "$D" l2 > "$OUT/summary.json" && python - "$OUT/summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if isinstance(v, dict) and "TCC_HIT_sum" in v and v["TCC_HIT_sum"] + v["TCC_MISS_sum"] > 1e7:
        print("%-60s hit %8.1fM miss %8.1fM  launches %d" % (k[:60], v["TCC_HIT_sum"] / 1e6, v["TCC_MISS_sum"] / 1e6, v.get("launches_seen", 0)))
PY
