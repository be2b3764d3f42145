#!/bin/bash
# Hardware counters behind the labeling experiment (profiles/r5_experiments.txt): vector-L1 accesses vs L1 -> L2 read
# requests, and L2 hits / misses, per graphop kernel, for the labelings given (ss sd ds dd; first letter rows).
#   bash tools/labeling_pmc.sh <outdir> [cases...]
OUT=$1; shift
CASES=${@:-ss sd}
mkdir -p "$OUT"
HERE=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$HERE"
for c in $CASES; do
  for grp in "l1:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "l2:TCC_HIT_sum TCC_MISS_sum"; do
    name=${c}_${grp%%:*}
    timeout -k 5 280 rocprofv3 --pmc ${grp#*:} --output-format csv -d "$OUT" -o "$name" -- python tools/labeling_experiment.py --cases $c --steps 1 > "$OUT/$name.log" 2>&1
    echo "[labeling_pmc] $name rc=$?"
    D=$(dirname "$(find "$OUT" -name "${name}_counter_collection.csv" | head -1)")
    python tools/pmc_summary.py "$D" "$name" > "$OUT/${name}_summary.json"
  done
done
