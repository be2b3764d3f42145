#!/bin/bash
# One rocprofv3 --pmc pass (at most two counters of one hardware block) over one step of tools/tune_sweep.py,
# summarised per graphop kernel.   bash tools/pmc_pair.sh <outdir> <name> "<counters>" [tune_sweep args...]
OUT=$1; NAME=$2; CTRS=$3; shift 3
mkdir -p "$OUT"
HERE=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$HERE"
timeout -k 5 200 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT" -o "$NAME" -- python tools/tune_sweep.py --steps 1 "$@" > "$OUT/$NAME.log" 2>&1
echo "[pmc_pair] $NAME rc=$?"
D=$(dirname "$(find "$OUT" -name "${NAME}_counter_collection.csv" | head -1)")
python tools/pmc_summary.py "$D" "$NAME" > "$OUT/${NAME}_summary.json"
