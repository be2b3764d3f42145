#!/bin/bash
# One rocprofv3 --pmc pass with the given counter pair over one attention step:
#   bash tools/pmc_pair.sh <outdir> <counterA> <counterB> [tune_sweep args...]
OUT=$1; A=$2; B=$3; shift 3
mkdir -p "$OUT"
HERE=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$HERE"
timeout -k 5 200 rocprofv3 --pmc $A $B --output-format csv -d "$OUT" -o pair -- python tools/tune_sweep.py --steps 1 "$@" > "$OUT/pair.log" 2>&1
echo "[pair] rc=$?"
D=$(dirname "$(find "$OUT" -name 'pair_counter_collection.csv' | head -1)")
python - "$D/pair_counter_collection.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "graphop::" not in n: continue
    acc[n.split("graphop::")[1].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if any(sum(v) / len(v) > 5e6 for v in cs.values()):
        print("%-58s" % k[:58], "  ".join("%s %.1fM" % (c, sum(v) / len(v) / 1e6) for c, v in sorted(cs.items())))
PY
