#!/usr/bin/env python3
"""profiles/pmc_traffic.json from a tools/pmc_summary.py summary: HBM-side bytes per launch for each
kernel family, corrected as MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE (KiB) x2 for
wide 16-B/lane coalesced reads on gfx950, WRITE_SIZE (KiB) as is; separate --pmc passes.
usage: tools/make_traffic_json.py <pmc_summary.json> <workload-tag> [bench-line.json] > profiles/pmc_traffic.json
With the bench line of one of the counter runs (its roofline.passes name the kernel family of every pass; the passes of
a step run in bench.PASS_TAGS order) the k-th dispatch of a family inside a step is attributed to its pass tag:
"passes": {tag: bytes per launch} -- the row- and column-major launches of one instantiation are told apart."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernels_sha  # noqa: E402  (hash of the kernel sources the counters were collected on)

summ = json.load(open(sys.argv[1]))
by_dispatch = summ.pop("__by_dispatch__", {})
fam = {}
for name, v in summ.items():
    if not name or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    f = re.sub(r"<.*", "", name)          # the family bench.py names: k_spmm_wown_f32, k_sddmm_wown_f32, ...
    e = fam.setdefault(f, {"bytes": 0.0, "n": 0, "variants": {}})
    b = v["FETCH_SIZE"] * 2048 + v["WRITE_SIZE"] * 1024
    e["bytes"] += b; e["n"] += 1
    e["variants"][name] = {"fetch_KiB": v["FETCH_SIZE"], "write_KiB": v["WRITE_SIZE"], "hbm_bytes_per_launch": b,
                           "l2_hit_rate": v.get("l2_hit_rate")}
out = {"workload": sys.argv[2], "kernels_sha": kernels_sha(),
       "note": "FETCH_SIZE*2 (gfx950 wide-read correction) + WRITE_SIZE, KiB->bytes, mean over the kernel "
               "variants of the family; counters are L2 fabric-side requests, Infinity-Cache hits included; "
               "4-B id streams are uncalibrated (MI355X_MICROARCH.md, HBM)",
       "kernels": {f: {"hbm_bytes_per_launch": int(e["bytes"] / e["n"]), "variants": e["variants"]} for f, e in fam.items()}}
if len(sys.argv) > 3 and by_dispatch:
    from bench import PASS_TAGS
    txt = open(sys.argv[3]).read()
    line = json.loads([l for l in txt.splitlines() if l.startswith("{")][-1])
    tag_kernel = {t: p["kernel"] for t, p in line["roofline"]["passes"].items()}
    per_tag, skipped = {}, []
    for f in sorted(set(tag_kernel.values())):
        tags = [t for t in PASS_TAGS if tag_kernel.get(t) == f]      # this family's launches inside one step, in order
        fs, ws = by_dispatch.get(f, {}).get("FETCH_SIZE", []), by_dispatch.get(f, {}).get("WRITE_SIZE", [])
        # the two counters come from separate runs of the same command: same number of dispatches, a whole number of steps
        if not tags or not fs or len(fs) != len(ws) or len(fs) % len(tags) != 0:
            skipped.append(f)
            continue
        for k, t in enumerate(tags):
            b = [fs[i] * 2048 + ws[i] * 1024 for i in range(k, len(fs), len(tags))]
            per_tag[t] = {"hbm_bytes_per_launch": int(sum(b) / len(b)), "kernel": f, "launches_seen": len(b)}
    out["passes"] = per_tag
    if skipped:
        out["passes_skipped"] = skipped
print(json.dumps(out, indent=1))
