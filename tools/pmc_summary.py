#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (one counter set per pass) per graphop kernel.

usage: tools/pmc_summary.py <dir> [prefixes...]     e.g.  tools/pmc_summary.py gpurun_out/pmc_r1 fetch write l2
Prints mean counter value per launch for every graphop kernel.  FETCH_SIZE / WRITE_SIZE are in
KiB (MI355X_MICROARCH.md, HBM section); on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide
(16 B/lane) coalesced reads, so the float4 row gathers are reported x2 ("fetch_corrected").
"""
import collections
import csv
import json
import os
import sys


def load(path, by_dispatch=None):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "graphop::" not in name:
            continue
        short = name.split("graphop::")[1].split("(")[0]
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if by_dispatch is not None and r.get("Dispatch_Id"):
            # per kernel FAMILY (template arguments stripped: what bench.py's pass records name) in dispatch order,
            # so that tools/make_traffic_json.py can tell the launches of one step apart (round-4 advice)
            fam = short.split("<")[0]
            by_dispatch[fam][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return acc


def main():
    d = sys.argv[1]
    prefixes = sys.argv[2:] or ["fetch", "write", "l2"]
    out = collections.defaultdict(dict)
    by_dispatch = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in prefixes:
        f = os.path.join(d, p + "_counter_collection.csv")
        if not os.path.exists(f):
            continue
        for k, cs in load(f, by_dispatch).items():
            for c, vals in cs.items():
                out[k][c] = sum(vals) / len(vals)
                out[k]["launches_seen"] = len(vals)
    for k, v in out.items():
        if "FETCH_SIZE" in v:
            v["fetch_bytes"] = v["FETCH_SIZE"] * 1024
            v["fetch_bytes_x2_wide_read_correction"] = v["FETCH_SIZE"] * 2048
        if "WRITE_SIZE" in v:
            v["write_bytes"] = v["WRITE_SIZE"] * 1024
        if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v:
            v["l2_hit_rate"] = v["TCC_HIT_sum"] / max(1.0, v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
    # a counter summed over the shader engines / XCDs appears once per dispatch in the CSV; if it appears several
    # times (one row per dimension) the rows of one dispatch are added up
    seq = {}
    for fam, cs in by_dispatch.items():
        seq[fam] = {}
        for c, pairs in cs.items():
            per = collections.OrderedDict()
            for did, v in sorted(pairs):
                per[did] = per.get(did, 0.0) + v
            seq[fam][c] = list(per.values())
    out["__by_dispatch__"] = seq
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
