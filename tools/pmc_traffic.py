#!/usr/bin/env python3
"""Fold the FETCH_SIZE / WRITE_SIZE summaries (tools/prof_summary.py) of two separate rocprofv3 --pmc passes into
per-family HBM traffic.  Units and corrections per MI355X_MICROARCH.md §HBM: both counters are in KiB; on gfx950
FETCH_SIZE tallies 128-B requests at 64 B, so reads are doubled; WRITE_SIZE is exact.

usage: pmc_traffic.py <pmc_fetch_size.csv> <pmc_write_size.csv> <steps in each pass> <out.json> [config key]
With a config key the result is merged into out.json under configs[key] (the layout bench.py reads)."""
import csv
import json
import sys
from collections import defaultdict

import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_families import family_of as fam      # kernel -> family TABLE (round 2 matched name prefixes and booked
                                                  # conv3x3_halo / bn_fold_wgrad / *_k1 / smallc under "other")


def main():
    fpath, wpath, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    rd, wr, n = defaultdict(float), defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(fpath)):
        rd[fam(r["kernel"])] += float(r["FETCH_SIZE_total"]) * 1024 * 2
        n[fam(r["kernel"])] += int(r["launches"])
    for r in csv.DictReader(open(wpath)):
        wr[fam(r["kernel"])] += float(r["WRITE_SIZE_total"]) * 1024
    res = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py "
                     "--steps 1 --warmup 1 --profile-steps 0 --no-cpu-baseline",
           "corrections": "KiB -> bytes; FETCH_SIZE x2 (gfx950 128-B requests tallied at 64 B); WRITE_SIZE exact",
           "steps_per_pass": steps, "families": {}}
    for f in rd:
        res["families"][f] = {"launches_per_step": n[f] / steps, "read_bytes_per_step": rd[f] / steps,
                              "write_bytes_per_step": wr[f] / steps,
                              "bytes_per_launch": (rd[f] + wr[f]) / max(n[f], 1)}
    if len(sys.argv) > 5:
        key = sys.argv[5]
        try:
            doc = json.load(open(out))
        except (OSError, ValueError):
            doc = {"configs": {}}
        doc.setdefault("configs", {})[key] = res
        res = doc
        json.dump(doc, open(out, "w"), indent=1)
        print(key, json.dumps({f: v for f, v in doc["configs"][key]["families"].items() if f.startswith("conv")}))
        return
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["families"]["conv"]))


if __name__ == "__main__":
    main()
