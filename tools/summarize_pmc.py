#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counter_collection.csv files -> one JSON (profiles/*/pmc_summary.json).

    python tools/summarize_pmc.py OUT.json DIR [DIR ...]      # DIRs: rocprofv3 -d outputs of separate --pmc passes
Kernel names are shortened to the function name; `launches` = dispatches seen, `mean_per_launch` = counter value
summed over the dispatch's XCDs / launches (rocprofv3 reports one row per dispatch and counter).
"""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    m = re.search(r"(k_\w+|__amd_rocclr_\w+)", name)
    return m.group(1) if m else name[:40]


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            per_dispatch = collections.defaultdict(float)
            names = {}
            for r in csv.DictReader(open(f)):
                key = (r["Dispatch_Id"], r["Counter_Name"])
                per_dispatch[key] += float(r["Counter_Value"])
                names[r["Dispatch_Id"]] = short(r["Kernel_Name"])
            for (disp, ctr), v in per_dispatch.items():
                acc[names[disp]][ctr].append(v)
    res = {k: {c: {"launches": len(v), "mean_per_launch": sum(v) / len(v)} for c, v in ctrs.items()} for k, ctrs in acc.items()}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out, "kernels:", len(res))


if __name__ == "__main__":
    main()
