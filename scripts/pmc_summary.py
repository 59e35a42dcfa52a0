#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc passes (one counter_collection.csv per pass) as JSON.

    python scripts/pmc_summary.py out.json pass1_counter_collection.csv [pass2_counter_collection.csv ...]

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB per dispatch.
"""
import collections
import csv
import json
import sys


def main():
    out, files = sys.argv[1], sys.argv[2:]
    res = collections.OrderedDict()
    for f in files:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        with open(f) as fh:
            for r in csv.DictReader(fh):
                agg[r["Counter_Name"]][r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for cname, kernels in agg.items():
            d = res.setdefault(cname, collections.OrderedDict())
            for k, v in sorted(kernels.items(), key=lambda kv: -sum(kv[1])):
                if not (k.startswith("void k_") or k.startswith("k_")):
                    continue
                d[k] = {"dispatches": len(v), "mean_KiB": sum(v) / len(v), "min_KiB": min(v), "max_KiB": max(v)}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
