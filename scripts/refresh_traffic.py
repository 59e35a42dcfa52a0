#!/usr/bin/env python3
"""Writes profiles/traffic_latest.json from two rocprofv3 --pmc passes over bench.py (FETCH_SIZE and WRITE_SIZE,
each in a run of its own, as the counters cannot share a pass with a trace):

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o p -- python3 bench.py --steps 1 \
        --warmup 1 --no-cpu-baseline --no-hbm-only
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o p -- python3 bench.py ... (same)
    python scripts/refresh_traffic.py gpurun_out/pmc_fetch/p_counter_collection.csv \
        gpurun_out/pmc_write/p_counter_collection.csv [kernel_stats.csv [name it is committed under]]

gfx950: FETCH_SIZE tallies 64 B per 128-B read request, so the bytes read are twice the counter (checked against
the cell kernel's known compulsory reads); WRITE_SIZE is exact.  Both are reported in KiB per dispatch.  The file
carries a hash of the kernel sources so that bench.py can tell a figure taken with other kernels."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def means(path, counter):
    agg = collections.defaultdict(list)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024.0)
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def main():
    import bench
    fetch, nf = means(sys.argv[1], "FETCH_SIZE")
    write, _ = means(sys.argv[2], "WRITE_SIZE")

    def pick(d, prefix):
        ks = [k for k in d if prefix in k]
        # the variant launched most often is the one inside the timed loop
        return max(ks, key=lambda k: nf.get(k, 0)) if ks else None
    kc, kv = pick(fetch, "k_sub_cells_split"), pick(fetch, "k_sub_vertices")
    out = {
        "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, QU60, bench.py --steps 1 --warmup 1 "
                  "--no-cpu-baseline --no-hbm-only (scripts/refresh_traffic.py)",
        "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request, so read bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact",
        "kernel_source_hash": bench.kernel_source_hash(), "ir_source_hash": bench.ir_source_hash(),
        "cell_kernel": kc, "cell_kernel_hbm_bytes_per_launch": 2.0 * fetch[kc] + write.get(kc, 0.0),
        "cell_kernel_fetch_raw_bytes": fetch[kc], "cell_kernel_write_bytes": write.get(kc, 0.0),
        "vertex_kernel_hbm_bytes_per_launch": 2.0 * fetch[kv] + write.get(kv, 0.0) if kv else None,
        "advection_hbm_bytes_per_step": {k: 2.0 * fetch[k] + write.get(k, 0.0) for k in fetch if "k_ir_" in k},
    }
    if len(sys.argv) > 3:  # rocprofv3 --kernel-trace --stats of the same command: average durations of the two kernels
        with open(sys.argv[3]) as fh:
            for r in csv.DictReader(fh):
                if r["Name"] == kc:
                    out["cell_kernel_avg_ns_rocprof"] = float(r["AverageNs"])
                if r["Name"] == kv:
                    out["vertex_kernel_avg_ns_rocprof"] = float(r["AverageNs"])
        out["kernel_stats_file"] = sys.argv[4] if len(sys.argv) > 4 else os.path.basename(sys.argv[3])
    with open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
