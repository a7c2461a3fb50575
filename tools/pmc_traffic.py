#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 PMC passes (csv output).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dirF> -o f -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d <dirW> -o w -- python3 bench.py ...
    tools/pmc_traffic.py <dirF>/f_counter_collection.csv <dirW>/w_counter_collection.csv [clips fs level seconds] > profiles/rNN_pmc_traffic.json

The four trailing numbers name the workload of the profiled command (default: bench.py's N = 1 default, 1024 16000 5 10);
bench.py only attaches `roofline.traffic` from a summary whose workload equals the one it runs.

Counter values are KiB.  Correction from /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): on gfx950
FETCH_SIZE reports half the bytes of a coalesced streaming read, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE
is used as is."""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] != counter or "wsa::" not in row["Kernel_Name"]:
                continue
            name = row["Kernel_Name"].replace("void ", "").split("(")[0]
            acc[name][0] += float(row["Counter_Value"])
            acc[name][1] += 1
    return {k: v[0] / v[1] for k, v in acc.items() if v[1]}


def main():
    f = per_kernel(sys.argv[1], "FETCH_SIZE")
    w = per_kernel(sys.argv[2], "WRITE_SIZE")
    wl = sys.argv[3:7] if len(sys.argv) >= 7 else ["1024", "16000", "5", "10"]
    out = {"workload": {"clips": int(wl[0]), "fs": int(wl[1]), "level": int(wl[2]), "seconds": float(wl[3])},
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), averaged per launch",
           "units": "counter values are KiB; bytes = value*1024",
           "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a coalesced streaming "
                         "read -> read bytes = 2*FETCH_SIZE*1024; WRITE_SIZE used as is",
           "kernels": {}}
    for k in sorted(set(f) | set(w)):
        fk, wk = f.get(k, 0.0), w.get(k, 0.0)
        out["kernels"][k] = {"FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk, "hbm_bytes_per_launch": int(2 * fk * 1024 + wk * 1024)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
