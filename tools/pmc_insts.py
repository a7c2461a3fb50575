#!/usr/bin/env python3
"""Wave-instructions per kernel launch by class + the shader clock under this load, from one rocprofv3 PMC pass (csv output).

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES \\
              --output-format csv -d <dir> -o p -- python3 bench.py --in-flight 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extra
    tools/pmc_insts.py <dir>/.../p_counter_collection.csv [clips fs level seconds] > profiles/rNN_pmc_insts.json

Counts are millions per launch (averaged over the launches of the pass).  clock_GHz_under_load = SQ_BUSY_CYCLES (summed over the chip's 32
shader engines) / 32 / kernel duration, averaged over the kernels that run longer than 50 us, weighted by duration: the clock the chip
sustains while these kernels run (the pass serialises kernels and runs ~3 % below an unprofiled run: MI355X_MICROARCH.md, DVFS).
bench.py's `issue_bound` object reads this file."""
import csv
import json
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dur = defaultdict(lambda: [0.0, 0])
    with open(sys.argv[1], newline="") as fh:
        for row in csv.DictReader(fh):
            if "wsa::" not in row["Kernel_Name"]:
                continue
            k = row["Kernel_Name"].replace("void ", "").split("(")[0]
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
            if row["Counter_Name"] == "SQ_BUSY_CYCLES":
                d = dur[k]; d[0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"]); d[1] += 1
    wl = sys.argv[2:6] if len(sys.argv) >= 6 else ["1024", "16000", "5", "10"]
    kernels = {k: {c: round(v[0] / v[1] / 1e6, 3) for c, v in sorted(cs.items())} for k, cs in acc.items()}
    num = den = 0.0
    for k, cs in kernels.items():
        if "SQ_BUSY_CYCLES" in cs and dur[k][1]:
            ns = dur[k][0] / dur[k][1]
            kernels[k]["duration_us"] = round(ns / 1e3, 2)
            if ns > 50e3:
                num += cs["SQ_BUSY_CYCLES"] * 1e6 / 32; den += ns
    out = {"workload": {"clips": int(wl[0]), "fs": int(wl[1]), "level": int(wl[2]), "seconds": float(wl[3])},
           "source": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES (one pass) "
                     "of bench.py --in-flight 1, averaged per launch; millions of wave-instructions",
           "clock_GHz_under_load": round(num / den, 3) if den else None,
           "kernels": kernels}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
