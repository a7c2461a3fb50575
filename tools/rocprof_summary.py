#!/usr/bin/env python3
"""Condense a rocprofv3 rocpd database (or *_kernel_stats.csv) into the short per-kernel summary
that is committed under profiles/.   usage: tools/rocprof_summary.py <results.db|stats.csv> [prefix]"""
import csv
import sqlite3
import sys


def from_db(path, prefix):
    db = sqlite3.connect(path)
    rows = list(db.execute("select * from top_kernels"))
    out = []
    for name, calls, total_us, avg_us, pct in rows:
        if prefix is None or prefix in name:
            out.append((name[:90], int(calls), float(total_us), float(avg_us), float(pct)))
    return out


def main():
    path = sys.argv[1]
    prefix = sys.argv[2] if len(sys.argv) > 2 else "wsa::"
    rows = from_db(path, prefix)
    print(f"# rocprofv3 --kernel-trace --stats summary ({path}); kernels matching '{prefix}'")
    print(f"{'kernel':<92}{'calls':>7}{'total_us':>14}{'avg_us':>12}{'pct_of_gpu_time':>17}")
    for r in rows:
        print(f"{r[0]:<92}{r[1]:>7}{r[2]:>14.1f}{r[3]:>12.2f}{r[4]:>17.2f}")


if __name__ == "__main__":
    main()
