#!/usr/bin/env python3
"""Print a few fields of a bench.py JSON line read from stdin (helper for tuning sweeps): tools/bench_field.py LABEL"""
import json
import sys
d = json.loads(sys.stdin.read())
rp = d.get("repeats") or {}
print(sys.argv[1] if len(sys.argv) > 1 else "", round(d["ms_per_step"], 4), "min/max", round(rp.get("min", 0), 4), round(rp.get("max", 0), 4), round(d["value"] / 1e8, 3),
      "single", round(d["single_batch"]["ms_per_step"], 3), {k: round(v, 3) for k, v in d["stage_ms"].items()})
