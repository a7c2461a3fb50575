#!/bin/bash
export WSA_TUNING_ENV=1   # libwsa reads its tuning switches only when this is set (csrc/api.hip Tuning::from_env)
# tuning helper: SQ activity counters per kernel launch.  usage (GPU box): tools/pmc_util.sh [bench args]
export TMPDIR=/tmp
ROOT=$(pwd); out=/tmp/pmc_util; rm -rf $out
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $out -o p -- python3 $ROOT/bench.py --in-flight 1 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-extra "$@" > /dev/null 2>&1)
python3 - $(find $out -name '*counter_collection.csv' | head -1) <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for row in csv.DictReader(open(sys.argv[1], newline="")):
    if "wsa::" not in row["Kernel_Name"]: continue
    k = row["Kernel_Name"].replace("void ", "").split("(")[0]
    a = acc[k][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    print(k, {c: round(v[0] / v[1] / 1e6, 2) for c, v in sorted(cs.items())}, "(millions per launch)")
PY
