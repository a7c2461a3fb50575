#!/bin/bash
export WSA_TUNING_ENV=1   # libwsa reads its tuning switches only when this is set (csrc/api.hip Tuning::from_env)
# instruction counts of the peaks probe's launches, per dbg value (TUNING=1 build).  usage (GPU box): tools/pmc_probe.sh "0 1 2"
export TMPDIR=/tmp
ROOT=$(pwd); out=/tmp/pmc_probe; rm -rf $out
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out -o p -- python3 $ROOT/tools/peaks_probe.py $1 > /tmp/probe.out 2>&1)
cat /tmp/probe.out | tail -8
python3 - $(find $out -name '*counter_collection.csv' | head -1) <<'PY'
import csv, sys
from collections import defaultdict, OrderedDict
rows = [r for r in csv.DictReader(open(sys.argv[1], newline="")) if "peaks_kernel" in r["Kernel_Name"]]
by = OrderedDict()
for r in rows:
    by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by)
# 21 launches per dbg value
for g in range(0, len(ids), 21):
    grp = ids[g + 1:g + 21]
    if not grp: continue
    acc = defaultdict(float)
    for i in grp:
        for k, v in by[i].items(): acc[k] += v / len(grp)
    print("group", g // 21, {k: round(v / 1e6, 2) for k, v in sorted(acc.items())})
PY
