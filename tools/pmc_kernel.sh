#!/bin/bash
export WSA_TUNING_ENV=1   # libwsa reads its tuning switches only when this is set (csrc/api.hip Tuning::from_env)
# tuning helper: instruction counts and wait / LDS counters of the kernels whose name contains $1 (two PMC passes of bench.py --in-flight 1).
# usage (GPU box): tools/pmc_kernel.sh <substring> [bench args]      PMC_PROG="tools/resample_probe.py 44100" tools/pmc_kernel.sh resample: another program instead of bench.py
pat=${1:-wsa::}; shift
export TMPDIR=/tmp
ROOT=$(pwd)
for pass in 1 2; do
  out=/tmp/pmc_k$pass; rm -rf $out
  if [ $pass = 1 ]; then C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
  else C="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; fi
  if [ -n "$PMC_PROG" ]; then (cd /tmp && rocprofv3 --kernel-trace --pmc $C --output-format csv -d $out -o p -- python3 $ROOT/$PMC_PROG > /dev/null 2>&1)
  else (cd /tmp && rocprofv3 --kernel-trace --pmc $C --output-format csv -d $out -o p -- python3 $ROOT/bench.py --in-flight 1 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-extra "$@" > /dev/null 2>&1); fi
  python3 - "$pat" $(find $out -name '*counter_collection.csv' | head -1) <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for row in csv.DictReader(open(sys.argv[2], newline="")):
    if sys.argv[1] not in row["Kernel_Name"]: continue
    k = row["Kernel_Name"].replace("void ", "").split("(")[0]
    a = acc[k][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    print(k, {c: round(v[0] / v[1] / 1e6, 3) for c, v in sorted(cs.items())}, "(millions per launch)")
PY
done
