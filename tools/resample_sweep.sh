#!/bin/bash
export WSA_TUNING_ENV=1   # libwsa reads its tuning switches only when this is set (csrc/api.hip Tuning::from_env)
# K0 block-shape sweep (GPU box): tools/resample_sweep.sh <fs_in> "S:J[:C] S:J[:C] ..."   (outputs per block row : per lane and run [: runs per block])
export TMPDIR=/tmp
ROOT=$(pwd)
[ -f "$ROOT/tools/resample_probe.py" ] || { echo "run from the repository root (tools/resample_probe.py not found under $ROOT)" >&2; exit 1; }
fs=$1; shift
for sj in $1; do
  IFS=: read S J C <<< "$sj"
  rm -rf /tmp/p; (cd /tmp && WSA_RS_S=$S WSA_RS_J=$J WSA_RS_C=${C:-0} rocprofv3 --kernel-trace --stats -d /tmp/p -o r -- python3 $ROOT/tools/resample_probe.py $fs > /dev/null 2>&1)
  echo "fs $fs S $S J $J C ${C:-default}: $(python3 tools/rocprof_summary.py $(find /tmp/p -name '*.db' | head -1) | grep resample_kernel | awk '{print $(NF-1)}') us"
done
