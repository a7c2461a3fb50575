#!/bin/bash
export WSA_TUNING_ENV=1   # libwsa reads its tuning switches only when this is set (csrc/api.hip Tuning::from_env)
# tuning helper: per-kernel durations of back-to-back steps with parts of the tracker switched off (WSA_DBG bits:
# 1 no finalize, 2 no accumulate, 4 no features, 8 no straighten).   usage (GPU box): tools/dbg_sweep.sh "0 1 2 3"
export TMPDIR=/tmp
ROOT=$(pwd)
for d in ${1:-0 1 2 3}; do
  out=/tmp/dbgsweep_$d; rm -rf $out
  (cd /tmp && WSA_DBG=$d rocprofv3 --kernel-trace --stats -d $out -o r -- python3 $ROOT/bench.py --in-flight 1 --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline > /dev/null 2>&1)
  echo "== WSA_DBG=$d"
  python3 $ROOT/tools/rocprof_summary.py $(find $out -name '*.db' | head -1) | grep -v "^#"
done
