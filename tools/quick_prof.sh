#!/bin/bash
export WSA_TUNING_ENV=1   # libwsa reads its tuning switches only when this is set (csrc/api.hip Tuning::from_env)
# per-kernel times of back-to-back steps (kernels alone) and of the default pipelined run.  usage (GPU box): tools/quick_prof.sh <tag> [bench args]
tag=${1:-x}; shift
export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/qp_$tag; mkdir -p $O
for mode in in_flight_1 default; do
  d=/tmp/qp_$mode; rm -rf $d
  extra=""; [ $mode = in_flight_1 ] && extra="--in-flight 1"
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $d -o r -- python3 $ROOT/bench.py --no-cpu-baseline --no-extra $extra "$@" > $O/bench_$mode.json 2> $O/bench_$mode.err)
  python3 tools/rocprof_summary.py $(find $d -name '*.db' | head -1) > $O/kernel_stats_$mode.txt
done
cat $O/kernel_stats_in_flight_1.txt
python3 tools/bench_field.py default < $O/bench_default.json 2>/dev/null || tail -c 600 $O/bench_default.json
