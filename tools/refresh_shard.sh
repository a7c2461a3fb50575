#!/bin/bash
export WSA_TUNING_ENV=1   # libwsa reads its tuning switches only when this is set (csrc/api.hip Tuning::from_env)
# The per-GPU shard of BASELINE config 4 (12 500 clips x 10 s) as one batch on one GPU: bench line + per-kernel times alone / pipelined (GPU box).
# usage: tools/refresh_shard.sh <tag>  -> gpurun_out/prof_<tag>/bench_shard.json, shard_kernel_stats_{in_flight_1,default}.txt
tag=${1:-x}
export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/prof_$tag; mkdir -p $O
python3 bench.py --clips 12500 --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline --no-extra > $O/bench_shard.json 2>> $O/bench.err
for mode in default in_flight_1; do
  d=/tmp/prof_shard_$mode; rm -rf $d
  extra=""; [ $mode = in_flight_1 ] && extra="--in-flight 1"
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $d -o r -- python3 $ROOT/bench.py --clips 12500 --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-extra $extra > /dev/null 2>&1)
  python3 tools/rocprof_summary.py $(find $d -name '*.db' | head -1) > $O/shard_kernel_stats_$mode.txt
done
ls -la $O | grep shard
