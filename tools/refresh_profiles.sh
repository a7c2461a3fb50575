#!/bin/bash
export WSA_TUNING_ENV=1   # libwsa reads its tuning switches only when this is set (csrc/api.hip Tuning::from_env)
# Regenerates the files profiles/README.md lists (GPU box).  usage: tools/refresh_profiles.sh <tag>   -> gpurun_out/prof_<tag>/
tag=${1:-x}
export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/prof_$tag; mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --level 13 --no-cpu-baseline > $O/bench_level13.json 2>> $O/bench.err
python3 bench_stream.py 2>> $O/bench.err | tail -1 > $O/stream_bench.json
for mode in default in_flight_1 stream; do
  d=/tmp/prof_$mode; rm -rf $d
  case $mode in
    default) (cd /tmp && rocprofv3 --kernel-trace --stats -d $d -o r -- python3 $ROOT/bench.py --no-cpu-baseline --no-extra > /dev/null 2>&1) ;;
    in_flight_1) (cd /tmp && rocprofv3 --kernel-trace --stats -d $d -o r -- python3 $ROOT/bench.py --no-cpu-baseline --no-extra --in-flight 1 > /dev/null 2>&1) ;;
    stream) (cd /tmp && rocprofv3 --kernel-trace --stats -d $d -o r -- python3 $ROOT/bench_stream.py --steps 2000 > /dev/null 2>&1) ;;
  esac
  out=kernel_stats_$mode.txt; [ $mode = stream ] && out=stream_kernel_stats.txt
  python3 tools/rocprof_summary.py $(find $d -name '*.db' | head -1) > $O/$out
done
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pmc_$c; rm -rf $d
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o p -- python3 $ROOT/bench.py --in-flight 1 --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-extra > /dev/null 2>&1)
done
python3 tools/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find /tmp/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1) 1024 16000 5 10 > $O/pmc_traffic.json
tools/pmc_insts.sh $O/pmc_insts.json
ls -la $O
