#!/bin/bash
export WSA_TUNING_ENV=1   # libwsa reads its tuning switches only when this is set (csrc/api.hip Tuning::from_env)
# kernel timeline of the pipelined bench under given environment switches (GPU box): tools/timeline_run.sh <tag> [VAR=value ...]
# writes gpurun_out/<tag>_timeline.txt (tools/timeline.py over a rocprofv3 --kernel-trace database of 40 steps)
tag=$1; shift
export TMPDIR=/tmp
ROOT=$(pwd); d=/tmp/tl_$tag; rm -rf $d
for kv in "$@"; do export "$kv"; done
(cd /tmp && rocprofv3 --kernel-trace -d $d -o r -- python3 $ROOT/bench.py --no-cpu-baseline --no-extra --steps 40 --warmup 3 --repeats 1 > /dev/null 2>&1)
python3 tools/timeline.py $(find $d -name "*.db" | head -1) 25 4 all > gpurun_out/${tag}_timeline.txt 2>&1
tail -3 gpurun_out/${tag}_timeline.txt
