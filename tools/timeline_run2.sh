#!/bin/bash
export WSA_TUNING_ENV=1   # libwsa reads its tuning switches only when this is set (csrc/api.hip Tuning::from_env)
# kernel timeline of tools/timeline_probe.py (GPU box): tools/timeline_run2.sh <tag> <probe args...>
tag=$1; shift
export TMPDIR=/tmp
ROOT=$(pwd); d=/tmp/tl_$(basename $tag); rm -rf $d
(cd /tmp && rocprofv3 --kernel-trace -d $d -o r -- python3 $ROOT/tools/timeline_probe.py "$@" > $ROOT/gpurun_out/${tag}_probe.txt 2>&1)
python3 tools/timeline.py $(find $d -name "*.db" | head -1) 25 4 all > gpurun_out/${tag}_timeline.txt 2>&1
cat gpurun_out/${tag}_probe.txt | tail -1; tail -2 gpurun_out/${tag}_timeline.txt
