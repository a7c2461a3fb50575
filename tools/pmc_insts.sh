#!/bin/bash
export WSA_TUNING_ENV=1   # libwsa reads its tuning switches only when this is set (csrc/api.hip Tuning::from_env)
# SQ instruction counters per kernel launch + the clock under load (one PMC pass).  usage (GPU box): tools/pmc_insts.sh [out.json] [bench args]
# prints the JSON that profiles/rNN_pmc_insts.json holds (bench.py's issue_bound object reads the newest one)
export TMPDIR=/tmp
ROOT=$(pwd); out=/tmp/pmc_insts; rm -rf $out
dst=${1:-/dev/stdout}; shift
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $out -o p -- python3 $ROOT/bench.py --in-flight 1 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-extra "$@" > /dev/null 2>&1)
python3 $ROOT/tools/pmc_insts.py $(find $out -name '*counter_collection.csv' | head -1) > $dst
