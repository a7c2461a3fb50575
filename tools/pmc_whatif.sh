#!/bin/bash
export WSA_TUNING_ENV=1   # libwsa reads its tuning switches only when this is set (csrc/api.hip Tuning::from_env)
# Wave-instructions of the back-end kernels under what-if switches (one PMC pass of `bench.py --in-flight 1` each; TUNING=1 build for WSA_DBG bits 1 .. 64).
# usage (GPU box): tools/pmc_whatif.sh "LABEL:VAR=v VAR2=v" ...      e.g.  "base:WSA_X=1" "no features:WSA_DBG=4 WSA_LIB_DIR=$PWD/webspeechanalyzer_amd/lib_tune"
# BENCH_ARGS adds bench arguments (--level 13)
export TMPDIR=/tmp
ROOT=$(pwd)
for spec in "$@"; do
  label="${spec%%:*}"; vars="${spec#*:}"
  out=/tmp/pmc_whatif; rm -rf $out
  (cd /tmp && env $vars rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $out -o p -- python3 $ROOT/bench.py --in-flight 1 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-extra $BENCH_ARGS > /dev/null 2>&1)
  python3 $ROOT/tools/pmc_insts.py $(find $out -name '*counter_collection.csv' | head -1) | python3 -c "
import json,sys
d=json.load(sys.stdin); out=[]
for k,v in d['kernels'].items():
    n=sum(v[c] for c in ['SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS','SQ_INSTS_VMEM_RD','SQ_INSTS_VMEM_WR'])
    if n>1: out.append('%s %.1f (V %.1f S %.1f L %.1f) %.0f us' % (k.replace('wsa::','').split('<')[0].replace('tracker_kernel_',''), n, v['SQ_INSTS_VALU'], v['SQ_INSTS_SALU'], v['SQ_INSTS_LDS'], v['duration_us']))
print('$label |', ' | '.join(out))"
done
