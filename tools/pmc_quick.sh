#!/bin/bash
export WSA_TUNING_ENV=1   # libwsa reads its tuning switches only when this is set (csrc/api.hip Tuning::from_env)
# usage: pmc_one.sh  -> FETCH/WRITE per kernel
export TMPDIR=/tmp
ROOT=$(pwd)
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pmc_$c; rm -rf $d
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o p -- python3 $ROOT/bench.py --in-flight 1 --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-extra > /dev/null 2>&1)
done
python3 tools/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find /tmp/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1) 1024 16000 5 10 | python3 -c "
import json,sys
d=json.load(sys.stdin)
t=0
for k,v in d['kernels'].items():
    print('%-40s fetch %8.1f MB  write %8.1f MB' % (k, 2*v['FETCH_SIZE_KiB']/1024*1.048576, v['WRITE_SIZE_KiB']/1024*1.048576)); t+=v['hbm_bytes_per_launch']
print('total %.1f MB' % (t/1e6))"
