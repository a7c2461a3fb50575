#!/bin/bash
export WSA_TUNING_ENV=1   # libwsa reads its tuning switches only when this is set (csrc/api.hip Tuning::from_env)
# tuning helper: FETCH_SIZE / WRITE_SIZE per kernel launch (KiB as counted; see tools/pmc_traffic.py for the gfx950 correction)
export TMPDIR=/tmp
ROOT=$(pwd)
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pmcq_$c; rm -rf $d
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o p -- python3 $ROOT/bench.py --in-flight 1 --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline > /dev/null 2>&1)
done
python3 tools/pmc_traffic.py $(find /tmp/pmcq_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find /tmp/pmcq_WRITE_SIZE -name '*counter_collection.csv' | head -1) | python3 -c "
import json,sys; d=json.load(sys.stdin)
for k,v in d['kernels'].items():
    if 'peaks' in k or 'gate' in k or 'tracker' in k: print(k, 'fetch(raw) %.0f MB write %.0f MB' % (v['FETCH_SIZE_KiB']*1024/1e6, v['WRITE_SIZE_KiB']*1024/1e6))"
