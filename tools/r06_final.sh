#!/bin/bash
# round 6's last GPU call: the suite and the fuzz on the final tree, then the standard profile set (tools/r06_refresh.sh)
export WSA_TUNING_ENV=1; export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/prof_r06; mkdir -p $O
python -m pytest tests -q -m gpu > $O/suite_full.txt 2>&1; grep -h 'passed\|failed' $O/suite_full.txt | tail -1 > $O/gpu_suite.txt; cat $O/gpu_suite.txt
python tools/fuzz_gpu.py 30000 31200 > $O/fuzz_a.txt 2>&1; grep -h seeds $O/fuzz_a.txt
WSA_FUZZ_LEVEL=13 python tools/fuzz_gpu.py 32000 32400 > $O/fuzz_b.txt 2>&1; grep -h seeds $O/fuzz_b.txt
WSA_FUZZ_LEVEL=10 python tools/fuzz_gpu.py 33000 33200 > $O/fuzz_c.txt 2>&1; grep -h seeds $O/fuzz_c.txt
WSA_FUZZ_LEVEL=3 python tools/fuzz_gpu.py 34000 34200 > $O/fuzz_d.txt 2>&1; grep -h seeds $O/fuzz_d.txt
python tools/fuzz_gpu.py 700 850 stream > $O/fuzz_e.txt 2>&1; grep -h seeds $O/fuzz_e.txt
bash tools/r06_refresh.sh > $O/refresh_all.log 2>&1
grep -h 'passed\|failed' $O/suite_full.txt | tail -1 > $O/gpu_suite.txt
tail -3 $O/refresh_all.log
