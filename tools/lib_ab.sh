#!/bin/bash
# A/B of two builds (webspeechanalyzer_amd/lib_old against lib): parity subset, wave-instructions per kernel, the headline at K = 20 / 100, the 12 500-clip shard.  (GPU box)
export WSA_TUNING_ENV=1; export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/lib_ab; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k 'not fuzz' > $O/t1.txt 2>&1; grep -h 'passed\|failed' $O/t1.txt
tools/pmc_whatif.sh "new:WSA_X=1" "old lib:WSA_LIB_DIR=$ROOT/webspeechanalyzer_amd/lib_old" > $O/whatif.txt 2>&1; cut -c1-420 $O/whatif.txt
tools/ab_k.sh "old:WSA_LIB_DIR=$ROOT/webspeechanalyzer_amd/lib_old" "new:WSA_X=1" > $O/ab.txt 2>&1; cut -c1-90 $O/ab.txt
for l in lib_old lib lib_old lib; do WSA_LIB_DIR=$ROOT/webspeechanalyzer_amd/$l python3 bench.py --clips 12500 --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline --no-extra 2>/dev/null | python3 tools/bench_field.py shard_$l | cut -c1-70; done
