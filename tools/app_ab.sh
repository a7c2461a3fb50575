#!/bin/bash
# A/B of two builds (lib_old against lib) at the application's settings and on the headline: tools/app_ab.sh   (GPU box)
export WSA_TUNING_ENV=1; export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/app_ab; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py tests/test_config1.py -x -q -m gpu -k 'not fuzz' > $O/t1.txt 2>&1; grep -n 'passed\|failed' $O/t1.txt
for fs in 16000 48000; do
  for lv in 13 5; do
    echo "new: $(python3 tools/app_defaults_probe.py $fs 5 15 $lv 2>&1 | tail -2 | head -1 | cut -c1-200)"
    echo "old: $(WSA_LIB_DIR=$ROOT/webspeechanalyzer_amd/lib_old python3 tools/app_defaults_probe.py $fs 5 15 $lv 2>&1 | tail -2 | head -1 | cut -c1-200)"
  done
  d=/tmp/appd_$fs; rm -rf $d
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $d -o r -- python3 $ROOT/tools/app_defaults_probe.py $fs 5 > /dev/null 2>&1)
  python3 tools/rocprof_summary.py $(find $d -name '*.db' | head -1) > $O/kernel_stats_app_$fs.txt; cat $O/kernel_stats_app_$fs.txt | cut -c1-140
done
tools/ab_k.sh "old:WSA_LIB_DIR=$ROOT/webspeechanalyzer_amd/lib_old" "new:WSA_X=1" > $O/ab.txt 2>&1; cat $O/ab.txt | cut -c1-90
