export WSA_TUNING_ENV=1; export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/r6j; mkdir -p $O
for fs in 16000 48000; do
  python3 tools/app_defaults_probe.py $fs 5 > $O/probe_$fs.txt 2>&1; tail -2 $O/probe_$fs.txt
  d=/tmp/appd_$fs; rm -rf $d
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $d -o r -- python3 $ROOT/tools/app_defaults_probe.py $fs 5 > /dev/null 2>&1)
  python3 tools/rocprof_summary.py $(find $d -name '*.db' | head -1) > $O/kernel_stats_app_$fs.txt; cat $O/kernel_stats_app_$fs.txt | cut -c1-140
  for c in FETCH_SIZE WRITE_SIZE; do
    e=/tmp/appd_${fs}_$c; rm -rf $e
    (cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $e -o p -- python3 $ROOT/tools/app_defaults_probe.py $fs 2 > /dev/null 2>&1)
  done
  python3 tools/pmc_traffic.py $(find /tmp/appd_${fs}_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find /tmp/appd_${fs}_WRITE_SIZE -name '*counter_collection.csv' | head -1) 1024 $fs 13 10 > $O/pmc_traffic_app_$fs.json
  python3 -c "
import json,sys
d=json.load(open('$O/pmc_traffic_app_$fs.json'))
for k,v in d['kernels'].items():
    print('%-46s fetch %8.1f MB  write %8.1f MB' % (k[:46], 2*v['FETCH_SIZE_KiB']/1024*1.048576, v['WRITE_SIZE_KiB']/1024*1.048576))"
done
python3 tools/app_defaults_probe.py 16000 5 25 13 | tail -2
# what-if ms table (pipelined, tuning build)
T=$ROOT/webspeechanalyzer_amd/lib_tune
tools/ab.sh "base:WSA_LIB_DIR=$T" "no finalize:WSA_LIB_DIR=$T WSA_DBG=1" "no features:WSA_LIB_DIR=$T WSA_DBG=4" "no accumulate:WSA_LIB_DIR=$T WSA_DBG=2" "no accumulate no finalize:WSA_LIB_DIR=$T WSA_DBG=3" "no emission:WSA_LIB_DIR=$T WSA_DBG=1048576" > $O/whatif_ms.txt 2>&1; cat $O/whatif_ms.txt | cut -c1-80
