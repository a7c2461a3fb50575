export WSA_TUNING_ENV=1
export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/prof_r06; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k 'persistent_front_end' > $O/t_fe.txt 2>&1; grep -n 'passed\|failed' $O/t_fe.txt
tools/pmc_insts.sh $O/pmc_insts.json; cp $O/pmc_insts.json profiles/r06_pmc_insts.json
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pmc_$c; rm -rf $d
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o p -- python3 $ROOT/bench.py --in-flight 1 --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-extra > /dev/null 2>&1)
done
python3 tools/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find /tmp/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1) 1024 16000 5 10 > $O/pmc_traffic.json; cp $O/pmc_traffic.json profiles/r06_pmc_traffic.json
tools/refresh_profiles.sh r06 > $O/refresh.log 2>&1
python3 bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2>> $O/bench.err
python3 bench.py --level 12 --no-cpu-baseline --no-extra > $O/bench_level12.json 2>> $O/bench.err
tools/pmc_util.sh > $O/pmc_util.txt 2>&1
tools/pmc_lds.sh > $O/pmc_lds.txt 2>&1
tools/refresh_shard.sh r06 >> $O/refresh.log 2>&1
true
true
python3 tools/bench_field.py final < $O/bench.json
python3 tools/bench_field.py steps20 < $O/bench_steps20.json
