#!/bin/bash
# the GPU suite + build()/smoke() on the tree as it stands (the driver's round-end order), then the default bench line
export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/final; mkdir -p $O
python -m pytest tests -q -m gpu > $O/suite_full.txt 2>&1; grep -h 'passed\|failed' $O/suite_full.txt | tail -1 | tee $O/gpu_suite.txt
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('build + smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/final/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['cpu_baseline'].get('c_port', {}).get('back_end_only'))
PY
python bench.py --steps 20 --warmup 5 > $O/bench20.json 2>> $O/bench.err; python -c "
import json; d = json.loads(open('gpurun_out/final/bench20.json').read().strip().splitlines()[-1]); print('K=20', d['ms_per_step'])"
