#!/bin/bash
# round 6, session c: the whole GPU suite on the tree with the resident GCC kernel, the stage-0 look-back, prepared launches; full bench
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c; mkdir -p $O
export TMPDIR=/tmp
cd $R
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
echo "== bench (all legs)"; timeout 900 python bench.py --steps 20 --warmup 5 --full-json $O/bench_full.json 2>$O/bench.err | tail -1 > $O/bench_line.json; wc -c $O/bench_line.json
python - <<'PY'
import json
r=json.load(open('gpurun_out/r6c/bench_line.json'))
print('value',r['value'],'ms',r['ms_per_step'],'kernel',r['roofline']['kernel_ms'],r['roofline']['kernel_ms_repeats'],'frac',r['roofline']['frac'])
print('packed',r.get('packed_rows'),'psd',r.get('psd_output'))
for k,v in r['legs'].items(): print(k, v['value'], v['ms_per_step'], {kk:vv for kk,vv in v.get('roofline',{}).items() if kk in('frac','traffic','hbm_frac','f64_frac','kernel_ms')})
PY
echo "== bank times"; for cfg in "--bpo 3 --log2-samples 22 --chunk 1024" "--bpo 24 --log2-samples 20 --chunk 512"; do timeout 300 python tools/bench_octbank.py $cfg --iters 20 2>/dev/null | tail -1 | cut -c1-120; done
