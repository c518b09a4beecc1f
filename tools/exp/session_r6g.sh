#!/bin/bash
# round 6, session g: does the pre-warm's batch size change what the first timed region sees?  (headline only, three runs each)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6g; mkdir -p $O
cd $R
for b in 4 32 4 32 4 32; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-legs --cpu-budget 0 --prewarm-batch $b 2>/dev/null | tail -1 | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print('batch $b', 'ms_per_step', round(r['ms_per_step'],5), 'repeats', r['roofline']['kernel_ms_repeats'], 'host_fixed_us', round(r['roofline']['host_fixed_us'],1), 'value', round(r['value']/1e9,4))"
done | tee $O/prewarm_ab.txt
echo "== full bench"
timeout 900 python bench.py --steps 20 --warmup 5 --full-json $O/bench_full.json 2>$O/bench.err | tail -1 > $O/bench_line.json; wc -c $O/bench_line.json
python - <<'PY'
import json
r=json.load(open('gpurun_out/r6g/bench_line.json'))
print('value',r['value'],'ms',r['ms_per_step'],'kernel',r['roofline']['kernel_ms'],r['roofline']['kernel_ms_repeats'],'frac',r['roofline']['frac'], 'fixed', r['roofline']['host_fixed_us'])
for k,v in r['legs'].items(): print(k, v['value'], v['ms_per_step'], {kk:vv for kk,vv in v.get('roofline',{}).items() if kk in('frac','hbm_frac','f64_frac','kernel_ms')})
PY
