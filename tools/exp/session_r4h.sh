#!/bin/bash
# round 4, checkpoint session: whole GPU suite, bench line with every leg, PMC traffic of the headline refreshed, counters of the pk kernel
set -u
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -6
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 3 2>gpurun_out/bench.err | tail -1 > gpurun_out/r4h_bench.json; python - <<'PY'
import json
r=json.load(open('gpurun_out/r4h_bench.json'))
print('headline', r['value'], r['roofline']['frac'], 'traffic', r['roofline']['traffic'], 'parity', r.get('parity',{}).get('gate'))
for k,v in r['legs'].items():
    rf=v.get('roofline',{})
    print(k, '%.4g'%v['value'], v['unit'], 'ms %.4f'%v['ms_per_step'], 'frac', rf.get('frac'), 'cpu' if 'cpu_baseline' in v else '')
print('cpu_baseline', r['cpu_baseline']['value'], r['cpu_baseline'].get('all_cores',{}).get('value'))
PY
echo "== PMC traffic of the headline"
rm -rf gpurun_out/prof/pmc_bench
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/prof/pmc_bench/$n -o p --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --cpu-budget 0 --prewarm-ms 0 --no-legs > $R/gpurun_out/prof/pmc_$n.log 2>&1 ); echo "pmc $n rc=$?"
done
python tools/pmc_traffic.py gpurun_out/prof/pmc_bench && cp profiles/pmc_traffic.json gpurun_out/r4h_pmc_traffic.json
echo "== counters of the N = 16384 kernel"
bash tools/gpu_pmc.sh r04_n16384 0 3 16384 8192 32 20 > /dev/null 2>&1; python tools/prof_summary.py pmc gpurun_out/pmc_r04_n16384 stft_pk > gpurun_out/r04_stft16384_pmc.txt; cat gpurun_out/r04_stft16384_pmc.txt | cut -c1-110
