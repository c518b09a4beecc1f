#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_pitch_gpu.py -x -q 2>&1 | tail -2
python tools/bench_all.py 2>/dev/null | grep T1 | cut -c1-330
python bench.py --steps 50 --warmup 5 --cpu-budget 0 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.3e frac %.3f kernel_ms %.4f'%(d['value'], d['roofline']['frac'], d['roofline']['kernel_ms']), d.get('psd_output'), d.get('same_batch',{}).get('kernel_ms'), '%.3e'%d['octave_bands']['value'])"
