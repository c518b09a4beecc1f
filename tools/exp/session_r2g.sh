#!/bin/bash
for rep in 1 2; do
for v in base prev r1head; do
  if [ $v = base ]; then LP=""; else LP=$PWD/tools/variants/$v; fi
  echo -n "$v: "; LD_LIBRARY_PATH=$LP python bench.py --steps 50 --warmup 5 --cpu-budget 0 --no-legs 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['roofline']['kernel_ms_repeats'], r['psd_output']['kernel_ms'], r.get('parity',{}).get('epilogue_mismatched'))"
done
done
FRT_BENCH_SETS=4 tools/bin/stft_selftest bench 1024 384 1 26 0 0 40 | tail -1 | cut -c65-90
