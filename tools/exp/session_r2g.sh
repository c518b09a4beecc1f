#!/bin/bash
for rep in 1 2; do
for v in base keepq r1head; do
  if [ $v = base ]; then LP=""; else LP=$PWD/tools/variants/$v; fi
  echo -n "$v: "; LD_LIBRARY_PATH=$LP python bench.py --steps 50 --warmup 5 --cpu-budget 0 --no-legs 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['roofline']['kernel_ms_repeats'], r['psd_output']['kernel_ms'], r.get('parity',{}).get('epilogue_mismatched'))"
done
done
