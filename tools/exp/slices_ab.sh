# zero-state K-slicing at the low-rate stages: default (slices until one wave per CU) against no slicing at all
for rep in 1 2; do
echo -n "default: "; timeout 60 python tools/bench_octbank.py --chunk 1024 --iters 40 2>&1 | tail -1 | python -c "import sys,json; print('%.4f ms' % json.loads(sys.stdin.read())['ms'])"
echo -n "FRT_ZS_MAX_SLICES=1: "; FRT_ZS_MAX_SLICES=1 timeout 60 python tools/bench_octbank.py --chunk 1024 --iters 40 2>&1 | tail -1 | python -c "import sys,json; print('%.4f ms' % json.loads(sys.stdin.read())['ms'])"
done
