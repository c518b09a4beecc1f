#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel stats + PMC traffic.  Outputs -> gpurun_out/.
# (Round 5: the shipped library reads no environment switches; A/B of kernel generations are tools/exp sessions on variant builds.)
set -u
R=$GRAFT_REPO_ROOT
TAG=${1:-r06}
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
echo "== selftest"; timeout 300 tools/bin/stft_selftest check | tail -2
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench (image, all legs)"; timeout 900 python bench.py --steps 20 --warmup 5 --full-json gpurun_out/${TAG}_bench_full.json 2>gpurun_out/bench_image.err | tail -1 | tee gpurun_out/${TAG}_bench_image.json | cut -c1-400; wc -c gpurun_out/${TAG}_bench_image.json
echo "== bench (psd)"; timeout 600 python bench.py --steps 50 --warmup 5 --kind psd --cpu-budget 0 --no-legs 2>&1 | tail -1 | tee gpurun_out/${TAG}_bench_psd.json | cut -c1-300
echo "== bench (packed rows, for comparison)"; timeout 600 python bench.py --steps 50 --warmup 5 --layout packed --cpu-budget 0 --no-legs 2>&1 | tail -1 | tee gpurun_out/${TAG}_bench_packed.json | cut -c1-300
echo "== rocprofv3 kernel stats of the bench command"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/stats -o bench -- python $R/bench.py --steps 20 --warmup 3 --cpu-budget 0 > $R/gpurun_out/prof/stats.log 2>&1 )
python tools/prof_summary.py stats gpurun_out/prof/stats/bench_results.db > gpurun_out/${TAG}_bench_kernel_stats.txt 2>/dev/null; head -12 gpurun_out/${TAG}_bench_kernel_stats.txt | cut -c1-200
echo "== rocprofv3 PMC passes of the bench command (separate passes, kernel-trace only)"
rm -rf gpurun_out/prof/pmc_bench
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/prof/pmc_bench/$n -o p --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --cpu-budget 0 --prewarm-ms 0 --no-legs > $R/gpurun_out/prof/pmc_$n.log 2>&1 ); echo "pmc $n rc=$?"
done
python tools/pmc_traffic.py gpurun_out/prof/pmc_bench && cp profiles/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
python tools/prof_summary.py pmc gpurun_out/prof/pmc_bench stft_kernel > gpurun_out/${TAG}_bench_pmc_traffic.txt
echo "== bench once more with the traffic figure of this session"
timeout 900 python bench.py --steps 50 --warmup 5 --cpu-budget 0 --no-legs 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(json.dumps(r['roofline']))"
echo "== large frames: rates, PMC summaries (N = 16384 and 4096)"
( export FRT_BENCH_SETS=4; for cfg in "16384 8192 32 20 0" "16384 8192 32 20 3" "8192 4096 32 21 0" "8192 4096 32 21 3" "4096 2048 16 22 0" "4096 1024 16 22 3" "2048 1024 8 24 0" "2048 512 8 24 3"; do tools/bin/stft_selftest bench $cfg 0 40 | tail -1; done ) > gpurun_out/${TAG}_stft_big_bench.txt 2>&1; cat gpurun_out/${TAG}_stft_big_bench.txt | cut -c1-150
bash tools/gpu_pmc.sh ${TAG}_n16384 0 3 16384 8192 32 20 > /dev/null 2>&1; python tools/prof_summary.py pmc gpurun_out/pmc_${TAG}_n16384 stft_pk > gpurun_out/${TAG}_stft16384_pmc.txt
echo "== kernel stats of the screen-space / widget / GCC kernels (their GPU tests under rocprofv3)"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/widgets -o w -- python -m pytest $R/tests/test_pipeline_gpu.py $R/tests/test_widgets_gpu.py $R/tests/test_gcc_gpu.py -q -m gpu -p no:cacheprovider > $R/gpurun_out/prof/widgets.log 2>&1 )
python tools/prof_summary.py stats gpurun_out/prof/widgets/w_results.db > gpurun_out/${TAG}_widgets_kernel_stats.txt 2>/dev/null; head -5 gpurun_out/${TAG}_widgets_kernel_stats.txt | cut -c1-160
echo "== GCC-PHAT against the batch size"
python tools/bench_gcc.py --pairs 1 4 16 32 40 48 56 64 100 160 256 1024 2>/dev/null | grep -v "^{" > gpurun_out/${TAG}_gcc_batch.txt; cat gpurun_out/${TAG}_gcc_batch.txt
echo "== banks and latency"
python tools/bench_firbank.py > gpurun_out/${TAG}_banks.json 2>/dev/null; cut -c1-250 gpurun_out/${TAG}_banks.json
python tools/stream_latency.py > gpurun_out/${TAG}_stream_latency.json 2>gpurun_out/stream_latency.err; head -c 300 gpurun_out/${TAG}_stream_latency.json
echo "== exact IIR bank: the launches of one call (27 bands, chunks of 1024; 216 bands)"
( cd /tmp && rm -rf /tmp/iirt && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/iirt -- python $R/tools/exp/iir_stage_times.py 8 3 22 > /dev/null 2>&1; python $R/tools/exp/iir_stage_times.py --parse /tmp/iirt ) > gpurun_out/${TAG}_iir_launches_call.txt 2>&1; tail -3 gpurun_out/${TAG}_iir_launches_call.txt
( cd /tmp && rm -rf /tmp/iirt24 && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/iirt24 -- python $R/tools/exp/iir_stage_times.py 8 24 20 512 > /dev/null 2>&1; python $R/tools/exp/iir_stage_times.py --parse /tmp/iirt24 ) > gpurun_out/${TAG}_iir_launches_call_bpo24.txt 2>&1; tail -3 gpurun_out/${TAG}_iir_launches_call_bpo24.txt
echo "== overlap-add bank: the launches of one call (27 bands; 216 bands)"
for cfg in "8 3 22" "8 24 20"; do
  tag=$(echo $cfg | tr ' ' '_'); OUT=/tmp/olat_$tag; rm -rf $OUT
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $R/tools/exp/fir_only.py $cfg 6 > /dev/null 2>&1 )
  python - $OUT <<'PY' > gpurun_out/${TAG}_ola_launches_$tag.txt
import csv, glob, sys
rows=[]
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-34:], int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])))
rows.sort()
ends=[i for i,r in enumerate(rows) if "energy_finish" in r[2]]
a,b=ends[-2]+1,ends[-1]+1
t0=rows[a][0]
for s,e,n,gx,gy,gz in rows[a:b]: print(f"{(s-t0)/1e3:8.1f} us  {n:34s} grid {gx:6d} x {gy:3d} x {gz:3d}  {(e-s)/1e3:7.1f} us")
print(f"{b-a} launches, span {(rows[b-1][1]-t0)/1e3:.1f} us")
PY
  tail -2 gpurun_out/${TAG}_ola_launches_$tag.txt
done
echo "== PMC traffic of every leg's kernels (tools/leg_traffic.py -> profiles/r06_leg_traffic.json)"
bash tools/gpu_leg_traffic.sh ${TAG}_leg_traffic 2>&1 | tail -14
echo "== bench once more: every leg with the traffic figures of this session"
timeout 900 python bench.py --steps 20 --warmup 5 --full-json gpurun_out/${TAG}_bench_full.json 2>>gpurun_out/bench_image.err | tail -1 > gpurun_out/${TAG}_bench_image.json; wc -c gpurun_out/${TAG}_bench_image.json
echo "== GCC phases (experiments build)"
for p in 1 256 1024; do FRT_GCC_PROFILE=1 FRT_LIB_VARIANT=bx timeout 300 python tools/exp/gcc_variant_bench.py --pairs $p --iters 2 2>&1 | grep "resident_kernel phases" | tail -1; done | tee gpurun_out/${TAG}_gcc_phases.txt
