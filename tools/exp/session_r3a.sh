#!/bin/bash
# round 3, session a: pipe rates + MFMA layout, parity of the MFMA passes (both D-row hypotheses), A/B against the vector-ALU build
export FRT_BENCH_SETS=4
tools/bin/f32pipes 2>&1 | tee gpurun_out/r03_f32pipes.txt
echo "== check base (DROW=0)"; tools/bin/stft_selftest check | grep -E "N= 1024|SELFTEST"
echo "== check drow1"; LD_LIBRARY_PATH=$PWD/tools/variants/drow1 tools/bin/stft_selftest check | grep -E "N= 1024|SELFTEST"
echo "== check nomf"; LD_LIBRARY_PATH=$PWD/tools/variants/nomf tools/bin/stft_selftest check | grep -E "SELFTEST"
for rep in 1 2; do
bash tools/exp/ab_variants.sh "base drow1 nomf" "1024 512 1 26 3 0 40" "1024 512 1 26 0 0 40"
done
echo "== no ring (register window for PSD)"
FRT_STFT_NO_RING=1 bash tools/exp/ab_variants.sh "base nomf" "1024 512 1 26 0 0 40"
echo "== ring for image"
FRT_STFT_RING_IMAGE=1 bash tools/exp/ab_variants.sh "base nomf" "1024 512 1 26 3 0 40"
