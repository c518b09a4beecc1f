#!/bin/bash
# per-stage launch durations of the batched overlap-add bank (ola_wave_kernel / ola_batch_kernel)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "8 3 22" "8 24 20"; do
  for nw in 0 1; do
    rm -rf /tmp/olat
    if [ $nw = 1 ]; then export FRT_OLA_NO_WAVE=1; else unset FRT_OLA_NO_WAVE; fi
    timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/olat -- python $R/tools/exp/ola_stage_times.py $cfg > /dev/null 2>&1
    echo "== $cfg no_wave=$nw"
    python $R/tools/exp/ola_stage_times.py --parse /tmp/olat
  done
done
