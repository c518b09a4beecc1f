#!/bin/bash
# round 5, session t: does the headline kernel's rate depend on where the output rows lie relative to the samples (the buffers' offsets in
# the memory's channel interleave)?  bench.py's runs of one session differ by 3 % (0.1145 / 0.1177 ms) with nothing changed but the allocator's
# placement.  stft_selftest bench 1024 512 1 26 3 (colour, split rows), ONE buffer set and four, FRT_BENCH_OUT_SHIFT swept.
set -u
R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp
B=tools/bin/stft_selftest
S="s/bench p32 N=1024 hop=512 C=1 T=2^26 F=131071 //; s/algorithmic.*of 8 TB.s)//"
FRT_BENCH_SHOW_PTRS=1 FRT_BENCH_SETS=4 $B bench 1024 512 1 26 3 0 5 32 1 | grep "^x "
for rep in 1 2; do
for sets in 4 1; do
  for sh in 0 4096 8192 16384 32768 65536 131072 262144 524288 1048576 2097152 1060864 3145728; do
    echo -n "sets=$sets out_shift=$sh: "; FRT_BENCH_SETS=$sets FRT_BENCH_OUT_SHIFT=$sh timeout 60 $B bench 1024 512 1 26 3 0 40 32 1 | tail -1 | sed "$S"
  done
done
done
