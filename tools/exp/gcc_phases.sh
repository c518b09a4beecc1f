#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in bx pf ltwpf; do for p in 1 1024; do echo "$v $p"; FRT_GCC_PROFILE=1 FRT_LIB_VARIANT=$v timeout 300 python tools/exp/gcc_variant_bench.py --pairs $p --iters 2 2>&1 | grep "resident_kernel phases" | tail -1; done; done
