#!/bin/bash
# round 3, session r: GCC-PHAT with the compile-time 6000-point plan and the register-resident cross spectrum
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gcc_gpu.py -m gpu -q > gpurun_out/r3r_tests.log 2>&1
tail -15 gpurun_out/r3r_tests.log
timeout 600 python tools/bench_gcc.py > gpurun_out/r3r_gcc.txt 2>&1
grep -v "^{" gpurun_out/r3r_gcc.txt | tail -12
