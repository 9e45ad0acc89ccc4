#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/soak3
timeout -k 10 1000 python3 tests/soak_gpu.py --cases 6000 --seed 4242 > gpurun_out/soak3/soak_6000_seed4242.txt 2>&1; tail -2 gpurun_out/soak3/soak_6000_seed4242.txt
timeout -k 10 600 python3 tests/soak_gpu.py --cases 0 --host-large 150 --seed 99 > gpurun_out/soak3/soak_host_large_150_seed99.txt 2>&1; tail -2 gpurun_out/soak3/soak_host_large_150_seed99.txt
