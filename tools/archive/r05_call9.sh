#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_final.txt 2>&1; tail -6 gpurun_out/pytest_gpu_final.txt
bash tools/profile_round.sh r05final > gpurun_out/profile_round.log 2>&1; tail -30 gpurun_out/profile_round.log
