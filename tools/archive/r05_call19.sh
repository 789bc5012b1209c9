#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_dit.py -q -x -k "latency" > gpurun_out/pytest_call19.txt 2>&1; tail -4 gpurun_out/pytest_call19.txt
timeout 200 python tools/latency_gemm_probe.py > gpurun_out/latency_gemm_probe.txt 2>&1; grep "kernel 7" gpurun_out/latency_gemm_probe.txt
timeout 400 python tools/latency_probe.py > gpurun_out/latency_probe.txt 2>&1; grep "64x64" gpurun_out/latency_probe.txt
