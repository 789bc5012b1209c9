#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_dit.py -q -x -k "latency or oracle or golden" > gpurun_out/pytest_call22.txt 2>&1; tail -3 gpurun_out/pytest_call22.txt
timeout 400 python tools/latency_probe.py > gpurun_out/latency_probe.txt 2>&1; grep "64x64" gpurun_out/latency_probe.txt
