#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c10; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_cosched.py tests/test_gpu_unet.py tests/test_gpu_cli.py -q -m gpu -x 2>&1 | tail -n 25 | tee gpurun_out/c10/pytest.txt
