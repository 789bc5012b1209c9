#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp; O=$R/gpurun_out/r06lat2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vae.py tests/test_gpu_downstream.py -q -m gpu -x 2>&1 | tail -n 4 | tee $O/pytest_vae.txt
timeout 200 python tools/vae_b1_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/vae_b1.txt
timeout 900 python -m lfm_amd.test_flow_latent --model_type DiT-L/2 --num_classes 1 --label_dropout 0. --method euler --step_size 0.02 --measure_time --random_weights --generator device --image_size 256 --num_in_channels 4 --num_out_channels 4 2>&1 | grep -v amdgpu.ids | tail -n 2 | tee $O/cli_measure_time.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/vaeprof -o v -- python $R/tools/vae_b1_probe.py 20 > /dev/null 2>&1
cp $(find $O/vaeprof -name "*kernel_stats.csv" | head -1) $O/vae_b1_kernel_stats.csv; rm -rf $O/vaeprof
head -n 8 $O/vae_b1_kernel_stats.csv | cut -c1-160
