#!/bin/bash
# round 6, call 12: evidence -- configs 3-6 (bench lines two lanes / one lane, kernel stats), latency mode (probe + the CLI's own --measure_time line + batch-1 VAE decode profile)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
bash tools/configs_evidence.sh r06cfg 2>&1 | tail -n 60
O=$R/gpurun_out/r06lat; mkdir -p $O
timeout 600 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/latency_probe.txt
# the number the reference's flag prints (test_flow_latent.py:223-246): 300 x run_sampling(1, ...) = 50-step Euler solve AND batch-1 VAE decode
timeout 900 python -m lfm_amd.test_flow_latent --model_type DiT-L/2 --num_classes 1 --label_dropout 0. --method euler --step_size 0.02 --measure_time --random_weights --generator device --image_size 256 --num_in_channels 4 --num_out_channels 4 2>&1 | grep -v amdgpu.ids | tail -n 3 | tee $O/cli_measure_time.txt
timeout 200 python tools/vae_b1_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/vae_b1.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/vaeprof -o v -- python $R/tools/vae_b1_probe.py 20 > /dev/null 2>&1
cp $(find $O/vaeprof -name "*kernel_stats.csv" | head -1) $O/vae_b1_kernel_stats.csv; rm -rf $O/vaeprof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b1prof -o v -- python $R/tools/fwd_probe_b1.py > /dev/null 2>&1
cp $(find $O/b1prof -name "*kernel_stats.csv" | head -1) $O/dit_b1_kernel_stats.csv; rm -rf $O/b1prof
head -n 14 $O/vae_b1_kernel_stats.csv | cut -c1-200; head -n 10 $O/dit_b1_kernel_stats.csv | cut -c1-200
