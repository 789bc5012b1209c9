#!/bin/bash
# 1024-token attention without the staging-offset spills: tests + before/after timing (LFM_HIP_LIBRARY selects the library build)
# the 'before' library (not tracked): git checkout ce23b04 -- lfm_amd/csrc/attention_kernel.h; bash tools/build_variant.sh att_before, restore the file
mkdir -p gpurun_out
{
python -m pytest tests/test_gpu_dit.py -q -k "attention or 1024_tokens" 2>&1 | tail -3
python -m pytest tests/test_gpu_cli.py -q -k "512" 2>&1 | tail -2
for i in 1 2; do
echo "--- before"; LFM_HIP_LIBRARY=tools/variants/att_before/liblfm_hip.so python tools/attn_1024_time.py
echo "--- after";  python tools/attn_1024_time.py
done
echo "--- hot shape, after"; python tools/attn_time.py
echo "--- hot shape, before"; LFM_HIP_LIBRARY=tools/variants/att_before/liblfm_hip.so python tools/attn_time.py
} > gpurun_out/r05_call16.log 2>&1
tail -40 gpurun_out/r05_call16.log
