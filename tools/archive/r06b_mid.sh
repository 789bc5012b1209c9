R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/f14
timeout 400 python tools/linear_shapes_probe.py 30 > gpurun_out/f14/linear_after.log 2>&1; grep "^M" gpurun_out/f14/linear_after.log | cut -c1-75
for c in 5 6; do timeout 300 python bench.py --config $c --steps 3 --warmup 1 --in-flight 1 --no-roofline 2>/dev/null | grep '^{' > gpurun_out/f14/config${c}_one_lane.json; python3 -c "
import json; d=json.load(open('gpurun_out/f14/config${c}_one_lane.json')); print('config', $c, 'one lane', round(d['value'],1), d.get('clock_mhz_under_mfma_load'))"; done
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_edm.py tests/test_gpu_vae.py -x -q 2>&1 | tail -3
