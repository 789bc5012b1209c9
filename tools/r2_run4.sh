cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
timeout 900 python -m pytest tests/test_gpu_downstream.py tests/test_gpu_vae.py tests/test_gpu_unet.py tests/test_gpu_edm.py tests/test_gpu_cli.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r2d/tests.log
cat gpurun_out/r2d/tests.log
timeout 300 python bench.py --steps 4 --warmup 2 > gpurun_out/r2d/bench2.json 2> gpurun_out/r2d/bench2.err; cat gpurun_out/r2d/bench2.json; tail -3 gpurun_out/r2d/bench2.err
for c in 5 4 3; do timeout 300 python bench.py --config $c --steps 2 --warmup 1 > gpurun_out/r2d/bench$c.json 2> gpurun_out/r2d/bench$c.err; cat gpurun_out/r2d/bench$c.json; tail -3 gpurun_out/r2d/bench$c.err; done
