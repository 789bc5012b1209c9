import sys, torch
sys.path.insert(0, ".")
from lfm_amd import hip
dev = torch.device("cuda:0")
n = torch.cuda.get_device_properties(dev).multi_processor_count
ticks = torch.zeros(n, dtype=torch.int64, device=dev)
for it in (200, 20000, 20000, 5000):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); hip.check(hip.lib().lfm_clock_probe(n, it, hip.ptr(ticks), hip.stream_ptr(dev)), "p"); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(it, "ms", ms, "ticks max", int(ticks.max()), "min", int(ticks.min()), "ticks/us", float(ticks.max()) / (ms * 1e3), "mfma/simd", 2 * it * 64, "cycles/mfma if 2.0GHz", ms * 1e-3 * 2.0e9 / (2 * it * 64))
