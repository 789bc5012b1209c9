"""Batch-1 latency mode (--measure_time): DiT-L/2 velocity evaluation and the 50-step graphed solve for ONE latent."""
import sys, statistics, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
from lfm_amd.models import DiT_models
from lfm_amd.solvers import odeint, sample_torchdiffeq_euler_fused
dev = torch.device("cuda:0")
for name, flags, skinny in (("DiT-L/2", 0, 1), ("DiT-L/2", 0, 2), ("DiT-L/2", 0, 0), ("DiT-B/2", 0, 1), ("DiT-B/2", 0, 2), ("DiT-XL/2", 0, 1), ("DiT-XL/2", 0, 2)):
    hip.gemm_select(flags << 4)  # flag 512: split-K off
    hip.set_option(hip.OPT_SKINNY_GEMM, skinny); torch.manual_seed(0)  # 1 (default): 64x64 tiles; 2: all rows x 16 columns; 0: the split-K 128x128 path of rounds 2-4
    m = DiT_models[name](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0).to(dev).eval()
    for B in (1,):
        x = torch.randn(B, 4, 32, 32, device=dev); t = torch.tensor(0.5, device=dev)
        for _ in range(3): m(t, x)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): m(t, x)
        e.record(); torch.cuda.synchronize()
        eager = s.elapsed_time(e) / 20
        # what --measure_time runs (test_flow_latent.py:223-246 -> sample_from_model -> the graph-captured fixed-grid solver, per-grid conditioning tables)
        sample_torchdiffeq_euler_fused(m, x, 0.02, {})  # captures the graph
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            s.record(); sample_torchdiffeq_euler_fused(m, x, 0.02, {}); e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
        nparam = sum(p.numel() for p in m.parameters())
        step = statistics.median(ts) / 50
        tag = (" (no split-K)" if flags else "") + (" [64x64 tiles]" if skinny == 1 else " [all rows x 16 columns]" if skinny == 2 else " [split-K 128x128 path]")
        print(f"{name}{tag} batch {B}: eager forward {eager*1e3:7.1f} us | 50-step Euler solve {statistics.median(ts):7.2f} ms = {step*1e3:7.1f} us/step "
              f"| fp16 weights {nparam*2/1e6:6.1f} MB => {nparam*2/step/1e9:6.2f} TB/s if weight-streaming bound", flush=True)
