"""Driver of tools/ubench/pkfma_opsel.hip: the self-checking packed-FMA victim alone, then while a DiT-L/2 evaluation loop runs on a second stream
(the load under which the folded fc1 epilogue leaves its solo result).  Prints mismatch counts per (instruction form, half, lane quarter).
usage: python tools/pkfma_opsel_probe.py [rounds]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, ".")
from lfm_amd.models import DiT_models  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "_bin", "libpkfma_opsel.so"))
L.pk_victim_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
torch.manual_seed(0)
inp = torch.randn(65536, device=dev)
sink = torch.zeros(4, device=dev)
FORMS = ["op_sel:[0,1,0] (low half takes the pair's HIGH register)", "op_sel_hi:[1,0,1] (high half takes the LOW register)", "natural halves ((b, b) built by v_mov)"]


def init(m):
    for p in m.parameters():
        if not bool(p.any()):
            torch.nn.init.normal_(p, std=0.02)
    return m.to(dev).eval()


big = init(DiT_models["DiT-L/2"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0))
x = torch.randn(64, 4, 32, 32, device=dev)
t = torch.tensor(0.5, device=dev)
big(t, x)
torch.cuda.synchronize()


def run(label, co, trans, lds):
    cnt = torch.zeros(24, dtype=torch.int32, device=dev)
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    for _ in range(rounds):
        cur = torch.cuda.current_stream(dev)
        sa.wait_stream(cur)
        sb.wait_stream(cur)
        if co:
            with torch.cuda.stream(sb):
                for _ in range(3):
                    big(t, x)
        with torch.cuda.stream(sa):
            for _ in range(40):  # ~0.7 ms each: many launches = many "first rounds" under the other stream's kernels
                rc = L.pk_victim_launch(inp.data_ptr(), cnt.data_ptr(), sink.data_ptr(), 256, 400, trans, lds, C.c_void_p(sa.cuda_stream))
                assert rc == 0, rc
        torch.cuda.synchronize()
    c = cnt.view(3, 2, 4).tolist()
    total = rounds * 40 * 256 * 512 * 400 * 4
    print(f"{label}: {total:.3e} checks per form and half")
    for f in range(3):
        print(f"    {FORMS[f]}: low half, lanes 0-15 / 16-31 / 32-47 / 48-63: {c[f][0]}   high half: {c[f][1]}", flush=True)


for trans in (1, 0):
    for lds in (133120, 1024):
        run(f"alone              (GELU tail {trans}, LDS {lds})", False, trans, lds)
        run(f"under a DiT-L/2 loop (GELU tail {trans}, LDS {lds})", True, trans, lds)
