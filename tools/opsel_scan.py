"""Every packed-fp32 instruction of the built library that carries an op_sel operand (a half of the packed operation takes its source from the OTHER
register of the pair), by kernel.  The form with the LOW half taking the HIGH register (op_sel:[..1..]) is the one that misbehaves under co-scheduling
(common.h: fma_v); tests/test_host_logic.py asserts that the shipped library holds none.   usage: python tools/opsel_scan.py [path to .so]"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lfm_amd._build import OBJDUMP, opsel_scan as scan  # noqa: E402  (the build itself runs it; this is the command-line front end)


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "lfm_amd", "_lib", "liblfm_hip.so")
    res = scan(lib)
    for (k, ins, sel), n in sorted(res.items(), key=lambda x: -x[1]):
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        print(f"{n:5d}  {ins} op_sel:[{sel}]  {name[:150]}")
    print(f"{sum(res.values())} packed-fp32 instructions with op_sel in {lib}")
