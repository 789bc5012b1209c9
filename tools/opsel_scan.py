"""Every packed-fp32 instruction of the built library that carries an op_sel operand (a half of the packed operation takes its source from the OTHER
register of the pair), by kernel.  The form with the LOW half taking the HIGH register (op_sel:[..1..]) is the one that misbehaves under co-scheduling
(common.h: fma_v); tests/test_host_logic.py asserts that the shipped library holds none.   usage: python tools/opsel_scan.py [path to .so]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def scan(lib):
    """-> {(kernel, instruction, 'op_sel:[..]'): count} over all gfx950 code objects bundled in `lib`"""
    out = collections.Counter()
    with tempfile.TemporaryDirectory() as td:
        tmp = os.path.join(td, os.path.basename(lib))
        os.symlink(os.path.abspath(lib), tmp)
        subprocess.run([OBJDUMP, "--offloading", tmp], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=td)
        for f in sorted(os.listdir(td)):
            if "amdgcn" not in f:
                continue
            dis = subprocess.run([OBJDUMP, "-d", os.path.join(td, f)], check=True, capture_output=True, text=True).stdout
            cur = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
                if m:
                    cur = m.group(1)
                    continue
                if "v_pk_" in line and "_f32" in line:
                    s = re.search(r"op_sel:\[([0-9,]+)\]", line)
                    if s and "1" in s.group(1):
                        out[(cur, line.split()[0], s.group(1))] += 1
    return out


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "lfm_amd", "_lib", "liblfm_hip.so")
    res = scan(lib)
    for (k, ins, sel), n in sorted(res.items(), key=lambda x: -x[1]):
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        print(f"{n:5d}  {ins} op_sel:[{sel}]  {name[:150]}")
    print(f"{sum(res.values())} packed-fp32 instructions with op_sel in {lib}")
