"""Build liblfm_hip.so for gfx950 with hipcc (in-tree; cross-compiles without a GPU).

    python -m lfm_amd._build [--force] [--verbose]
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "_lib")
LIB = os.path.join(LIBDIR, "liblfm_hip.so")
STAMP = os.path.join(LIBDIR, "liblfm_hip.stamp")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# No -ffast-math (round 4, profiles/r04_fastmath_ab.txt: parity unchanged to four digits, bench within noise with or without it): the fp32 timestep
# sinusoid, softmax, GELU and LayerNorm statistics compile under IEEE rules; the approximate instructions this path wants (v_exp_f32, v_rcp_f32) are
# written out where they are used.  LFM_FAST_MATH=1 restores the flag for an A/B (tools/fastmath_ab.sh).
# -fno-slp-vectorize (round 5): the SLP vectorizer re-packs scalar fp32 code into v_pk_*_f32 and, where the two halves come from different places, gives the
# instruction an op_sel operand -- the form that reads its operand as 0.0 in lanes 48-63 under co-scheduling (csrc/common.h: fma_v; profiles/
# r05_cosched_root_cause.txt).  Without the pass the library holds NO op_sel'd packed-fp32 instruction (tests/test_host_logic.py asserts it on the built
# code objects); the packed GELU / epilogue arithmetic is written with vector types and stays packed.  A/B on one box: 109.4 vs 109.4 img/s (one batch in
# flight), 114.4 vs 114.4 (two) -- profiles/r05_noslp_ab.txt.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-macro-redefined", "-fno-slp-vectorize"]
if os.environ.get("LFM_FAST_MATH") == "1":
    FLAGS += ["-ffast-math", "-fno-finite-math-only"]
if os.environ.get("LFM_MEASURE") == "1":  # measurement builds: the s_memtime-stamped GEMM epilogues and the attention phase / trace variants (tools/)
    FLAGS.append("-DLFM_MEASURE")


OBJDUMP = os.path.join(os.path.dirname(os.path.dirname(HIPCC)), "lib", "llvm", "bin", "llvm-objdump")


def opsel_scan(lib):
    """-> {(kernel, instruction, 'op_sel:[..]'): count}: every packed-fp32 instruction of the gfx950 code objects bundled in `lib` with an op_sel operand (a half
    of the packed operation takes its source from the OTHER register of the pair) -- the form that read its operand as 0.0 in lanes 48-63 under co-scheduling
    (csrc/common.h: fma_v; profiles/r05_cosched_root_cause.txt).  The build refuses a library that holds one; tests/test_host_logic.py asserts it again."""
    import collections
    import re
    import tempfile

    out = collections.Counter()
    with tempfile.TemporaryDirectory() as td:
        tmp = os.path.join(td, "lib.so")
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
                    sel = re.search(r"op_sel:\[([0-9,]+)\]", line)
                    if sel and "1" in sel.group(1):
                        out[(cur, line.split()[0], sel.group(1))] += 1
    return out


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), "rb").read())
    inc = os.path.join(os.path.dirname(HERE), "include", "lfm_hip.h")
    h.update(open(inc, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    """Compile every .hip under csrc/ into ONE shared library.  Returns the library path.  Safe under torchrun: the stamp check and the
    build run under an exclusive file lock (N ranks starting on stale sources compile once), and the library is linked to a temporary
    name and renamed into place (a concurrent dlopen never sees a half-written file)."""
    import fcntl

    os.makedirs(LIBDIR, exist_ok=True)
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read() == dig:
        return LIB
    try:
        lock = open(os.path.join(LIBDIR, ".build.lock"), "w")
    except PermissionError:  # a read-only install: nothing can be (re)built here anyway
        if os.path.exists(LIB):
            if not (os.path.exists(STAMP) and open(STAMP).read() == dig):
                import warnings

                warnings.warn(f"{LIB} was built from other sources than the ones next to it (stamp mismatch) and this install is read-only: using it as "
                              "shipped; hip.lib() still refuses a library of another ABI version")
            return LIB
        raise
    with lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read() == dig:
                return LIB  # another process built it while we waited
            return _build_locked(dig, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(dig, verbose):
    if not os.path.exists(HIPCC):
        raise RuntimeError(f"hipcc not found at {HIPCC}; cannot build liblfm_hip.so")
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
        cmd = [HIPCC, *FLAGS, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        if verbose and out:
            print(out.decode())
    tmp = LIB + f".tmp{os.getpid()}"
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout.decode()}")
    if os.path.exists(OBJDUMP) and os.environ.get("LFM_SKIP_OPSEL_SCAN") != "1":  # the co-scheduling guard, at BUILD time (round-5 advisor finding: the fix depends on code generation)
        found = opsel_scan(tmp)
        if found:
            os.unlink(tmp)
            raise RuntimeError("the built library holds packed-fp32 instructions with op_sel (csrc/common.h: fma_v / row_affine4):\n" + "\n".join(
                f"  {n} x {ins} op_sel:[{sel}] in {k}" for (k, ins, sel), n in sorted(found.items(), key=lambda x: -x[1])[:10]))
    os.replace(tmp, LIB)
    open(STAMP, "w").write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
