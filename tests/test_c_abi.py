"""The C ABI as a C caller sees it (include/lfm_hip.h), without a GPU:
  * a C program compiled against the header reports sizeof / offsetof of the call structs; the ctypes mirrors (lfm_amd/hip.py) and the binding shown to
    a maintainer of the reference must describe exactly that layout -- a struct that is one pointer short makes the library read past it;
  * the C program links against the library and every declared entry point resolves;
  * per-call settings (ABI 4: lfm_dit_call.fold_ln / .gemm_select) are scoped to the call and the calling thread, the library-wide switches stay defaults.
"""
import ctypes as C
import os
import re
import subprocess
import threading

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "lfm_hip.h")

C_PROBE = r"""
#include <stdio.h>
#include <stddef.h>
#include <string.h>
#include "lfm_hip.h"
#define SZ(T) printf("sizeof " #T " %zu\n", sizeof(T))
#define OFF(T, f) printf("offsetof " #T " " #f " %zu\n", offsetof(T, f))
int main(void) {
  SZ(lfm_dit_shape); SZ(lfm_dit_weights); SZ(lfm_dit_call); SZ(lfm_vae_resnet); SZ(lfm_vae_weights); SZ(lfm_vae_enc_weights);
  OFF(lfm_dit_shape, depth); OFF(lfm_dit_shape, hidden); OFF(lfm_dit_shape, heads); OFF(lfm_dit_shape, patch); OFF(lfm_dit_shape, in_ch);
  OFF(lfm_dit_shape, res); OFF(lfm_dit_shape, mlp_hidden); OFF(lfm_dit_shape, label_rows);
  OFF(lfm_dit_weights, pos_embed); OFF(lfm_dit_weights, patch_w); OFF(lfm_dit_weights, patch_b); OFF(lfm_dit_weights, t_w0); OFF(lfm_dit_weights, t_b0);
  OFF(lfm_dit_weights, t_w2); OFF(lfm_dit_weights, t_b2); OFF(lfm_dit_weights, y_table); OFF(lfm_dit_weights, ada_w); OFF(lfm_dit_weights, ada_b);
  OFF(lfm_dit_weights, qkv_w); OFF(lfm_dit_weights, qkv_b); OFF(lfm_dit_weights, proj_w); OFF(lfm_dit_weights, proj_b); OFF(lfm_dit_weights, fc1_w);
  OFF(lfm_dit_weights, fc1_b); OFF(lfm_dit_weights, fc2_w); OFF(lfm_dit_weights, fc2_b); OFF(lfm_dit_weights, final_w); OFF(lfm_dit_weights, final_b);
  OFF(lfm_dit_weights, patch_w16);
  OFF(lfm_dit_call, batch); OFF(lfm_dit_call, x); OFF(lfm_dit_call, t); OFF(lfm_dit_call, t_len); OFF(lfm_dit_call, y); OFF(lfm_dit_call, cfg);
  OFF(lfm_dit_call, cfg_scale); OFF(lfm_dit_call, out); OFF(lfm_dit_call, axpy_base); OFF(lfm_dit_call, axpy_dt); OFF(lfm_dit_call, cond_table);
  OFF(lfm_dit_call, cond_step); OFF(lfm_dit_call, cond_offset); OFF(lfm_dit_call, cond_rows); OFF(lfm_dit_call, fold_ln); OFF(lfm_dit_call, gemm_select);
  /* a zero-initialised call means "library defaults" */
  if (lfm_abi_version() != LFM_ABI_VERSION) { printf("abi mismatch: library %d, header %d\n", lfm_abi_version(), LFM_ABI_VERSION); return 3; }  /* what a C caller does first */
  lfm_dit_call c; memset(&c, 0, sizeof c);
  int sel = -1, fold = -1;
  int rc = lfm_dit_call_settings(&c, &sel, &fold);
  printf("abi %d settings rc %d sel %d fold %d strerror %s\n", lfm_abi_version(), rc, sel, fold, lfm_strerror(LFM_ERR_SHAPE));
  c.gemm_select = LFM_CALL_GEMM_SELECT(5); c.fold_ln = LFM_CALL_OFF;
  rc = lfm_dit_call_settings(&c, &sel, &fold);
  printf("percall rc %d sel %d fold %d\n", rc, sel, fold);
  return 0;
}
"""


@pytest.fixture(scope="module")
def lib():
    from lfm_amd import hip

    return hip.lib()


@pytest.fixture(scope="module")
def c_report(tmp_path_factory, lib):
    """Compile the probe with the plain host C compiler against include/lfm_hip.h, link liblfm_hip.so, run it."""
    from lfm_amd import hip

    td = tmp_path_factory.mktemp("cabi")
    src = td / "probe.c"
    src.write_text(C_PROBE)
    exe = td / "probe"
    libdir = os.path.dirname(hip.LIB_PATH)
    r = subprocess.run(["cc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", libdir, "-llfm_hip",
                        f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True, env={**os.environ, "LD_LIBRARY_PATH": "/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", "")})
    assert out.returncode == 0, out.stderr
    rep = {"sizeof": {}, "offsetof": {}, "lines": out.stdout.splitlines()}
    for line in out.stdout.splitlines():
        w = line.split()
        if w[0] == "sizeof":
            rep["sizeof"][w[1]] = int(w[2])
        elif w[0] == "offsetof":
            rep["offsetof"].setdefault(w[1], {})[w[2]] = int(w[3])
    return rep


def _check_struct(rep, cname, ctype):
    assert C.sizeof(ctype) == rep["sizeof"][cname], (cname, C.sizeof(ctype), rep["sizeof"][cname])
    offs = rep["offsetof"][cname]
    assert [f[0] for f in ctype._fields_] == list(offs), (cname, "field order / names differ from the header")
    for f in ctype._fields_:
        assert getattr(ctype, f[0]).offset == offs[f[0]], (cname, f[0])


def test_ctypes_mirrors_match_the_header_as_compiled_by_cc(c_report):
    from lfm_amd import autoencoder, hip

    _check_struct(c_report, "lfm_dit_shape", hip.DitShape)
    _check_struct(c_report, "lfm_dit_weights", hip.DitWeights)
    _check_struct(c_report, "lfm_dit_call", hip.DitCall)
    for cname, pyname in (("lfm_vae_resnet", "VaeResnet"), ("lfm_vae_weights", "VaeWeights"), ("lfm_vae_enc_weights", "VaeEncWeights")):
        ct = getattr(autoencoder, pyname, None) or getattr(hip, pyname, None)
        if ct is not None:
            assert C.sizeof(ct) == c_report["sizeof"][cname], cname


def test_c_caller_sees_abi_4_and_call_scoped_settings(c_report):
    from lfm_amd import hip

    line = [x for x in c_report["lines"] if x.startswith("abi ")][0].split()
    assert int(line[1]) == hip.ABI_VERSION == 4
    assert line[3:9] == ["rc", "0", "sel", "0", "fold", "1"], line  # zero-initialised call = library defaults (automatic kernels, fold on)
    per = [x for x in c_report["lines"] if x.startswith("percall ")][0].split()
    assert per[1:] == ["rc", "0", "sel", "5", "fold", "0"], per


def _header_struct_fields(name):
    hdr = open(HDR).read()
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            fields.append(re.findall(r"[A-Za-z_][A-Za-z_0-9]*", decl)[-1])
    return fields


def test_integration_md_binding_matches_the_header():
    """The reference-side ctypes stub of INTEGRATION.md is executed (with ctypes only) and its three Structures are compared with the header."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", md, flags=re.S)
    stub = [b for b in blocks if "class lfm_dit_weights" in b]
    assert len(stub) == 1
    # keep the struct definitions only (the DiT.forward sketch below them needs torch and a GPU)
    code = stub[0].split("_lib.lfm_dit_workspace_bytes")[0]
    code = code.replace('_lib = C.CDLL("liblfm_hip.so")', "_lib = None").replace("import ctypes as C, torch", "import ctypes as C")
    ns = {}
    exec(code, ns)
    from lfm_amd import hip

    for cname, mirror in (("lfm_dit_shape", hip.DitShape), ("lfm_dit_weights", hip.DitWeights), ("lfm_dit_call", hip.DitCall)):
        doc = ns[cname]
        assert [f[0] for f in doc._fields_] == _header_struct_fields(cname), cname
        assert [(f[0], f[1]) for f in doc._fields_] == [(f[0], f[1]) for f in mirror._fields_], cname
        assert C.sizeof(doc) == C.sizeof(mirror)
    assert "21 device pointers" in stub[0]


def test_every_declared_entry_point_resolves(lib):
    hdr = re.sub(r"#ifdef LFM_MEASURE.*?#endif /\* LFM_MEASURE \*/", "", open(HDR).read(), flags=re.S)
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)  # declarations only
    names = sorted(set(re.findall(r"\b(lfm_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), n


def test_per_call_settings_are_scoped_to_the_call_and_the_thread(lib):
    """Thread A evaluates calls that carry their own kernel selection / fold switch, thread B flips the LIBRARY defaults meanwhile and evaluates plain
    calls: neither ever sees the other's choice (lfm_dit_call_settings runs the scope code of lfm_dit_forward without launching)."""
    from lfm_amd import hip

    errors = []
    stop = threading.Event()

    def settings(call):
        sel, fold = C.c_int(-1), C.c_int(-1)
        rc = lib.lfm_dit_call_settings(C.byref(call), C.byref(sel), C.byref(fold))
        return rc, sel.value, fold.value

    def a():
        call = hip.DitCall()
        call.gemm_select = hip.call_gemm_select(6 | (256 << 4))
        call.fold_ln = hip.CALL_OFF
        for _ in range(20000):
            if settings(call) != (0, 6 | (256 << 4), 0):
                errors.append(("per-call", settings(call)))
                break
        stop.set()

    def b():
        plain = hip.DitCall()
        i = 0
        while not stop.is_set():
            want = (1, 0)[i & 1]
            fold = i & 1
            assert lib.lfm_gemm_select(want) == 0 and lib.lfm_set_option(hip.OPT_FOLD_LN, fold) == 0
            got = settings(plain)
            if got != (0, want, fold):
                errors.append(("default", want, fold, got))
                break
            i += 1

    ta, tb = threading.Thread(target=a), threading.Thread(target=b)
    try:
        ta.start()
        tb.start()
        ta.join()
        stop.set()
        tb.join()
    finally:
        lib.lfm_gemm_select(0)
        lib.lfm_set_option(hip.OPT_FOLD_LN, 1)
    assert not errors, errors
    bad = hip.DitCall()
    bad.gemm_select = hip.call_gemm_select(3)  # not a kernel id
    assert settings(bad)[0] == -5
    bad = hip.DitCall()
    bad.fold_ln = 7
    assert settings(bad)[0] == -5
