"""The C ABI: every symbol declared in include/adelie_hip.h is exported by libadelie_hip.so, and the ctypes mirror
of adelie_hip_grpnet_args has the layout the C compiler gives the struct.  No GPU needed."""
import ctypes
import os
import re
import subprocess
import tempfile

from adelie_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "adelie_hip.h")


def _declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(adelie_hip_[a-z0-9_]+)\s*\(", txt)) - {"adelie_hip_poll_fn"})


def test_library_exports_every_declared_symbol():
    b = _abi.hip_backend()
    syms = _declared_symbols()
    assert len(syms) >= 28
    missing = [s for s in syms if not hasattr(b.lib, s)]
    assert not missing, missing
    assert sorted("adelie_hip_" + s for s in _abi.HIP_SYMBOLS) == syms
    hdr = open(os.path.join(ROOT, "include", "adelie_hip.h")).read()
    declared = int(re.search(r"#define\s+ADELIE_HIP_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert b.fn("abi_version")() == declared == _abi.ABI_VERSION


def test_ctypes_struct_matches_c_layout():
    fields = [f[0] for f in _abi.GrpnetArgs._fields_]
    prog = "#include <stdio.h>\n#include <stddef.h>\n#include \"adelie_hip.h\"\nint main(){\n"
    prog += 'printf("%zu\\n", sizeof(adelie_hip_grpnet_args));\n'
    for f in fields:
        prog += f'printf("%zu\\n", offsetof(adelie_hip_grpnet_args, {f}));\n'
    prog += "return 0;}\n"
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write(prog)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        out = [int(x) for x in subprocess.check_output([exe]).split()]
    assert out[0] == ctypes.sizeof(_abi.GrpnetArgs)
    for f, off in zip(fields, out[1:]):
        assert getattr(_abi.GrpnetArgs, f).offset == off, f


def test_no_cpu_fallback_without_device():
    """Creating a design without a visible GPU must fail loudly (the product never routes to a CPU path)."""
    import numpy as np
    import pytest

    b = _abi.hip_backend()
    if b.fn("device_count")() > 0:
        pytest.skip("a GPU is visible")
    import adelie_amd as ad

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ad.matrix.dense(np.asfortranarray(np.zeros((4, 2))))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "adelie_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".sh")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in txt.lower(), (fn,)
