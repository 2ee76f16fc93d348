"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/rb_capi.h declares; calls that need a device fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "rb_capi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rb_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from rnabloom import _native as N
    lib = C.CDLL(N.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "librb_hip.so does not export %s" % n
    bound = {s[0] for s in N.SYMBOLS}
    assert set(names) == bound, set(names) ^ bound


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from rnabloom import _native as N
    p = N.GraphParams(1000, 1000, 1000, 2, 2, 2, 25, 0, 1, 0, 0, 0, 0)
    h = C.c_void_p()
    rc = N.lib.rb_graph_create(C.byref(p), C.byref(h))
    assert rc != 0 and b"hipGetDeviceCount" in N.lib.rb_last_error()
    b = C.c_void_p()
    assert N.lib.rb_batch_create_ascii(0, None, None, None, 0, 3, C.byref(b)) != 0


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "rna-bloom_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "rb_oracle" not in txt.replace("oracle/rb_oracle.c rbo_rng31", "") or f.endswith(".hpp"), f
                assert "import rbo" not in txt and "from oracle" not in txt, f


def test_expected_size_matches_reference_formula():
    from rnabloom import _native as N
    import math
    for n, fpr, h in [(16_500_000, 0.01, 2), (450_000_000, 0.01, 2), (1000, 0.05, 3), (10 ** 9, 0.01, 1)]:
        f32 = C.c_float(fpr).value
        r = -h / math.log(1 - math.exp(math.log(f32) / h))
        assert N.lib.rb_expected_size(n, fpr, h) == math.ceil(n * r)
