"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol include/*.h declares,
and the product path fails loudly (never falls back to the CPU) when no HIP device is present."""
import ctypes as C
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    syms = []
    for hdr in glob.glob(os.path.join(ROOT, "include", "*.h")):
        txt = open(hdr).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        for m in re.finditer(r"\b(flvis_[a-z0-9_]+)\s*\(", txt):
            syms.append(m.group(1))
    return sorted(set(syms))


def test_library_exports_every_declared_symbol():
    import flvis_amd
    lib = flvis_amd.load_library()
    syms = declared_symbols()
    assert len(syms) >= 10
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, "declared in include/*.h but not exported: %s" % missing
    assert b"gfx950" in lib.flvis_version()


def test_no_cpu_fallback_without_gpu():
    import torch
    import flvis_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = flvis_amd.load_library()
    h = C.c_void_p(0)
    rc = lib.flvis_hip_create(0, C.c_void_p(0), C.byref(h))
    assert rc == flvis_amd.FLVIS_ERR_NO_DEVICE and not h.value
    with pytest.raises(flvis_amd.FlvisError):
        flvis_amd.Context(0)


def test_product_does_not_reference_oracle():
    bad = []
    for path in glob.glob(os.path.join(ROOT, "flvis_amd", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
            txt = open(path, errors="replace").read()
            if re.search(r"oracle/|libflvis_ref|ref_api\.h|ref_math\.hpp", txt):
                bad.append(path)
    assert not bad, "product sources must not use the oracle: %s" % bad
