"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every
symbol include/vslam_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "vslam_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vslam_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(capi):
    lib = ctypes.CDLL(capi.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 10
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_no_cpu_fallback_without_gpu(capi):
    """Without a device the product must fail loudly, never compute on the CPU."""
    if capi.device_count() > 0:
        return
    try:
        capi.Extractor(752, 480, 1500)
    except capi.VslamError as e:
        assert e.status == capi.ERR_NO_DEVICE
    else:
        raise AssertionError("extractor creation must fail without a GPU")


def test_product_does_not_reference_oracle():
    """Nothing under gtsam-vslam_amd/ may import, link or include the oracle."""
    pkg = os.path.join(ROOT, "gtsam-vslam_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in txt.lower().replace("no oracle", ""), os.path.join(dp, f)
