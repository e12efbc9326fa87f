"""CPU tests of the drop-in boundary: libpgorb.so loads, exports every symbol that
include/pgorb.h declares, and fails loudly (no CPU fallback) when there is no HIP device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "pgorb.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pgorb_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported():
    from pilotguru_amd import _lib
    L = _lib.lib()
    syms = _header_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(L, s), "libpgorb.so does not export %s" % s
    assert sorted(_lib.SYMBOLS) == syms, "pilotguru_amd/_lib.py SYMBOLS out of sync with include/pgorb.h"


def test_struct_layouts_match_header():
    from pilotguru_amd import KEYPOINT_DTYPE, _lib
    assert KEYPOINT_DTYPE.itemsize == 28                      # cv::KeyPoint (OpenCV 2.4)
    assert [KEYPOINT_DTYPE.fields[n][1] for n in ("x", "y", "size", "angle", "response", "octave", "class_id")] \
        == [0, 4, 8, 12, 16, 20, 24]
    assert C.sizeof(_lib.PgorbParams) == 40


def test_no_device_fails_loudly_without_fallback():
    from tests.conftest import has_gpu
    if has_gpu():
        pytest.skip("a GPU is present")
    import pilotguru_amd as pg
    from pilotguru_amd._lib import PGORB_E_NODEVICE, PgorbError
    with pytest.raises(PgorbError) as e:
        pg.ORBextractor(1000, 1.2, 8, 20, 7)
    assert e.value.code == PGORB_E_NODEVICE
    assert "no CPU path" in str(e.value)


def test_descriptor_distance_host_helper(oracle):
    import pilotguru_amd as pg
    rng = np.random.RandomState(2)
    a = rng.randint(0, 256, (40, 32)).astype(np.uint8)
    b = rng.randint(0, 256, (40, 32)).astype(np.uint8)
    for i in range(40):
        assert pg.ORBmatcher.DescriptorDistance(a[i], b[i]) == oracle.descriptor_distance(a[i], b[i])
    assert pg.ORBmatcher.DescriptorDistance(bytes(32), bytes([255] * 32)) == 256


def test_product_never_imports_oracle():
    """The product package must not reference oracle/ (only tests, smoke and bench's CPU leg may)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pilotguru_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", ".cc", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                code = "\n".join(l for l in txt.splitlines()
                                 if not l.strip().startswith(("//", "#", "*", "/*")) and "oracle/" not in l.split("//")[-1:][0]
                                 or "import" in l or "include" in l)
                assert not re.search(r"(from|import)\s+oracle", code), f
                assert not re.search(r'#include\s+"[^"]*oracle', code), f
                assert "liborb_oracle" not in txt, f
