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
    # ... and nothing named pgorb_* leaves the library without a prototype (the header IS the boundary)
    import shutil
    import subprocess
    if shutil.which("nm"):
        out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
        exported = sorted({l.split()[-1] for l in out.splitlines() if " T pgorb_" in l})
        assert exported == syms, "exports without a prototype: %s" % sorted(set(exported) - set(syms))


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


def _comm_library_in_subprocess(preamble):
    import json
    import subprocess
    import sys
    code = preamble + r"""
import ctypes as C, json, os
L = C.CDLL(os.path.join(%r, "pilotguru_amd", "libpgorb.so"))          # (the bare library: importing the package imports torch)
buf = C.create_string_buffer(1024); pre = C.c_int(0)
rc = L.pgorb_comm_library(buf, len(buf), C.byref(pre))
mapped = sorted({ln.split()[-1] for ln in open('/proc/self/maps') if 'librccl.so' in ln.rsplit('/', 1)[-1]})
print(json.dumps({"rc": rc, "path": buf.value.decode(), "pre": pre.value, "mapped": mapped}))
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       universal_newlines=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_librccl_is_the_one_the_process_already_holds():
    """csrc/comm.hip asks the loader for an already-mapped librccl first (RTLD_NOLOAD): inside a torch process that is torch's
    bundled torch/lib/librccl.so, and the process ends up with exactly ONE RCCL (two builds in one process would hand the
    ncclUniqueId of one to the other -- a failure only a multi-rank run would show)."""
    import torch
    bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if not os.path.exists(bundled):
        pytest.skip("this torch build does not bundle librccl")
    out = _comm_library_in_subprocess("import torch\nimport torch.distributed\n"
                                      "import ctypes as _C\n_C.CDLL(%r, mode=_C.RTLD_GLOBAL)\n" % bundled)
    assert out["rc"] == 0, out
    assert os.path.realpath(out["path"]) == os.path.realpath(bundled), out
    assert out["pre"] == 1 and len(out["mapped"]) == 1, out


def test_librccl_loaded_by_name_when_the_process_has_none():
    out = _comm_library_in_subprocess("")
    if out["rc"] != 0:
        # no librccl on the loader's path in this environment: the error is a code and a message, not a crash (ADVICE r5:
        # dlerror() read twice was a NULL dereference exactly here)
        assert "librccl.so not found" in out["path"], out
        return
    assert out["pre"] == 0 and len(out["mapped"]) == 1 and "librccl" in out["path"], out
