"""CPU: libyolo_hip.so loads and exports every symbol include/yolo_hip.h declares (no compute calls)."""

import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "yolo_hip.h")
LIB = os.path.join(ROOT, "yolo-v1_amd", "yolo", "libyolo_hip.so")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(yolo_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(LIB):
        import __graft_entry__ as g
        g.build()
    return ctypes.CDLL(LIB)


def test_every_declared_symbol_is_exported(built):
    names = _declared()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(built, n)]
    assert not missing, missing


def test_python_binding_covers_the_header():
    from yolo import _hip
    declared = set(_declared()) - {"yolo_hip_last_error"}
    assert declared == set(_hip._SIGS), (declared ^ set(_hip._SIGS))


def test_abi_version_and_argument_errors(built):
    built.yolo_hip_abi_version.restype = ctypes.c_int
    assert built.yolo_hip_abi_version() == 2
    # argument validation happens before any HIP call -> safe without a GPU
    built.yolo_decode.restype = ctypes.c_int
    rc = built.yolo_decode(None, 1, 7, 2, 20, ctypes.c_double(0.5), None, None, None)
    assert rc == -1
    built.yolo_hip_last_error.restype = ctypes.c_char_p
    assert b"yolo_decode" in built.yolo_hip_last_error()


def test_struct_layouts_match_the_header():
    """sizeof of the ctypes mirrors == what the C compiler lays out (checked by compiling a probe)."""
    import subprocess
    import tempfile
    from yolo import _hip
    probe = '#include <stdio.h>\n#include "yolo_hip.h"\nint main(){printf("%zu %zu %zu\\n", sizeof(yolo_igemm_desc), sizeof(yolo_wgrad_desc), sizeof(yolo_pool_desc));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "p.c")
        open(src, "w").write(probe)
        exe = os.path.join(d, "p")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    assert sizes == [ctypes.sizeof(_hip.IgemmDesc), ctypes.sizeof(_hip.WgradDesc), ctypes.sizeof(_hip.PoolDesc)]


def test_gpu_tensor_without_library_fails_loudly(monkeypatch):
    from yolo import _hip
    monkeypatch.setattr(_hip, "_LIB", None)
    monkeypatch.setattr(_hip, "LIB_PATH", "/nonexistent/libyolo_hip.so")
    with pytest.raises(RuntimeError, match="no CPU/eager fallback"):
        _hip.lib()


def test_binding_refuses_a_library_of_another_abi_version(monkeypatch, built):
    """descriptors are passed by pointer: a library built from another header must not be bound"""
    from yolo import _hip
    monkeypatch.setattr(_hip, "_LIB", None)
    monkeypatch.setattr(_hip, "ABI_VERSION", 99)
    with pytest.raises(RuntimeError, match="ABI version"):
        _hip.lib()


def test_inline_assembly_blocks_have_their_wait_states():
    """hipcc pads data hazards only between instructions it emitted itself.  The kernels' hand-written vector-memory instructions take their
    addresses from SGPRs that hipcc may reload from a spill lane (v_readlane) right in front of the asm block -- 5 wait states are needed and nobody
    inserts them (a GPU fault in round 3 until the block carried its own s_nop).  tools/check_asm_hazards.py disassembles the kernels and checks
    every asm block for this pattern; no GPU needed (hipcc cross-compiles)."""
    import shutil
    import subprocess
    import sys
    if not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    files = [os.path.join(ROOT, "yolo-v1_amd", "csrc", f) for f in ("igemm_persist.hip", "wgrad_wide.hip", "wgrad_pipe.hip")]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_hazards.py")] + files, capture_output=True, text=True)
    assert r.returncode == 0 and "inline-assembly hazards: 0" in r.stdout, r.stdout + r.stderr
