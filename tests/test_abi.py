"""The C-ABI library loads without a GPU and exports exactly what include/seunet_hip.h declares."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "seunet_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(seunet_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_survey_entry_points():
    syms = declared_symbols()
    for must in ("seunet_version", "seunet_last_error", "seunet_conv3d_fwd", "seunet_conv3d_wgrad",
                 "seunet_gate_epilogue_fwd", "seunet_gate_epilogue_bwd", "seunet_stats_finalize", "seunet_maxpool_fwd",
                 "seunet_maxpool_bwd", "seunet_head_fwd", "seunet_head_bwd", "seunet_loss_sums", "seunet_loss_grad",
                 "seunet_net_workspace_bytes", "seunet_net_forward", "seunet_net_backward"):
        assert must in syms, must


def test_library_exports_every_declared_symbol_and_binding_covers_them():
    import seunet_amd
    from seunet_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()                                    # no GPU needed to dlopen + resolve
    syms = declared_symbols()
    assert sorted(_lib.PROTOTYPES.keys()) == syms        # binding table == header
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (seunet_[a-z0-9_]+)", out))
    assert set(syms) <= exported, sorted(set(syms) - exported)
    assert lib.seunet_version() >= 100
    assert isinstance(_lib.last_error(), str)


def test_pure_host_entry_points_without_gpu():
    import ctypes as C
    import seunet_amd
    from seunet_amd import _lib
    from seunet_amd.SE_UNet import make_desc, registry
    import seunet_oracle as orc
    lib = _lib.load()
    for inch, wm in ((2, 1), (1, 1), (2, 2)):
        desc = make_desc(1, inch, 1, 64, 64, 64, wm, _lib.BF16, 0, 0.01)
        assert registry(desc) == orc.parameter_registry(inch, 1, wm)
    d = make_desc(4, 2, 1, 128, 128, 128, 1, _lib.BF16, 0, 0.01)
    assert 8e9 < lib.seunet_net_workspace_bytes(C.byref(d)) < 40e9
    bad = make_desc(1, 2, 1, 100, 128, 128, 1, _lib.BF16, 0, 0.01)      # not a multiple of 8
    assert lib.seunet_net_workspace_bytes(C.byref(bad)) == 0
    assert "multiples of 8" in _lib.last_error()
    three = make_desc(1, 2, 3, 64, 64, 64, 1, _lib.BF16, 0, 0.01)       # n_classes = 3: the general head path (side maps kept)
    one = make_desc(1, 2, 1, 64, 64, 64, 1, _lib.BF16, 0, 0.01)
    assert lib.seunet_net_workspace_bytes(C.byref(three)) > lib.seunet_net_workspace_bytes(C.byref(one)) > 0
    assert registry(three) == orc.parameter_registry(2, 3, 1)
    bad = make_desc(1, 2, 9, 64, 64, 64, 1, _lib.BF16, 0, 0.01)
    assert lib.seunet_net_workspace_bytes(C.byref(bad)) == 0 and "n_classes" in _lib.last_error()
    assert lib.seunet_conv_wpack_bytes(_lib.BF16, 27, 64, 32) == 27 * 4 * 32 * 32
    assert lib.seunet_conv3d_wgrad_workspace_bytes(27, 64, 32) > 0


def test_header_is_plain_c_and_links_from_a_c_program(tmp_path):
    """include/seunet_hip.h is the boundary a non-Python binding (cgo / JNI / a C host) talks to: it must compile as strict
    C99 and the library must link from a C program with nothing but the header (no torch, no HIP headers).  The program
    calls host-only entry points, so this runs without a GPU."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "se-unet-airseg_amd")
    if not os.path.exists(os.path.join(lib_dir, "libseunet_hip.so")) or shutil.which("gcc") is None:
        pytest.skip("needs the built library and gcc")
    src = tmp_path / "abi_c.c"
    src.write_text(
        '#include "seunet_hip.h"\n#include <stdio.h>\n'
        "int main(void) {\n"
        "  seunet_net_desc d = {1, 2, 1, 64, 64, 64, 1, SEUNET_BF16, SEUNET_CONV_MFMA, 0.01f, 1e-5f};\n"
        '  printf("%d %d %d\\n", seunet_version(), seunet_net_param_count(&d), seunet_net_workspace_bytes(&d) > 0);\n'
        "  d.n_classes = 9;\n"
        '  if (seunet_net_workspace_bytes(&d) != 0) return 2;\n'
        '  printf("%s\\n", seunet_last_error());\n'
        "  return 0;\n}\n")
    exe = tmp_path / "abi_c"
    cc = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), str(src),
                         "-o", str(exe), "-L", lib_dir, "-lseunet_hip", "-Wl,-rpath," + lib_dir], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True)
    assert run.returncode == 0, (run.stdout, run.stderr)
    lines = run.stdout.strip().splitlines()
    assert lines[0].split()[1:] == ["117", "1"] and "n_classes" in lines[1]
