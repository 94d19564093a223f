"""The C-ABI library loads on a CPU-only box and exports every symbol include/cpe.h declares; the ctypes
mirror has the same struct sizes; without a GPU the product fails loudly instead of falling back."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

from cheetah_pose_estimation_amd import _lib, abi, skeleton, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "cpe.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cpe_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    _lib.build_library()
    lib = _lib.load()
    names = _declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), n


def test_struct_sizes_match_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "cpe.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(cpe_skeleton), sizeof(cpe_camera), sizeof(cpe_priors), sizeof(cpe_options), sizeof(cpe_stats));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes == [C.sizeof(abi.Skeleton), C.sizeof(abi.Camera), C.sizeof(abi.Priors), C.sizeof(abi.Options), C.sizeof(abi.Stats)]


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    sk = skeleton.build_skeleton("phantom", 25)
    with pytest.raises(_lib.CpeError, match="no HIP device|no CPU fallback"):
        _lib.Handle(sk, synth.make_cameras(6))


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under the package may import or load it."""
    pkg = os.path.join(ROOT, "cheetah_pose_estimation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "cpe_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f
