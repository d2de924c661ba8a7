"""CPU: the C-ABI library loads and exports every symbol include/mmf_hip.h declares; the ctypes
table lists exactly those symbols; the product package refuses to run without its library."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "mmf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mmf_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from multimotionfusion_amd import _capi, build
    lib_path = build.build(verbose=False)  # hipcc cross-compiles gfx950 without a GPU
    lib = ctypes.CDLL(lib_path)
    syms = declared_symbols()
    assert len(syms) >= 35
    for name in syms:
        assert hasattr(lib, name), f"{name} declared in mmf_hip.h but not exported by libmmf_hip.so"
    assert sorted(_capi.SIGNATURES) == syms, set(_capi.SIGNATURES) ^ set(syms)
    assert lib.mmf_abi_version() == 3


def test_no_device_is_reported_not_emulated():
    """Without a HIP device the context cannot be created: there is no CPU fallback."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from multimotionfusion_amd import _capi
    lib = _capi.load()
    h = ctypes.c_void_p()
    rc = lib.mmf_ctx_create(0, None, 1, ctypes.byref(h))
    assert rc != 0 and b"HIP device" in lib.mmf_last_error()
    from multimotionfusion_amd.cudafuncs import Context
    with pytest.raises(RuntimeError):
        Context(0)


def test_xcd_block_mapping_is_a_permutation_with_one_contiguous_run_per_xcd():
    """csrc/device_math.hpp: xcd_block.  The workgroups of a launch go to the eight XCDs round-robin by their index; the surfel
    passes whose neighbouring blocks share cache lines give every XCD ONE contiguous run of the blocks.  Host logic, no device:
    the mapping must be a permutation of the blocks (every block worked on exactly once -- clean's ordered compaction and the
    rasterising passes rely on it), order preserving within an XCD, and contiguous per XCD."""
    from multimotionfusion_amd import _capi
    lib = _capi.load()
    import random
    rng = random.Random(7)
    for nb in [1, 7, 15, 16, 17, 23, 24, 63, 64, 65, 300, 961, 1200, 2161] + [rng.randrange(16, 5000) for _ in range(60)]:
        to = [lib.mmf_debug_xcd_block(b, nb) for b in range(nb)]
        assert sorted(to) == list(range(nb)), nb
        if nb < 16:
            assert to == list(range(nb))
            continue
        for x in range(8):
            mine = [to[b] for b in range(x, nb, 8)]  # the blocks of XCD x, in dispatch order
            assert mine == list(range(mine[0], mine[0] + len(mine))), (nb, x)
        starts = [to[x] for x in range(8)]
        assert starts == sorted(starts) and starts[0] == 0, (nb, starts)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under multimotionfusion_amd/ may reference it."""
    pkg = os.path.join(REPO, "multimotionfusion_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                for line in open(os.path.join(root, f), errors="replace"):
                    code = line.split("#")[0] if f.endswith(".py") else line
                    if f.endswith(".py"):
                        assert "from oracle" not in code and "import oracle" not in code, (f, line)
                    elif line.lstrip().startswith("#include"):
                        assert "oracle" not in line, (f, line)  # comments may cite the oracle, code may not use it


def test_cpp_shims_compile_and_link(tmp_path):
    """The C++ classes with the reference's names (RGBDOdometry, Model, MultiMotionFusion) build
    with plain g++ against the C ABI (no Eigen / OpenCV / GL needed)."""
    import subprocess
    from multimotionfusion_amd import build
    build.build(verbose=False)
    pkg = os.path.join(REPO, "multimotionfusion_amd")
    exe = tmp_path / "shim_link_check"
    cmd = ["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", os.path.join(REPO, "tests", "cpp", "shim_link_check.cpp"),
           "-D__HIP_PLATFORM_AMD__", "-isystem", "/opt/rocm/include", "-o", str(exe), f"-L{pkg}", "-lmmf_hip", "-lamdhip64", f"-Wl,-rpath,{pkg}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert out.strip() == "abi 3"


def test_compute_fusion_weight_matches_the_oracle_bit_for_bit():
    """Model::computeFusionWeight (Model.cpp:876-891, with rodrigues2's JacobiSVD, :1301-1342) is host arithmetic:
    the library's C++ restatement (csrc/pose_algebra.hpp) against the oracle's C one (oracle/mmf_oracle_pose.c) on
    random pose pairs, including the increments a tracked frame produces (mm / mrad) and large ones."""
    import numpy as np
    from multimotionfusion_amd import _capi, synth
    from oracle import oracle as orc
    lib = _capi.load()
    rng = np.random.default_rng(5)
    seen = set()
    for k in range(3000):
        P = synth.make_pose(rng.normal(size=3) * 0.5, rng.normal(size=3)).astype(np.float32)
        scale = 10.0 ** rng.uniform(-7, 0)
        L = (P.astype(np.float64) @ synth.make_pose(rng.normal(size=3) * scale, rng.normal(size=3) * scale)).astype(np.float32)
        if k % 7 == 0:
            L = P.copy()  # overridePose: pose == lastPose
        mult = float(np.float32(rng.choice([1.0, 3.0, 100.0])))
        out = np.zeros(1, np.float32)
        assert lib.mmf_compute_fusion_weight(_capi.fptr(np.ascontiguousarray(P.reshape(16))), _capi.fptr(np.ascontiguousarray(L.reshape(16))),
                                             mult, _capi.fptr(out)) == 0
        ref = np.float32(orc.compute_fusion_weight(P, L, mult))
        assert out[0].view(np.uint32) == ref.view(np.uint32), (k, out[0], ref)
        seen.add(float(out[0] / mult))
    assert min(seen) == 0.5 and max(seen) == 1.0 and len(seen) > 20  # both clamps and the range between were hit
