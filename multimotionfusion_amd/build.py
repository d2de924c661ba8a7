"""Builds the gfx950 HIP library (libmmf_hip.so) in-tree with hipcc.

`python -m multimotionfusion_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles
for gfx950 without a GPU, so this runs in the CPU-only authoring container as well.
"""
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libmmf_hip.so")

# -ffp-contract=off: the per-pixel arithmetic must round exactly like the CPU oracle's
# (no fused multiply-add), see csrc/device_math.hpp.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared",
               "-Wall", "-Wextra", "-I/opt/rocm/include", "-ldl"]
SOURCES = ["mmf_hip.hip"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def _newest_source_mtime():
    newest = 0.0
    for root in (CSRC, os.path.join(REPO_DIR, "include")):
        for name in os.listdir(root):
            newest = max(newest, os.path.getmtime(os.path.join(root, name)))
    return newest


def build(force=False, verbose=True):
    """Compile csrc/*.hip into libmmf_hip.so unless it is already newer than every source."""
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= _newest_source_mtime():
        return LIB_PATH
    cmd = [_hipcc()] + HIPCC_FLAGS + ["-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print("[mmf build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
