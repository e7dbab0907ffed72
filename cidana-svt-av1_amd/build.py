"""Build the product in-tree:

  cidana-svt-av1_amd/libsvt_hip_dsp.so   HIP kernels + C ABI, gfx950 only

hipcc cross-compiles gfx950 without a GPU.  Nothing is installed outside the
repo; the .so is git-ignored but travels with the gpurun snapshot.  (The CPU
checkers are test infrastructure and are built by __graft_entry__.build().)
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "libsvt_hip_dsp.so")
SOURCES = ["csrc/svt_hip_dsp.hip"]
import glob as _glob
# every header the translation unit can include (kernel_cfl.h / kernel_ois.h were missing from a hand-kept list)
DEPS = (["csrc/svt_hip_dsp.hip", "../include/svt_hip_dsp.h"]
        + sorted(os.path.relpath(p, os.path.dirname(os.path.abspath(__file__)))
                 for p in _glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "*.h"))
                 + _glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "gen", "*.h"))))
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17", "-fwrapv",
               "-Wall", "-Wno-unused-function"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_product(force=False, verbose=True):
    gen = os.path.join(PKG, "csrc", "gen", "txfm1d_gen.h")
    tools = [os.path.join(PKG, "tools", f) for f in ("txfm_net.py", "gen_device.py")]
    if _stale(gen, tools):
        subprocess.check_call([sys.executable, os.path.join(PKG, "tools", "gen_device.py")])
    deps = [os.path.join(PKG, d) for d in DEPS]
    if force or _stale(LIB, deps):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc] + HIPCC_FLAGS + ["-o", LIB] + [os.path.join(PKG, s) for s in SOURCES]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=PKG)
    return LIB


def build_all(force=False, verbose=True):
    return build_product(force=force, verbose=verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
