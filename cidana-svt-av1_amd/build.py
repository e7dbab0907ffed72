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
SOURCES = ["csrc/svt_hip_core.hip", "csrc/svt_hip_txfm.hip", "csrc/svt_hip_pixel.hip", "csrc/svt_hip_intra.hip", "csrc/svt_hip_picture.hip", "csrc/svt_hip_frame.hip", "csrc/host_tables.cpp", "csrc/y4m_reader.cpp", "csrc/host_err.cpp"]
HOST_ONLY = [s for s in SOURCES if s.endswith(".cpp")]          # no HIP header, no device: also built under sanitizers (tests/test_host_sanitizers.py)
OBJ_DIR = os.path.join(PKG, "build_obj")            # git-ignored; objects do not travel, the linked .so does
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fwrapv", "-Wall", "-Wno-unused-function"]
HOST_FLAGS = ["-x", "c++", "-O2", "-fPIC", "-std=c++17", "-Wall"]        # host-only units: no device pass


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _deps_of(obj):
    """headers a translation unit really included, from hipcc's -MD depfile (every listed file that exists)"""
    d = obj[:-2] + ".d"
    if not os.path.exists(d):
        return None
    toks = open(d).read().replace("\\\n", " ").split()
    return [t for t in toks[1:] if os.path.exists(t)]


def build_product(force=False, verbose=True):
    """One hipcc job per translation unit, all in parallel (the 1-D transform headers alone take a minute to compile),
    then one link.  A unit is rebuilt when its source, any header of its depfile, or this file changed."""
    gen = os.path.join(PKG, "csrc", "gen", "txfm1d_gen.h")
    tools = [os.path.join(PKG, "tools", f) for f in ("txfm_net.py", "gen_device.py")]
    if _stale(gen, tools):
        subprocess.check_call([sys.executable, os.path.join(PKG, "tools", "gen_device.py")])
    # the quantiser look-up header of host_tables.cpp: generator or its data newer than the header -> regenerate (the depfile
    # rule then sees a changed header and rebuilds the unit)
    qgen = os.path.join(PKG, "csrc", "gen", "qlookup_gen.h")
    if _stale(qgen, [os.path.join(PKG, "tools", "gen_qlookup.py"), os.path.join(PKG, "qlookup_data.py")]):
        subprocess.check_call([sys.executable, os.path.join(PKG, "tools", "gen_qlookup.py")])
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    jobs, objs = [], []
    for src in SOURCES:
        sp = os.path.join(PKG, src)
        obj = os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + ".o")
        objs.append(obj)
        deps = _deps_of(obj)
        if force or deps is None or _stale(obj, deps + [sp, os.path.abspath(__file__)]):
            cmd = [hipcc] + (HIPCC_FLAGS if src.endswith(".hip") else HOST_FLAGS) + ["-MD", "-c", sp, "-o", obj]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            jobs.append((src, subprocess.Popen(cmd, cwd=PKG)))
    failed = [src for src, p in jobs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for " + ", ".join(failed))
    if jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=PKG)
    return LIB


def build_all(force=False, verbose=True):
    return build_product(force=force, verbose=verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
