"""CPU only: the host-only translation units of libsvt_hip_dsp (cidana-svt-av1_amd/build.py HOST_ONLY: the table builders, the
intra availability helper, the y4m header / multi-threaded frame reader, the error text) compiled with g++ under AddressSanitizer +
UndefinedBehaviorSanitizer and, separately, ThreadSanitizer, and driven by tests/c/host_sanitize_driver.cpp ("a TSAN/ASAN
host test of the shim"; the reference's hardening flags are CMakeLists.txt:33, its CI has a Valgrind job only).  GPU sanitizers are
not available on this pool and are never asked for: nothing here touches a device."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cidana-svt-av1_amd")
DRIVER = os.path.join(ROOT, "tests", "c", "host_sanitize_driver.cpp")
pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")


def host_only_sources():
    import importlib.util
    spec = importlib.util.spec_from_file_location("svt_build", os.path.join(PKG, "build.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    srcs = [os.path.join(PKG, s) for s in b.HOST_ONLY]
    assert len(srcs) >= 3 and all(os.path.exists(s) for s in srcs)
    for s in srcs:                                         # host-only means it: no HIP header may sneak in
        assert "hip_runtime" not in open(s).read() and "host_common.h" not in open(s).read(), s
    return srcs


def build(tmp, name, san_flags):
    exe = str(tmp / name)
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fno-omit-frame-pointer", "-Wall", "-Werror", "-DSVT_HIP_TEST_HOOKS", "-pthread",
           "-I", os.path.join(ROOT, "include")] + san_flags + [DRIVER] + host_only_sources() + ["-o", exe]
    pr = subprocess.run(cmd, capture_output=True, text=True)
    assert pr.returncode == 0, pr.stderr[-4000:]
    return exe


@pytest.fixture(scope="module")
def asan_exe(tmp_path_factory):
    return build(tmp_path_factory.mktemp("asan"), "driver_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])


@pytest.fixture(scope="module")
def tsan_exe(tmp_path_factory):
    return build(tmp_path_factory.mktemp("tsan"), "driver_tsan", ["-fsanitize=thread"])


def run(exe, args, **env):
    e = dict(os.environ, ASAN_OPTIONS="abort_on_error=0:detect_leaks=1:strict_string_checks=1", UBSAN_OPTIONS="print_stacktrace=1",
             TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1", **env)
    pr = subprocess.run([exe] + args, capture_output=True, text=True, timeout=600, env=e)
    assert pr.returncode == 0, (pr.stdout[-1500:], pr.stderr[-6000:])
    assert "ERROR: AddressSanitizer" not in pr.stderr and "runtime error" not in pr.stderr and "WARNING: ThreadSanitizer" not in pr.stderr, pr.stderr[-6000:]
    return pr.stdout


def test_asan_ubsan_random_y4m_headers(asan_exe):
    out = run(asan_exe, ["headers", "2000"])
    assert "2000 lines" in out


def test_asan_ubsan_big_frames_truncation_and_thread_start_failure(asan_exe, tmp_path):
    """>= 4 MiB frames through the four-thread pread path, with 0 .. 3 reader threads refused (the fallback branch reads the
    unstarted shares on the calling thread), a truncated last frame, a file ending inside "FRAME", small frames on the fread path"""
    out = run(asan_exe, ["frames", str(tmp_path)])
    assert "ok" in out


def test_asan_ubsan_availability_sweep_and_tables(asan_exe):
    out = run(asan_exe, ["avail"])
    assert "353528 tuples" in out
    assert "ok" in run(asan_exe, ["tables"])


def test_tsan_concurrent_readers_and_table_builders(tsan_exe, tmp_path):
    out = run(tsan_exe, ["threads", str(tmp_path)])
    assert "ok" in out


def test_tsan_frames_with_refused_threads(tsan_exe, tmp_path):
    assert "ok" in run(tsan_exe, ["frames", str(tmp_path)])
