"""CPU: the host layer of libvfgs_hip under AddressSanitizer + UndefinedBehaviorSanitizer and under ThreadSanitizer.

3,300 lines of C++ (vfgs_host.cpp, vfgs_fw_host.cpp, vfgs_cfg_host.cpp) hold stream rings, pinned staging buffers, worker threads
and a look-ahead that reads caller memory ahead of the line it was handed.  GPU sanitizers do not exist on the pool, so the host
sources are compiled with g++ and the sanitizers against tests/sanitize/hip_stub.cpp -- a model of the HIP runtime with deferred
stream execution, in which every copy and every "kernel launch" touches exactly the bytes the real one would -- and driven by
tests/sanitize/host_walks.cpp through the walks of tests/test_gpu_parity.py (lines in order, late edits, repeats, skips, setters
in the middle of a frame, the ring of stripes, promised frame heights over buffers of different pitches), the host stripe / frame
pipelines, replicas on several devices, batches, parts, lists, overlap regions, refusals.  The reference declares such a switch
and never wires it (/root/reference/CMakeLists.txt:25-29).  Values are the GPU suite's business; here the sanitizers check
addresses, lifetimes and threads, and the walks check that what a call hands back is what the (zero-grain) stub kernel made of it.
"""
import os
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "versatilefilmgrain_amd" / "csrc"
SAN = ROOT / "tests" / "sanitize"
HOST_SOURCES = [CSRC / "vfgs_host.cpp", CSRC / "vfgs_fw_host.cpp", CSRC / "vfgs_cfg_host.cpp"]
HIP_INCLUDE = Path(os.environ.get("ROCM_PATH", "/opt/rocm")) / "include"

pytestmark = pytest.mark.skipif(shutil.which("g++") is None or not (HIP_INCLUDE / "hip" / "hip_runtime_api.h").exists(),
                                reason="needs g++ and the HIP headers")


def build(tmp, name, flags):
    exe = tmp / name
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fno-omit-frame-pointer", *flags, "-D__HIP_PLATFORM_AMD__", f"-I{HIP_INCLUDE}",
           f'-DVFGS_FW_TABLES_PATH="{CSRC / "fw_tables.bin"}"', *map(str, HOST_SOURCES), str(SAN / "hip_stub.cpp"), str(SAN / "host_walks.cpp"),
           "-o", str(exe), "-pthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    return exe


def run(exe, env_extra, *walks):
    env = dict(os.environ, **env_extra)
    for k in ("VFGS_HIP_FRAME_HEIGHT", "VFGS_HIP_LINE_LOOKAHEAD", "LD_PRELOAD"):
        env.pop(k, None)
    r = subprocess.run([str(exe), *walks], capture_output=True, text=True, env=env, timeout=600)
    return r


def test_host_layer_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    exe = build(tmp_path, "walks_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])
    r = run(exe, {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert r.returncode == 0 and "ERROR" not in r.stderr and "runtime error" not in r.stderr, (r.stdout[-2000:], r.stderr[-6000:])
    assert r.stdout.count(" ok ") == 10 and "FAILED" not in r.stdout, r.stdout


def test_host_layer_under_thread_sanitizer(tmp_path):
    exe = build(tmp_path, "walks_tsan", ["-fsanitize=thread"])
    r = run(exe, {"TSAN_OPTIONS": "halt_on_error=0:second_deadlock_stack=1"})
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, (r.stdout[-2000:], r.stderr[-6000:])
    assert r.stdout.count(" ok ") == 10 and "FAILED" not in r.stdout, r.stdout


def test_the_harness_sees_a_kernel_that_touches_one_row_too_many(tmp_path):
    """The check of the checker: the same build with a stub kernel that walks one row past every plane must be reported."""
    bad = tmp_path / "hip_stub_bad.cpp"
    src = (SAN / "hip_stub.cpp").read_text()
    assert "r < row_first + pd.nrows; r++)" in src
    bad.write_text(src.replace("r < row_first + pd.nrows; r++)", "r < row_first + pd.nrows + 1; r++)")
                   .replace('"../../versatilefilmgrain_amd/csrc/', f'"{CSRC}/'))
    exe = tmp_path / "walks_bad"
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address", "-D__HIP_PLATFORM_AMD__", f"-I{HIP_INCLUDE}",
           f'-DVFGS_FW_TABLES_PATH="{CSRC / "fw_tables.bin"}"', *map(str, HOST_SOURCES), str(bad), str(SAN / "host_walks.cpp"), "-o", str(exe), "-pthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    r = run(exe, {}, "device_entries")
    assert r.returncode != 0 and "heap-buffer-overflow" in r.stderr, r.stderr[-3000:]
