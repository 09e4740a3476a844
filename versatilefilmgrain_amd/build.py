"""Build libvfgs_hip.so (gfx950 only) in-tree with hipcc.

The .so is git-ignored but travels to the GPU box with the snapshot; nothing is JIT-built
at run time.  `python -m versatilefilmgrain_amd.build` or `__graft_entry__.build()`.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import tempfile
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libvfgs_hip.so"
SOURCES = [CSRC / "vfgs_kernel.hip", CSRC / "vfgs_fw_kernel.hip", CSRC / "vfgs_host.cpp", CSRC / "vfgs_fw_host.cpp", CSRC / "vfgs_cfg_host.cpp"]
# bench-only streaming kernels (the copy ceiling bench.py reports): a library of their own, outside the product and its ABI
DIAG_SRC = PKG.parent / "tools" / "bench_diag.hip"
DIAG_LIB = PKG.parent / "tools" / "bin" / "libvfgs_bench_diag.so"
FW_TABLES = CSRC / "fw_tables.bin"   # model constants, linked into the library as data (oracle/dump_fw_tables.c)
HEADERS = [CSRC / "vfgs_layout.h", CSRC / "vfgs_fw_layout.h", FW_TABLES,
           PKG.parent / "include" / "vfgs_hip.h", PKG.parent / "include" / "vfgs_hip_fw.h"]
ARCH = "gfx950"


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found: libvfgs_hip.so can only be built with ROCm")
    return exe


def up_to_date() -> bool:
    if not LIB.exists():
        return False
    t = LIB.stat().st_mtime
    return all(p.stat().st_mtime <= t for p in SOURCES + HEADERS)


def build_diag(force: bool = False) -> Path:
    if DIAG_LIB.exists() and DIAG_LIB.stat().st_mtime >= DIAG_SRC.stat().st_mtime and not force:
        return DIAG_LIB
    DIAG_LIB.parent.mkdir(parents=True, exist_ok=True)
    r = subprocess.run([hipcc(), "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-shared", "-o", str(DIAG_LIB), str(DIAG_SRC)],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
    return DIAG_LIB


def build(force: bool = False, verbose: bool = False) -> Path:
    # (the bench-only diagnostics library is built where it is used -- bench.py, tools/dev/clocks_under_load.sh call
    # build_diag() -- not here: the product build must not depend on tools/ being present, or on that file compiling)
    if up_to_date() and not force:
        return LIB
    # (no -D beyond the tables path and the depth of a kernel translation unit: vfgs_layout.h refuses any tuning / ablation knob
    # without VFGS_DEV_BUILD, and HIPCC_COMPILE_FLAGS_APPEND could smuggle one in)
    if "VFGS_" in os.environ.get("HIPCC_COMPILE_FLAGS_APPEND", ""):
        raise RuntimeError("HIPCC_COMPILE_FLAGS_APPEND carries a VFGS_ flag: the product build takes none")
    common = [hipcc(), "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-Wall", "-Wno-unused-function", f'-DVFGS_FW_TABLES_PATH="{FW_TABLES}"']
    if verbose:
        common.insert(1, "-Rpass-analysis=kernel-resource-usage")
    # One object per translation unit, compiled side by side (110 -> 60 s).  The grain kernels are TWO units out of one source -- one code
    # object per sample depth (vfgs_kernel.hip, launch_grain): a process works at one depth and never loads the other's kernels.
    units = [(CSRC / "vfgs_kernel.hip", "vfgs_kernel_d10.o", ["-DVFGS_KERNEL_DEPTH=10"]), (CSRC / "vfgs_kernel.hip", "vfgs_kernel_d8.o", ["-DVFGS_KERNEL_DEPTH=8"])]
    units += [(src, src.stem + ".o", []) for src in SOURCES[1:]]
    with tempfile.TemporaryDirectory(prefix="vfgs_build_") as tmp:
        def compile_unit(u):
            src, obj, flags = u
            return subprocess.run(common + flags + ["-c", str(src), "-o", str(Path(tmp) / obj)], cwd=tmp, capture_output=True, text=True)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 2)) as pool:
            results = list(pool.map(compile_unit, units))
        for (src, obj, _), r in zip(units, results):
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src.name} ({obj}):\n{r.stdout}\n{r.stderr}")
            if verbose:
                print(r.stderr)
        out = Path(tmp) / LIB.name
        r = subprocess.run([hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(out)] + [str(Path(tmp) / obj) for _, obj, _ in units],
                           cwd=tmp, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc (link) failed:\n{r.stdout}\n{r.stderr}")
        os.replace(out, LIB) if out.stat().st_dev == LIB.parent.stat().st_dev else shutil.copyfile(out, LIB)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
