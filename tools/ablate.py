#!/usr/bin/env python3
"""Developer tool (not product, not test): build kernel variants with -DVFGS_ABLATE=<n>
/-DVFGS_... knobs and time them on the 4320p workload to see which resource bounds the
kernel.  Variants produce WRONG output by design; only the timing matters.
Usage: python tools/ablate.py "0" "1" "2,-DVFGS_WAVES=8" ...   (variant = ablate id[,extra flags])
"""
import ctypes as C
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402
import vfgs_testlib as T  # noqa: E402
from versatilefilmgrain_amd import hw  # noqa: E402

W, H = 7680, 4320


def build(variant, tmp):
    parts = variant.split(",")
    out = Path(tmp) / f"libvfgs_{abs(hash(variant))}.so"
    cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", f"-DVFGS_ABLATE={parts[0]}",
           *parts[1:], "-o", str(out), str(ROOT / "versatilefilmgrain_amd/csrc/vfgs_kernel.hip"),
           str(ROOT / "versatilefilmgrain_amd/csrc/vfgs_host.cpp")]
    subprocess.run(cmd, check=True, cwd=tmp)
    return out


def bench(path, batch, steps, trace="fgs_sei_10_420", rounds=3, oop=False):
    hw._lib = None
    lib = hw.load(path)
    h = hw.VfgsHip(device=0)
    T.replay(h, T.load_trace(trace))
    pool = max(2, 24 // batch)
    g = torch.Generator(device="cuda").manual_seed(1)
    Y = torch.randint(0, 1024, (pool, batch, H, W), dtype=torch.int16, device="cuda", generator=g)
    U = torch.randint(0, 1024, (pool, batch, H // 2, W // 2), dtype=torch.int16, device="cuda", generator=g)
    V = torch.randint(0, 1024, (pool, batch, H // 2, W // 2), dtype=torch.int16, device="cuda", generator=g)
    if oop:
        Yd, Ud, Vd = torch.empty_like(Y), torch.empty_like(U), torch.empty_like(V)
    st = torch.cuda.current_stream().cuda_stream
    best = 1e9
    for r in range(rounds + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps):
            s = i % pool
            if oop:
                h.add_grain_copy_dev(Y[s].data_ptr(), U[s].data_ptr(), V[s].data_ptr(), Yd[s].data_ptr(), Ud[s].data_ptr(),
                                     Vd[s].data_ptr(), W, H, 0, H, W, W // 2, batch, Y[s][0].numel() * 2, U[s][0].numel() * 2, st)
            else:
                h.add_grain_frames_dev(Y[s].data_ptr(), U[s].data_ptr(), V[s].data_ptr(), W, H, W, W // 2, batch,
                                       Y[s][0].numel() * 2, U[s][0].numel() * 2, st)
        e1.record()
        torch.cuda.synchronize()
        if r:
            best = min(best, e0.elapsed_time(e1) / steps / batch * 1e3)
    del Y, U, V
    return best


def main():
    variants = sys.argv[1:] or ["0"]
    with tempfile.TemporaryDirectory() as tmp:
        libs = [(v, build(v, tmp)) for v in variants]
        for v, p in libs:
            t1 = bench(p, 1, 100)
            t8 = bench(p, 8, 16)
            o1 = bench(p, 1, 100, oop=True)
            o8 = bench(p, 8, 16, oop=True)
            gb = 199.0656e6 / 1e3
            print(f"variant {v:34s} in-place b1 {t1:6.2f} us ({gb / t1:5.0f} GB/s) b8 {t8:6.2f} us ({gb / t8:5.0f} GB/s) | "
                  f"out-of-place b1 {o1:6.2f} us ({gb / o1:5.0f} GB/s) b8 {o8:6.2f} us ({gb / o8:5.0f} GB/s)", flush=True)


if __name__ == "__main__":
    main()
