#!/usr/bin/env python3
"""Developer tool (not product, not test): build kernel variants with -DVFGS_ABLATE=<n> and other
-DVFGS_* knobs and time them on the 4320p workload, INTERLEAVED over several rounds in one
process (medians; run-to-run drift on one box is +-2 us).  Variants with VFGS_ABLATE != 0 produce
WRONG output by design; only the timing matters.
Usage: python tools/ablate.py "0" "1" "0,-DVFGS_WAVES=16,-DVFGS_WG_PER_CU=1" ...   (variant = ablate id[,extra flags])
       env ROUNDS (default 5), MODES (default "ip1,ip8": ip = in place, op = out of place, 1/8 = frames per launch)
"""
import os
os.environ.setdefault("VFGS_ALLOW_DEV_BUILD", "1")   # this tool builds and loads developer variants
import os
import statistics
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402
import vfgs_testlib as T  # noqa: E402
from versatilefilmgrain_amd import hw  # noqa: E402

W, H = 7680, 4320
GB = 199.0656e6 / 1e3


def build(variant, tmp):
    parts = variant.split(",")
    out = Path(tmp) / f"libvfgs_{abs(hash(variant))}.so"
    src = ROOT / "versatilefilmgrain_amd/csrc"
    flags = []
    for x in parts[1:]:
        if x.startswith("src="):           # alternative source directory (e.g. an older kernel kept for A/B runs)
            src = ROOT / x[4:]
        else:
            flags.append(x)
    cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-w", "-DVFGS_DEV_BUILD", f"-DVFGS_ABLATE={parts[0]}",
           *flags, f"-I{ROOT / 'versatilefilmgrain_amd/csrc'}", f'-DVFGS_FW_TABLES_PATH="{ROOT / "versatilefilmgrain_amd/csrc/fw_tables.bin"}"',
           "-o", str(out), str(src / "vfgs_kernel.hip"), str(src / "vfgs_host.cpp")]
    cmd += [str(src / f) for f in ("vfgs_fw_kernel.hip", "vfgs_fw_host.cpp", "vfgs_cfg_host.cpp") if (src / f).exists()]
    subprocess.run(cmd, check=True, cwd=tmp)
    return out


def main():
    variants = sys.argv[1:] or ["0"]
    rounds = int(os.environ.get("ROUNDS", "5"))
    modes = os.environ.get("MODES", "ip1,ip8").split(",")
    pool, maxb = 3, 8
    g = torch.Generator(device="cuda").manual_seed(1)
    Y = torch.randint(0, 1024, (pool, maxb, H, W), dtype=torch.int16, device="cuda", generator=g)
    U = torch.randint(0, 1024, (pool, maxb, H // 2, W // 2), dtype=torch.int16, device="cuda", generator=g)
    V = torch.randint(0, 1024, (pool, maxb, H // 2, W // 2), dtype=torch.int16, device="cuda", generator=g)
    Yd, Ud, Vd = torch.empty_like(Y), torch.empty_like(U), torch.empty_like(V)
    st = torch.cuda.current_stream().cuda_stream
    yp, cp = Y[0][0].numel() * 2, U[0][0].numel() * 2
    with tempfile.TemporaryDirectory() as tmp:
        libs = []
        for v in variants:
            hw._lib = None
            hw.load(build(v, tmp))
            h = hw.VfgsHip(device=0)
            T.replay(h, T.load_trace("fgs_sei_10_420"))
            libs.append((v, h))

        def run(h, mode):
            oop, batch = mode.startswith("op"), int(mode[2:])
            steps = max(2, 48 // batch)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(steps):
                s = i % pool
                if oop:
                    h.add_grain_copy_dev(Y[s].data_ptr(), U[s].data_ptr(), V[s].data_ptr(), Yd[s].data_ptr(), Ud[s].data_ptr(),
                                         Vd[s].data_ptr(), W, H, 0, H, W, W // 2, batch, yp, cp, st)
                else:
                    h.add_grain_frames_dev(Y[s].data_ptr(), U[s].data_ptr(), V[s].data_ptr(), W, H, W, W // 2, batch, yp, cp, st)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / steps / batch * 1e3

        res = {(v, m): [] for v, _ in libs for m in modes}
        for r in range(rounds + 1):
            for v, h in libs:
                for m in modes:
                    t = run(h, m)
                    if r:
                        res[(v, m)].append(t)
        for v, _ in libs:
            line = f"variant {v:44s}"
            for m in modes:
                ts = res[(v, m)]
                med = statistics.median(ts)
                line += f" | {m}: med {med:6.2f} min {min(ts):6.2f} us ({GB / med:5.0f} GB/s)"
            print(line, flush=True)


if __name__ == "__main__":
    main()
