#!/usr/bin/env python3
"""Developer tool: does the grain kernel's speed depend on the picture content?
Same run, interleaved: uniform random samples / ramp + noise ("natural-like") / one flat value."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402

import vfgs_testlib as T  # noqa: E402
from versatilefilmgrain_amd import hw  # noqa: E402

W, H, B = 7680, 4320, 8


def make(kind, n):
    g = torch.Generator(device="cuda").manual_seed(5)

    def mk(hh, ww):
        if kind == "uniform":
            return torch.randint(0, 1024, (n, hh, ww), dtype=torch.int32, device="cuda", generator=g).to(torch.int16)
        if kind == "flat":
            return torch.full((n, hh, ww), 500, dtype=torch.int16, device="cuda")
        ramp = (torch.arange(ww, device="cuda").float()[None, :] / ww * 0.6 + torch.arange(hh, device="cuda").float()[:, None] / hh * 0.3 + 0.05) * 1024
        t = ramp[None].expand(n, hh, ww) + torch.randint(-4, 5, (n, hh, ww), device="cuda", generator=g)
        return t.clamp(0, 1023).to(torch.int16).contiguous()
    return mk(H, W), mk(H // 2, W // 2), mk(H // 2, W // 2)


def main():
    h = hw.VfgsHip(device=0)
    T.replay(h, T.load_trace(sys.argv[1] if len(sys.argv) > 1 else "fgs_sei_10_420"))
    kinds = ["uniform", "natural", "flat"]
    pristine = {k: [make(k, B) for _ in range(2)] for k in kinds}
    work = [tuple(torch.empty_like(t) for t in pristine["uniform"][0]) for _ in range(3)]
    st = torch.cuda.current_stream().cuda_stream
    res = {k: [] for k in kinds}
    for r in range(6):
        for k in kinds:
            # fresh content every round (in-place processing would otherwise drift towards the clip bounds)
            for i, wset in enumerate(work):
                for dst, src in zip(wset, pristine[k][i % 2]):
                    dst.copy_(src)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(6):
                Y, U, V = work[i % 3]
                h.add_grain_frames_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), W, H, W, W // 2, B, Y[0].numel() * 2, U[0].numel() * 2, st)
            e1.record()
            torch.cuda.synchronize()
            if r:
                res[k].append(e0.elapsed_time(e1) / 6 / B * 1e3)
    for k in kinds:
        v = sorted(res[k])
        print(f"{k:8s} us/frame median {v[len(v)//2]:.2f}  min {v[0]:.2f}  max {v[-1]:.2f}")


if __name__ == "__main__":
    main()
