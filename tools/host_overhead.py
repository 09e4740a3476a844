#!/usr/bin/env python3
"""Developer tool: host time per device call (seed state machine, LFSR stream refills, launch) for the
call shapes bench.py uses at 1 and at 8 ranks.  Calls are queued without waiting, so this is the rate at
which the host can feed the GPU; it must stay above the kernel rate or the multi-GPU run is host-bound."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402

import vfgs_testlib as T  # noqa: E402
from versatilefilmgrain_amd import hw  # noqa: E402

W, H = 7680, 4320


def main():
    h = hw.VfgsHip(device=0)
    T.replay(h, T.load_trace("fgs_sei_10_420"))
    st = torch.cuda.current_stream().cuda_stream
    import itertools
    for scaling, ranks in itertools.product(("weak", "strong"), (1, 2, 4, 8)):
        nbr = (H + 15) // 16
        rows = nbr // ranks + (1 if nbr % ranks else 0)
        part_h = min(rows * 16, H)
        frames = 8 * ranks if scaling == "weak" else 8        # bench.py: weak = ranks x batch frames per step, strong = batch frames
        Y = torch.zeros((part_h, W), dtype=torch.int16, device="cuda")
        U = torch.zeros((part_h // 2, W // 2), dtype=torch.int16, device="cuda")
        V = torch.zeros((part_h // 2, W // 2), dtype=torch.int16, device="cuda")
        calls = 3     # short bursts: the LFSR slot ring (4 slots) applies back-pressure on longer ones

        def once():
            # all frames alias one stripe (frame pitch 0): the content is irrelevant here
            h.add_grain_frames_part_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), W, H, 0, part_h, W, W // 2, frames, 0, 0, st)
        once()
        torch.cuda.synchronize()
        host, total = [], []
        for _ in range(8):
            t0 = time.perf_counter()
            for _ in range(calls):
                once()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            host.append(1e6 * (t1 - t0) / calls)
            total.append(1e6 * (t2 - t0) / calls)
        host.sort(); total.sort()
        hm, tm = host[len(host) // 2], total[len(total) // 2]
        print(f"{scaling:6s} ranks {ranks}: {frames:2d} frames x {part_h:4d} lines per call: host {hm:7.1f} us per call, "
              f"host+GPU {tm:7.1f} us per call (host = {100 * hm / tm:4.1f} % of it; it runs ahead of the GPU while that stays < 100 %)")


if __name__ == "__main__":
    main()
