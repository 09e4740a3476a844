"""One rank of the multi-GPU stripe split, as its own process (tests/test_gpu_multiproc.py starts two of them):
fresh process -> program the library -> its stripe of every frame of the batch in ONE
vfgs_hip_add_grain_frames_part_dev launch -> write the stripe's planes to <out>.npz.  No process group, no
collective: the ranks never talk to each other."""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    rank, world, key, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    import torch
    import vfgs_testlib as T
    from bench import split_rows
    from versatilefilmgrain_amd import hw

    g = json.loads((T.GOLDEN / "md5.json").read_text())["full"][key]
    sx, sy = {"420": (2, 2), "422": (2, 1), "444": (1, 1)}[g["format"]]
    W, H, nfr = g["width"], g["height"], g["frames"]
    torch.cuda.set_device(0)                       # both ranks share the box's one GPU; on a node each takes its own
    h = hw.VfgsHip(device=0)
    T.replay(h, T.load_trace(f'{g["cfg"]}_{g["depth"]}_{g["format"]}'))
    frames, _ = T.lcg_frames(W, H, g["depth"], sx, sy, nfr)
    nbr = (H + 15) // 16
    row0, nrows = split_rows(nbr, world)[rank]
    y0, y1 = row0 * 16, min((row0 + nrows) * 16, H)
    ph = y1 - y0
    f0 = frames[0]
    # the rank's memory holds ONLY its stripe of each frame
    Y = torch.from_numpy(np.stack([f.Y[y0:y1] for f in frames])).cuda()
    U = torch.from_numpy(np.stack([f.U[y0 // sy:y1 // sy] for f in frames])).cuda()
    V = torch.from_numpy(np.stack([f.V[y0 // sy:y1 // sy] for f in frames])).cuda()
    sz = f0.Y.itemsize
    h.add_grain_frames_part_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), W, H, y0, ph, f0.stride, f0.cstride, nfr,
                                Y[0].numel() * sz, U[0].numel() * sz, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    np.savez(out, Y=Y.cpu().numpy(), U=U.cpu().numpy(), V=V.cpu().numpy(), y0=y0, y1=y1, seeds=np.array(h.seed_state(), dtype=np.uint32))


if __name__ == "__main__":
    main()
