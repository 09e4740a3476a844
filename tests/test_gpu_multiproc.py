"""GPU (MI355X): the multi-GPU stripe split run by SEPARATE PROCESSES of the HIP path (one per rank, as bench.py --gpus N
runs them on a node; here both use the box's one device).  Each rank holds and processes only its stripe of every frame
in one launch, never talks to the other rank, and the assembled frames must be the reference CLI's output."""
import json
import subprocess
import sys

import numpy as np
import pytest

import vfgs_testlib as T

pytestmark = pytest.mark.gpu
MD5 = json.loads((T.GOLDEN / "md5.json").read_text())


@pytest.mark.parametrize("key,world", [("cfg5_4320p_10b_420_fgs_sei", 2), ("cfg1_1080p_10b_420_fgs_sei", 3)])
def test_stripe_split_across_processes_equals_reference_md5(tmp_path, key, world):
    g = MD5["full"][key]
    procs = []
    for r in range(world):
        out = tmp_path / f"rank{r}.npz"
        procs.append((out, subprocess.Popen([sys.executable, str(T.ROOT / "tests" / "gpu_rank_worker.py"), str(r), str(world), key, str(out)],
                                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    parts = []
    for out, p in procs:
        log, _ = p.communicate(timeout=600)
        assert p.returncode == 0, log
        parts.append(np.load(out))
    # assemble: the stripes tile the picture exactly, in rank order
    assert parts[0]["y0"] == 0 and parts[-1]["y1"] == g["height"]
    for a, b in zip(parts, parts[1:]):
        assert a["y1"] == b["y0"]
    sx, sy = {"420": (2, 2), "422": (2, 1), "444": (1, 1)}[g["format"]]
    frames = []
    for i in range(g["frames"]):
        f = T.Frame(g["width"], g["height"], g["depth"], sx, sy)
        for p in parts:
            y0, y1 = int(p["y0"]), int(p["y1"])
            f.Y[y0:y1] = p["Y"][i]
            f.U[y0 // sy:y1 // sy] = p["U"][i]
            f.V[y0 // sy:y1 // sy] = p["V"][i]
        frames.append(f)
    assert T.md5_frames(frames) == g["output_md5"]
    # every rank ends with the registers of a process that did the whole frames
    ora = T.OracleHW()
    T.replay(ora, T.load_trace(f'{g["cfg"]}_{g["depth"]}_{g["format"]}'))
    dummy, _ = T.lcg_frames(g["width"], g["height"], g["depth"], sx, sy, 1)
    for _ in range(g["frames"]):
        ora.add_grain_frame(dummy[0].copy())
    for p in parts:
        assert tuple(int(x) for x in p["seeds"]) == ora.seed_state()
