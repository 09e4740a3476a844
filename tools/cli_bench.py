#!/usr/bin/env python3
"""Developer tool: the reference's own command line program, unchanged (vfgs_main.c + vfgs_fw.c + yuv.c), linked against
libvfgs_hip.so (oracle/_ref/vfgs_hip_cli) next to the all-reference binary (oracle/_ref/vfgs_ref): seconds per frame INCLUDING
the program's file I/O (stdio on /dev/shm), from the difference of two runs with different frame counts (process start, GPU
initialisation and the first, line-by-line walk of the buffer cancel out).  Outputs must be byte-identical."""
import hashlib
import json
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import vfgs_testlib as T  # noqa: E402


def run(exe, w, h, n, inp, out, env=None):
    import os
    t0 = time.perf_counter()
    subprocess.run([str(exe), "-w", str(w), "-h", str(h), "-b", "10", "-n", str(n), "-r", "12345", str(inp), str(out)], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900, env=dict(os.environ, **(env or {})))
    return time.perf_counter() - t0


def main():
    cli, ref = T.REF_DIR / "vfgs_hip_cli", T.REF_DIR / "vfgs_ref"
    assert cli.exists() and ref.exists(), "oracle/_ref binaries were not prebuilt (make -C oracle ref where /root/reference exists)"
    shm = Path("/dev/shm")
    for (w, h, n1, n2) in ((1920, 1080, 10, 210), (3840, 2160, 6, 56), (7680, 4320, 3, 15)):
        frames, _ = T.lcg_frames(w, h, 10, 2, 2, 3)
        inp = shm / f"vfgs_cli_in_{w}.yuv"
        with open(inp, "wb") as f:
            for i in range(n2):
                f.write(frames[i % 3].picture_bytes())
        res = {"size": f"{w}x{h}", "depth": 10, "cfg": "built-in default SEI (vfgs_main.c:69-110)", "frames": [n1, n2]}
        md5 = {}
        for name, exe in (("reference", ref), ("hip", cli)):
            out = shm / f"vfgs_cli_out_{name}.yuv"
            run(exe, w, h, n1, inp, out)            # (also warms the page cache)
            t2 = min(run(exe, w, h, n2, inp, out) for _ in range(2))
            md5[name] = hashlib.md5(out.read_bytes()).hexdigest()
            # short runs: plain and (for the library) with the frame height promised from outside, INTERLEAVED -- round 5 ran the
            # three promised runs behind everything else and read an order effect of the box as a cost of the promise
            # (profiles/r05_cli_short_runs.log vs profiles/r06_promise_probe_*.log)
            t1s, tps = [], []
            for _ in range(3):
                t1s.append(run(exe, w, h, n1, inp, out))
                md5[name + "_short"] = hashlib.md5(out.read_bytes()).hexdigest()
                if name == "hip":
                    tps.append(run(exe, w, h, n1, inp, out, {"VFGS_HIP_FRAME_HEIGHT": str(h)}))
                    md5["hip_promised"] = hashlib.md5(out.read_bytes()).hexdigest()
            t1 = min(t1s)
            per = (t2 - t1) / (n2 - n1)
            res[f"{name}_ms_per_frame_incl_file_io"] = round(per * 1e3, 2)
            res[f"{name}_frames_per_s"] = round(1 / per, 1)
            res[f"{name}_process_s_for_{n1}_frames"] = round(t1, 3)
            if name == "hip":
                # the same unchanged binary with the frame height promised from outside: the first walk is computed ahead too
                res[f"hip_process_s_for_{n1}_frames_with_VFGS_HIP_FRAME_HEIGHT"] = round(min(tps), 3)
                res["hip_short_runs_s"] = {"plain": [round(t, 3) for t in t1s], "promised": [round(t, 3) for t in tps]}
            out.unlink()
        inp.unlink()
        res["identical_output"] = md5["reference"] == md5["hip"] and md5["reference_short"] == md5["hip_short"] == md5["hip_promised"]
        res["speedup"] = round(res["reference_ms_per_frame_incl_file_io"] / res["hip_ms_per_frame_incl_file_io"], 1)
        print(json.dumps(res), flush=True)
        if not res["identical_output"]:
            sys.exit(1)


if __name__ == "__main__":
    main()
