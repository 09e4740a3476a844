#!/usr/bin/env python3
"""Generate the golden fixtures from the REAL reference (run in the build container only).

Needs /root/reference and the in-place reference builds of oracle/Makefile
(oracle/_ref/vfgs_ref, vfgs_ref_x, vfgs_trace).  Writes *data only*:

  tests/golden/traces/<cfg>_<depth>_<fmt>.npz   what the reference firmware programs into the
                                                hardware layer (setter calls + payloads) for
                                                `vfgs -b <depth> -f <fmt> -r 12345 -c <cfg>`
  tests/golden/md5.json                         md5 of the reference CLI's output on the LCG
                                                synthetic input
  tests/golden/frames/*.npz                     a few complete small reference outputs
  tests/golden/fwcfg/<cfg>_<depth>_<fmt>.npz    the parameter structures (fgs_sei / fgs_afgs1,
                                                vfgs_fw.h:51-89) the reference CLI hands to its
                                                firmware for that command line, in call order:
                                                the INPUT of the firmware-layer tests, whose
                                                expected output is the trace of the same name

Nothing here is read at test time from /root/reference; the fixtures are.
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vfgs_testlib as T  # noqa: E402

REF = Path("/root/reference")
CFG = REF / "cfg"
SEED = 12345
SMALL = (192, 144, 3)  # width, height, frames

FULL_SIZE = [  # BASELINE.json configs (name, w, h, depth, fmt, cfg, frames)
    ("cfg1_1080p_10b_420_fgs_sei", 1920, 1080, 10, "420", "fgs_sei", 10),
    ("cfg2_1080p_10b_420_ff_test1", 1920, 1080, 10, "420", "fgs_sei_ff_test1", 10),
    ("cfg3_2160p_10b_420_ar_test1", 3840, 2160, 10, "420", "fgs_sei_ar_test1", 2),
    ("cfg4_2160p_8b_444_afgs1_test1", 3840, 2160, 8, "444", "fgs_afgs1_test1", 2),
    ("cfg5_4320p_10b_420_fgs_sei", 7680, 4320, 10, "420", "fgs_sei", 2),
]

NON_420 = [  # (cfg, depth, fmt): driven with --no-check (vfgs_main.c:235 rejects them otherwise)
    ("fgs_afgs1_test1", 8, "444"), ("fgs_afgs1_test1", 10, "444"),
    ("fgs_sei", 10, "444"), ("fgs_sei", 10, "422"), ("fgs_sei", 8, "422"),
    ("fgs_sei_ff_test6", 10, "444"), ("fgs_sei_ff_test6", 8, "422"), ("fgs_sei_ff_test6", 8, "444"),
    ("fgs_afgs1_test3", 10, "422"),
]

# the two other file syntaxes the CLI reads (vfgs_main.c:309-434 grain table, :490-514 SEI dump)
OTHER_SYNTAX = [("fgs_afgs1_test1.tbl", 10, "420"), ("fgs_afgs1_test1.tbl", 8, "420"), ("fgs_sei_dump.txt", 10, "420")]

SUB = {"420": (2, 2), "422": (2, 1), "444": (1, 1)}


def cfg_path(cfg):
    return CFG / (cfg if "." in cfg else f"{cfg}.cfg")


def job_name(cfg, depth, fmt):
    return f"{(cfg or 'default').replace('.', '_')}_{depth}_{fmt}"


def write_input(path, w, h, depth, fmt, nframes):
    sx, sy = SUB[fmt]
    frames, _ = T.lcg_frames(w, h, depth, sx, sy, nframes)
    with open(path, "wb") as f:
        for fr in frames:
            f.write(fr.picture_bytes())


def run_ref(w, h, depth, fmt, cfg, nframes, inp, out, outdepth=None):
    exe = "vfgs_ref" if fmt == "420" else "vfgs_ref_x"
    cmd = [str(T.REF_DIR / exe), "-w", str(w), "-h", str(h), "-b", str(depth), "-f", fmt,
           "-n", str(nframes), "-r", str(SEED)]
    if outdepth:
        cmd += ["--outdepth", str(outdepth)]
    if fmt != "420":
        cmd.append("--no-check")
    if cfg:
        cmd += ["-c", str(cfg_path(cfg))]
    cmd += [inp, out]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
    return hashlib.md5(Path(out).read_bytes()).hexdigest()


def record_trace(depth, fmt, cfg, dst):
    with tempfile.NamedTemporaryFile(suffix=".trace", delete=False) as t:
        tname = t.name
    cmd = [str(T.REF_DIR / "vfgs_trace"), "--program-only", "--no-check", "-b", str(depth), "-f", fmt, "-r", str(SEED)]
    if fmt == "420":
        cmd.remove("--no-check")
    if cfg:
        cmd += ["-c", str(cfg_path(cfg))]
    env = dict(os.environ, VFGS_TRACE_OUT=tname)
    subprocess.run(cmd, check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    T.save_trace_npz(T.parse_trace_file(tname), dst)
    os.unlink(tname)


def record_fwcfg(depth, fmt, cfg, dst):
    with tempfile.NamedTemporaryFile(suffix=".cfgdump", delete=False) as t:
        tname = t.name
    os.unlink(tname)
    cmd = [str(T.REF_DIR / "vfgs_ref_x"), "--program-only", "--no-check", "--dump-cfg", tname,
           "-b", str(depth), "-f", fmt, "-r", str(SEED)]
    if fmt == "420":
        cmd.remove("--no-check")
    if cfg:
        cmd += ["-c", str(cfg_path(cfg))]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    raw = Path(tname).read_bytes()
    os.unlink(tname)
    kinds, blobs, off = [], {}, 0
    while off < len(raw):
        kind, n = np.frombuffer(raw, dtype=np.uint32, count=2, offset=off)
        blobs[f"cfg{len(kinds)}"] = np.frombuffer(raw, dtype=np.uint8, count=int(n), offset=off + 8).copy()
        kinds.append(int(kind))
        off += 8 + int(n)
    np.savez_compressed(dst, kinds=np.array(kinds, dtype=np.int32), seed=np.array([SEED], dtype=np.uint32), **blobs)


DERIVED_440 = ["fgs_sei_10_444", "fgs_afgs1_test1_8_444", "fgs_afgs1_test1_10_444", "fgs_sei_ff_test6_10_444"]


def derive_440():
    """The hardware layer accepts csubx = 1, csuby = 2 (vfgs_hw.c:382-388) but the reference CLI has no such
    format (-f 420 | 422 | 444).  Programming sequences for it are therefore derived from the recorded 4:4:4
    ones: identical setter calls, only vfgs_set_chroma_subsampling(1, 1) becomes (1, 2) -- which also changes
    what the following vfgs_set_chroma_pattern calls copy (vfgs_hw.c:320-325).  Expected outputs come from the
    real hardware layer driven directly (tests/test_oracle_vs_reference.py), not from the CLI."""
    for name in DERIVED_440:
        rec = T.load_trace(name)
        out = [(op, a, 2 if op == T.OP_CHROMA_SUBSAMPLING else b, p) for op, a, b, p in rec]
        assert any(op == T.OP_CHROMA_SUBSAMPLING and (a, b) == (1, 2) for op, a, b, _ in out), name
        T.save_trace_npz(out, T.TRACES / f"{name[:-3]}440.npz")
        print("derived", f"{name[:-3]}440")


def fwcfg_only():
    (T.GOLDEN / "fwcfg").mkdir(parents=True, exist_ok=True)
    cfgs = sorted(p.stem for p in CFG.glob("*.cfg"))
    jobs = [(c, d, "420") for c in [None] + cfgs for d in (8, 10)] + NON_420 + OTHER_SYNTAX
    for cfg, depth, fmt in jobs:
        name = job_name(cfg, depth, fmt)
        record_fwcfg(depth, fmt, cfg, T.GOLDEN / "fwcfg" / f"{name}.npz")
        print("fwcfg", name, flush=True)


def cfg_corpus():
    """The configuration files themselves (INPUT data of the cfg reader tests), as one archive."""
    files = sorted(p for p in CFG.iterdir() if p.suffix in (".cfg", ".tbl", ".txt"))
    np.savez_compressed(T.GOLDEN / "cfg_corpus.npz", **{p.name: np.frombuffer(p.read_bytes(), dtype=np.uint8) for p in files})
    print("cfg corpus:", len(files), "files")


def extras_only():
    """Add the grain-table / SEI-dump command lines to the existing fixtures without redoing the rest."""
    w, h, n = SMALL
    md5 = json.loads((T.GOLDEN / "md5.json").read_text())
    with tempfile.TemporaryDirectory() as tmp:
        for cfg, depth, fmt in OTHER_SYNTAX:
            name = job_name(cfg, depth, fmt)
            inp, out = f"{tmp}/in.yuv", f"{tmp}/out.yuv"
            write_input(inp, w, h, depth, fmt, n)
            record_trace(depth, fmt, cfg, T.TRACES / f"{name}.npz")
            record_fwcfg(depth, fmt, cfg, T.GOLDEN / "fwcfg" / f"{name}.npz")
            md5["small"][name] = run_ref(w, h, depth, fmt, cfg, n, inp, out)
            print(name, md5["small"][name], flush=True)
    (T.GOLDEN / "md5.json").write_text(json.dumps(md5, indent=1, sort_keys=True) + "\n")
    derive_440()
    cfg_corpus()


def missing_only():
    """Fixtures of the jobs of NON_420 / OTHER_SYNTAX that have no trace yet (a job added later), without redoing the rest."""
    w, h, n = SMALL
    md5 = json.loads((T.GOLDEN / "md5.json").read_text())
    with tempfile.TemporaryDirectory() as tmp:
        for cfg, depth, fmt in NON_420 + OTHER_SYNTAX:
            name = job_name(cfg, depth, fmt)
            if (T.TRACES / f"{name}.npz").exists():
                continue
            inp, out = f"{tmp}/in.yuv", f"{tmp}/out.yuv"
            write_input(inp, w, h, depth, fmt, n)
            record_trace(depth, fmt, cfg, T.TRACES / f"{name}.npz")
            record_fwcfg(depth, fmt, cfg, T.GOLDEN / "fwcfg" / f"{name}.npz")
            md5["small"][name] = run_ref(w, h, depth, fmt, cfg, n, inp, out)
            print(name, md5["small"][name], flush=True)
    (T.GOLDEN / "md5.json").write_text(json.dumps(md5, indent=1, sort_keys=True) + "\n")


def main():
    if "--missing-only" in sys.argv:
        return missing_only()
    if "--fwcfg-only" in sys.argv:
        return fwcfg_only()
    if "--extras-only" in sys.argv:
        return extras_only()
    T.build_oracle()
    (T.GOLDEN / "traces").mkdir(parents=True, exist_ok=True)
    (T.GOLDEN / "frames").mkdir(parents=True, exist_ok=True)
    cfgs = sorted(p.stem for p in CFG.glob("*.cfg"))
    md5 = {"seed": SEED, "lcg_seed": 1, "small": {}, "small_outdepth8": {}, "inputs": {}, "full": {}}
    w, h, n = SMALL

    with tempfile.TemporaryDirectory() as tmp:
        inputs = {}
        for depth in (8, 10):
            for fmt in SUB:
                p = f"{tmp}/in_{depth}_{fmt}.yuv"
                write_input(p, w, h, depth, fmt, n)
                inputs[(depth, fmt)] = p
                md5["inputs"][f"{w}x{h}_{depth}_{fmt}_x{n}"] = hashlib.md5(Path(p).read_bytes()).hexdigest()

        jobs = [(c, d, "420") for c in [None] + cfgs for d in (8, 10)] + NON_420 + OTHER_SYNTAX
        for cfg, depth, fmt in jobs:
            name = job_name(cfg, depth, fmt)
            record_trace(depth, fmt, cfg, T.TRACES / f"{name}.npz")
            (T.GOLDEN / "fwcfg").mkdir(parents=True, exist_ok=True)
            record_fwcfg(depth, fmt, cfg, T.GOLDEN / "fwcfg" / f"{name}.npz")
            out = f"{tmp}/out.yuv"
            md5["small"][name] = run_ref(w, h, depth, fmt, cfg, n, inputs[(depth, fmt)], out)
            if name in ("fgs_sei_10_420", "fgs_afgs1_test1_8_444", "fgs_sei_ff_test6_8_422"):
                sx, sy = SUB[fmt]
                raw = np.frombuffer(Path(out).read_bytes(), dtype=np.uint16 if depth > 8 else np.uint8)
                per = w * h + 2 * (w // sx) * (h // sy)
                np.savez_compressed(T.GOLDEN / "frames" / f"{name}_{w}x{h}.npz", out=raw[:per])  # first frame only
            print(name, md5["small"][name], flush=True)
            # 10-bit in, 8-bit out (yuv_to_8bit, yuv.c:216-258) for the fused-output extension; stock CLI only (4:2:0)
            if depth == 10 and fmt == "420" and cfg in ("fgs_sei", "fgs_afgs1_test1", "fgs_sei_ff_test6", "fgs_sei_ar_test1"):
                md5["small_outdepth8"][name] = run_ref(w, h, depth, fmt, cfg, n, inputs[(depth, fmt)], out, outdepth=8)

        for name, fw, fh, depth, fmt, cfg, nf in FULL_SIZE:
            inp, out = f"{tmp}/full_in.yuv", f"{tmp}/full_out.yuv"
            write_input(inp, fw, fh, depth, fmt, nf)
            md5["full"][name] = {
                "width": fw, "height": fh, "depth": depth, "format": fmt, "cfg": cfg, "frames": nf,
                "input_md5": hashlib.md5(Path(inp).read_bytes()).hexdigest(),
                "output_md5": run_ref(fw, fh, depth, fmt, cfg, nf, inp, out),
            }
            print(name, md5["full"][name], flush=True)
            os.unlink(inp)
            os.unlink(out)

    (T.GOLDEN / "md5.json").write_text(json.dumps(md5, indent=1, sort_keys=True) + "\n")
    derive_440()
    cfg_corpus()


if __name__ == "__main__":
    main()
