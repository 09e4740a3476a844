#!/usr/bin/env python3
"""Developer helper (build container): copies csrc/ to a scratch directory and cuts work OUT of the grain kernels by textual edits,
for timing-only builds (WRONG output) that show what the row walk costs without its LDS reads / without its arithmetic.

usage: tools/dev/make_ablation_src.py <outdir> [patch:<file>] [nolut] [nopat] [nograin] [noprologue] [pad:N] [git:<rev>]
  patch:F  apply a patch of csrc/ first (git diff format, e.g. tools/dev/r05_w16_int16_bank.patch)
  nolut    8-bit LUT "gather" takes the address itself (address arithmetic stays, no LDS read)
  nopat    one-pattern forms: pattern values from registers instead of the LDS read
  nograin  no grain_unit at all: the rows are only walked (loads, lane rotation, stores)
  nograin_luma / nograin_chroma  the same for one plane type only: which planes' arithmetic costs what
  noprologue  (with nograin) no table image, no block parameters, no barriers: the bare walk
  pad:N    every table image N bytes larger (staged and never read): what the size of the image costs
  git:REV  take the sources from that revision instead of the working tree
The shipped source carries no such switches (vfgs_layout.h); build the result with tools/dev/build_variant.sh SRC=<outdir>.
"""
import subprocess, sys, os, shutil
out = sys.argv[1]
opts = set(sys.argv[2:])
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
csrc = os.path.join(root, "versatilefilmgrain_amd", "csrc")
os.makedirs(out, exist_ok=True)
rev = [o[4:] for o in opts if o.startswith("git:")]
for f in os.listdir(csrc):
    if rev:
        data = subprocess.run(["git", "-C", root, "show", f"{rev[0]}:versatilefilmgrain_amd/csrc/{f}"], capture_output=True).stdout
        open(os.path.join(out, f), "wb").write(data)
    else:
        shutil.copy(os.path.join(csrc, f), os.path.join(out, f))
for o in opts:
    if o.startswith("patch:"):
        subprocess.run(["patch", "-s", "-p3", "-d", out, "-i", os.path.abspath(o[6:])], check=True)
p = os.path.join(out, "vfgs_kernel.hip")
s = open(p).read()
w16 = "mad_half_i16" in s      # the patched source of tools/dev/r05_w16_int16_bank.patch
def rep(a, b):
    global s
    assert s.count(a) == 1, (s.count(a), a)
    s = s.replace(a, b)
if "nolut" in opts and not w16:
    for k in ("(v << 2)", "(v >> 6)", "(v >> 14)", "(v >> 22)"):
        rep("*(const uint32_t*)(lds + ((%s & 0x3fcu) | k1));" % k, "((%s & 0x3fcu) | k1) * 0x00010101u;" % k)
if "nolut" in opts and w16:
    rep("e[4 * q + 0] = lds[v & 0xffu];", "e[4 * q + 0] = v & 0xffu;")
    rep("e[4 * q + 1] = lds[(v >> 8) & 0xffu];", "e[4 * q + 1] = (v >> 8) & 0xffu;")
    rep("e[4 * q + 2] = lds[(v >> 16) & 0xffu];", "e[4 * q + 2] = (v >> 16) & 0xffu;")
    rep("e[4 * q + 3] = lds[v >> 24];", "e[4 * q + 3] = v >> 24;")
if "nopat" in opts and not w16:
    rep("d = *(const uint32_t*)(lds + adq + M::col(q));", "d = (adq + M::col(q)) * 0x9e3779b1u;")
    rep("const uint32_t lo = *(const uint32_t*)(lds + (a4 & ~3u)), hi = *(const uint32_t*)(lds + (a4 & ~3u) + 4);", "const uint32_t lo = a4 * 0x9e3779b1u, hi = lo ^ (a4 << 7);")
if "nopat" in opts and w16:
    rep("else { const u32x2 t = *(const u32x2*)(lds + a8); raw = t.x; raw_hi = t.y; }",
        "else { raw = a8 * 0x9e3779b1u; raw_hi = raw ^ (a8 << 7); }")
    rep("if (ALIGN2) { raw = *(const uint32_t*)(lds + a8); raw_hi = *(const uint32_t*)(lds + a8 + 4); }",
        "if (ALIGN2) { raw = a8 * 0x9e3779b1u; raw_hi = raw ^ (a8 << 7); }")
if "nograin" in opts:
    rep("if (NU * g + u < tsegs)\n", "if (false)\n")
    rep("if (valid)\n\t\t\t\t{\n\t\t\t\t\tconst int j = base", "if (false)\n\t\t\t\t{\n\t\t\t\t\tconst int j = base")
for which, cond in (("nograin_luma", "comp != 0"), ("nograin_chroma", "comp == 0")):
    if which in opts:
        rep("if (NU * g + u < tsegs)\n", "if (%s && NU * g + u < tsegs)\n" % cond)
        rep("if (valid)\n\t\t\t\t{\n\t\t\t\t\tconst int j = base", "if (%s && valid)\n\t\t\t\t{\n\t\t\t\t\tconst int j = base" % cond)
if "noprologue" in opts:
    assert s.count("if (!PERSIST || first_task)\n") == 2
    s = s.replace("if (!PERSIST || first_task)\n", "if (false)\n")
    rep("\tparam_loads(0, wc0, wu0);\n", "\tfor (int i = 0; i < NPE; i++) { wc0[i] = u32x2{0, 0}; wu0[i] = u32x2{0, 0}; }\n")
    rep("\tparam_table(0, wc0, wu0);\n\t__syncthreads();\n", "")
open(p, "w").write(s)
pad = [o[4:] for o in opts if o.startswith("pad:")]
if pad:
    p2 = os.path.join(out, "vfgs_layout.h")
    t = open(p2).read()
    for a in ("L.y_bytes = L.y_bank + 64 * L.y_rs + L.y_neg;", "L.c_bytes = L.c_bank + L.ch * L.c_rs + L.c_neg;"):
        assert t.count(a) == 1
        t = t.replace(a, a[:-1] + " + " + pad[0] + ";")
    open(p2, "w").write(t)
print("wrote", out, sorted(opts))
