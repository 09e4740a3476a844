"""TEST INFRASTRUCTURE -- numpy restatement of the reference firmware's two grain pattern generators
(/root/reference/src/vfgs_fw.c).  Only tests may import this; the product generates patterns on the
GPU (versatilefilmgrain_amd/csrc/vfgs_fw_kernel.hip) and has no CPU path.

Pinned (tests/test_fw_oracle.py) against the pattern bytes the REAL reference firmware programmed,
recorded in tests/golden/traces, for frequency-filtered, SEI-AR and AFGS1 configurations.

  ff_pattern(n, fh, fv, seed_index)                      vfgs_fw.c:362-408 + :297-360
  ar_pattern(chroma, coef[4][7], scale, shift, seed_idx) vfgs_fw.c:464-501 (coefficient layout of :412)
  sei_ar_taps / afgs1_taps                               vfgs_fw.c:427-436 / :461-464

The constant tables (Gaussian samples, seeds, 64-point DCT-II basis) are read from the blob the
library links in (csrc/fw_tables.bin, layout in oracle/dump_fw_tables.c).
"""
from pathlib import Path

import numpy as np

_BLOB = (Path(__file__).resolve().parent.parent / "versatilefilmgrain_amd" / "csrc" / "fw_tables.bin").read_bytes()
GAUSS = np.frombuffer(_BLOB[:2048], dtype=np.int8).astype(np.int64)
SEEDS = np.frombuffer(_BLOB[2048:3072], dtype=np.uint32)
DCT64 = np.frombuffer(_BLOB[3072:], dtype=np.int8).reshape(64, 64).astype(np.int64)


def prng(x):
    """vfgs_fw.c:284-295, the variant that equals the hardware layer's LFSR."""
    return ((x >> 1) | ((((x >> 1) ^ (x >> 29)) & 1) << 31)) & 0xFFFFFFFF


def ff_pattern(n, fh, fv, seed_index):
    """n = 64 (luma) or 32 (chroma): band-limited noise, then two integer basis passes, clip +-127."""
    gw = n // 16
    fh, fv = gw * (fh + 1), gw * (fv + 1)
    B = np.zeros((n, n), dtype=np.int64)
    r = int(SEEDS[seed_index])
    for l in range(n):
        for k in range(0, n, gw):
            if k < fh and l < fv:
                for j in range(gw):
                    B[l, k + j] = GAUSS[(r + j) & 2047]
            r = prng(r)
    B[0, 0] = 0
    D = DCT64[::64 // n, :n]                          # the 32-point basis is every other row (vfgs_fw.c:343)
    X = ((256 if n == 64 else 128) + D.T @ B) >> (9 if n == 64 else 8)
    return np.clip((256 + X @ D) >> 9, -127, 127).astype(np.int8)


def ar_pattern(chroma, coef, scale, shift, seed_index):
    """Causal 4x7 filter in raster order + Gaussian noise per sample; returns the cropped 64x64 / 32x32 window."""
    sub = 2 if chroma else 1
    w, h = (44, 38) if chroma else (82, 73)
    c = np.asarray(coef, dtype=np.int64).reshape(4, 7)
    buf = np.zeros((h, w), dtype=np.int64)
    r = int(SEEDS[seed_index])
    for y in range(h):
        for x in range(w):
            g = 0
            if y >= 3 and 3 <= x < w - 3:
                g = int((c[:3] * buf[y - 3:y, x - 3:x + 4]).sum() + (c[3, :3] * buf[y, x - 3:x]).sum())
                g = (g + (1 << (scale - 1))) >> scale
            g += (int(GAUSS[r & 2047]) + (1 << (shift - 1))) >> shift
            r = prng(r)
            buf[y, x] = max(-127, min(127, g))
    n, off = 64 // sub, 3 + 6 // sub
    return buf[off:off + n, off:off + n].astype(np.int8)


def sei_ar_taps(v, scale):
    """SEI auto-regressive model values -> tap matrix (vfgs_fw.c:427-436), int16 storage as in the reference."""
    c = np.zeros((4, 7), dtype=np.int64)
    c[3, 2] = v[1]
    c[2, 3] = (v[1] * v[4]) >> scale
    c[2, 2] = c[2, 4] = (v[3] * v[4]) >> scale
    c[3, 1] = v[5]
    c[1, 3] = (v[5] * v[4] * v[4]) >> (2 * scale)
    return ((c + 32768) % 65536 - 32768)


def afgs1_taps(ar, lag):
    """AV1 coefficient order -> tap matrix (vfgs_fw.c:461-464)."""
    c = np.zeros((4, 7), dtype=np.int64)
    k = 0
    for j in range(-lag, 1):
        for i in range(-lag, lag + 1):
            if not (i < 0 or j < 0):
                break
            c[3 + j, 3 + i] = ar[k]
            k += 1
    return c
