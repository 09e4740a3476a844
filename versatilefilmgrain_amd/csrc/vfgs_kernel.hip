// gfx950 (MI355X, CDNA4) film grain kernels.  Written for wave64 / LDS / HBM3E directly;
// there is no other target.
//
// What is computed is the closed form of the reference hardware layer
// (/root/reference/src/vfgs_hw.c:140-312; derivation in DESIGN.md):
//
//   out = clip( in + round( scale[in>>bs] * F( s * pat[slot[in>>bs]][oy+r][ox+i] (+ overlap) ), shift ) )
//
// where (s, ox, oy) come from a 32-bit window of the LFSR bit stream at bit
// (block_row * blocks_per_line + block), F is the 3-tap filter at block edges, and the
// window of the block row above feeds the 2-line overlap.
//
// Work decomposition (see DESIGN.md "kernel" and "Work items" below):
//   * one WAVEFRONT owns one work item = one row of one plane x 4 segments (<= 4 KiB
//     contiguous); each grain block is served by a fixed lane pair of that wavefront, which
//     derives the block's LFSR window, sign and pattern offsets in registers;
//   * every lane moves 16 bytes (10-bit) / 8 bytes (8-bit) = 8 samples per access, so one
//     wave-instruction reads or writes one contiguous <= 1 KiB row segment;
//   * segments are shifted by HALF A BLOCK (8 samples) against the block grid, so every
//     block edge -- the only place where a sample depends on its horizontal neighbours --
//     lies strictly inside a segment: no halo, no inter-wave exchange, in-place is race free;
//   * pattern banks (slot-interleaved) and LUTs are staged once per workgroup in LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "vfgs_layout.h"

namespace vfgs {

struct TableLayoutBase { static constexpr int LUMA_OFF = 0; };

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------
// small device helpers
//
// Issue rates measured on MI355X (tools/valu_rate.hip): plain VOP2 integer ops (add, and,
// shifts) take 2 cycles per wave-instruction; VOP3-only, packed-16, SDWA and DPP forms
// (v_perm_b32, v_mad_*, v_pk_*, v_bfe_*, v_add3) take 4.  The per-sample sequence below is
// chosen against those prices (DESIGN.md "instruction budget").

__device__ __forceinline__ uint32_t stream_window(const uint32_t* __restrict__ s, uint32_t bit)
{
	// 32-bit window of the LFSR bit stream = the register after `bit` steps (vfgs_hw.c:74-79)
	const uint32_t* p = s + (bit >> 5);
	uint32_t lo = p[0], hi = p[1];
	return __builtin_amdgcn_alignbit(hi, lo, bit & 31);
}

// The same in two steps: the two dwords are fetched together with the item's sample loads
// (so that no later wait on them drains younger loads or older stores: vmcnt retires in
// order), the window is cut out when the block parameters are needed.
__device__ __forceinline__ void stream_fetch(const uint32_t* __restrict__ s, uint32_t bit, uint32_t (&raw)[2])
{
	const uint32_t* p = s + (bit >> 5);
	raw[0] = p[0];
	raw[1] = p[1];
}

__device__ __forceinline__ uint32_t stream_cut(const uint32_t (&raw)[2], uint32_t bit)
{
	return __builtin_amdgcn_alignbit(raw[1], raw[0], bit & 31);
}

struct BlockParam {
	uint32_t addr;  // LDS byte offset of bank[.][oy][ox][slot 0]
	int sign;       // +1 / -1
};

// vfgs_hw.c:99-138 -- bit fields of the register per component.
template <int COMP, int SUBX, int SUBY, int RS>
__device__ __forceinline__ BlockParam block_param(uint32_t v, uint32_t bank_off)
{
	uint32_t fx, fy, sb;
	if (COMP == 0)      { fx = v & 0x3ff;         fy = (v >> 14) & 0x3ff;            sb = v >> 31; }
	else if (COMP == 1) { fx = (v >> 10) & 0x3ff; fy = (v >> 24) | ((v & 3u) << 8);  sb = (v >> 2) & 1; }
	else                { fx = (v >> 20) & 0x3ff; fy = (v >> 4) & 0x3ff;             sb = (v >> 15) & 1; }
	uint32_t ox = (__umul24(fx, 13u) >> 10) * (4 / SUBX);
	uint32_t oy = (__umul24(fy, 12u) >> 10) * (4 / SUBY);
	BlockParam r;
	r.addr = __umul24(oy, (uint32_t)RS) + ox * kSlots + bank_off;
	r.sign = sb ? -1 : 1;
	return r;
}

// One sample's pattern value out of its 8-byte slot group {hi,lo}: the LUT entry's top byte is
// the v_perm_b32 selector (slot 0..7, or 0x0c = constant 0) for result byte 3; the arithmetic
// shift then sign-extends it (the other three result bytes are don't-care).
__device__ __forceinline__ int pick_slot(uint32_t hi, uint32_t lo, uint32_t lut_entry)
{
	return (int)__builtin_amdgcn_perm(hi, lo, lut_entry) >> 24;
}

// 24-bit multiplies: the compiler would otherwise pick 32-bit multiplies for these.
__device__ __forceinline__ int mad24(int a, int b, int c)
{
	return __mul24(a, b) + c;
}

// x.i16[0] * y.i16[0] + c  (one VOP3 instruction; y is a LUT entry whose low half is the signed scale)
__device__ __forceinline__ int mad_i16(int x, uint32_t y, int c)
{
	int r;
	asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(c));
	return r;
}

// x * y.i24 + c: the LUT entry's low 24 bits are the signed scale, pre-shifted so that the product's
// HIGH half is the scaled grain (see grain_unit); the selector byte above them is ignored by the i24 multiply
__device__ __forceinline__ int mad_i24(int x, uint32_t y, int c)
{
	int r;
	asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "s"(c));
	return r;
}

// x * y + c on 24-bit operands, all in VGPRs (or inline constants)
__device__ __forceinline__ int mad_vvv(int x, int y, int c)
{
	int r;
	asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(c));
	return r;
}

__device__ __forceinline__ int swap_lane_pairs(int v)
{
	// quad_perm:[1,0,3,2]: lane 2m <-> lane 2m+1
	return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);
}

// Global memory goes through raw buffer instructions: the row base is a wave-uniform scalar
// offset, the lane supplies the byte offset inside the row, and a lane that must not touch
// memory supplies kOOB, which the hardware range check (offset >= num_records) turns into
// "load returns 0 / store is dropped".  No exec-mask branches around loads and stores.
constexpr uint32_t kOOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const uint8_t* base, uint32_t row_bytes)
{
	return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, row_bytes, 0x00020000);
}

// A lane's 8 consecutive samples of one row, as two independently addressable halves of 4
// (SPLIT: the halves belong to different grain blocks and can be valid independently).
// DEPTH 10: 16 bytes in memory, kept as 4 dwords of two uint16 each.
// DEPTH  8:  8 bytes in memory, widened to the same 4 x (2 x uint16) form.
template <int DEPTH, bool SPLIT>
__device__ __forceinline__ void load_unit(__amdgpu_buffer_rsrc_t rs, uint32_t v0, uint32_t v1, uint32_t soff, uint32_t (&w)[4])
{
	if (DEPTH > 8)
	{
		if (!SPLIT)
		{
			const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, v0, soff, VFGS_LDAUX);
			w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w;
		}
		else
		{
			const u32x2 t0 = __builtin_amdgcn_raw_buffer_load_b64(rs, v0, soff, VFGS_LDAUX);
			const u32x2 t1 = __builtin_amdgcn_raw_buffer_load_b64(rs, v1, soff, VFGS_LDAUX);
			w[0] = t0.x; w[1] = t0.y; w[2] = t1.x; w[3] = t1.y;
		}
	}
	else
	{
		uint32_t r0, r1;
		if (!SPLIT)
		{
			const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, v0, soff, VFGS_LDAUX);
			r0 = t.x; r1 = t.y;
		}
		else
		{
			r0 = __builtin_amdgcn_raw_buffer_load_b32(rs, v0, soff, VFGS_LDAUX);
			r1 = __builtin_amdgcn_raw_buffer_load_b32(rs, v1, soff, VFGS_LDAUX);
		}
		w[0] = __builtin_amdgcn_perm(0, r0, 0x0c010c00);
		w[1] = __builtin_amdgcn_perm(0, r0, 0x0c030c02);
		w[2] = __builtin_amdgcn_perm(0, r1, 0x0c010c00);
		w[3] = __builtin_amdgcn_perm(0, r1, 0x0c030c02);
	}
}

template <int DEPTH, bool SPLIT>
__device__ __forceinline__ void store_unit(__amdgpu_buffer_rsrc_t rs, uint32_t v0, uint32_t v1, uint32_t soff, const uint32_t (&w)[4])
{
#if VFGS_ABLATE == 5   // (almost) never store: keeps the math alive, drops the write traffic
	if (!(w[0] == 0x12345678u && w[3] == 0x9abcdef0u)) return;
#endif
	if (DEPTH > 8)
	{
		if (!SPLIT)
		{
			const u32x4 t = {w[0], w[1], w[2], w[3]};
			__builtin_amdgcn_raw_buffer_store_b128(t, rs, v0, soff, VFGS_STAUX);
		}
		else
		{
			const u32x2 t0 = {w[0], w[1]}, t1 = {w[2], w[3]};
			__builtin_amdgcn_raw_buffer_store_b64(t0, rs, v0, soff, VFGS_STAUX);
			__builtin_amdgcn_raw_buffer_store_b64(t1, rs, v1, soff, VFGS_STAUX);
		}
	}
	else
	{
		const uint32_t r0 = __builtin_amdgcn_perm(w[1], w[0], 0x06040200);
		const uint32_t r1 = __builtin_amdgcn_perm(w[3], w[2], 0x06040200);
		if (!SPLIT)
		{
			const u32x2 t = {r0, r1};
			__builtin_amdgcn_raw_buffer_store_b64(t, rs, v0, soff, VFGS_STAUX);
		}
		else
		{
			__builtin_amdgcn_raw_buffer_store_b32(r0, rs, v0, soff, VFGS_STAUX);
			__builtin_amdgcn_raw_buffer_store_b32(r1, rs, v1, soff, VFGS_STAUX);
		}
	}
}

// Fused output narrowing (the step after the path in the reference CLI, yuv_to_8bit,
// yuv.c:216-258: out8 = (uint8)((v + 2) >> 2)): 10-bit results are stored as 8-bit samples.
template <bool SPLIT>
__device__ __forceinline__ void store_unit_narrow(__amdgpu_buffer_rsrc_t rs, uint32_t v0, uint32_t v1, const uint32_t (&w)[4])
{
	uint32_t n[4];
#pragma unroll
	for (int k = 0; k < 4; k++)
		n[k] = ((w[k] + 0x00020002u) >> 2) & 0x00ff00ffu;     // both halves <= 1022: no carry across the halves
	store_unit<8, SPLIT>(rs, v0, v1, 0, n);
}

// ---------------------------------------------------------------------------------------
// The per-lane grain pipeline for 8 samples of one row.
//
//   w          in/out: samples, 4 x (2 x uint16)
//   lut0,lut1  LDS byte offsets of the 256-entry LUT used by samples 0-3 / 4-7
//   a0,a1      LDS byte offsets of the pattern data of samples 0-3 / 4-7 (current block row)
//
// Two forms of the same arithmetic (vfgs_hw.c:211-229, 250-267):
//
// OVERLAP = true (the two lines under a block-row boundary): pattern values are blended as
//   P = (Pcur * m + Pup * n + 16) >> 5 with m = sign_cur * w_cur, n = sign_up * w_up, so P is
//   the signed grain; lut0/lut1 are the +scale tables; rel = 1, c0 = c1 = 2.
//   Lanes on other lines of the same access carry (m, n) = (32 * sign, 0): (32 P + 16) >> 5 == P.
//
// OVERLAP = false: the block sign s is folded into the scale instead of the pattern value:
//   lut0/lut1 point at the table of sign * scale, P~ = s * P is used unsigned-by-sign, and
//   round(scale * P, shift) == (s*scale) * P~ ... exactly (s*s == 1).  The 3-tap edge filter
//   F = (l1 + 3 l0 + r0 + 2) >> 2 on true values becomes, in the P~ domain of the lane whose
//   sample is filtered,  F~ = (a~ + 3 b~ + rel * c~ + (s > 0 ? 2 : 1)) >> 2  with rel = s * s'
//   the relative sign of the two blocks: for s = -1, -((-A + 2) >> 2) == (A + 1) >> 2.
//
//   EDGE16 true : block edge between this lane and its pair lane (16-sample blocks); `first`
//                 lanes hold the right-hand block's first sample in slot 0, their partners
//                 the left-hand block's last sample in slot 7
//          false: block edge between samples 3 and 4 of this lane (8-sample blocks)
template <int DEPTH, bool OVERLAP, bool EDGE16>
__device__ __forceinline__ void grain_unit(const uint8_t* lds, uint32_t (&w)[4], uint32_t lut0, uint32_t lut1,
                                            uint32_t a0, uint32_t a1, int m0, int m1,
                                            uint32_t u0, uint32_t u1, int n0, int n1,
                                            bool edge_on, bool first, int rel, int c0, int c1,
                                            int scale_shift, int half, uint32_t lo2, uint32_t hi2)
{
#if VFGS_ABLATE == 1 || (VFGS_ABLATE >= 11 && VFGS_ABLATE <= 13)   // copy only (11: + no table staging, 12: + no LFSR loads, 13: both)
	return;
#endif
	uint32_t e[8];
	int P[8];

	// LUT gather: intensity = sample >> bs, as a uint8 (vfgs_hw.c:157,211); entry address = 4*intensity
#pragma unroll
	for (int k = 0; k < 4; k++)
	{
		const uint32_t idx = (DEPTH > 8) ? (w[k] & 0x03fc03fcu) : ((w[k] & 0x00ff00ffu) << 2);
		const uint32_t lut = k < 2 ? lut0 : lut1;
#if VFGS_ABLATE == 8 || VFGS_ABLATE == 10   // timing only: no LUT gather
		e[2 * k]     = (idx << 22) | 37u | lut;
		e[2 * k + 1] = (idx << 6) | 53u | lut;
#else
		e[2 * k]     = *(const uint32_t*)(lds + lut + (idx & 0xffffu));
		e[2 * k + 1] = *(const uint32_t*)(lds + lut + (idx >> 16));
#endif
	}

	// pattern fetch: 4 samples x 8 slots = 32 bytes per half
	{
#if VFGS_ABLATE == 9 || VFGS_ABLATE == 10    // timing only: no pattern fetch from LDS
		const u32x4 c0_ = {a0, a1, a0 * 3, a1 * 5}, c1_ = {a0 ^ a1, a0 + a1, a0 * 7, a1 * 9};
		const u32x4 c2_ = {a1, a0, a1 * 3, a0 * 5}, c3_ = {a1 ^ 77, a0 + 99, a1 * 7, a0 * 9};
#else
		const u32x4 c0_ = *(const u32x4*)(lds + a0), c1_ = *(const u32x4*)(lds + a0 + 16);
		const u32x4 c2_ = *(const u32x4*)(lds + a1), c3_ = *(const u32x4*)(lds + a1 + 16);
#endif
		P[0] = pick_slot(c0_.y, c0_.x, e[0]); P[1] = pick_slot(c0_.w, c0_.z, e[1]);
		P[2] = pick_slot(c1_.y, c1_.x, e[2]); P[3] = pick_slot(c1_.w, c1_.z, e[3]);
		P[4] = pick_slot(c2_.y, c2_.x, e[4]); P[5] = pick_slot(c2_.w, c2_.z, e[5]);
		P[6] = pick_slot(c3_.y, c3_.x, e[6]); P[7] = pick_slot(c3_.w, c3_.z, e[7]);
	}
	if (OVERLAP)
	{
		const u32x4 c0_ = *(const u32x4*)(lds + u0), c1_ = *(const u32x4*)(lds + u0 + 16);
		const u32x4 c2_ = *(const u32x4*)(lds + u1), c3_ = *(const u32x4*)(lds + u1 + 16);
		int Q[8];
		Q[0] = pick_slot(c0_.y, c0_.x, e[0]); Q[1] = pick_slot(c0_.w, c0_.z, e[1]);
		Q[2] = pick_slot(c1_.y, c1_.x, e[2]); Q[3] = pick_slot(c1_.w, c1_.z, e[3]);
		Q[4] = pick_slot(c2_.y, c2_.x, e[4]); Q[5] = pick_slot(c2_.w, c2_.z, e[5]);
		Q[6] = pick_slot(c3_.y, c3_.x, e[6]); Q[7] = pick_slot(c3_.w, c3_.z, e[7]);
#pragma unroll
		for (int k = 0; k < 8; k++)
			P[k] = mad_vvv(Q[k], k < 4 ? n0 : n1, mad_vvv(P[k], k < 4 ? m0 : m1, 16)) >> 5;
	}

	// 3-tap filter across the block edge (vfgs_hw.c:250-259), on unfiltered neighbours
	// (all values are small: explicit 24-bit multiply-adds; the compiler otherwise reaches for 32/64-bit
	// multiplies and turns the selects into a branch that copies all eight P registers)
	if (EDGE16)
	{
		const int mine = first ? P[0] : P[7];
		const int inner = first ? P[1] : P[6];
		const int theirs = swap_lane_pairs(mine);
		int f = mad_vvv(rel, theirs, mad_vvv(3, mine, inner + c0)) >> 2;
		f = edge_on ? f : mine;
		P[0] = first ? f : P[0];
		P[7] = first ? P[7] : f;
	}
	else
	{
		const int l1 = P[2], l0 = P[3], r0 = P[4], r1 = P[5];
		const int fl = mad_vvv(rel, r0, mad_vvv(3, l0, l1 + c0)) >> 2;
		const int fr = mad_vvv(rel, l0, mad_vvv(3, r0, r1 + c1)) >> 2;
		P[3] = edge_on ? fl : l0;
		P[4] = edge_on ? fr : r0;
	}

	// scale, add, clip (vfgs_hw.c:263-267)
#pragma unroll
	for (int k = 0; k < 4; k++)
	{
		// round(scale * P, shift) (vfgs_hw.c:263) = (scale * 2^(16-shift) * P + 2^15) >> 16 exactly; the LUT holds
		// scale * 2^(16-shift), so the shift is free: the pack below simply takes the high halves
		const int g0 = mad_i24(P[2 * k], e[2 * k], 0x8000);
		const int g1 = mad_i24(P[2 * k + 1], e[2 * k + 1], 0x8000);
		const uint32_t gp = __builtin_amdgcn_perm((uint32_t)g1, (uint32_t)g0, 0x07060302);
		uint32_t v = w[k];
		if (DEPTH > 8)  // a 16-bit container may hold anything: keep the add inside int16 (result is clipped anyway)
			v = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, v), __builtin_bit_cast(u16x2, 0x70007000u)));
		s16x2 s = __builtin_bit_cast(s16x2, v) + __builtin_bit_cast(s16x2, gp);
		s = __builtin_elementwise_max(s, __builtin_bit_cast(s16x2, lo2));
		s = __builtin_elementwise_min(s, __builtin_bit_cast(s16x2, hi2));
		w[k] = __builtin_bit_cast(uint32_t, s);
	}
}

// ---------------------------------------------------------------------------------------
// Work items.
//
// A row of nblk grain blocks is cut into "units" of 8 samples: unit j covers samples
// [8j, 8j+8), j = 0 .. 2*nblk-1; even j = first half of block j/2, odd j = second half of block
// (j-1)/2.  The pair (odd j, j+1) straddles the edge between two blocks and always stays in one
// wave access ("segment"): segment s owns units [s*upt - 1, s*upt - 1 + upt), upt even, lane i
// <-> unit s*upt - 1 + i.  Planes with 8-sample blocks (subsampled chroma): one lane owns the 8
// samples around block edge m, i.e. the second half of block m-1 and the first half of block m;
// segment s owns edges [s*upt, s*upt + upt).
//
// One work item = ONE row of ONE plane x 4 consecutive segments (<= 4 KiB contiguous), owned by
// one wavefront.  A wave therefore streams from one plane in long runs (the 16-/8-byte shifted
// segment ends produce two partial cache lines per run instead of two per segment), the overlap
// decision (lines j = 0, 1 of a block row) is wave-uniform, and the three planes need no common
// tile geometry.  Items of a frame are numbered Y rows first, then Cb rows, then Cr rows.

// vfgs_hw.c:99-138 with the component as wave-uniform data: the x field is 10 bits at `sx`, the
// y field the low 10 bits of the register rotated right by `sy` (component 1 takes bits 31:24
// and 1:0 -- exactly a rotation by 24), the sign bit at `sb`.
template <int SUBX, int SUBY, int RS>
__device__ __forceinline__ BlockParam block_param_rt(uint32_t v, uint32_t bank_off, int sx, int sy, int sb)
{
	const uint32_t fx = (v >> sx) & 0x3ff;
	const uint32_t fy = __builtin_amdgcn_alignbit(v, v, sy) & 0x3ff;
	const uint32_t ox = (__umul24(fx, 13u) >> 10) * (4 / SUBX);
	const uint32_t oy = (__umul24(fy, 12u) >> 10) * (4 / SUBY);
	BlockParam r;
	r.addr = __umul24(oy, (uint32_t)RS) + ox * kSlots + bank_off;
	r.sign = ((v >> sb) & 1) ? -1 : 1;
	return r;
}

struct ItemDesc {   // all wave-uniform
	int f;       // frame of the batch
	int plane;   // 0 = Y, 1 = Cb, 2 = Cr
	int row;     // absolute row of that plane
	int tile;    // group of 4 segments along the row
};

// PHASE 0 issues the item's global loads (samples into w, one dword per lane of the LFSR stream
// slices of the item's block row into sw); PHASE 1 computes and stores.
template <int DEPTH, int BW, int SUBX, int SUBY, int RS, bool OUT8, bool SPLITC, int PHASE>
__device__ __forceinline__ void plane_item(const KernelArgs& a, uint8_t* lds, uint32_t scratch, const ItemDesc d, const int lane,
                                           uint32_t (&w)[4][4], uint32_t (&sw)[2])
{
	constexpr int SZ = DEPTH > 8 ? 2 : 1;
	constexpr bool SPLIT = SPLITC && (BW != 16);
	const bool luma = (d.plane == 0);
	const int nunits = 2 * a.nblk;
	const int last = a.nblk - 1;
	const int upt = (BW == 16) ? a.upt_y : a.upt_c;
	const int segs = (BW == 16) ? a.segs_y : a.segs_c;

	// wave-uniform row data
	const int y = d.row * SUBY;                        // luma line this row belongs to
	const int R = y >> 4;                              // block row
	const int jrow = y & 15;
	const int rloc = d.row & (16 / SUBY - 1);          // row inside the block row
	const bool has_up = (R > 0) && (jrow <= 1);        // vfgs_hw.c:175,180
	const int k = R - (a.y0 >> 4);                     // block row inside the stripe
	const uint32_t cur_bit = a.cur_bit0 + (uint32_t)d.f * a.frame_bit_step + (uint32_t)(k * a.nblk);
	const uint32_t up_bit = (k > 0) ? cur_bit - (uint32_t)a.nblk : a.up_bit0 + (uint32_t)d.f * a.frame_bit_step;
	const uint32_t cur_w0 = cur_bit >> 5, up_w0 = up_bit >> 5;

	const uint32_t pitch = (uint32_t)((luma ? a.stride : a.cstride) * SZ);
	const uint32_t dpitch = OUT8 ? (uint32_t)(luma ? a.dstride : a.dcstride) : pitch;
	const int prow0 = luma ? a.y0 : a.y0 / SUBY;       // row the plane pointers address
	const uint8_t* sbase = (luma ? a.Y : (d.plane == 1 ? a.U : a.V)) + (uint64_t)d.f * (luma ? a.y_frame_pitch : a.c_frame_pitch);
	uint8_t* dbase = (luma ? a.dY : (d.plane == 1 ? a.dU : a.dV)) + (uint64_t)d.f * (luma ? a.dy_frame_pitch : a.dc_frame_pitch);
	// num_records = exact extent of the plane's stripe: the hardware bounds-checks every access
	const __amdgpu_buffer_rsrc_t srs = make_rsrc(sbase, luma ? a.y_extent : a.c_extent);
	const __amdgpu_buffer_rsrc_t drs = make_rsrc(dbase, luma ? a.dy_extent : a.dc_extent);
	const uint32_t rowb = (uint32_t)(d.row - prow0) * pitch, drowb = (uint32_t)(d.row - prow0) * dpitch;

	// per-lane geometry of the 4 segments
	uint32_t vo0[4], vo1[4], do0[4], do1[4];
	int bl[4], br[4];            // blocks left / right of the lane's edge (BW == 16: both = the lane's block)
	bool first[4], edge[4];
#pragma unroll
	for (int g = 0; g < 4; g++)
	{
		const int seg = d.tile * 4 + g;
		const bool sok = (seg < segs) && (lane < upt);
		if (BW == 16)
		{
			const int ju = seg * upt - 1 + lane;
			const bool ok = sok && (ju >= 0) && (ju < nunits);
			first[g] = !(ju & 1);
			bl[g] = br[g] = min(max(ju >> 1, 0), last);
			const int jl = first[g] ? ju - 1 : ju;          // left unit of this lane pair
			edge[g] = sok && (jl >= 0) && (jl + 1 < nunits);
			vo0[g] = ok ? rowb + (uint32_t)(8 * ju * SZ) : kOOB;
			vo1[g] = 0;
			do0[g] = !OUT8 ? vo0[g] : (ok ? drowb + (uint32_t)(8 * ju) : kOOB);
			do1[g] = 0;
		}
		else
		{
			const int m = seg * upt + lane;                 // block edge index
			const bool in = sok && (m <= a.nblk);
			const bool h0 = in && (m - 1 >= 0), h1 = in && (m <= last);
			first[g] = false;
			bl[g] = min(max(m - 1, 0), last);
			br[g] = min(m, last);
			edge[g] = h0 && h1;
			const int xc0 = 8 * m - 4;
			vo0[g] = h0 ? rowb + (uint32_t)(xc0 * SZ) : kOOB;
			vo1[g] = h1 ? rowb + (uint32_t)((xc0 + 4) * SZ) : kOOB;
			if (!SPLIT && !(h0 && h1)) vo0[g] = kOOB;       // (cannot happen in a non-edge item)
			do0[g] = !OUT8 ? vo0[g] : (h0 ? drowb + (uint32_t)xc0 : kOOB);
			do1[g] = !OUT8 ? vo1[g] : (h1 ? drowb + (uint32_t)(xc0 + 4) : kOOB);
			if (OUT8 && !SPLIT && !(h0 && h1)) do0[g] = kOOB;
		}
	}

	if (PHASE == 0)
	{
#pragma unroll
		for (int g = 0; g < 4; g++)
			load_unit<DEPTH, SPLIT>(srs, vo0[g], vo1[g], 0, w[g]);
		// LFSR stream slices of this block row (and the one above): lane l takes dword w0 + l; the
		// windows of all blocks of the row lie inside the first (nblk + 63) / 32 dwords
#if VFGS_ABLATE != 12 && VFGS_ABLATE != 13
		sw[0] = a.stream[cur_w0 + lane];
		sw[1] = a.stream[up_w0 + lane];
#endif
		return;
	}

	// ---- block parameters: windows out of the stream slices (via a per-wave LDS scratch) ------
#if VFGS_ABLATE != 12 && VFGS_ABLATE != 13
	*(uint32_t*)(lds + scratch + 4 * lane) = sw[0];
	*(uint32_t*)(lds + scratch + 256 + 4 * lane) = sw[1];
#endif
	const int comp = d.plane;
	const int fsx = comp == 0 ? 0 : (comp == 1 ? 10 : 20);
	const int fsy = comp == 0 ? 14 : (comp == 1 ? 24 : 4);
	const int fsb = comp == 0 ? 31 : (comp == 1 ? 2 : 15);
	const uint32_t bank = luma ? (uint32_t)TableLayoutBase::LUMA_OFF : a.chroma_off;
	const uint32_t lutp = a.lut_off + (uint32_t)comp * 2048;
	const uint32_t lo2 = (uint32_t)(luma ? a.ylo : a.clo) * 0x10001u, hi2 = (uint32_t)(luma ? a.yhi : a.chi) * 0x10001u;
	const int half = 1 << (a.scale_shift - 1);
	const uint32_t rowoff = (uint32_t)rloc * RS;
	int wc = 32, wu_ = 0;                                   // vfgs_hw.c:173-188
	if (has_up) { if (jrow == 0) { wc = SUBY > 1 ? 20 : 12; wu_ = SUBY > 1 ? 20 : 24; } else { wc = 24; wu_ = 12; } }

	auto window = [&](uint32_t slot, uint32_t w0, uint32_t bit) {
		const uint32_t idx = (bit >> 5) - w0;
		const uint32_t lo = *(const uint32_t*)(lds + scratch + slot + 4 * idx);
		const uint32_t hi = *(const uint32_t*)(lds + scratch + slot + 4 * idx + 4);
		return __builtin_amdgcn_alignbit(hi, lo, bit & 31);
	};

#if VFGS_ABLATE == 7
	BlockParam keep0, keep1;
#endif
#pragma unroll
	for (int g = 0; g < 4; g++)
	{
		if (d.tile * 4 + g >= segs)
			break;                                          // wave-uniform
#if VFGS_ABLATE == 7   // timing only: block parameters of segment 0 reused for segments 1..3 (what sharing them would save)
		static_assert(true, "");
		BlockParam c0, c1;
		if (g == 0)
		{
			c0 = block_param_rt<SUBX, SUBY, RS>(window(0, cur_w0, cur_bit + bl[g]), bank, fsx, fsy, fsb);
			c1 = (BW == 16) ? c0 : block_param_rt<SUBX, SUBY, RS>(window(0, cur_w0, cur_bit + br[g]), bank, fsx, fsy, fsb);
			keep0 = c0; keep1 = c1;
		}
		else { c0 = keep0; c1 = keep1; }
#else
		const BlockParam c0 = block_param_rt<SUBX, SUBY, RS>(window(0, cur_w0, cur_bit + bl[g]), bank, fsx, fsy, fsb);
		const BlockParam c1 = (BW == 16) ? c0 : block_param_rt<SUBX, SUBY, RS>(window(0, cur_w0, cur_bit + br[g]), bank, fsx, fsy, fsb);
#endif
		uint32_t h0, h1;
		if (BW == 16) { h0 = first[g] ? 0u : 8u * kSlots; h1 = h0 + 4 * kSlots; }
		else { h0 = 4 * kSlots; h1 = 0; }                   // samples 4..7 of the left block, 0..3 of the right block
		const uint32_t a0 = c0.addr + rowoff + h0, a1 = c1.addr + rowoff + h1;
		if (has_up)
		{
			const BlockParam u0 = block_param_rt<SUBX, SUBY, RS>(window(256, up_w0, up_bit + bl[g]), bank, fsx, fsy, fsb);
			const BlockParam u1 = (BW == 16) ? u0 : block_param_rt<SUBX, SUBY, RS>(window(256, up_w0, up_bit + br[g]), bank, fsx, fsy, fsb);
			const uint32_t uoff = (16 / SUBY) * RS + rowoff;
			grain_unit<DEPTH, true, BW == 16>(lds, w[g], lutp, lutp, a0, a1, __mul24(c0.sign, wc), __mul24(c1.sign, wc),
			                                  u0.addr + uoff + h0, u1.addr + uoff + h1, __mul24(u0.sign, wu_), __mul24(u1.sign, wu_),
			                                  edge[g], first[g], 1, 2, 2, a.scale_shift, half, lo2, hi2);
		}
		else
		{
			const int rel = (BW == 16) ? __mul24(c0.sign, swap_lane_pairs(c0.sign)) : __mul24(c0.sign, c1.sign);
			grain_unit<DEPTH, false, BW == 16>(lds, w[g], lutp + (c0.sign < 0 ? 1024u : 0u), lutp + (c1.sign < 0 ? 1024u : 0u), a0, a1,
			                                   0, 0, 0, 0, 0, 0, edge[g], first[g], rel, c0.sign < 0 ? 1 : 2, c1.sign < 0 ? 1 : 2,
			                                   a.scale_shift, half, lo2, hi2);
		}
		if (OUT8) store_unit_narrow<SPLIT>(drs, do0[g], do1[g], w[g]);
		else      store_unit<DEPTH, SPLIT>(drs, do0[g], do1[g], 0, w[g]);
	}
}

template <int DEPTH, int CSUBX, int CSUBY, bool OUT8>
__global__ __launch_bounds__(kWavesPerWG * 64, (kWavesPerWG * kWGPerCU + 3) / 4) void grain_kernel(const KernelArgs a)
{
	using L = TableLayout<CSUBX, CSUBY>;
	constexpr int CBW = 16 / CSUBX;
	constexpr int kScratch = 512;      // per wave: two 256-byte LFSR stream slices

	__shared__ __attribute__((aligned(16))) uint8_t lds[L::BYTES + kWavesPerWG * kScratch];

	// stage banks + LUTs: global (L2 resident) -> LDS, 16 bytes per lane per step
#if VFGS_ABLATE != 6 && VFGS_ABLATE != 11 && VFGS_ABLATE != 13
	for (int i = threadIdx.x * 16; i < L::BYTES; i += kWavesPerWG * 64 * 16)
		*(u32x4*)(lds + i) = *(const u32x4*)(a.tables + i);
	__syncthreads();
#endif

	const int lane = threadIdx.x & 63;
	// wave-uniform by construction; telling the compiler keeps item decoding, row offsets and
	// buffer descriptors in SGPRs (otherwise every buffer instruction gets a waterfall loop)
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t scratch = L::BYTES + wave * kScratch;

	// persistent: the workgroups of one launch share the items round-robin, kWavesPerWG
	// consecutive items (neighbouring tiles / rows of one plane) per workgroup and step
	const int step = gridDim.x * kWavesPerWG;
	const int per_frame = a.items_y + 2 * a.items_c;
	auto decode = [&](int item) {
		ItemDesc d;
		d.f = item / per_frame;
		int r = item - d.f * per_frame;
		if (r < a.items_y) { d.plane = 0; d.row = a.y0 + r / a.tiles_y; d.tile = r % a.tiles_y; }
		else
		{
			r -= a.items_y;
			d.plane = 1 + (r >= a.items_c);
			if (r >= a.items_c) r -= a.items_c;
			d.row = a.crow_first + r / a.tiles_c;
			d.tile = r % a.tiles_c;
		}
		return d;
	};
	// a subsampled-chroma lane at the left / right picture edge owns only one valid half: items
	// that contain such a lane move chroma in two 8-byte halves (wave-uniform)
	auto run = [&](const ItemDesc d, auto phase, uint32_t (&w)[4][4], uint32_t (&sw)[2]) {
		constexpr int PH = decltype(phase)::value;
		if (d.plane == 0)
			plane_item<DEPTH, 16, 1, 1, L::LRS, OUT8, false, PH>(a, lds, scratch, d, lane, w, sw);
		else if (CBW == 16 || (d.tile != 0 && d.tile != a.tiles_c - 1))
			plane_item<DEPTH, CBW, CSUBX, CSUBY, L::CRS, OUT8, false, PH>(a, lds, scratch, d, lane, w, sw);
		else
			plane_item<DEPTH, CBW, CSUBX, CSUBY, L::CRS, OUT8, true, PH>(a, lds, scratch, d, lane, w, sw);
	};

	const std::integral_constant<int, 0> LOAD;
	const std::integral_constant<int, 1> COMP;
#if VFGS_PIPE
	// two register sets: the next item's loads are issued before the current item is computed
	uint32_t wa[4][4], swa[2], wb[4][4], swb[2];
	int item = blockIdx.x * kWavesPerWG + wave;
	if (item >= a.nitems)
		return;
	ItemDesc d = decode(item);
	run(d, LOAD, wa, swa);
	for (;;)
	{
		int next = item + step;
		ItemDesc dn = d;
		if (next < a.nitems) { dn = decode(next); run(dn, LOAD, wb, swb); }
		run(d, COMP, wa, swa);
		if (next >= a.nitems) break;
		item = next; d = dn;
		next = item + step;
		if (next < a.nitems) { dn = decode(next); run(dn, LOAD, wa, swa); }
		run(d, COMP, wb, swb);
		if (next >= a.nitems) break;
		item = next; d = dn;
	}
#else
	uint32_t w[4][4], sw[2];
	for (int item = blockIdx.x * kWavesPerWG + wave; item < a.nitems; item += step)
	{
		const ItemDesc d = decode(item);
		run(d, LOAD, w, sw);
		run(d, COMP, w, sw);
	}
#endif
}

// ---------------------------------------------------------------------------------------
// host-side launcher (called from vfgs_host.cpp)

template <int DEPTH, int CSUBX, int CSUBY, bool OUT8>
static hipError_t launch_t(const KernelArgs& a, int grid, hipStream_t stream)
{
	hipLaunchKernelGGL((grain_kernel<DEPTH, CSUBX, CSUBY, OUT8>), dim3(grid), dim3(kWavesPerWG * 64), 0, stream, a);
	return hipGetLastError();
}

hipError_t launch_grain(const KernelArgs& a, int depth, int csubx, int csuby, bool out8, int grid, hipStream_t stream)
{
#define VFGS_CASE(D, X, Y) if (depth == D && csubx == X && csuby == Y && !out8) return launch_t<D, X, Y, false>(a, grid, stream)
	VFGS_CASE(10, 2, 2); VFGS_CASE(10, 2, 1); VFGS_CASE(10, 1, 1); VFGS_CASE(10, 1, 2);
	VFGS_CASE(8, 2, 2);  VFGS_CASE(8, 2, 1);  VFGS_CASE(8, 1, 1);  VFGS_CASE(8, 1, 2);
#undef VFGS_CASE
#define VFGS_CASE8(X, Y) if (depth == 10 && csubx == X && csuby == Y && out8) return launch_t<10, X, Y, true>(a, grid, stream)
	VFGS_CASE8(2, 2); VFGS_CASE8(2, 1); VFGS_CASE8(1, 1); VFGS_CASE8(1, 2);
#undef VFGS_CASE8
	return hipErrorInvalidValue;
}

int table_bytes(int csubx, int csuby)
{
	if (csubx == 2 && csuby == 2) return TableLayout<2, 2>::BYTES;
	if (csubx == 2 && csuby == 1) return TableLayout<2, 1>::BYTES;
	if (csubx == 1 && csuby == 1) return TableLayout<1, 1>::BYTES;
	return TableLayout<1, 2>::BYTES;
}

}  // namespace vfgs
