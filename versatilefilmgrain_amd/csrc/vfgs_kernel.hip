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
// Work decomposition (see DESIGN.md "kernel"):
//   * one WAVEFRONT owns one work item = up to 32 grain blocks (<= 512 luma samples) x 4 luma
//     lines of Y, plus the co-located Cb/Cr samples; each grain block is served by a fixed
//     lane pair of that wavefront, which derives the block's LFSR window in registers;
//   * every lane moves 16 bytes (10-bit) / 8 bytes (8-bit) = 8 samples per access, so one
//     wave-instruction reads or writes one contiguous <= 1 KiB row segment;
//   * items are shifted by HALF A BLOCK (8 luma samples) against the block grid, so every
//     block edge -- the only place where a sample depends on its horizontal neighbours --
//     lies strictly inside an item: no halo, no inter-wave exchange, in-place is race free;
//   * pattern banks (slot-interleaved) and LUTs are staged once per workgroup in LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vfgs_layout.h"

namespace vfgs {

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
#if VFGS_ABLATE == 1
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
		e[2 * k]     = *(const uint32_t*)(lds + lut + (idx & 0xffffu));
		e[2 * k + 1] = *(const uint32_t*)(lds + lut + (idx >> 16));
	}

	// pattern fetch: 4 samples x 8 slots = 32 bytes per half
	{
		const u32x4 c0_ = *(const u32x4*)(lds + a0), c1_ = *(const u32x4*)(lds + a0 + 16);
		const u32x4 c2_ = *(const u32x4*)(lds + a1), c3_ = *(const u32x4*)(lds + a1 + 16);
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
			P[k] = mad24(Q[k], k < 4 ? n0 : n1, mad24(P[k], k < 4 ? m0 : m1, 16)) >> 5;
	}

	// 3-tap filter across the block edge (vfgs_hw.c:250-259), on unfiltered neighbours
	if (EDGE16)
	{
		const int mine = first ? P[0] : P[7];
		const int inner = first ? P[1] : P[6];
		const int theirs = swap_lane_pairs(mine);
		int f = (inner + 3 * mine + __mul24(rel, theirs) + c0) >> 2;
		f = edge_on ? f : mine;
		P[0] = first ? f : P[0];
		P[7] = first ? P[7] : f;
	}
	else
	{
		const int l1 = P[2], l0 = P[3], r0 = P[4], r1 = P[5];
		P[3] = edge_on ? ((l1 + 3 * l0 + __mul24(rel, r0) + c0) >> 2) : l0;
		P[4] = edge_on ? ((__mul24(rel, l0) + 3 * r0 + r1 + c1) >> 2) : r0;
	}

	// scale, add, clip (vfgs_hw.c:263-267)
#pragma unroll
	for (int k = 0; k < 4; k++)
	{
		const int g0 = mad_i16(P[2 * k], e[2 * k], half) >> scale_shift;
		const int g1 = mad_i16(P[2 * k + 1], e[2 * k + 1], half) >> scale_shift;
		const uint32_t gp = __builtin_amdgcn_perm((uint32_t)g1, (uint32_t)g0, 0x05040100);
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
// Geometry of one work item along x.
//
// A row of nblk grain blocks is cut into "units" of 8 samples: unit j covers luma samples
// [8j, 8j+8), j = 0 .. 2*nblk-1; even j = first half of block j/2, odd j = second half of block
// (j-1)/2.  The pair (odd j, j+1) straddles the edge between two blocks and always stays in
// one item: item tx owns units [tx*L - 1, tx*L - 1 + L), L even (a.upt), lane i <-> unit tx*L-1+i.
// Subsampled chroma (8-sample blocks): one lane owns the 8 chroma samples around block edge
// m, i.e. the second half of block m-1 and the first half of block m; 32 lanes per row.

// One work item (see "Geometry" above), in four phases so that the kernel can interleave two
// items: LOAD_Y / LOAD_C issue the global loads of the luma / chroma rows into registers,
// COMP_Y / COMP_C compute and store them.  The kernel's order per item is
//     LOAD_C(i), COMP_Y(i), LOAD_Y(i+1), COMP_C(i)
// so a wave always has loads in flight while it computes, with no more registers than one
// item's worth of data (the luma registers are refilled as soon as they have been stored).
// SPLITC = the item touches the left or right picture edge, where a subsampled-chroma lane can
// own only one valid half: those items move chroma in two 8-byte halves per lane, all others
// in one 16-byte access.
enum { LOAD_Y = 0, LOAD_C = 1, COMP_Y = 2, COMP_C = 3 };

template <int DEPTH, int CSUBX, int CSUBY, bool OUT8, bool SPLITC, int PHASE>
__device__ __forceinline__ void do_item(const KernelArgs& a, const uint8_t* lds, const int item, const int lane,
                                        uint32_t (&wy)[4][4], uint32_t (&wu)[(4 / CSUBY) / ((CSUBX == 1) ? 1 : 2)][4],
                                        uint32_t (&wv)[(4 / CSUBY) / ((CSUBX == 1) ? 1 : 2)][4],
                                        uint32_t (&sy)[2][2], uint32_t (&sc)[4][2])
{
	using L = TableLayout<CSUBX, CSUBY>;
	constexpr int SZ = DEPTH > 8 ? 2 : 1;
	constexpr int CBW = 16 / CSUBX;               // chroma block width in samples
	constexpr int CROWS = 4 / CSUBY;              // chroma rows per item (4 luma lines)
	constexpr int CRPL = (CBW == 16) ? 1 : 2;     // chroma rows per wave access
	constexpr int CNL = CROWS / CRPL;             // chroma accesses per plane per item
	constexpr uint32_t LUTY = L::LUT_OFF, LUTU = L::LUT_OFF + 2048, LUTV = L::LUT_OFF + 4096;   // [+scale | -scale] each
	constexpr bool SPLIT = SPLITC && (CBW != 16);
	constexpr bool LUMA = (PHASE == LOAD_Y || PHASE == COMP_Y);
	constexpr bool COMP = (PHASE == COMP_Y || PHASE == COMP_C);

	const int nunits = 2 * a.nblk;
	const int last = a.nblk - 1;
	const int half = 1 << (a.scale_shift - 1);

	// item -> (frame f, block row k of the stripe, line quad p, tile tx); tx fastest
	int t = item;
	const int tx = t % a.ntx; t /= a.ntx;
	const int p = t & 3;      t >>= 2;
	const int k = t % a.nbr;
	const int f = t / a.nbr;
	const int R = (a.y0 >> 4) + k;                // absolute block row (y >> 4)
	const bool has_up = (R > 0) && (p == 0);      // lines j = 0, 1 of a block row below the first (vfgs_hw.c:175,180)

	// whole item outside the stripe (only for stripes that are not multiples of 16 lines)?
	if (COMP && (16 * R + 4 * p + 3 < a.y0 || 16 * R + 4 * p >= a.y0 + a.nlines))
		return;

	const uint32_t yrow = (uint32_t)(a.stride * SZ), crow = (uint32_t)(a.cstride * SZ);
	const int j0 = tx * a.upt - 1;                // first unit of this item
	const int ju = j0 + lane;                     // this lane's unit (luma, and chroma when CBW == 16)
	const bool l_ok = (lane < a.upt) && (ju >= 0) && (ju < nunits);
	const bool l_first = !(ju & 1);               // first half of its block
	const uint32_t cur_bit = a.cur_bit0 + (uint32_t)f * a.frame_bit_step + (uint32_t)(k * a.nblk);
	const uint32_t up_bit = (k > 0) ? cur_bit - (uint32_t)a.nblk : a.up_bit0 + (uint32_t)f * a.frame_bit_step;
	const int yblk = min(max(ju >> 1, 0), last);

	if (LUMA)
	{
		// one descriptor per plane of this frame's stripe; num_records = its exact extent, so the
		// hardware bounds-checks every access of the item
		const __amdgpu_buffer_rsrc_t sY = make_rsrc(a.Y + (uint64_t)f * a.y_frame_pitch, a.y_extent);
		const __amdgpu_buffer_rsrc_t dY = make_rsrc(a.dY + (uint64_t)f * a.dy_frame_pitch, a.dy_extent);
		uint32_t voy[4], dvo[4];
#pragma unroll
		for (int r = 0; r < 4; r++)
		{
			const int yabs = 16 * R + 4 * p + r;
			const bool rok = (yabs >= a.y0) && (yabs < a.y0 + a.nlines);          // wave-uniform
			// a line outside the stripe is computed but neither read nor written (all lanes out of range)
			voy[r] = (rok && l_ok) ? (uint32_t)(yabs - a.y0) * yrow + (uint32_t)(8 * ju * SZ) : kOOB;
			dvo[r] = !OUT8 ? voy[r] : ((rok && l_ok) ? (uint32_t)(yabs - a.y0) * (uint32_t)a.dstride + (uint32_t)(8 * ju) : kOOB);
			if (PHASE == LOAD_Y) load_unit<DEPTH, false>(sY, voy[r], 0, 0, wy[r]);
		}
		if (PHASE == LOAD_Y)
		{
			stream_fetch(a.stream, cur_bit + yblk, sy[0]);
			if (has_up) stream_fetch(a.stream, up_bit + yblk, sy[1]);
			return;
		}

		const uint32_t ylo2 = (uint32_t)a.ylo * 0x10001u, yhi2 = (uint32_t)a.yhi * 0x10001u;
		const uint32_t vy = stream_cut(sy[0], cur_bit + yblk);
		const BlockParam ycur = block_param<0, 1, 1, L::LRS>(vy, L::LUMA_OFF);
		// edge between this lane pair: left unit must exist (>= 0), right unit must exist (< nunits)
		const int jl = l_first ? ju - 1 : ju;
		const bool edge_on = (lane < a.upt) && (jl >= 0) && (jl + 1 < nunits);
		const uint32_t hoff = l_first ? 0u : 8u * kSlots;
		const uint32_t base = ycur.addr + hoff;
		if (has_up)   // wave-uniform: lines 0 and 1 blend with the block row above
		{
			const BlockParam yup = block_param<0, 1, 1, L::LRS>(stream_cut(sy[1], up_bit + yblk), L::LUMA_OFF);
			const uint32_t ubase = yup.addr + 16u * L::LRS + hoff;
#pragma unroll
			for (int r = 0; r < 2; r++)
			{
				const uint32_t rowoff = (uint32_t)(4 * p + r) * L::LRS;   // scalar
				const int wc = (r == 0) ? 12 : 24, wu_ = (r == 0) ? 24 : 12;   // vfgs_hw.c:177-183, suby == 1
				grain_unit<DEPTH, true, true>(lds, wy[r], LUTY, LUTY, base + rowoff, base + rowoff + 4 * kSlots,
				                              ycur.sign * wc, ycur.sign * wc,
				                              ubase + rowoff, ubase + rowoff + 4 * kSlots, yup.sign * wu_, yup.sign * wu_,
				                              edge_on, l_first, 1, 2, 2, a.scale_shift, half, ylo2, yhi2);
				if (OUT8) store_unit_narrow<false>(dY, dvo[r], 0, wy[r]);
				else      store_unit<DEPTH, false>(dY, dvo[r], 0, 0, wy[r]);
			}
		}
		const int rel = __mul24(ycur.sign, swap_lane_pairs(ycur.sign));    // relative sign of the two blocks at the edge
		const uint32_t luts = LUTY + (ycur.sign < 0 ? 1024u : 0u);
		const int cs = ycur.sign < 0 ? 1 : 2;
#pragma unroll
		for (int r = 0; r < 4; r++)
		{
			if (r < 2 && has_up)
				continue;
			const uint32_t rowoff = (uint32_t)(4 * p + r) * L::LRS;   // scalar
			grain_unit<DEPTH, false, true>(lds, wy[r], luts, luts, base + rowoff, base + rowoff + 4 * kSlots,
			                               0, 0, 0, 0, 0, 0,
			                               edge_on, l_first, rel, cs, cs, a.scale_shift, half, ylo2, yhi2);
			if (OUT8) store_unit_narrow<false>(dY, dvo[r], 0, wy[r]);
			else      store_unit<DEPTH, false>(dY, dvo[r], 0, 0, wy[r]);
		}
	}
	else
	{
		const __amdgpu_buffer_rsrc_t sU = make_rsrc(a.U + (uint64_t)f * a.c_frame_pitch, a.c_extent);
		const __amdgpu_buffer_rsrc_t sV = make_rsrc(a.V + (uint64_t)f * a.c_frame_pitch, a.c_extent);
		const __amdgpu_buffer_rsrc_t dU = make_rsrc(a.dU + (uint64_t)f * a.dc_frame_pitch, a.dc_extent);
		const __amdgpu_buffer_rsrc_t dV = make_rsrc(a.dV + (uint64_t)f * a.dc_frame_pitch, a.dc_extent);
		uint32_t cv0[CNL], cv1[CNL], dc0[CNL], dc1[CNL];
		int crl[CNL];                                  // chroma row inside the block row, per access
		int cm;                                        // CBW == 8: block edge index m; CBW == 16: unit index
		bool c_first = false;
		bool h0, h1;
		int xc0;
		if (CBW == 16)
		{
			cm = ju;
			h0 = h1 = l_ok;
			c_first = l_first;
			xc0 = 8 * ju;
		}
		else
		{
			const int cu = lane & 31;
			cm = (j0 + 1) / 2 + cu;
			const bool in = cu < a.upt / 2;
			h0 = in && (cm - 1 >= 0) && (cm - 1 <= last);
			h1 = in && (cm <= last);
			xc0 = 8 * cm - 4;
		}
#pragma unroll
		for (int c = 0; c < CNL; c++)
		{
			crl[c] = CROWS * p + CRPL * c + ((CBW == 16) ? 0 : (lane >> 5));
			const int prow = R * (16 / CSUBY) + crl[c];          // absolute chroma row
			const int yabs = prow * CSUBY;
			const bool rok = (yabs >= a.y0) && (yabs < a.y0 + a.nlines);
			const uint32_t rowb = (uint32_t)(prow - a.y0 / CSUBY) * crow;
			cv0[c] = (rok && h0) ? rowb + (uint32_t)(xc0 * SZ) : kOOB;
			cv1[c] = (rok && h1) ? rowb + (uint32_t)((xc0 + 4) * SZ) : kOOB;
			const uint32_t drowb = (uint32_t)(prow - a.y0 / CSUBY) * (uint32_t)a.dcstride;
			dc0[c] = !OUT8 ? cv0[c] : ((rok && h0) ? drowb + (uint32_t)xc0 : kOOB);
			dc1[c] = !OUT8 ? cv1[c] : ((rok && h1) ? drowb + (uint32_t)(xc0 + 4) : kOOB);
			if (PHASE == LOAD_C)
			{
				load_unit<DEPTH, SPLIT>(sU, cv0[c], cv1[c], 0, wu[c]);
				load_unit<DEPTH, SPLIT>(sV, cv0[c], cv1[c], 0, wv[c]);
			}
		}
		int cbl, cbr;                                  // blocks left / right of the lane's edge
		if (CBW == 16) { cbl = cbr = yblk; }
		else { cbl = min(max(cm - 1, 0), last); cbr = min(cm, last); }
		if (PHASE == LOAD_C)
		{
			stream_fetch(a.stream, cur_bit + cbl, sc[0]);
			if (CBW != 16) stream_fetch(a.stream, cur_bit + cbr, sc[1]);
			if (has_up)
			{
				stream_fetch(a.stream, up_bit + cbl, sc[2]);
				if (CBW != 16) stream_fetch(a.stream, up_bit + cbr, sc[3]);
			}
			return;
		}

		const uint32_t clo2 = (uint32_t)a.clo * 0x10001u, chi2 = (uint32_t)a.chi * 0x10001u;
		const uint32_t vcl = stream_cut(sc[0], cur_bit + cbl);
		const uint32_t vcr = (CBW == 16) ? vcl : stream_cut(sc[1], cur_bit + cbr);
		const BlockParam ucur0 = block_param<1, CSUBX, CSUBY, L::CRS>(vcl, L::CHROMA_OFF);
		const BlockParam vcur0 = block_param<2, CSUBX, CSUBY, L::CRS>(vcl, L::CHROMA_OFF);
		BlockParam ucur1 = ucur0, vcur1 = vcur0;
		if (CBW != 16)
		{
			ucur1 = block_param<1, CSUBX, CSUBY, L::CRS>(vcr, L::CHROMA_OFF);
			vcur1 = block_param<2, CSUBX, CSUBY, L::CRS>(vcr, L::CHROMA_OFF);
		}
		bool edge_on;
		uint32_t g0, g1;
		int relu, relv;
		if (CBW == 16)
		{
			const int jl = c_first ? ju - 1 : ju;
			edge_on = (lane < a.upt) && (jl >= 0) && (jl + 1 < nunits);
			g0 = c_first ? 0u : 8u * kSlots;
			g1 = g0 + 4 * kSlots;
			relu = __mul24(ucur0.sign, swap_lane_pairs(ucur0.sign));
			relv = __mul24(vcur0.sign, swap_lane_pairs(vcur0.sign));
		}
		else
		{
			edge_on = ((lane & 31) < a.upt / 2) && (cm - 1 >= 0) && (cm <= last);
			g0 = 4 * kSlots;   // samples 4..7 of the left block
			g1 = 0;            // samples 0..3 of the right block
			relu = __mul24(ucur0.sign, ucur1.sign);
			relv = __mul24(vcur0.sign, vcur1.sign);
		}
		const uint32_t lutu0 = LUTU + (ucur0.sign < 0 ? 1024u : 0u), lutu1 = LUTU + (ucur1.sign < 0 ? 1024u : 0u);
		const uint32_t lutv0 = LUTV + (vcur0.sign < 0 ? 1024u : 0u), lutv1 = LUTV + (vcur1.sign < 0 ? 1024u : 0u);
		const int ru0 = ucur0.sign < 0 ? 1 : 2, ru1 = ucur1.sign < 0 ? 1 : 2;
		const int rv0 = vcur0.sign < 0 ? 1 : 2, rv1 = vcur1.sign < 0 ? 1 : 2;
#pragma unroll
		for (int c = 0; c < CNL; c++)
		{
			const uint32_t rowoff = __umul24((uint32_t)crl[c], (uint32_t)L::CRS);
			const uint32_t ua0 = ucur0.addr + rowoff + g0, ua1 = ucur1.addr + rowoff + g1;
			const uint32_t va0 = vcur0.addr + rowoff + g0, va1 = vcur1.addr + rowoff + g1;
			// only accesses whose first row is line j = row * CSUBY <= 1 of the block row can hold overlap lines
			if (has_up && (CRPL * c * CSUBY <= 1))
			{
				const uint32_t wl = stream_cut(sc[2], up_bit + cbl);
				const uint32_t wr = (CBW == 16) ? wl : stream_cut(sc[3], up_bit + cbr);
				const BlockParam uup0 = block_param<1, CSUBX, CSUBY, L::CRS>(wl, L::CHROMA_OFF);
				const BlockParam vup0 = block_param<2, CSUBX, CSUBY, L::CRS>(wl, L::CHROMA_OFF);
				const BlockParam uup1 = (CBW == 16) ? uup0 : block_param<1, CSUBX, CSUBY, L::CRS>(wr, L::CHROMA_OFF);
				const BlockParam vup1 = (CBW == 16) ? vup0 : block_param<2, CSUBX, CSUBY, L::CRS>(wr, L::CHROMA_OFF);
				const int jj = crl[c] * CSUBY;
				int wc = 32, wu_ = 0;
				if (jj == 0) { wc = CSUBY > 1 ? 20 : 12; wu_ = CSUBY > 1 ? 20 : 24; }
				else if (jj == 1) { wc = 24; wu_ = 12; }
				const uint32_t uoff = (16 / CSUBY) * L::CRS + rowoff;
				// lanes without overlap must not read past the bank: point their (unused) read at the current row
				const uint32_t uu0 = wu_ ? uup0.addr + uoff + g0 : ua0, uu1 = wu_ ? uup1.addr + uoff + g1 : ua1;
				const uint32_t vu0 = wu_ ? vup0.addr + uoff + g0 : va0, vu1 = wu_ ? vup1.addr + uoff + g1 : va1;
				grain_unit<DEPTH, true, CBW == 16>(lds, wu[c], LUTU, LUTU, ua0, ua1, __mul24(ucur0.sign, wc), __mul24(ucur1.sign, wc),
				                                   uu0, uu1, __mul24(uup0.sign, wu_), __mul24(uup1.sign, wu_),
				                                   edge_on, c_first, 1, 2, 2, a.scale_shift, half, clo2, chi2);
				grain_unit<DEPTH, true, CBW == 16>(lds, wv[c], LUTV, LUTV, va0, va1, __mul24(vcur0.sign, wc), __mul24(vcur1.sign, wc),
				                                   vu0, vu1, __mul24(vup0.sign, wu_), __mul24(vup1.sign, wu_),
				                                   edge_on, c_first, 1, 2, 2, a.scale_shift, half, clo2, chi2);
			}
			else
			{
				grain_unit<DEPTH, false, CBW == 16>(lds, wu[c], lutu0, lutu1, ua0, ua1, 0, 0, 0, 0, 0, 0,
				                                    edge_on, c_first, relu, ru0, ru1, a.scale_shift, half, clo2, chi2);
				grain_unit<DEPTH, false, CBW == 16>(lds, wv[c], lutv0, lutv1, va0, va1, 0, 0, 0, 0, 0, 0,
				                                    edge_on, c_first, relv, rv0, rv1, a.scale_shift, half, clo2, chi2);
			}
			if (OUT8)
			{
				store_unit_narrow<SPLIT>(dU, dc0[c], dc1[c], wu[c]);
				store_unit_narrow<SPLIT>(dV, dc0[c], dc1[c], wv[c]);
			}
			else
			{
				store_unit<DEPTH, SPLIT>(dU, dc0[c], dc1[c], 0, wu[c]);
				store_unit<DEPTH, SPLIT>(dV, dc0[c], dc1[c], 0, wv[c]);
			}
		}
	}
}

template <int DEPTH, int CSUBX, int CSUBY, bool OUT8>
__global__ __launch_bounds__(kWavesPerWG * 64, (kWavesPerWG * kWGPerCU + 3) / 4) void grain_kernel(const KernelArgs a)
{
	using L = TableLayout<CSUBX, CSUBY>;
	constexpr int CNL_ = (4 / CSUBY) / ((CSUBX == 1) ? 1 : 2);

	__shared__ __attribute__((aligned(16))) uint8_t lds[L::BYTES];

	// stage banks + LUTs: global (L2 resident) -> LDS, 16 bytes per lane per step
#if VFGS_ABLATE != 6
	for (int i = threadIdx.x * 16; i < L::BYTES; i += kWavesPerWG * 64 * 16)
		*(u32x4*)(lds + i) = *(const u32x4*)(a.tables + i);
	__syncthreads();
#endif

	const int lane = threadIdx.x & 63;
	// wave-uniform by construction; telling the compiler keeps item decoding, row offsets and
	// buffer descriptors in SGPRs (otherwise every buffer instruction gets a waterfall loop)
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

	// persistent: the workgroups of one launch share the items round-robin, kWavesPerWG
	// consecutive items (neighbouring tiles of one line quad) per workgroup and step.
	// A lane of the first tile of a row has no block to its left; a tile whose last block edge
	// index exceeds the last block has lanes with no block to their right: those tiles take the
	// SPLIT form of do_item (wave-uniform test).
#if VFGS_CHUNKED
	// each workgroup walks one contiguous run of items (tiles along a row, then the next rows)
	const int step = kWavesPerWG;
	const int per_wg = ((a.nitems + (int)gridDim.x - 1) / (int)gridDim.x + kWavesPerWG - 1) / kWavesPerWG * kWavesPerWG;
	int item = blockIdx.x * per_wg + wave;
	const int item_end = min(a.nitems, (int)(blockIdx.x + 1) * per_wg);
#define VFGS_NITEMS item_end
#else
	const int step = gridDim.x * kWavesPerWG;
	int item = blockIdx.x * kWavesPerWG + wave;
#define VFGS_NITEMS a.nitems
#endif
	if (item >= VFGS_NITEMS)
		return;
	auto is_split = [&](int it) { const int tx = it % a.ntx; return tx == 0 || (tx + 1) * (a.upt / 2) > a.nblk; };

	uint32_t wy[4][4], wu[CNL_][4], wv[CNL_][4];
	uint32_t sy[2][2], sc[4][2];      // raw LFSR stream dwords of the item's blocks (luma / chroma; current, upper row)
#define VFGS_PHASE(PH, IT)                                                                                      \
	do {                                                                                                        \
		if (is_split(IT)) do_item<DEPTH, CSUBX, CSUBY, OUT8, true, PH>(a, lds, IT, lane, wy, wu, wv, sy, sc);  \
		else              do_item<DEPTH, CSUBX, CSUBY, OUT8, false, PH>(a, lds, IT, lane, wy, wu, wv, sy, sc); \
	} while (0)

#if VFGS_PIPE
	VFGS_PHASE(LOAD_Y, item);
	for (;;)
	{
		const int next = item + step;
		VFGS_PHASE(LOAD_C, item);
		VFGS_PHASE(COMP_Y, item);
		if (next < VFGS_NITEMS)
			VFGS_PHASE(LOAD_Y, next);      // the luma registers are free again: refill them now
		VFGS_PHASE(COMP_C, item);
		if (next >= VFGS_NITEMS)
			break;
		item = next;
	}
#else
	for (; item < VFGS_NITEMS; item += step)
	{
		VFGS_PHASE(LOAD_Y, item);
		VFGS_PHASE(LOAD_C, item);
		VFGS_PHASE(COMP_Y, item);
		VFGS_PHASE(COMP_C, item);
	}
#endif
#undef VFGS_PHASE
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
