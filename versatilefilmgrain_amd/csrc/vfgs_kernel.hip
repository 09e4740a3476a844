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
// Work decomposition ("row walk"; one kernel family since round 4):
//   * every lane moves 16 bytes per access (8 samples at 10 bit, 16 samples at 8 bit), one wave access = one line-aligned
//     1 KiB "position" of a row;
//   * the lanes COMPUTE bytes shifted left of the block grid by half a block, so every block edge -- the only place where a
//     sample depends on its horizontal neighbours -- lies inside a lane or between the two lanes of a pair: no halo, no
//     inter-wave exchange, in place is race free; the shift between what a lane moves and what it computes is a rotation by
//     one lane in registers (DPP);
//   * a WAVEFRONT owns whole rows and streams them through a ring of four register sets; a WORKGROUP of 4 waves covers a few
//     consecutive rows of ONE block row, is NOT persistent, copies only its plane's banks + LUTs to LDS and computes the block
//     parameters of its block row (LFSR window -> sign, pattern offsets) once, into LDS; workgroups are numbered in memory
//     order, so the chip sweeps the frames front to back and the hardware dispatcher balances the load;
//   * a component whose pattern LUT selects one slot for every intensity is served from a packed one-byte-per-sample bank
//     (ONEY / ONEC kernels, vfgs_layout.h);
//   * rows of more than 512 grain blocks (8192 luma samples) are walked in parts of 512 blocks, the parameter table refilled
//     between the parts; the 8-bit output of a 10-bit path (yuv.c:216-258) is a narrowing in the store (OUT8 kernels);
//   * the frames of a launch lie at a constant pitch behind the plane pointers of the arguments, or anywhere: then their plane
//     pointers are a table IN the kernel arguments (FrameTable, vfgs_layout.h) and a workgroup reads its frame's with scalar loads.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <type_traits>

#include "vfgs_layout.h"

namespace vfgs {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------
// small device helpers
//
// Issue rates measured on MI355X (tools/valu_rate.hip): plain VOP2 integer ops (add, and,
// shifts) and f32 FMA take ~2.4 cycles per wave-instruction; integer VOP3-only, packed-16, SDWA
// and DPP forms (v_perm_b32, v_mad_*, v_pk_*, v_bfe_*, v_add3, v_and_or) take ~4.4.

// One sample's pattern value out of its 8-byte slot group {hi,lo}: the LUT entry's top byte is
// the v_perm_b32 selector (slot 0..7, or 0x0c = constant 0) for result byte 3; the arithmetic
// shift then sign-extends it (the other three result bytes are don't-care).
__device__ __forceinline__ int pick_slot(uint32_t hi, uint32_t lo, uint32_t lut_entry)
{
	return (int)__builtin_amdgcn_perm(hi, lo, lut_entry) >> 24;
}

// x * y.i24 + c: the LUT entry's low 24 bits are the signed scale, pre-shifted so that the product's
// HIGH half is the scaled grain (see grain_unit); the selector byte above them is ignored by the i24 multiply
__device__ __forceinline__ int mad_i24(int x, uint32_t y, int c)
{
	int r;
	asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "s"(c));
	return r;
}

// x * y + c on 24-bit operands, all in VGPRs (or inline constants): the compiler would otherwise pick
// 32-bit multiplies for these
__device__ __forceinline__ int mad_vvv(int x, int y, int c)
{
	int r;
	asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(c));
	return r;
}

// sext(byte B of d) * y.i24: the one-pattern bank holds four pattern bytes per dword; SDWA selects and sign-extends one of them inside
// the multiply (VOP2), which saves the separate extraction where the value itself is not needed (everywhere but at block edges)
__device__ __forceinline__ int mul_byte_i24(const int b, uint32_t d, uint32_t y)       // (b is a constant after unrolling: one case survives)
{
	int r;
	switch (b)
	{
	case 0: asm("v_mul_i32_i24_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(d), "v"(y)); break;
	case 1: asm("v_mul_i32_i24_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(d), "v"(y)); break;
	case 2: asm("v_mul_i32_i24_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(d), "v"(y)); break;
	default: asm("v_mul_i32_i24_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(d), "v"(y)); break;
	}
	return r;
}

// ---- packed 16-bit helpers (8-bit one-pattern form, vfgs_layout.h "packed 16-bit form") ----
// two samples per instruction: {a.lo * b.lo + c.lo, a.hi * b.hi + c.hi} in 16 bits (the host has proven that nothing overflows)
__device__ __forceinline__ uint32_t pk_mad_i16(uint32_t a, uint32_t b, uint32_t c)
{
	uint32_t r;
	asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
	return r;
}
// both halves shifted right arithmetically by the corresponding half of `sh`
__device__ __forceinline__ uint32_t pk_ashr_i16(uint32_t sh, uint32_t v)
{
	uint32_t r;
	asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(r) : "s"(sh), "v"(v));
	return r;
}
// x.i16 * (HI ? y.hi : y.lo).i16 + c in 32 bits: a block edge's filtered value times its sample's scale, which sits in one half
// of the pair's register
template <bool HI>
__device__ __forceinline__ int mad_i32_i16(int x, uint32_t y, int c)
{
	int r;
	if (HI) asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[0,1,0,0]" : "=v"(r) : "v"(x), "v"(y), "s"(c));
	else asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "s"(c));
	return r;
}
__device__ __forceinline__ int dot2_i16(uint32_t a, uint32_t b, int c)
{
	return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), c, false);
}
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b)      // mask ? a : b, bit by bit
{
	uint32_t r;
	asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
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
// "load returns 0 / store is dropped".  No exec-mask branches around loads and stores, and a
// whole access can be switched off by a descriptor with zero records (wave-uniform).
constexpr uint32_t kOOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const uint8_t* base, uint32_t bytes)
{
	return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}

// ---------------------------------------------------------------------------------------
// Lanes.
//
// A lane holds 16 bytes of one row = NS samples (8 at 10 bit, 16 at 8 bit) as 4 dwords; lane
// position p of a row covers samples [NS*p - SHIFT, NS*p - SHIFT + NS).  The lane's samples fall
// into NR "runs" that belong to consecutive grain blocks, and into NQ = NS/4 "quads" of 4 samples
// (a quad never straddles a block; its pattern data is two ds_read_b128):
//
//   PAIR  (NS 8, BW 16): the lane is half a block: p - 1 = unit index, block = (p - 1) >> 1; odd p =
//                        first half.  The block edge lies between lanes 2m (last sample in slot 7)
//                        and 2m + 1 (first sample in slot 0): one DPP quad_perm swap.
//   NS 8,  BW 8        : runs = second half of block p-1 | first half of block p; edge between samples 3|4
//   NS 16, BW 16       : the same with 8-sample halves; edge between samples 7|8
//   NS 16, BW 8        : runs = last 4 of block 2p-1 | block 2p | first 4 of block 2p+1; edges 3|4 and 11|12
template <int NS, int BW>
struct LaneMap {
	static constexpr bool PAIR = (NS == 8 && BW == 16);
	static constexpr int SHIFT = PAIR ? 8 : BW / 2;                 // samples
	static constexpr int NR = PAIR ? 1 : NS / BW + 1;
	static constexpr int NQ = NS / 4;
	static constexpr int NE = NR - 1;                               // block edges inside the lane
	static constexpr int BPL = PAIR ? 1 : NS / BW;                  // whole blocks a lane advances by
	__device__ static constexpr int run(int q) { return PAIR ? 0 : (4 * q + BW - SHIFT) / BW; }
	__device__ static constexpr int col(int q) { return PAIR ? 4 * q : (4 * q + BW - SHIFT) % BW; }   // column of the quad's first sample in its block
	__device__ static constexpr int edge_quad(int e) { return PAIR ? 0 : ((e + 1) * BW - (BW - SHIFT)) / 4 - 1; }   // quad left of inner edge e
};

// A block's parameters in ONE dword per run (computed once per workgroup and task into the LDS parameter table; a lane reads
// the 1-3 entries of its runs per position and rebuilds k2, the rounding constants and the relative signs of the edge filter
// from them with a handful of instructions):
//   bits 15:0  LDS byte address of bank[.][oy][ox (+ the lane's column for PAIR)][slot 0], row 0 of the block row; one-pattern
//              form, current row of blocks: of the NEGATED bank if the block's sign is negative (grain_unit)
//   bit  31    the block's sign is negative
template <int NR>
struct RunParam {
	uint32_t pa[NR];
};

// vfgs_hw.c:99-138 with the component as wave-uniform data: the x field is 10 bits at `sx`, the
// y field the low 10 bits of the register rotated right by `sy` (component 1 takes bits 31:24
// and 1:0 -- exactly a rotation by 24), the sign bit at `sb`.  Returns the LDS address; sign in *neg.
template <int SUBX, int SUBY, int RS, int SB>
__device__ __forceinline__ uint32_t block_param(uint32_t v, uint32_t bank_off, int sx, int sy, int sb, bool* neg)
{
	const uint32_t fx = (v >> sx) & 0x3ff;
	const uint32_t fy = __builtin_amdgcn_alignbit(v, v, sy) & 0x3ff;
	const uint32_t ox = (__umul24(fx, 13u) >> 10) * (4 / SUBX);
	const uint32_t oy = (__umul24(fy, 12u) >> 10) * (4 / SUBY);
	*neg = (v >> sb) & 1;
	return __umul24(oy, (uint32_t)RS) + ox * SB + bank_off;     // SB = bytes per sample position: 1 (8-bit one-pattern form: 2) or one per slot
}

// ---------------------------------------------------------------------------------------
// The per-lane grain pipeline for the NS samples of one row.
//
//   w          in/out: samples, 4 dwords (10 bit: 2 x uint16 each; 8 bit: 4 x uint8 each)
//   rp         block parameters of the lane's runs; rowoff = (row inside the block row) * RS
//
// Two forms of the same arithmetic (vfgs_hw.c:211-229, 250-267):
//
// OVERLAP = true (the two lines under a block-row boundary): pattern values are blended as
//   P = (Pcur * m + Pup * n + 16) >> 5 with m = sign_cur * w_cur, n = sign_up * w_up, so P is
//   the signed grain; the +scale tables are used; rel = 1, c = 2.
//
// OVERLAP = false: the block sign s is folded into the scale instead of the pattern value:
//   k2 points at the table of sign * scale, P~ = s * P is used unsigned-by-sign, and
//   round(scale * P, shift) == (s*scale) * P~ ... exactly (s*s == 1).  The 3-tap edge filter
//   F = (l1 + 3 l0 + r0 + 2) >> 2 on true values becomes, in the P~ domain of the lane whose
//   sample is filtered,  F~ = (a~ + 3 b~ + rel * c~ + (s > 0 ? 2 : 1)) >> 2  with rel = s * s'
//   the relative sign of the two blocks: for s = -1, -((-A + 2) >> 2) == (A + 1) >> 2.
template <int DEPTH, int BW, bool OVERLAP, bool ONE, bool ALIGN2, int NEG>
__device__ __forceinline__ void grain_unit(const uint8_t* lds, uint32_t (&w)[4],
                                            const RunParam<LaneMap<DEPTH == 8 ? 16 : 8, BW>::NR>& rp,
                                            const RunParam<LaneMap<DEPTH == 8 ? 16 : 8, BW>::NR>& up,
                                            const uint32_t lutb, const uint32_t rowoff, const uint32_t uprowoff, const int wcur, const int wup,
                                            const bool (&edge_on)[LaneMap<DEPTH == 8 ? 16 : 8, BW>::PAIR ? 1 : LaneMap<DEPTH == 8 ? 16 : 8, BW>::NE],
                                            const bool first, const uint32_t lo2, const uint32_t hi2, const int pkshift)
{
	constexpr int NS = DEPTH == 8 ? 16 : 8;
	using M = LaneMap<NS, BW>;
	constexpr int NR = M::NR, NQ = M::NQ;
	// 8-bit one-pattern form: int16 bank + table of scale bytes (vfgs_layout.h "packed 16-bit form"); pkshift = the scale shift
	constexpr bool PK = DEPTH == 8 && ONE && kPk16;
	static_assert(!(PK && M::PAIR), "lane pairs are 10-bit lanes");
	// unpack the block parameters
	int sg[NR];              // 0 / -1: the block's sign is negative
	uint32_t ad[NR], k2s[NR];
#pragma unroll
	for (int r = 0; r < NR; r++)
	{
		sg[r] = (int)rp.pa[r] >> 31;
		// one-pattern form: the address of a negative block (of the CURRENT row of blocks; `up` is only used on overlap lines)
		// points into the negated bank (vfgs_layout.h); the overlap lines blend true values by signed weights: back to the bank
		ad[r] = (rp.pa[r] & 0xffffu) + rowoff - ((ONE && OVERLAP) ? ((uint32_t)sg[r] & (uint32_t)NEG) : 0u);
		k2s[r] = ONE ? 0u : (OVERLAP ? lutb : (((uint32_t)sg[r] & 0x04000400u) | lutb));     // OVERLAP: the +scale table
	}
	if constexpr (PK && !OVERLAP)
	{
		// ---- two samples per instruction (every line but the two overlap lines of a block row) ----
		const uint32_t rnd = 1u << (pkshift - 1);
		const uint32_t rpk = rnd * 0x10001u, shpk = (uint32_t)pkshift * 0x10001u;
		const uint8_t* lut = lds + (lutb & 0xffffu);
		uint32_t S[NS / 2], Pp[NS / 2], G[NS / 2];
		// scale gather (vfgs_hw.c:211,239): the sample IS the address; a pair's scales in the two halves of one register
#pragma unroll
		for (int q = 0; q < NQ; q++)
		{
			const uint32_t v = w[q];
			const uint32_t s0 = lut[v & 0xffu], s1 = lut[(v >> 8) & 0xffu], s2 = lut[(v >> 16) & 0xffu], s3 = lut[v >> 24];
			S[2 * q] = s0 | (s1 << 16);
			S[2 * q + 1] = s2 | (s3 << 16);
		}
		// pattern fetch: four int16 = one 8-byte read (two dwords where the block offsets are multiples of 2 samples, ALIGN2)
#pragma unroll
		for (int q = 0; q < NQ; q++)
		{
			uint32_t a8 = ad[M::run(q)] + M::col(q) * 2;
#ifdef VFGS_PK_NO_READ2      // developer A/B: keep the compiler from pairing two quads' 8-byte reads into one ds_read2_b64 (8 LDS cycles where two ds_read_b64 take 2 each)
			if (!ALIGN2 && (q & 1)) asm volatile("" : "+v"(a8));
#endif
			if (ALIGN2) { Pp[2 * q] = *(const uint32_t*)(lds + a8); Pp[2 * q + 1] = *(const uint32_t*)(lds + a8 + 4); }
			else { const u32x2 t = *(const u32x2*)(lds + a8); Pp[2 * q] = t.x; Pp[2 * q + 1] = t.y; }
		}
		// round(scale * P, shift) (vfgs_hw.c:263) for both samples of a pair
#pragma unroll
		for (int m = 0; m < NS / 2; m++) G[m] = pk_ashr_i16(shpk, pk_mad_i16(Pp[m], S[m], rpk));
		// the two samples at a block edge (vfgs_hw.c:250-259): 3-tap filter on the unfiltered neighbours as dot products of the
		// packed pairs, 32-bit product with the sample's scale (the filtered value may reach +-159), result dropped into its half
#pragma unroll
		for (int ed = 0; ed < M::NE; ed++)
		{
			const int s = 4 * M::edge_quad(ed), pa = s / 2 + 1, pb = s / 2 + 2;      // pairs (l1, l0) | (r0, r1)
			const uint32_t A = Pp[pa], B = Pp[pb];
			const int fl = dot2_i16(B, 0x00000001u, dot2_i16(A, 0x00030001u, 2)) >> 2;     // (l1 + 3 l0 + r0 + 2) >> 2
			const int fr = dot2_i16(A, 0x00010000u, dot2_i16(B, 0x00010003u, 2)) >> 2;     // (l0 + 3 r0 + r1 + 2) >> 2
			const int gl = mad_i32_i16<true>(fl, S[pa], (int)rnd), gr = mad_i32_i16<false>(fr, S[pb], (int)rnd);
			const uint32_t mA = edge_on[ed] ? 0xffff0000u : 0u, mB = mA >> 16;
			G[pa] = bfi(mA, (uint32_t)gl << (16 - pkshift), G[pa]);
			G[pb] = bfi(mB, (uint32_t)(gr >> pkshift), G[pb]);
		}
		// add, clip (vfgs_hw.c:264-267), two samples per instruction
#pragma unroll
		for (int d = 0; d < 4; d++)
		{
			const uint32_t v01 = __builtin_amdgcn_perm(0, w[d], 0x0c010c00), v23 = __builtin_amdgcn_perm(0, w[d], 0x0c030c02);
			s16x2 a = __builtin_bit_cast(s16x2, v01) + __builtin_bit_cast(s16x2, G[2 * d]);
			s16x2 b = __builtin_bit_cast(s16x2, v23) + __builtin_bit_cast(s16x2, G[2 * d + 1]);
			a = __builtin_elementwise_min(__builtin_elementwise_max(a, __builtin_bit_cast(s16x2, lo2)), __builtin_bit_cast(s16x2, hi2));
			b = __builtin_elementwise_min(__builtin_elementwise_max(b, __builtin_bit_cast(s16x2, lo2)), __builtin_bit_cast(s16x2, hi2));
			w[d] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, b), __builtin_bit_cast(uint32_t, a), 0x06040200);
		}
		return;
	}
	uint32_t e[NS];
	int P[NS];
	uint32_t pdw[NQ];        // one-pattern form: the four pattern bytes of quad q as they came out of LDS

	// LUT gather: intensity = sample >> bs, as a uint8 (vfgs_hw.c:157,211); entry address = 4 * intensity | table
#pragma unroll
	for (int q = 0; q < NQ; q++)
	{
		const uint32_t k2 = k2s[M::run(q)];
		if (DEPTH > 8)
		{
#pragma unroll
			for (int h = 0; h < 2; h++)
			{
				const uint32_t idx = (w[2 * q + h] & 0x03fc03fcu) | k2;
				e[4 * q + 2 * h]     = *(const uint32_t*)(lds + (idx & 0xffffu));
				e[4 * q + 2 * h + 1] = *(const uint32_t*)(lds + (idx >> 16));
			}
		}
		else if (PK)
		{
			// (overlap lines of the packed 16-bit form: the scale byte itself)
			const uint32_t v = w[q];
			const uint8_t* lut = lds + (lutb & 0xffffu);
			e[4 * q + 0] = lut[v & 0xffu];
			e[4 * q + 1] = lut[(v >> 8) & 0xffu];
			e[4 * q + 2] = lut[(v >> 16) & 0xffu];
			e[4 * q + 3] = lut[v >> 24];
		}
		else
		{
			const uint32_t v = w[q], k1 = k2 & 0xffffu;
			e[4 * q + 0] = *(const uint32_t*)(lds + (((v << 2) & 0x3fcu) | k1));
			e[4 * q + 1] = *(const uint32_t*)(lds + (((v >> 6) & 0x3fcu) | k1));
			e[4 * q + 2] = *(const uint32_t*)(lds + (((v >> 14) & 0x3fcu) | k1));
			e[4 * q + 3] = *(const uint32_t*)(lds + (((v >> 22) & 0x3fcu) | k1));
		}
	}

	// pattern fetch.  General form: 4 samples x 8 slots = 32 bytes per quad, the sample's slot picked by v_perm_b32.
	// One-pattern form: 4 samples = one dword (at a 2-byte aligned address where the block offsets are multiples of 2
	// samples, ALIGN2: cut out of two aligned dwords), each value sign-extended out of its byte.
	auto fetch4 = [&](uint32_t adq, int q, int (&out)[4], uint32_t& raw) {
		if (PK)
		{
			const uint32_t a8 = adq + M::col(q) * 2;
			uint32_t p01, p23;
			if (ALIGN2) { p01 = *(const uint32_t*)(lds + a8); p23 = *(const uint32_t*)(lds + a8 + 4); }
			else { const u32x2 t = *(const u32x2*)(lds + a8); p01 = t.x; p23 = t.y; }
			raw = p01;
			out[0] = (int)(p01 << 16) >> 16; out[1] = (int)p01 >> 16;
			out[2] = (int)(p23 << 16) >> 16; out[3] = (int)p23 >> 16;
		}
		else if (ONE)
		{
			uint32_t d;
			if (ALIGN2)
			{
				const uint32_t a4 = adq + M::col(q);
				const uint32_t lo = *(const uint32_t*)(lds + (a4 & ~3u)), hi = *(const uint32_t*)(lds + (a4 & ~3u) + 4);
				d = __builtin_amdgcn_alignbit(hi, lo, (a4 & 2u) * 8);
			}
			else
				d = *(const uint32_t*)(lds + adq + M::col(q));
			raw = d;
#pragma unroll
			for (int i = 0; i < 4; i++) out[i] = (int)(d << (24 - 8 * i)) >> 24;
		}
		else
		{
			const u32x4 c0 = *(const u32x4*)(lds + adq + M::col(q) * kSlots), c1 = *(const u32x4*)(lds + adq + M::col(q) * kSlots + 16);
			out[0] = pick_slot(c0.y, c0.x, e[4 * q + 0]); out[1] = pick_slot(c0.w, c0.z, e[4 * q + 1]);
			out[2] = pick_slot(c1.y, c1.x, e[4 * q + 2]); out[3] = pick_slot(c1.w, c1.z, e[4 * q + 3]);
		}
	};
#pragma unroll
	for (int q = 0; q < NQ; q++)
	{
		int v4[4];
		pdw[q] = 0;
		fetch4(ad[M::run(q)], q, v4, pdw[q]);
#pragma unroll
		for (int i = 0; i < 4; i++) P[4 * q + i] = v4[i];
	}
	if (OVERLAP)
	{
#pragma unroll
		for (int q = 0; q < NQ; q++)
		{
			const int r = M::run(q);
			const uint32_t uad = (up.pa[r] & 0xffffu) + uprowoff;
			int Q[4];
			uint32_t rawq;
			fetch4(uad, q, Q, rawq);
			const int m = (wcur ^ sg[r]) - sg[r];                            // sign_cur * weight_cur
			const int usg = (int)up.pa[r] >> 31;
			const int n = (wup ^ usg) - usg;                                 // sign_up * weight_up
#pragma unroll
			for (int i = 0; i < 4; i++) P[4 * q + i] = mad_vvv(Q[i], n, mad_vvv(P[4 * q + i], m, 16)) >> 5;
		}
	}

	// 3-tap filter across the block edge (vfgs_hw.c:250-259), on unfiltered neighbours
	// (all values are small: explicit 24-bit multiply-adds; the compiler otherwise reaches for 32/64-bit
	// multiplies and turns the selects into a branch that copies all the P registers)
	if (M::PAIR)
	{
		const int mine = first ? P[0] : P[7];
		const int inner = first ? P[1] : P[6];
		const int theirs = swap_lane_pairs(mine);
		int f;
		if (OVERLAP || ONE) f = (theirs + mad_vvv(3, mine, inner + 2)) >> 2;
		else
		{
			const int x = sg[0] ^ swap_lane_pairs(sg[0]);                    // 0: the two blocks have the same sign, -1: opposite
			f = (((theirs ^ x) - x) + mad_vvv(3, mine, inner + 2 + sg[0])) >> 2;
		}
		f = edge_on[0] ? f : mine;
		P[0] = first ? f : P[0];
		P[7] = first ? P[7] : f;
	}
	else
	{
		int fl[NR], fr[NR];
#pragma unroll
		for (int ed = 0; ed < M::NE; ed++)
		{
			const int s = 4 * M::edge_quad(ed);              // l1 = P[s+2], l0 = P[s+3] | r0 = P[s+4], r1 = P[s+5]
			if (OVERLAP || ONE)
			{
				fl[ed] = (P[s + 4] + mad_vvv(3, P[s + 3], P[s + 2] + 2)) >> 2;
				fr[ed] = (P[s + 3] + mad_vvv(3, P[s + 4], P[s + 5] + 2)) >> 2;
			}
			else
			{
				const int x = sg[ed] ^ sg[ed + 1];
				fl[ed] = (((P[s + 4] ^ x) - x) + mad_vvv(3, P[s + 3], P[s + 2] + 2 + sg[ed])) >> 2;
				fr[ed] = (((P[s + 3] ^ x) - x) + mad_vvv(3, P[s + 4], P[s + 5] + 2 + sg[ed + 1])) >> 2;
			}
		}
#pragma unroll
		for (int ed = 0; ed < M::NE; ed++)
		{
			const int s = 4 * M::edge_quad(ed);
			P[s + 3] = edge_on[ed] ? fl[ed] : P[s + 3];
			P[s + 4] = edge_on[ed] ? fr[ed] : P[s + 4];
		}
	}

	// scale, add, clip (vfgs_hw.c:263-267)
	// round(scale * P, shift) (vfgs_hw.c:263) = (scale * 2^(16-shift) * P + 2^15) >> 16 exactly; the LUT holds
	// scale * 2^(16-shift), so the shift is free: the pack below simply takes the high halves
	// scaled grain of sample i, before the final >> 16: from the value itself where the edge filter may have changed it, straight
	// from the pattern byte (one-pattern form, no overlap blend) everywhere else
	auto edge_sample = [](int i) {
		if (M::PAIR) return i == 0 || i == NS - 1;
		for (int ed = 0; ed < M::NE; ed++)
			if (i == 4 * M::edge_quad(ed) + 3 || i == 4 * M::edge_quad(ed) + 4) return true;
		return false;
	};
	int G[NS];
#pragma unroll
	for (int i = 0; i < NS; i++)
	{
		// (8 bit only: +1..2 % there, 8 VGPRs fewer and no spill left in the all-one-pattern kernels; at 10 bit the same change
		// lets the compiler reach six waves per SIMD, which these kernels do not like: -1..-2.5 %, profiles/r04_ab5_sdwa_multiply.log)
		if (PK) G[i] = mad_i24(P[i], e[i], 1 << (pkshift - 1)) << (16 - pkshift);     // (overlap lines only; < 2^24: |P| <= 199, scale <= 255)
		else if (DEPTH == 8 && ONE && !OVERLAP && !edge_sample(i)) G[i] = mul_byte_i24(i % 4, pdw[i / 4], e[i]) + 0x8000;
		else G[i] = mad_i24(P[i], e[i], 0x8000);
	}
	auto clip2 = [&](uint32_t v, int g0, int g1) {
		const uint32_t gp = __builtin_amdgcn_perm((uint32_t)g1, (uint32_t)g0, 0x07060302);
		if (DEPTH > 8)  // a 16-bit container may hold anything: keep the add inside int16 (result is clipped anyway)
			v = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, v), __builtin_bit_cast(u16x2, 0x70007000u)));
		s16x2 s = __builtin_bit_cast(s16x2, v) + __builtin_bit_cast(s16x2, gp);
		s = __builtin_elementwise_max(s, __builtin_bit_cast(s16x2, lo2));
		s = __builtin_elementwise_min(s, __builtin_bit_cast(s16x2, hi2));
		return __builtin_bit_cast(uint32_t, s);
	};
	if (DEPTH > 8)
	{
#pragma unroll
		for (int d = 0; d < 4; d++)
			w[d] = clip2(w[d], G[2 * d], G[2 * d + 1]);
	}
	else
	{
#pragma unroll
		for (int d = 0; d < 4; d++)
		{
			const uint32_t v01 = __builtin_amdgcn_perm(0, w[d], 0x0c010c00), v23 = __builtin_amdgcn_perm(0, w[d], 0x0c030c02);
			const uint32_t s01 = clip2(v01, G[4 * d], G[4 * d + 1]);
			const uint32_t s23 = clip2(v23, G[4 * d + 2], G[4 * d + 3]);
			w[d] = __builtin_amdgcn_perm(s23, s01, 0x06040200);
		}
	}
}


// ---------------------------------------------------------------------------------------
// memory helpers

template <int AUX>
__device__ __forceinline__ void load_seg(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff, uint32_t (&w)[4])
{
	const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, AUX);
	w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w;
}

// A buffer store of more than 64 bits reads its data registers AFTER it has issued; a VALU write to them in the next
// cycle changes what the last lanes (12..15 of every 16) store.  The compiler's hazard recognizer inserts the wait state
// only for stores WITHOUT a scalar offset register (the documented form of the hazard); on gfx950 the stores with one
// need it too (found the hard way: DESIGN.md 4 "store data hazard").  Two wait states after every such store.
__device__ __forceinline__ void store_data_hazard()
{
	__builtin_amdgcn_sched_barrier(0);     // nothing may move between the store and the wait states
	asm volatile("s_nop 1");
	__builtin_amdgcn_sched_barrier(0);
}

// one unit of the destination: 16 bytes, or 8 where a 10-bit source is narrowed to 8 bit (DW = 2)
template <int DW, int AUX>
__device__ __forceinline__ void store_unit(__amdgpu_buffer_rsrc_t rs, uint32_t voff, const uint32_t (&w)[DW])
{
	if constexpr (DW == 4)
	{
		const u32x4 t = {w[0], w[1], w[2], w[3]};
		__builtin_amdgcn_raw_buffer_store_b128(t, rs, voff, 0, AUX);
		store_data_hazard();
	}
	else
	{
		const u32x2 t = {w[0], w[1]};
		__builtin_amdgcn_raw_buffer_store_b64(t, rs, voff, 0, AUX);
	}
}

// register sets of a wave's ring for a plane of this depth and form (vfgs_layout.h VFGS_RING*)
template <int DEPTH, bool ONE>
constexpr int ring_depth() { return (DEPTH == 8 && ONE && kPk16) ? VFGS_RING_PK : ((DEPTH > 8 && ONE) ? VFGS_RING_ONE10 : VFGS_RING); }

// ---------------------------------------------------------------------------------------
// Row walk: one workgroup's share of one plane.
//
// tools/skeleton2.hip (profiles/r03_skeleton2_*.log) priced what round 2's tiled kernels paid besides their bytes: every
// vector-memory INSTRUCTION queues for the CU's saturated memory pipeline, and a 4 KiB tile cost 11 of them where 8 move
// data (narrow accesses at both tile edges), plus 8 LFSR loads per wave: 0.73 -> 0.67 of 8 TB/s.
// Here a wave owns whole ROWS instead of a tile of several rows:
//   * a workgroup = 4 waves = 4 x rw_rpw rows of ONE block row (wave w: rows w, w + 4, ...); a wave streams its row position
//     by position (64 aligned 16-byte units each) through a ring of four register sets -- the refill of a set is the position
//     four steps ahead, across row ends -- so a row costs one load and one store per KiB and nothing else;
//   * the block parameters of the row's blocks (this block row's LFSR registers and, for the workgroup that holds the overlap
//     lines, those of the block row above) are computed ONCE per workgroup, one or two blocks per thread, and kept in LDS
//     behind the table image; a lane reads its 1-3 entries per position;
//   * lanes compute the half-block shifted bytes: rotation by one lane with DPP wave_shr / wave_shl, the hand-over between
//     consecutive positions of a row with wave_ror / wave_rol (the previous position's lane 63 waits in lane 0 of a register
//     and enters as the `old` operand of the shift: no v_readlane, no scalar round trip); a position's units are stored one
//     step later, once lane 0 of the next position has delivered the last dwords of its lane 63;
//   * row bases and position offsets live in the buffer descriptor (base, num_records = bytes of the row left), so the
//     hardware range check covers every access of every lane: lanes behind the row's end load 0 and store nothing
//     (8-bit 4:2:x rows of an odd number of blocks end in HALF a unit: the check works per dword);
//   * rows of more than kTileBlocks blocks (WIDE kernels): the table holds one PART of
//     the row (kTileBlocks blocks) at a time; all waves walk part 0 of their row, the workgroup refills the table for part 1
//     between two barriers, and so on -- the ring of register sets simply runs on (the host gives such workgroups one row per wave);
//   * OUT8 (10-bit source, 8-bit destination, yuv.c:216-258): out8 = (v + 2) >> 2 is applied to a lane's results, which
//     halves them (DW = 2 dwords per unit); everything behind the computation -- rotation back, stores, descriptors -- works
//     on those halves with the destination's own pitches.
template <int DEPTH, int BW, int SUBX, int SUBY, int RS, int IMG_BYTES, bool ONE, int NEG, int NARROW, bool OUT8, bool WIDE, bool PERSIST>
__device__ __forceinline__ void run_plane_rw(const KernelArgs& a, const FrameTable& ft, const PlaneDesc& pd, uint8_t* lds, const int comp, const int f_in, const int r_in,
                                             const uint32_t img_off, const uint32_t bank_off, const uint32_t lut_off, const int lane, const int wave)
{
	constexpr int NS = DEPTH == 8 ? 16 : 8;
	constexpr int SZ = DEPTH > 8 ? 2 : 1;
	constexpr int SB = ONE ? ((DEPTH == 8 && kPk16) ? 2 : 1) : kSlots;     // bytes of a bank per sample position (vfgs_layout.h)
	using M = LaneMap<NS, BW>;
	constexpr int NR = M::NR;
	constexpr int RPB = 16 / SUBY;                       // rows of this plane per block row
	constexpr int NEF = M::PAIR ? 1 : M::NE;
	constexpr int K = M::SHIFT * SZ / 4;                 // dwords of a lane that lie in the memory unit before the lane's own
	constexpr int DW = OUT8 ? 2 : 4;                     // dwords of a unit in the destination ...
	constexpr int KD = OUT8 ? K / 2 : K;                 // ... and how many of a lane's result dwords belong to the unit before its own
	static_assert(!OUT8 || (DEPTH == 10 && K % 2 == 0), "the narrowed destination exists for 10-bit sources");
	constexpr int LDA = VFGS_LDAUX_ALIGNED, STA = VFGS_STAUX_ALIGNED;
	constexpr int LPB = M::PAIR ? 1 : M::BPL;            // blocks per lane step (PAIR: half a block, see idx0 below)
	constexpr int BPS = M::PAIR ? 32 : 64 * M::BPL;      // grain blocks a position advances by
	// positions per group = register sets of the ring = how many positions behind its load a position is stored (vfgs_layout.h)
	// (narrow one-pattern planes at 10 bit -- rows of one or two positions: the chroma of 1080p -- walk with a ring of their own depth: vfgs_layout.h)
	constexpr int NU = (DEPTH > 8 && ONE && NARROW != 0) ? VFGS_RING_NARROW10 : ring_depth<DEPTH, ONE>();
	constexpr int GPP = kTileBlocks / (NU * BPS);        // groups per part of a row (a part = kTileBlocks blocks = one parameter table)
	static_assert(GPP * NU * BPS == kTileBlocks, "a part is a whole number of groups");
	constexpr uint32_t PT_CUR = IMG_BYTES, PT_UP = IMG_BYTES + kParamTableBytes;
	static_assert(IMG_BYTES % 16 == 0, "table image in whole 16-byte units");
	static_assert(!(PERSIST && (WIDE || NARROW != 0)), "persistent workgroups walk ordinary rows");
	const int pt = comp ? 1 : 0;
	auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };

	// One TASK = what a workgroup of the non-persistent launch does: the rows of one part of one block row of frame f.  A
	// persistent workgroup (PERSIST: general-form luma of small pictures, where the 36 KB table image is more than the 15-30 KB
	// of samples it serves) runs several tasks with ONE staging of the image: only the block parameters are per task (their
	// LFSR words and the task's first four positions are requested before the workgroup meets at the barrier that frees the
	// parameter table).
	auto task = [&](const int f, const int r, const bool first_task) {
	// ---- the workgroup's place: block row of the stripe, part of it; the wave's rows -------------------------------
	const int split = r & (pd.rw_splits - 1);
	const int kbr = uni(r >> pd.rw_lsplits);             // block row inside the stripe
	const int Rabs = (a.y0 >> 4) + kbr;                  // absolute block row
	const int row_first = (a.y0 + SUBY - 1) / SUBY;      // first row of the stripe in this plane; the plane pointers address row y0 / SUBY
	const int prow0 = a.y0 / SUBY;
	const int alo = max(row_first, Rabs * RPB), ahi = min(row_first + pd.nrows, (Rabs + 1) * RPB);
	const int rpw = pd.rw_rpw;
#if VFGS_RW_CONSEC
	const int RSTR = 1;                                  // a wave's rows are consecutive (a contiguous stream where the rows are)
	const int base = uni(Rabs * RPB + split * (kWavesPerWG * rpw) + wave * rpw);     // my rows: base + RSTR * k, k in [k0, k1)
	int k0 = max(0, alo - base), k1 = min(rpw, max(0, ahi - base));
#else
	constexpr int RSTR = kWavesPerWG;                    // the waves of a workgroup walk down its rows side by side
	const int base = uni(Rabs * RPB + split * (kWavesPerWG * rpw) + wave);
	int k0 = (max(0, alo - base) + kWavesPerWG - 1) / kWavesPerWG, k1 = min(rpw, (max(0, ahi - base) + kWavesPerWG - 1) / kWavesPerWG);
#endif
	if (kbr >= a.nbrows || k1 < k0) k1 = k0;
	k0 = uni(k0); k1 = uni(k1);
	const bool wg_up = (Rabs > 0) && (split == 0);       // this workgroup holds the overlap lines of its block row (vfgs_hw.c:175,180)

	const int last = a.nblk - 1;
	const uint32_t cur_bit = a.cur_bit0 + (uint32_t)f * a.frame_bit_step + (uint32_t)(kbr * a.nblk);
	const uint32_t up_bit = (kbr > 0) ? cur_bit - (uint32_t)a.nblk : a.up_bit0 + (uint32_t)f * a.frame_bit_step;

	// ---- in flight together: the table image, the LFSR words of the row's blocks, my first four positions ----------
	// (in this order: a wave's loads return in issue order; fixed instruction stream)
	constexpr int STEP = kWavesPerWG * 64 * 16;
	constexpr int NIT = (IMG_BYTES + STEP - 1) / STEP;
	u32x4 tmp[NIT];
	if (!PERSIST || first_task)
	{
		const __amdgpu_buffer_rsrc_t irs = make_rsrc(a.tables + img_off, IMG_BYTES);
#pragma unroll
		for (int it = 0; it < NIT; it++)      // threads beyond the image re-read (and re-write) its last unit
			tmp[it] = __builtin_amdgcn_raw_buffer_load_b128(irs, min((uint32_t)(threadIdx.x * 16 + it * STEP), (uint32_t)(IMG_BYTES - 16)), 0, 0);
	}
	// parameter table of the part that begins at block B0: entry e = block B0 + e - 1 (clamped into the row); thread t fills
	// entries t, t + 256, ...
	constexpr int NPE = (kParamEntries + kWavesPerWG * 64 - 1) / (kWavesPerWG * 64);
	const __amdgpu_buffer_rsrc_t strs = make_rsrc((const uint8_t*)a.stream, a.stream_bytes);
	const __amdgpu_buffer_rsrc_t strs_up = make_rsrc((const uint8_t*)a.stream, wg_up ? a.stream_bytes : 0);
	auto param_loads = [&](const int B0, u32x2 (&wc)[NPE], u32x2 (&wu)[NPE]) {
#pragma unroll
		for (int i = 0; i < NPE; i++)
		{
			const int e = (int)threadIdx.x + i * kWavesPerWG * 64;
			const uint32_t blk = (uint32_t)min(max(B0 + e - 1, 0), last);
			const bool need = e < a.nblk - B0 + 4 && e < kParamEntries;
			wc[i] = __builtin_amdgcn_raw_buffer_load_b64(strs, need ? ((cur_bit + blk) >> 5) * 4 : kOOB, 0, 0);
			wu[i] = __builtin_amdgcn_raw_buffer_load_b64(strs_up, need ? ((up_bit + blk) >> 5) * 4 : kOOB, 0, 0);
		}
	};
	const int fsx = comp == 0 ? 0 : (comp == 1 ? 10 : 20);
	const int fsy = comp == 0 ? 14 : (comp == 1 ? 24 : 4);
	const int fsb = comp == 0 ? 31 : (comp == 1 ? 2 : 15);
	auto param_table = [&](const int B0, const u32x2 (&wc)[NPE], const u32x2 (&wu)[NPE]) {
#pragma unroll
		for (int i = 0; i < NPE; i++)
		{
			// (rows of up to 252 blocks -- 2160p and narrower -- need the first round only, 4320p two of the three: a wave-uniform
			// branch around arithmetic and LDS writes; the LFSR loads stay unconditional, switched off by their offsets, so
			// that the waits can still be counted)
			if (i > 0 && i * kWavesPerWG * 64 >= a.nblk - B0 + 4) break;
			const int e = (int)threadIdx.x + i * kWavesPerWG * 64;
			const uint32_t blk = (uint32_t)min(max(B0 + e - 1, 0), last);
			bool neg;
			const uint32_t vc = __builtin_amdgcn_alignbit(wc[i].y, wc[i].x, (cur_bit + blk) & 31);
			const uint32_t pc = (block_param<SUBX, SUBY, RS, SB>(vc, bank_off, fsx, fsy, fsb, &neg) + ((ONE && neg) ? (uint32_t)NEG : 0u)) | (neg ? 0x80000000u : 0u);
			const uint32_t vu = __builtin_amdgcn_alignbit(wu[i].y, wu[i].x, (up_bit + blk) & 31);
			const uint32_t pu = block_param<SUBX, SUBY, RS, SB>(vu, bank_off, fsx, fsy, fsb, &neg) | (neg ? 0x80000000u : 0u);
			if (e < kParamEntries)
			{
				*(uint32_t*)(lds + PT_CUR + e * 4) = pc;
				*(uint32_t*)(lds + PT_UP + e * 4) = pu;
			}
		}
	};
	u32x2 wc0[NPE], wu0[NPE];
	param_loads(0, wc0, wu0);
	// A row = rw_segs wave accesses ("positions": its units and the one behind them), walked in groups of four: the four
	// register sets.  One buffer descriptor serves a whole group: base = the first byte of the group, num_records = the
	// bytes the row has left from there (at most the group's 4 KiB), the position's 1 KiB step sits in the instruction's
	// immediate offset: the hardware range check switches off exactly the lanes behind the row's end -- and every lane of a
	// group that does not exist -- and no access of any lane can leave the row.  (Measured on gfx950: a scalar offset
	// operand IS part of what is checked against num_records, so the row offset has to go into the base.)
	const int tsegs = pd.rw_segs;
	const int ngroups = (tsegs + NU - 1) / NU;
	// (frames at a constant pitch, or a list of frames anywhere: their pointers sit in the kernel arguments, one scalar load)
	const uint8_t* sbase = a.listed ? ft.src[comp][f] : a.src[comp] + (uint64_t)f * pd.fpitch;
	uint8_t* dbase = a.listed ? ft.dst[comp][f] : a.dst[comp] + (uint64_t)f * pd.dfpitch;
	const uint32_t lane16 = (uint32_t)lane * 16;
	const uint32_t laned = (uint32_t)lane * (4 * DW);    // ... in the destination
	constexpr uint32_t UB = kMaxUnits * 16, UBD = kMaxUnits * 4 * DW;   // bytes of a position, source / destination
	constexpr uint32_t GB = NU * UB, GBD = NU * UBD;     // bytes of a group
	auto row_off = [&](int k) { return (uint32_t)((base + RSTR * k - prow0) * (int)pd.pitch); };
	// (only the narrowed destination has a geometry of its own: otherwise the host passes the source's, and saying so here
	// saves the scalar registers of a second set of row offsets)
	auto row_offd = [&](int k) { return OUT8 ? (uint32_t)((base + RSTR * k - prow0) * (int)pd.dpitch) : row_off(k); };
	auto left = [&](int g) { const uint32_t o = (uint32_t)g * GB; return o < pd.rowbytes ? min(pd.rowbytes - o, GB) : 0u; };
	auto leftd = [&](int g) { const uint32_t o = (uint32_t)g * GBD, rb = OUT8 ? pd.drowbytes : pd.rowbytes; return o < rb ? min(rb - o, GBD) : 0u; };
	uint32_t w[NU][4];
	if constexpr (NARROW == 0)
	{
		const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(sbase + (k0 < k1 ? row_off(k0) : 0u), k0 < k1 ? left(0) : 0u);
#pragma unroll
		for (int u = 0; u < NU; u++) load_seg<LDA>(rs0, lane16 + u * UB, 0, w[u]);
	}
	else
	{
		// rows of NARROW positions: the four register sets hold 4 / NARROW consecutive rows of the wave
#pragma unroll
		for (int u = 0; u < NU; u++)
		{
			const int k = k0 + u / NARROW;
			const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(sbase + (k < k1 ? row_off(k) : 0u), k < k1 ? pd.rowbytes : 0u);
			load_seg<LDA>(rs0, lane16 + (u % NARROW) * UB, 0, w[u]);
		}
	}
	if (!PERSIST || first_task)
	{
#pragma unroll
		for (int it = 0; it < NIT; it++)
			*(u32x4*)(lds + min((uint32_t)(threadIdx.x * 16 + it * STEP), (uint32_t)(IMG_BYTES - 16))) = tmp[it];
	}
	else
		__syncthreads();       // every wave is done with the previous task's parameter table

	// ---- block parameters of the row's first part (once per workgroup and task) --------------------------------------
	param_table(0, wc0, wu0);
	__syncthreads();
	// (WIDE is a kernel of its own: the part loop keeps the LFSR descriptors and a few more values alive through the walk, 5-7
	// VGPRs and a dozen spilled SGPRs that every picture would pay for -- measured with -Rpass-analysis=kernel-resource-usage)
	const int nparts = WIDE ? (a.nblk + kTileBlocks - 1) / kTileBlocks : 1;     // (workgroup-uniform: every wave meets the barriers below)
	static_assert(!(WIDE && NARROW != 0), "rows walked in parts are not narrow");
	if (k0 >= k1 && nparts == 1)
		return;

	// ---- per lane constants ------------------------------------------------------------------------------------------
	const uint32_t lutb = lut_off * 0x10001u;
	const uint32_t lo2 = a.lo2[pt], hi2 = a.hi2[pt];
	const bool first = M::PAIR && (lane & 1);                              // PAIR: odd lane positions hold the first half of a block
	const uint32_t pairoff = M::PAIR ? (first ? 0u : 8u * SB) : 0u;
	const uint32_t idx0 = (M::PAIR ? (uint32_t)(lane + 1) >> 1 : (uint32_t)lane * LPB) * 4;   // byte offset of my first entry in position 0 of a part
	const int cl = M::PAIR ? lane - 1 - (lane & 1) : lane * LPB - 1;       // PAIR: left unit of my lane pair; else: block of run 0 (both for position 0)
	constexpr int SSTEP = M::PAIR ? 64 : BPS;                              // what `cl` advances by per position
	// DPP moves by one lane; lanes without a source lane (lane 0 / lane 63) keep `old`; the rotations wrap around
	auto lane_up = [](uint32_t old, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x138, 0xf, 0xf, false); };    // wave_shr:1: lane l <- lane l - 1
	auto lane_down = [](uint32_t old, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x130, 0xf, 0xf, false); };  // wave_shl:1: lane l <- lane l + 1
	auto rot_up = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x13c, 0xf, 0xf, false); };      // wave_ror:1: lane 0 <- lane 63
	auto rot_down = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x134, 0xf, 0xf, false); };    // wave_rol:1: lane 63 <- lane 0
	// a lane's results as they go to memory: the 4 dwords themselves, or narrowed to 8 bit (yuv_to_8bit, yuv.c:216-258:
	// out8 = (v + 2) >> 2; both halves of a dword are <= 1023 + 2: no carry across them)
	auto results = [](const uint32_t (&t)[4], uint32_t (&o)[DW]) {
		if constexpr (OUT8)
		{
			uint32_t n[4];
#pragma unroll
			for (int d = 0; d < 4; d++) n[d] = ((t[d] + 0x00020002u) >> 2) & 0x00ff00ffu;
			o[0] = __builtin_amdgcn_perm(n[1], n[0], 0x06040200);
			o[1] = __builtin_amdgcn_perm(n[3], n[2], 0x06040200);
		}
		else
		{
#pragma unroll
			for (int d = 0; d < 4; d++) o[d] = t[d];
		}
	};

	// ---- the walk ----------------------------------------------------------------------------------------------------
	uint32_t carry[4] = {0, 0, 0, 0};      // in lane 0: the last K dwords of lane 63 of the previous position of the row
	uint32_t outp[DW] = {};                // the previous position's units: dwords KD.. of its lanes (the first DW - KD dwords of a unit)
	uint32_t tp[KD > 0 ? KD : 1] = {};     // ... and the first KD dwords its lanes computed: they belong one lane down
	__amdgpu_buffer_rsrc_t pdst = make_rsrc(dbase, 0);     // where the previous GROUP's last position goes
	if constexpr (NARROW != 0)
	{
		// Rows of one or two positions (chroma of 1080p 4:2:0 at 10 bit, of 2160p at 8 bit: 2 KiB and less).  Walked row by
		// row, half of the ring would stay empty and a wave would keep 2 KiB in flight where the chip needs 4: so a group is
		// 4 / NARROW consecutive rows of the wave.  Position q of the wave's sequence = row k0 + q / NARROW, position
		// q % NARROW of it; everything that depends on the row is wave-uniform data of the slot; the overlap lines are a
		// wave-uniform branch per slot (not two walks: a group may hold one of each kind).
		constexpr int P = NARROW, RPG = NU / P;
		const int ngr = (k1 - k0 + RPG - 1) / RPG;
		for (int G = 0; G < ngr; G++)
		{
			const int kg = k0 + G * RPG;
#pragma unroll
			for (int u = 0; u < NU; u++)
			{
				const int p = u % P;                                   // position inside the row (compile time after unrolling)
				const int k = kg + u / P;                              // the slot's row
				const bool valid = k < k1, nvalid = k + RPG < k1;      // wave-uniform
				const __amdgpu_buffer_rsrc_t nsrc = make_rsrc(sbase + (nvalid ? row_off(k + RPG) : 0u), nvalid ? pd.rowbytes : 0u);
				const __amdgpu_buffer_rsrc_t cdst = make_rsrc(dbase + (valid ? row_offd(k) : 0u), valid ? (OUT8 ? pd.drowbytes : pd.rowbytes) : 0u);
				uint32_t t[4];
#pragma unroll
				for (int d = 0; d < K; d++) t[d] = lane_up(p == 0 ? 0u : carry[d], w[u][4 - K + d]);
#pragma unroll
				for (int d = 0; d < K; d++) carry[d] = rot_up(w[u][4 - K + d]);
#pragma unroll
				for (int d = K; d < 4; d++) asm volatile("v_mov_b32 %0, %1" : "=v"(t[d]) : "v"(w[u][d - K]));
				load_seg<LDA>(nsrc, lane16 + p * UB, 0, w[u]);
				if (valid)
				{
					const int j = base + RSTR * k - Rabs * RPB;        // row inside the block row
					const int jrow = j * SUBY;
					const uint32_t rowoff = (uint32_t)j * RS, uprowoff = (uint32_t)(RPB + j) * RS;
					const int wc_ = jrow == 0 ? (SUBY > 1 ? 20 : 12) : 24, wu_ = jrow == 0 ? (SUBY > 1 ? 20 : 24) : 12;
					const uint8_t* pe = lds + idx0;
					bool edge_on[NEF];
					if (M::PAIR)
					{
						const int jl = cl + p * SSTEP;
						edge_on[0] = jl >= 0 && jl + 1 < 2 * a.nblk;
					}
					else
					{
#pragma unroll
						for (int ed = 0; ed < M::NE; ed++) edge_on[ed] = (cl + p * SSTEP + ed >= 0) && (cl + p * SSTEP + ed < last);
					}
					RunParam<NR> rp, up;
#pragma unroll
					for (int rr = 0; rr < NR; rr++) rp.pa[rr] = *(const uint32_t*)(pe + PT_CUR + (p * BPS + rr) * 4) + pairoff;
					if (Rabs > 0 && jrow <= 1)     // blends in the block above (vfgs_hw.c:173-188, 223-229)
					{
#pragma unroll
						for (int rr = 0; rr < NR; rr++) up.pa[rr] = *(const uint32_t*)(pe + PT_UP + (p * BPS + rr) * 4) + pairoff;
						grain_unit<DEPTH, BW, true, ONE, ONE && SUBX == 2, NEG>(lds, t, rp, up, lutb, rowoff, uprowoff, wc_, wu_, edge_on, first, lo2, hi2, a.pk_shift);
					}
					else
					{
#pragma unroll
						for (int rr = 0; rr < NR; rr++) up.pa[rr] = 0u;
						grain_unit<DEPTH, BW, false, ONE, ONE && SUBX == 2, NEG>(lds, t, rp, up, lutb, rowoff, uprowoff, 0, 0, edge_on, first, lo2, hi2, a.pk_shift);
					}
				}
				uint32_t o[DW];
				results(t, o);
				// the previous position (of this row, or the last one of the row before: `pdst` is its row) is complete
#pragma unroll
				for (int d = 0; d < KD; d++) outp[DW - KD + d] = lane_down(rot_down(o[d]), tp[d]);
				store_unit<DW, STA>(pdst, laned + ((p + P - 1) % P) * UBD, outp);
				pdst = cdst;
#pragma unroll
				for (int d = KD; d < DW; d++) outp[d - KD] = o[d];
#pragma unroll
				for (int d = 0; d < KD; d++) tp[d] = o[d];
#if VFGS_SCHED_FENCE
				__builtin_amdgcn_sched_barrier(0);
#endif
			}
		}
#pragma unroll
		for (int d = 0; d < KD; d++) outp[DW - KD + d] = lane_down(0u, tp[d]);
		store_unit<DW, STA>(pdst, laned + (P - 1) * UBD, outp);
		return;
	}
	// one row, groups [g_lo, g_hi) of it (a part); `overlap` is a type so that the walk of the (rare) overlap lines is code of
	// its own: the hot loop carries neither their arithmetic nor a branch around it
	auto walk_row = [&](auto overlap, const int k, const int g_lo, const int g_hi) {
		constexpr bool OV = decltype(overlap)::value;
		const int j = base + RSTR * k - Rabs * RPB;    // row inside the block row
		const int jrow = j * SUBY;
		const uint32_t rowoff = (uint32_t)j * RS, uprowoff = (uint32_t)(RPB + j) * RS;
		const int wc_ = jrow == 0 ? (SUBY > 1 ? 20 : 12) : 24, wu_ = jrow == 0 ? (SUBY > 1 ? 20 : 24) : 12;
		const uint32_t ro = row_off(k), rod = row_offd(k);
		for (int g = g_lo; g < g_hi; g++)
		{
			// the group after this one (this row's next, or the next row's first): what the four refills fetch
			const bool lastg = g + 1 == ngroups;
			const int ng = lastg ? 0 : g + 1;
			const bool nvalid = !lastg || k + 1 < k1;
			const uint32_t nso = nvalid ? (lastg ? row_off(k + 1) : ro) + (uint32_t)ng * GB : 0u;
			const __amdgpu_buffer_rsrc_t nsrc = make_rsrc(sbase + nso, nvalid ? left(ng) : 0u);
			const __amdgpu_buffer_rsrc_t cdst = make_rsrc(dbase + (rod + (uint32_t)g * GBD), leftd(g));
			const uint8_t* pe = lds + idx0 + (uint32_t)((g - g_lo) * NU * BPS * 4);    // (the table holds the part that begins at group g_lo)
			const int clg = cl + g * NU * SSTEP;
#pragma unroll
			for (int u = 0; u < NU; u++)
			{
				const bool firsts = u == 0 && g == 0;                    // first position of the row
				// assemble my 16 bytes: the last K dwords of the unit of the lane before me, the first 4 - K of mine
				uint32_t t[4];
#pragma unroll
				for (int d = 0; d < K; d++) t[d] = lane_up(firsts ? 0u : carry[d], w[u][4 - K + d]);   // (in front of a row there is nothing)
#pragma unroll
				for (int d = 0; d < K; d++) carry[d] = rot_up(w[u][4 - K + d]);
				// (real copies: were t[] merely another name for these registers, the refill below would have to land somewhere
				// else and be copied back at the end of the loop -- behind a wait for all four refills)
#pragma unroll
				for (int d = K; d < 4; d++) asm volatile("v_mov_b32 %0, %1" : "=v"(t[d]) : "v"(w[u][d - K]));
				// the registers are free: refill them with the position four steps ahead
				load_seg<LDA>(nsrc, lane16 + u * UB, 0, w[u]);
				if (NU * g + u < tsegs)
				{
					bool edge_on[NEF];
					if (M::PAIR)
					{
						const int jl = clg + u * SSTEP;                    // left unit of this lane pair
						edge_on[0] = jl >= 0 && jl + 1 < 2 * a.nblk;
					}
					else
					{
#pragma unroll
						for (int ed = 0; ed < M::NE; ed++) edge_on[ed] = (clg + u * SSTEP + ed >= 0) && (clg + u * SSTEP + ed < last);
					}
					RunParam<NR> rp, up;
#pragma unroll
					for (int rr = 0; rr < NR; rr++) rp.pa[rr] = *(const uint32_t*)(pe + PT_CUR + (u * BPS + rr) * 4) + pairoff;
#pragma unroll
					for (int rr = 0; rr < NR; rr++) up.pa[rr] = OV ? *(const uint32_t*)(pe + PT_UP + (u * BPS + rr) * 4) + pairoff : 0u;
					grain_unit<DEPTH, BW, OV, ONE, ONE && SUBX == 2, NEG>(lds, t, rp, up, lutb, rowoff, uprowoff, OV ? wc_ : 0, OV ? wu_ : 0, edge_on, first, lo2, hi2, a.pk_shift);
				}
				uint32_t o[DW];
				results(t, o);
				// the previous position's units are complete once the KD dwords its lanes computed have moved one lane down; its
				// lane 63 takes them from my lane 0 (the previous position of a row's first one is the last of another row:
				// its lane 63 lies behind that row's end and is never stored)
#pragma unroll
				for (int d = 0; d < KD; d++) outp[DW - KD + d] = lane_down(rot_down(o[d]), tp[d]);
				if (u == 0) store_unit<DW, STA>(pdst, laned + (NU - 1) * UBD, outp);
				else store_unit<DW, STA>(cdst, laned + (u - 1) * UBD, outp);
#pragma unroll
				for (int d = KD; d < DW; d++) outp[d - KD] = o[d];
#pragma unroll
				for (int d = 0; d < KD; d++) tp[d] = o[d];
#if VFGS_SCHED_FENCE
				__builtin_amdgcn_sched_barrier(0);
#endif
			}
			pdst = cdst;
		}
	};
	for (int h = 0;;)
	{
		// part h of my rows (all of them -- every group -- where the row's blocks fit one table: every BASELINE size)
		const int g_lo = h * GPP, g_hi = (h + 1 == nparts) ? ngroups : (h + 1) * GPP;
		for (int k = k0; k < k1; k++)
		{
			const int jrow = (base + RSTR * k - Rabs * RPB) * SUBY;
			if (Rabs > 0 && jrow <= 1) walk_row(std::true_type(), k, g_lo, g_hi);      // blends in the block above (vfgs_hw.c:173-188, 223-229)
			else walk_row(std::false_type(), k, g_lo, g_hi);
		}
		if (++h >= nparts) break;
		// the table of the next part: every wave is done reading this one; the LFSR words come out of L2 while the ring's
		// refills are in flight (wide pictures only: one row per wave, so the walk simply continues where it stopped)
		__syncthreads();
		u32x2 wcn[NPE], wun[NPE];
		param_loads(h * kTileBlocks, wcn, wun);
		param_table(h * kTileBlocks, wcn, wun);
		__syncthreads();
	}
	// the last position of my last row
#pragma unroll
	for (int d = 0; d < KD; d++) outp[DW - KD + d] = lane_down(0u, tp[d]);
	store_unit<DW, STA>(pdst, laned + (NU - 1) * UBD, outp);
	};     // task

	if constexpr (!PERSIST)
		task(f_in, r_in, true);
	else
	{
		// my tasks: t, t + P, t + 2 P, ... of the launch's nframes x pd.wgs luma tasks (frame-major: the persistent workgroups sweep
		// the frames in memory order together); the host passes P as (frames, tasks) so that nothing is divided here
		int f = f_in, r = r_in;
		for (bool first_task = true; f < a.nframes; first_task = false)
		{
			task(f, r, first_task);
			f += a.persist_step_f;
			r += a.persist_step_r;
			if (r >= pd.wgs) { r -= pd.wgs; f++; }
		}
	}
}

// Waves per SIMD the kernels are allocated for.  The 8-bit all-one-pattern kernels: SIX since round 6 -- the packed 16-bit form with a
// ring of two register sets needs 79 registers (round 5's form needed 97..100 and was held at five with one register spilled in the
// prologue, profiles/r03_ab22_lds_probes_and_occupancy.log), and six are worth 1.4 .. 3.4 % together with the shorter ring
// (profiles/r06_ab5_ring_depth_six_waves.log); rows walked in parts stay at four, the general-form kernels are held at four by their LDS
// image, the others by spills.
#ifndef VFGS_PK_WAVES
#define VFGS_PK_WAVES 6       // waves per SIMD of the 8-bit all-one-pattern kernels (packed 16-bit form with a ring of two: 79 VGPRs, no spill)
#endif
template <int DEPTH, bool ONEY, bool ONEC, bool WIDE>
constexpr int rw_waves_per_simd()
{
#ifdef VFGS_ONE10_WAVES      // developer A/B: the 10-bit all-one-pattern kernels at another occupancy
	if (DEPTH == 10 && ONEY && ONEC && !WIDE) return VFGS_ONE10_WAVES;
#endif
	return (DEPTH == 8 && ONEY && ONEC && !WIDE && VFGS_WG_PER_CU == 4) ? (kPk16 ? VFGS_PK_WAVES : 5) : (kWavesPerWG * VFGS_WG_PER_CU + 3) / 4;
}

// in place or out of place; workgroups numbered frame -> plane -> block row -> part of the block row
template <int DEPTH, int CSUBX, int CSUBY, bool OUT8, bool ONEY, bool ONEC, bool WIDE, bool PERSIST>
__global__ __launch_bounds__(kWavesPerWG * 64, (rw_waves_per_simd<DEPTH, ONEY, ONEC, WIDE>())) void grain_rw_kernel(const KernelArgs a, const FrameTable ft)
{
	constexpr ImageLayout L = image_layout(CSUBX, CSUBY, ONEY, ONEC, DEPTH == 8);
	// (unused LDS behind the tables holds a kernel at no more than its class's workgroups per CU: vfgs_layout.h lds_allocation)
	__shared__ __attribute__((aligned(16))) uint8_t lds[lds_allocation(DEPTH == 10, ONEY, ONEC, WIDE, L.lds_bytes + kParamBytes)];

	const int lane = threadIdx.x & 63;
	// wave-uniform by construction; telling the compiler keeps the decoding, row offsets and buffer descriptors in SGPRs
	// (otherwise every buffer instruction gets a waterfall loop)
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	// grid: x = (workgroup inside the frame, frame of a group of 2^lfronts frames), y = group of frames.  Large frames are swept
	// two at a time: the workgroups of frames 2m and 2m + 1 are dealt out alternately (measured: +1.6 % at 4320p, nothing at 2160p;
	// more than two, or smaller frames: a loss -- profiles/r03_ab18_frame_fronts_in_one_launch.log)
	int f, r;
	if constexpr (PERSIST)
	{
		// grid: x = [persistent luma workgroups | one workgroup per chroma task, frame-major]; one division per workgroup
		static_assert(!ONEY && !WIDE, "persistence exists for the general-form luma image");
		if ((int)blockIdx.x < a.persist_wgs)
		{
			f = (int)blockIdx.x / a.pd[0].wgs;
			r = (int)blockIdx.x - f * a.pd[0].wgs;
			run_plane_rw<DEPTH, 16, 1, 1, L.y_rs, L.y_bytes, ONEY, L.y_neg, 0, OUT8, WIDE, true>(a, ft, a.pd[0], lds, 0, f, r, L.y_off, L.y_bank, 0, lane, wave);
			return;
		}
		const int x = (int)blockIdx.x - a.persist_wgs, per = 2 * a.pd[1].wgs;
		f = x / per;
		r = a.pd[0].wgs + (x - f * per);
	}
	else
	{
		f = (int)(blockIdx.y << a.lfronts) + (int)(blockIdx.x & ((1u << a.lfronts) - 1));
		r = (int)(blockIdx.x >> a.lfronts);
	}
	if (f >= a.nframes) return;
	if (r < a.pd[0].wgs)
		run_plane_rw<DEPTH, 16, 1, 1, L.y_rs, L.y_bytes, ONEY, L.y_neg, 0, OUT8, WIDE, false>(a, ft, a.pd[0], lds, 0, f, r, L.y_off, L.y_bank, 0, lane, wave);
	else
	{
		r -= a.pd[0].wgs;
		const int comp = 1 + (r >= a.pd[1].wgs);
		if (comp == 2) r -= a.pd[1].wgs;
		// horizontally subsampled chroma rows of one or two positions (2 KiB and less: 1080p at 10 bit, 2160p at 8 bit): several rows per group
		if (!WIDE && CSUBX == 2 && a.pd[1].rw_segs == 2 && ring_depth<DEPTH, ONEC>() >= 2)      // (a ring of one set cannot hold two positions of a row)
			run_plane_rw<DEPTH, 16 / CSUBX, CSUBX, CSUBY, L.c_rs, L.c_bytes, ONEC, L.c_neg, WIDE ? 0 : 2, OUT8, WIDE, false>(a, ft, a.pd[1], lds, comp, f, r, L.c_off[comp - 1], L.c_bank, L.c_lut[comp - 1], lane, wave);
		else if (!WIDE && CSUBX == 2 && a.pd[1].rw_segs == 1)
			run_plane_rw<DEPTH, 16 / CSUBX, CSUBX, CSUBY, L.c_rs, L.c_bytes, ONEC, L.c_neg, WIDE ? 0 : 1, OUT8, WIDE, false>(a, ft, a.pd[1], lds, comp, f, r, L.c_off[comp - 1], L.c_bank, L.c_lut[comp - 1], lane, wave);
		else
			run_plane_rw<DEPTH, 16 / CSUBX, CSUBX, CSUBY, L.c_rs, L.c_bytes, ONEC, L.c_neg, 0, OUT8, WIDE, false>(a, ft, a.pd[1], lds, comp, f, r, L.c_off[comp - 1], L.c_bank, L.c_lut[comp - 1], lane, wave);
	}
}

// ---------------------------------------------------------------------------------------
// host-side launcher (called from vfgs_host.cpp)

template <int DEPTH, int CSUBX, int CSUBY, bool OUT8, bool ONEY, bool ONEC, bool WIDE, bool PERSIST>
static hipError_t launch_t(const KernelArgs& a, const FrameTable& ft, int grid, hipStream_t stream)
{
	const dim3 g = PERSIST ? dim3((unsigned)grid) : dim3((unsigned)grid << a.lfronts, ((unsigned)a.nframes + (1u << a.lfronts) - 1) >> a.lfronts);
	hipLaunchKernelGGL((grain_rw_kernel<DEPTH, CSUBX, CSUBY, OUT8, ONEY, ONEC, WIDE, PERSIST>), g, dim3(kWavesPerWG * 64), 0, stream, a, ft);
	return hipGetLastError();
}

template <int DEPTH, int CSUBX, int CSUBY, bool OUT8>
static hipError_t launch_form(const KernelArgs& a, const FrameTable& ft, bool oney, bool onec, bool wide, bool persist, int grid, hipStream_t stream)
{
	if (wide)
	{
		// rows walked in parts: the general form for every format; at 4:2:0 and 4:4:4 also one-pattern chroma under either luma form
		// (the default SEI model: eight luma patterns, one chroma pattern; AFGS1 and the single-pattern SEI models: one each)
		if (persist) return hipErrorInvalidValue;
		if constexpr (CSUBX == CSUBY)
		{
			if (oney && onec) return launch_t<DEPTH, CSUBX, CSUBY, OUT8, true, true, true, false>(a, ft, grid, stream);
			if (!oney && onec) return launch_t<DEPTH, CSUBX, CSUBY, OUT8, false, true, true, false>(a, ft, grid, stream);
		}
		return (oney || onec) ? hipErrorInvalidValue : launch_t<DEPTH, CSUBX, CSUBY, OUT8, false, false, true, false>(a, ft, grid, stream);
	}
	if constexpr (DEPTH == 10)     // (the host asks for persistence at 10 bit only)
	{
		if (persist)
		{
			if (oney) return hipErrorInvalidValue;
			return onec ? launch_t<DEPTH, CSUBX, CSUBY, OUT8, false, true, false, true>(a, ft, grid, stream)
			            : launch_t<DEPTH, CSUBX, CSUBY, OUT8, false, false, false, true>(a, ft, grid, stream);
		}
	}
	else if (persist) return hipErrorInvalidValue;
	if (oney && onec) return launch_t<DEPTH, CSUBX, CSUBY, OUT8, true, true, false, false>(a, ft, grid, stream);
	if (oney) return launch_t<DEPTH, CSUBX, CSUBY, OUT8, true, false, false, false>(a, ft, grid, stream);
	if (onec) return launch_t<DEPTH, CSUBX, CSUBY, OUT8, false, true, false, false>(a, ft, grid, stream);
	return launch_t<DEPTH, CSUBX, CSUBY, OUT8, false, false, false, false>(a, ft, grid, stream);
}

// out8: the destination holds 8-bit samples of a 10-bit path; oney / onec: the image holds the one-pattern form for luma /
// chroma (vfgs_layout.h); wide: rows of more than kTileBlocks blocks (launch_form: which forms exist); persist: a.persist_wgs luma workgroups
// share the launch's luma tasks (10 bit, general-form luma, not wide), grid = persist_wgs + all chroma tasks; else grid =
// workgroups per frame
//
// The product compiles this file TWICE (versatilefilmgrain_amd/build.py: -DVFGS_KERNEL_DEPTH=10 and =8), side by side: one code object per
// sample depth, 56 + 32 kernels.  That halves the build (110 -> 60 s) and keeps the other depth's kernels off the device; it does NOT buy the
// start-up time round 5 hoped for: a code object's first launch costs 1.3 ms (profiles/r06_startup_probe_two_code_objects.jsonl), so the 15 ms
// of a process's first grain launch are the allocations and first transfers around it, not the 88 kernels.  Without the macro (the assembly
// listings) everything is one translation unit.
template <int D>
static hipError_t launch_depth(const KernelArgs& a, const FrameTable& ft, int csubx, int csuby, bool out8, bool oney, bool onec, bool wide, bool persist, int grid, hipStream_t stream)
{
#define VFGS_CASE(X, Y)                                                                                                     \
	if (csubx == X && csuby == Y)                                                                                           \
	{                                                                                                                       \
		if constexpr (D == 10) { if (out8) return launch_form<D, X, Y, true>(a, ft, oney, onec, wide, persist, grid, stream); } \
		return launch_form<D, X, Y, false>(a, ft, oney, onec, wide, persist, grid, stream);                                     \
	}
	VFGS_CASE(2, 2) VFGS_CASE(2, 1) VFGS_CASE(1, 1) VFGS_CASE(1, 2)
#undef VFGS_CASE
	return hipErrorInvalidValue;
}

#if !defined(VFGS_KERNEL_DEPTH) || VFGS_KERNEL_DEPTH == 8
hipError_t launch_grain_d8(const KernelArgs& a, const FrameTable& ft, int csubx, int csuby, bool out8, bool oney, bool onec, bool wide, bool persist, int grid, hipStream_t stream)
{
	return launch_depth<8>(a, ft, csubx, csuby, out8, oney, onec, wide, persist, grid, stream);
}
#else
hipError_t launch_grain_d8(const KernelArgs& a, const FrameTable& ft, int csubx, int csuby, bool out8, bool oney, bool onec, bool wide, bool persist, int grid, hipStream_t stream);
#endif

#if !defined(VFGS_KERNEL_DEPTH) || VFGS_KERNEL_DEPTH == 10
hipError_t launch_grain(const KernelArgs& a, const FrameTable* list, int depth, int csubx, int csuby, bool out8, bool oney, bool onec, bool wide, bool persist, int grid, hipStream_t stream)
{
	if ((out8 && depth != 10) || wide != (a.nblk > kTileBlocks)) return hipErrorInvalidValue;
	if ((a.listed != 0) != (list != nullptr) || (list && a.nframes > kListFrames)) return hipErrorInvalidValue;
	static const FrameTable no_list{};
	const FrameTable& ft = list ? *list : no_list;
	if (depth == 10) return launch_depth<10>(a, ft, csubx, csuby, out8, oney, onec, wide, persist, grid, stream);
	if (depth == 8) return launch_grain_d8(a, ft, csubx, csuby, out8, oney, onec, wide, persist, grid, stream);
	return hipErrorInvalidValue;
}

// the name of the instantiation launch_grain() dispatches for these arguments, as the profiler prints it (vfgs_hip_last_launch_info)
void describe_launch(char* out, size_t n, int depth, int csubx, int csuby, bool out8, bool oney, bool onec, bool wide, bool persist)
{
	auto b = [](bool v) { return v ? "true" : "false"; };
	snprintf(out, n, "grain_rw_kernel<%d,%d,%d,%s,%s,%s,%s,%s>", depth, csubx, csuby, b(out8), b(oney), b(onec), b(wide), b(persist));
}

ImageLayout layout_of(int csubx, int csuby, bool oney, bool onec, bool depth8) { return image_layout(csubx, csuby, oney, onec, depth8); }
#endif

}  // namespace vfgs
