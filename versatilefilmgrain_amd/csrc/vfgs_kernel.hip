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
// Work decomposition (DESIGN.md "kernel"):
//   * every lane moves 16 bytes per access (8 samples at 10 bit, 16 samples at 8 bit), one
//     wave access = one contiguous <= 1 KiB "segment" of a row, 4 segments = one "tile";
//   * the unit grid is shifted left of the block grid by half a block, so every block edge --
//     the only place where a sample depends on its horizontal neighbours -- lies inside a lane
//     or between the two lanes of a pair: no halo, no inter-wave exchange, in place is race free;
//   * a WAVEFRONT owns one tile and walks kRowsPerWave rows of ONE block row down that tile: the
//     lane geometry and the block parameters (LFSR window -> sign, pattern offsets) are computed
//     once and reused for every row; while a segment is being computed the same registers of the
//     segment after next are already being refilled ("rolling prefetch": the registers of a
//     segment are reloaded with the next row right after its store);
//   * a WORKGROUP of 4 waves covers all tiles of a few consecutive rows (full rows, contiguous in
//     memory), is NOT persistent and copies only its plane's banks + LUTs to LDS; workgroups are
//     numbered in memory order, so the chip sweeps the frames front to back with a compact window
//     and the hardware dispatcher balances the load;
//   * a component whose pattern LUT selects one slot for every intensity is served from a packed
//     one-byte-per-sample bank (ONEY / ONEC kernels, vfgs_layout.h).
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

// A block's parameters in ONE register per run (they stay resident for all rows of a wave; k2, the rounding constants and
// the relative signs of the edge filter are rebuilt from it for every row with a handful of instructions, which is
// what keeps the kernel at 6 waves per SIMD):
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
template <int SUBX, int SUBY, int RS, bool ONE>
__device__ __forceinline__ uint32_t block_param(uint32_t v, uint32_t bank_off, int sx, int sy, int sb, bool* neg)
{
	const uint32_t fx = (v >> sx) & 0x3ff;
	const uint32_t fy = __builtin_amdgcn_alignbit(v, v, sy) & 0x3ff;
	const uint32_t ox = (__umul24(fx, 13u) >> 10) * (4 / SUBX);
	const uint32_t oy = (__umul24(fy, 12u) >> 10) * (4 / SUBY);
	*neg = (v >> sb) & 1;
	return __umul24(oy, (uint32_t)RS) + ox * (ONE ? 1 : kSlots) + bank_off;     // bytes per sample position: 1 or one per slot
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
                                            const bool first, const uint32_t lo2, const uint32_t hi2)
{
	constexpr int NS = DEPTH == 8 ? 16 : 8;
	using M = LaneMap<NS, BW>;
	constexpr int NR = M::NR, NQ = M::NQ;
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
	uint32_t e[NS];
	int P[NS];

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
	auto fetch4 = [&](uint32_t adq, int q, int (&out)[4]) {
		if (ONE)
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
		fetch4(ad[M::run(q)], q, v4);
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
			fetch4(uad, q, Q);
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
	auto clip2 = [&](uint32_t v, int p0, int p1, uint32_t e0, uint32_t e1) {
		const int g0 = mad_i24(p0, e0, 0x8000);
		const int g1 = mad_i24(p1, e1, 0x8000);
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
			w[d] = clip2(w[d], P[2 * d], P[2 * d + 1], e[2 * d], e[2 * d + 1]);
	}
	else
	{
#pragma unroll
		for (int d = 0; d < 4; d++)
		{
			const uint32_t v01 = __builtin_amdgcn_perm(0, w[d], 0x0c010c00), v23 = __builtin_amdgcn_perm(0, w[d], 0x0c030c02);
			const uint32_t s01 = clip2(v01, P[4 * d], P[4 * d + 1], e[4 * d], e[4 * d + 1]);
			const uint32_t s23 = clip2(v23, P[4 * d + 2], P[4 * d + 3], e[4 * d + 2], e[4 * d + 3]);
			w[d] = __builtin_amdgcn_perm(s23, s01, 0x06040200);
		}
	}
}

// ---------------------------------------------------------------------------------------
// One workgroup's share of one plane.
//
// Partly valid lanes.  Where a lane is not half a block (every plane type but PAIR) the first lane of a row
// begins before the row and the last one ends behind it.  Such a lane LOADS the 16 bytes at its offset clamped
// into the row (all bytes it reads belong to the row; the surplus is its neighbour's data and is ignored), then
// rotates its dwords into place; it STORES only its own dwords, with dword (or 8-byte) stores that every
// segment issues and that are switched off (kOOB) in all other lanes.  The instruction stream therefore does
// not depend on where those lanes are -- which is what lets the compiler count outstanding refills (a
// wave-uniform branch around a memory instruction makes it wait for everything instead).

template <int AUX = VFGS_LDAUX>
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

template <int AUX = VFGS_STAUX>
__device__ __forceinline__ void store_b128(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff, const uint32_t (&w)[4])
{
	const u32x4 t = {w[0], w[1], w[2], w[3]};
	__builtin_amdgcn_raw_buffer_store_b128(t, rs, voff, soff, AUX);
	store_data_hazard();
}

// the first / last N dwords of a unit (aligned mode: the part of a unit that belongs to the neighbouring tile's wave)
template <int N, int AUX>
__device__ __forceinline__ void load_dwords(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff, uint32_t (&w)[4])
{
	if (N == 4) { const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, AUX); w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w; }
	else if (N == 2) { const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, AUX); w[0] = t.x; w[1] = t.y; }
	else w[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, AUX);
}

template <int N, int AUX>
__device__ __forceinline__ void store_dwords(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff, const uint32_t* w)
{
	if (N == 4) { const u32x4 t = {w[0], w[1], w[2], w[3]}; __builtin_amdgcn_raw_buffer_store_b128(t, rs, voff, soff, AUX); store_data_hazard(); }
	else if (N == 3) { const u32x3 t = {w[0], w[1], w[2]}; __builtin_amdgcn_raw_buffer_store_b96(t, rs, voff, soff, AUX); store_data_hazard(); }
	else if (N == 2) { const u32x2 t = {w[0], w[1]}; __builtin_amdgcn_raw_buffer_store_b64(t, rs, voff, soff, AUX); }
	else if (N == 1) __builtin_amdgcn_raw_buffer_store_b32(w[0], rs, voff, soff, AUX);
}

template <int DEPTH, int BW, int SUBX, int SUBY, int RS, bool OUT8, int IMG_BYTES, bool ONE, bool AL, int NEG>
__device__ __forceinline__ void run_plane(const KernelArgs& a, const PlaneDesc& pd, uint8_t* lds, const int comp, const int f, int r,
                                          const uint32_t img_off, const uint32_t bank_off, const uint32_t lut_off, const int lane, const int wave)
{
	constexpr int NS = DEPTH == 8 ? 16 : 8;
	constexpr int SZ = DEPTH > 8 ? 2 : 1;
	using M = LaneMap<NS, BW>;
	constexpr int NR = M::NR;
	constexpr int RPB = 16 / SUBY;                       // rows of this plane per block row
	constexpr int NEF = M::PAIR ? 1 : M::NE;
	constexpr bool PARTIAL = !AL && !M::PAIR;   // rows of this plane type begin and end with a partly valid lane
	constexpr int K = M::SHIFT * SZ / 4;                 // aligned mode: dwords of a lane that lie in the memory unit before the lane's own
	constexpr int LDA = AL ? VFGS_LDAUX_ALIGNED : VFGS_LDAUX, STA = AL ? VFGS_STAUX_ALIGNED : VFGS_STAUX;
	static_assert(!(AL && OUT8), "the narrowed destination keeps the shifted accesses");
	constexpr bool HALVES = !(NS == 16 && BW == 8);      // ... whose valid part is one 8-byte half (else: 1 or 3 dwords)
	const int pt = comp ? 1 : 0;

	// ---- the workgroup's place: block row group, part of the block row, column group; the wave's place inside it.
	// All wave-uniform; readfirstlane tells the compiler (runtime divisions run on the vector ALU), so that row offsets
	// and descriptors stay in SGPRs instead of waterfall loops around every buffer instruction.
	auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
	// (tiles_w, ppb, splits are powers of two -- shifts; only a picture wider than kWavesPerWG tiles has column groups)
	int colgroup = 0;
	if (pd.colgroups > 1) { colgroup = r % pd.colgroups; r /= pd.colgroups; }
	const int split = r & (pd.splits - 1);
	const int bgroup = r >> pd.lsplits;
	const int tile = uni(colgroup * pd.tiles_w + (wave & (pd.tiles_w - 1)));
	const int q = wave >> pd.ltiles_w;
	const int ph = q & (pd.ppb - 1);
	const int kbr = uni(bgroup * pd.bpw + (q >> pd.lppb));   // block row inside the stripe
	const int Rabs = (a.y0 >> 4) + kbr;                  // absolute block row
	const int row_first = (a.y0 + SUBY - 1) / SUBY;      // first row of the stripe in this plane; the plane pointers address row y0 / SUBY
	const int prow0 = a.y0 / SUBY;
	const bool active = (tile < pd.tiles) && (kbr < a.nbrows);
	// rows of this block row that belong to the stripe: [alo, ahi); mine: base + stp * k, k in [k0, k1)
	const int alo = max(row_first, Rabs * RPB), ahi = min(row_first + pd.nrows, (Rabs + 1) * RPB);
#if VFGS_SPLIT_INTERLEAVE
	// the parts of a block row are interleaved: the workgroups of one block row, dispatched back to back, sweep it together
	const int lstp = pd.lppb + pd.lsplits, stp = 1 << lstp;
	const int base = uni(Rabs * RPB + (split << pd.lppb) + ph);
#else
	const int lstp = pd.lppb, stp = pd.ppb;
	const int base = uni(Rabs * RPB + split * (RPB >> pd.lsplits) + ph);
#endif
	const int nk = (RPB >> pd.lsplits) >> pd.lppb;
	int k0 = max(0, alo - base + stp - 1) >> lstp, k1 = min(nk, (max(0, ahi - base) + stp - 1) >> lstp);
	if (!active) k1 = k0 = 0;
	k0 = uni(k0); k1 = uni(k1);

	// ---- lane geometry of the 4 segments (once per wave) -----------------------------------------
	const uint8_t* sbase = a.src[comp] + (uint64_t)f * pd.fpitch;
	const __amdgpu_buffer_rsrc_t drs = make_rsrc(a.dst[comp] + (uint64_t)f * pd.dfpitch, pd.dextent);
	const __amdgpu_buffer_rsrc_t strs = make_rsrc((const uint8_t*)a.stream, a.stream_bytes);
	const int last = a.nblk - 1;
	const uint32_t cur_bit = a.cur_bit0 + (uint32_t)f * a.frame_bit_step + (uint32_t)(kbr * a.nblk);
	const uint32_t up_bit = (kbr > 0) ? cur_bit - (uint32_t)a.nblk : a.up_bit0 + (uint32_t)f * a.frame_bit_step;
	const bool any_up = (Rabs > 0) && ((base + stp * k0 - Rabs * RPB) * SUBY <= 1);   // my first row is an overlap line (vfgs_hw.c:175,180)

	uint32_t vo[4];                    // byte offset inside a row of the 16 bytes the lane LOADS, or kOOB
	bool fullm[4];                     // the lane lies completely inside the row: it stores its 16 bytes at vo
	bool anypart[4];                   // the segment holds a partly valid lane (wave-uniform)
	bool edge_on[4][NEF];
	bool first = false;
	int blk[4][NR];                    // blocks of the lane's runs, clamped to the row
#pragma unroll
	for (int g = 0; g < 4; g++)
	{
		const int seg = tile * kSegsPerTile + g;
		const int p = seg * pd.upt + lane;                                 // lane position along the row
		const bool sok = (seg < pd.segs) && (lane < pd.upt);
		const int x = p * 16 - M::SHIFT * SZ;                              // first byte of the lane in the row
		const bool full = AL ? (sok && p * 16 + 16 <= (int)pd.rowbytes) : (sok && x >= 0 && x + 16 <= (int)pd.rowbytes);
		const bool part = PARTIAL && sok && !full && x + 16 > 0 && x < (int)pd.rowbytes;
		fullm[g] = full;
		anypart[g] = PARTIAL && __builtin_amdgcn_ballot_w64(part) != 0;
		// aligned mode: the lane MOVES memory unit p (bytes [16p, 16p + 16) of the row) and COMPUTES bytes [x, x + 16)
		vo[g] = full ? (uint32_t)(AL ? p * 16 : x) : (part ? (uint32_t)min(max(x, 0), (int)pd.rowbytes - 16) : kOOB);
		if (M::PAIR)
		{
			const int ju = p - 1;                                          // 8-sample unit of the row
			first = !(ju & 1);
			blk[g][0] = min(max(ju >> 1, 0), last);
			const int jl = first ? ju - 1 : ju;                            // left unit of this lane pair
			edge_on[g][0] = sok && (jl >= 0) && (jl + 1 < 2 * a.nblk);
		}
		else
		{
			const int b0 = p * M::BPL - 1;                                 // block of run 0
#pragma unroll
			for (int rr = 0; rr < NR; rr++) blk[g][rr] = min(max(b0 + rr, 0), last);
#pragma unroll
			for (int ed = 0; ed < M::NE; ed++) edge_on[g][ed] = sok && (b0 + ed >= 0) && (b0 + ed + 1 <= last);
		}
	}
	// a partly valid lane, after its load from the clamped offset: move its dwords to where they belong
	auto rotate_partial = [&](int g, uint32_t (&t)[4]) {
		const int seg = tile * kSegsPerTile + g;
		int x = (seg * pd.upt + lane) * 16 - M::SHIFT * SZ;
		asm volatile("" : "+v"(x));       // opaque to the optimiser: keeps this rarely needed arithmetic out of the registers of the row loop
		const bool part = !fullm[g] && vo[g] != kOOB;
		if (HALVES)
		{   // the lane holds bytes [x', x' + 16) with x' = x +- 8: its own half sits in the other half of the registers
			const uint32_t t0 = t[0], t1 = t[1];
			t[0] = part ? t[2] : t0; t[1] = part ? t[3] : t1;
			t[2] = part ? t0 : t[2]; t[3] = part ? t1 : t[3];
		}
		else
		{   // x' - x = +4 (first lane), -4 or -12 (last lane): rotate by one dword, and by two more where it is -12
			const int kk = ((x - (int)vo[g]) >> 2) & 3;                     // t[d] belongs at d - kk
			const bool r1 = part && (kk & 1), r2 = part && (kk & 2);
			uint32_t u[4];
#pragma unroll
			for (int d = 0; d < 4; d++) u[d] = r1 ? t[(d + 1) & 3] : t[d];
#pragma unroll
			for (int d = 0; d < 4; d++) t[d] = r2 ? u[(d + 2) & 3] : u[d];
		}
	};
	// ... and its stores: only the dwords inside the row (all other lanes: kOOB)
	auto store_partial = [&](int g, uint32_t soff, const uint32_t (&t)[4]) {
		const int seg = tile * kSegsPerTile + g;
		int x = (seg * pd.upt + lane) * 16 - M::SHIFT * SZ;
		asm volatile("" : "+v"(x));
		const bool part = !fullm[g] && vo[g] != kOOB;
		if (!OUT8)
		{
			if (HALVES)
			{   // the valid half: the upper one of the first lane (x < 0), the lower one of the last lane
				const bool upper = x < 0;
				const u32x2 d = {upper ? t[2] : t[0], upper ? t[3] : t[1]};
				__builtin_amdgcn_raw_buffer_store_b64(d, drs, part ? (uint32_t)(upper ? 0 : x) : kOOB, soff, VFGS_STAUX);
			}
			else
			{
#pragma unroll
				for (int d = 0; d < 4; d++)
				{
					const bool in = part && x + 4 * d >= 0 && x + 4 * d + 4 <= (int)pd.rowbytes;
					__builtin_amdgcn_raw_buffer_store_b32(t[d], drs, in ? (uint32_t)(x + 4 * d) : kOOB, soff, VFGS_STAUX);
				}
			}
		}
		else
		{   // narrowed destination: 2 dwords per lane, the valid one is the upper one of the first lane, the lower one of the last lane
			const bool upper = x < 0;
			__builtin_amdgcn_raw_buffer_store_b32(upper ? t[1] : t[0], drs, part ? (uint32_t)(upper ? 0 : x / 2) : kOOB, soff, VFGS_STAUX);
		}
	};

	// ---- in flight together: the table image, the LFSR windows of my blocks, the first row's samples -----
	// In THIS order: a wave's loads return in the order they were issued, so the table image and the LFSR words (L2 hits,
	// ~1 us under load) must not queue behind the first row's samples (HBM, several us under load) -- the workgroup's
	// barrier and the block parameters are then done by the time the samples arrive (measured with s_memrealtime marks:
	// the barrier was passed 5.8 us after wave start with the samples first).  The instruction stream is
	// fixed (clamped addresses and zero-record descriptors instead of branches) so that the waits can be counted.
	constexpr int STEP = kWavesPerWG * 64 * 16;
	constexpr int NIT = (IMG_BYTES + STEP - 1) / STEP;
	static_assert(IMG_BYTES % 16 == 0, "table image in whole 16-byte units");
	u32x4 tmp[NIT];
	{
		const __amdgpu_buffer_rsrc_t irs = make_rsrc(a.tables + img_off, IMG_BYTES);
#pragma unroll
		for (int it = 0; it < NIT; it++)      // threads beyond the image re-read (and re-write) its last unit
			tmp[it] = __builtin_amdgcn_raw_buffer_load_b128(irs, min((uint32_t)(threadIdx.x * 16 + it * STEP), (uint32_t)(IMG_BYTES - 16)), 0, 0);
	}
	u32x2 wcur[4][NR], wup[4][NR];
	{
		const __amdgpu_buffer_rsrc_t strs_up = make_rsrc((const uint8_t*)a.stream, any_up ? a.stream_bytes : 0);
#pragma unroll
		for (int g = 0; g < 4; g++)
#pragma unroll
			for (int rr = 0; rr < NR; rr++)
			{
				wcur[g][rr] = __builtin_amdgcn_raw_buffer_load_b64(strs, ((cur_bit + (uint32_t)blk[g][rr]) >> 5) * 4, 0, 0);
				wup[g][rr] = __builtin_amdgcn_raw_buffer_load_b64(strs_up, ((up_bit + (uint32_t)blk[g][rr]) >> 5) * 4, 0, 0);
			}
	}
	// aligned mode: the last K dwords of the unit before the tile are this wave's (lane 0 of segment 0 computes them), the
	// last K dwords of the tile's last unit are the next wave's
	uint32_t pre[4] = {0, 0, 0, 0};
	const uint32_t preoff = (AL && lane == 0 && tile > 0 && tile * kSegsPerTile < pd.segs) ? (uint32_t)(tile * (kSegsPerTile * kMaxUnits * 16) - K * 4) : kOOB;
	const uint32_t vos3 = (AL && lane == 63) ? kOOB : vo[3];
	const uint32_t tailoff = (AL && lane == 63) ? vo[3] : kOOB;
	uint32_t rowb = (uint32_t)uni((base + stp * k0 - prow0) * (int)pd.pitch), drowb = (uint32_t)uni((base + stp * k0 - prow0) * (int)pd.dpitch);
	const uint32_t rstep = (uint32_t)stp * pd.pitch, drstep = (uint32_t)stp * pd.dpitch;
	uint32_t w[4][4];
	{
		const __amdgpu_buffer_rsrc_t frs = make_rsrc(sbase, (k0 < k1) ? pd.extent : 0);
#pragma unroll
		for (int g = 0; g < 4; g++) load_seg<LDA>(frs, vo[g], rowb, w[g]);
		if (AL) load_dwords<K, LDA>(frs, preoff, rowb, pre);
	}
#pragma unroll
	for (int it = 0; it < NIT; it++)
		*(u32x4*)(lds + min((uint32_t)(threadIdx.x * 16 + it * STEP), (uint32_t)(IMG_BYTES - 16))) = tmp[it];
	__syncthreads();
	if (k0 >= k1)
		return;

	// ---- block parameters (once per wave) ------------------------------------------------------------
	const int fsx = comp == 0 ? 0 : (comp == 1 ? 10 : 20);
	const int fsy = comp == 0 ? 14 : (comp == 1 ? 24 : 4);
	const int fsb = comp == 0 ? 31 : (comp == 1 ? 2 : 15);
	const uint32_t lutb = lut_off * 0x10001u;
	const uint32_t lo2 = a.lo2[pt], hi2 = a.hi2[pt];
	const uint32_t pairoff = M::PAIR ? (first ? 0u : 8u * (ONE ? 1 : kSlots)) : 0u;
	RunParam<NR> rp[4];
#pragma unroll
	for (int g = 0; g < 4; g++)
#pragma unroll
		for (int rr = 0; rr < NR; rr++)
		{
			const uint32_t v = __builtin_amdgcn_alignbit(wcur[g][rr].y, wcur[g][rr].x, (cur_bit + (uint32_t)blk[g][rr]) & 31);
			bool neg;
			const uint32_t ad = block_param<SUBX, SUBY, RS, ONE>(v, bank_off, fsx, fsy, fsb, &neg) + pairoff + ((ONE && neg) ? (uint32_t)NEG : 0u);
			rp[g].pa[rr] = ad | (neg ? 0x80000000u : 0u);
		}

	// ---- rows ---------------------------------------------------------------------------------------
	// (all four segments always run -- lanes of segments beyond the row carry kOOB -- so that the instruction stream is
	// fixed and the compiler can count the outstanding refills instead of waiting for all of them)
	auto row = [&](auto overlap, const int k, const RunParam<NR> (&up)[4], const int wc_, const int wu_) {
		constexpr bool OV = decltype(overlap)::value;
		const int j = base + stp * k - Rabs * RPB;                        // row inside the block row
		const uint32_t rowoff = (uint32_t)j * RS, uprowoff = (uint32_t)(RPB + j) * RS;
		// the refill of the row after my last one goes through a descriptor with zero records: the hardware
		// drops it, the instruction stream (and the compiler's vmcnt counting) stays the same
		const __amdgpu_buffer_rsrc_t nrs = make_rsrc(sbase, (k + 1 < k1) ? pd.extent : 0);
#pragma unroll
		for (int g = 0; g < 4; g++)
		{
#if !VFGS_PREFETCH
			load_seg(make_rsrc(sbase, pd.extent), vo[g], rowb, w[g]);
#endif
			uint32_t t[4] = {w[g][0], w[g][1], w[g][2], w[g][3]};
			if (PARTIAL && anypart[g]) rotate_partial(g, t);
			grain_unit<DEPTH, BW, OV, ONE, ONE && SUBX == 2, NEG>(lds, t, rp[g], up[g], lutb, rowoff, uprowoff, wc_, wu_, edge_on[g], first, lo2, hi2);
			if (OUT8)
			{
				uint32_t n[4];
#pragma unroll
				for (int d = 0; d < 4; d++)   // yuv_to_8bit (yuv.c:216-258): out8 = (v + 2) >> 2; both halves <= 1022: no carry across the halves
					n[d] = ((t[d] + 0x00020002u) >> 2) & 0x00ff00ffu;
				uint32_t o[4] = {__builtin_amdgcn_perm(n[1], n[0], 0x06040200), __builtin_amdgcn_perm(n[3], n[2], 0x06040200), 0, 0};
				const u32x2 d2 = {o[0], o[1]};
				__builtin_amdgcn_raw_buffer_store_b64(d2, drs, fullm[g] ? vo[g] >> 1 : kOOB, drowb, VFGS_STAUX);
				if (PARTIAL) store_partial(g, drowb, o);
			}
			else
			{
				store_b128(drs, fullm[g] ? vo[g] : kOOB, drowb, t);
				if (PARTIAL) store_partial(g, drowb, t);
			}
#if VFGS_PREFETCH
			load_seg(nrs, vo[g], rowb + rstep, w[g]);
#endif
#if VFGS_SCHED_FENCE
			// keep the refill where it is: with registers to spare hipcc's scheduler otherwise computes all four segments
			// first and issues the four stores and the four refills together at the end of the row -- every row would then
			// start with a full memory latency
			__builtin_amdgcn_sched_barrier(0);
#endif
		}
		rowb += rstep;
		drowb += drstep;
	};

	// Aligned mode.  The memory side moves whole aligned units (line-aligned 1 KiB wave accesses, nontemporal: DESIGN.md 4);
	// the computation keeps the half-block shifted lanes.  A lane's 4 dwords are the last K dwords of the unit of the lane
	// BEFORE it (DPP wave_shr:1; lane 0 takes them from lane 63 of the previous segment through SGPRs, or from the unit in
	// front of the tile) and the first 4 - K of its own; results go back the same way, so the store of a segment waits for
	// lane 0 of the NEXT segment.  The registers of a segment are refilled as soon as the lanes have been assembled.
	auto row_al = [&](auto overlap, const int k, const RunParam<NR> (&up)[4], const int wc_, const int wu_) {
		constexpr bool OV = decltype(overlap)::value;
		const int j = base + stp * k - Rabs * RPB;
		const uint32_t rowoff = (uint32_t)j * RS, uprowoff = (uint32_t)(RPB + j) * RS;
		const __amdgpu_buffer_rsrc_t nrs = make_rsrc(sbase, (k + 1 < k1) ? pd.extent : 0);
		const bool is0 = lane == 0, is63 = lane == 63;
#if VFGS_LANE_SHIFT_DPP
		auto lane_up = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false); };    // wave_shr:1
		auto lane_down = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false); };  // wave_shl:1
#else
		const int a_prev = ((lane - 1) & 63) * 4, a_next = ((lane + 1) & 63) * 4;
		auto lane_up = [&](uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute(a_prev, (int)v); };
		auto lane_down = [&](uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute(a_next, (int)v); };
#endif
		uint32_t carry[4] = {0, 0, 0, 0};      // wave-uniform: the last K dwords of lane 63 of the previous segment
		uint32_t outp[4] = {0, 0, 0, 0};       // the previous segment's units, complete but for lane 63
#pragma unroll
		for (int g = 0; g < 4; g++)
		{
			uint32_t t[4];
#pragma unroll
			for (int d = 0; d < K; d++)
			{
				const uint32_t sh = lane_up(w[g][4 - K + d]);   // lane l <- lane l - 1
				t[d] = is0 ? (g == 0 ? pre[d] : carry[d]) : sh;
			}
#pragma unroll
			for (int d = 0; d < K; d++) carry[d] = (uint32_t)__builtin_amdgcn_readlane((int)w[g][4 - K + d], 63);
#pragma unroll
			for (int d = K; d < 4; d++) t[d] = w[g][d - K];
			load_seg<LDA>(nrs, vo[g], rowb + rstep, w[g]);
			if (g == 0) load_dwords<K, LDA>(nrs, preoff, rowb + rstep, pre);
			grain_unit<DEPTH, BW, OV, ONE, ONE && SUBX == 2, NEG>(lds, t, rp[g], up[g], lutb, rowoff, uprowoff, wc_, wu_, edge_on[g], first, lo2, hi2);
			if (g == 0)
			{
				store_dwords<K, STA>(drs, preoff, drowb, t);               // lane 0: the tail of the unit in front of the tile
			}
			else
			{
#pragma unroll
				for (int d = 0; d < K; d++)
				{
					const uint32_t l0 = (uint32_t)__builtin_amdgcn_readlane((int)t[d], 0);
					outp[4 - K + d] = is63 ? l0 : outp[4 - K + d];
				}
				store_b128<STA>(drs, vo[g - 1], drowb, outp);
			}
#pragma unroll
			for (int d = K; d < 4; d++) outp[d - K] = t[d];
#pragma unroll
			for (int d = 0; d < K; d++) outp[4 - K + d] = lane_down(t[d]);   // lane l <- lane l + 1
#if VFGS_SCHED_FENCE
			__builtin_amdgcn_sched_barrier(0);
#endif
		}
		store_b128<STA>(drs, vos3, drowb, outp);
		if (K < 4) store_dwords<4 - K, STA>(drs, tailoff, drowb, outp);       // lane 63: the head of the tile's last unit
		rowb += rstep;
		drowb += drstep;
	};
	auto row_any = [&](auto overlap, const int k, const RunParam<NR> (&up)[4], const int wc_, const int wu_) {
		if constexpr (AL) row_al(overlap, k, up, wc_, wu_); else row(overlap, k, up, wc_, wu_);
	};

	int k = k0;
	if (any_up)
	{
		// the lines j = 0, 1 of a block row (luma lines; a vertically subsampled plane has only j = 0) blend in
		// the block above (vfgs_hw.c:173-188, 223-229); they come first in my walk
		RunParam<NR> up[4];
#pragma unroll
		for (int g = 0; g < 4; g++)
#pragma unroll
			for (int rr = 0; rr < NR; rr++)
			{
				const uint32_t v = __builtin_amdgcn_alignbit(wup[g][rr].y, wup[g][rr].x, (up_bit + (uint32_t)blk[g][rr]) & 31);
				bool neg;
				const uint32_t ad = block_param<SUBX, SUBY, RS, ONE>(v, bank_off, fsx, fsy, fsb, &neg) + pairoff;
				up[g].pa[rr] = ad | (neg ? 0x80000000u : 0u);
			}
		for (; k < k1; k++)
		{
			const int jrow = (base + stp * k - Rabs * RPB) * SUBY;
			if (jrow > 1) break;
			const int wc_ = jrow == 0 ? (SUBY > 1 ? 20 : 12) : 24, wu_ = jrow == 0 ? (SUBY > 1 ? 20 : 24) : 12;
			row_any(std::true_type(), k, up, wc_, wu_);
		}
	}
	{
		RunParam<NR> none[4] = {};
		for (; k < k1; k++)
			row_any(std::false_type(), k, none, 0, 0);
	}
}

// ---------------------------------------------------------------------------------------
// Row walk: the product path for pictures whose rows hold at most kTileBlocks grain blocks (8192 luma samples).
//
// tools/skeleton2.hip (profiles/r03_skeleton2_*.log) priced what the tiled kernels above pay besides their bytes: every
// vector-memory INSTRUCTION queues for the CU's saturated memory pipeline, and a 4 KiB tile costs 11 of them where 8 move
// data (the aligned kernels' narrow accesses at both tile edges), plus 8 LFSR loads per wave: 0.73 -> 0.67 of 8 TB/s.
// Here a wave owns whole ROWS instead of a tile of several rows:
//   * a workgroup = 4 waves = 4 x rw_rpw rows of ONE block row (wave w: rows w, w + 4, ...; the same ~60 KB and the same
//     table image per workgroup as before); a wave streams its row segment by segment (64 aligned 16-byte units each)
//     through a ring of four register sets -- the refill of a set is the segment four steps ahead, across row ends --
//     so a row costs one load and one store per KiB and nothing else: no tile edges, no narrow accesses;
//   * the block parameters of the row's <= 512 blocks (this block row's LFSR registers and, for the workgroup that holds
//     the overlap lines, those of the block row above) are computed ONCE per workgroup, one or two blocks per thread, and
//     kept in LDS behind the table image; a lane reads its 1-3 entries per segment (the tiled kernels hold 4 segments x
//     1-3 runs x 2 in registers: the 8-bit 4:2:x kernels spilled on that);
//   * lanes compute the half-block shifted bytes as in the aligned kernels: rotation by one lane with DPP wave_shr / wave_shl,
//     the hand-over between consecutive segments of a row with wave_ror / wave_rol (the previous segment's lane 63 waits in
//     lane 0 of a register and enters as the `old` operand of the shift: no v_readlane, no scalar round trip); a segment's
//     units are stored one step later, once lane 0 of the next segment has delivered the last dwords of its lane 63;
//   * row bases and segment offsets live in the buffer descriptor (base, num_records = bytes of the row left), so the
//     hardware range check covers every access of every lane: lanes behind the row's end load 0 and store nothing.
template <int DEPTH, int BW, int SUBX, int SUBY, int RS, int IMG_BYTES, bool ONE, int NEG, int NARROW>
__device__ __forceinline__ void run_plane_rw(const KernelArgs& a, const PlaneDesc& pd, uint8_t* lds, const int comp, const int f, const int r,
                                             const uint32_t img_off, const uint32_t bank_off, const uint32_t lut_off, const int lane, const int wave)
{
	constexpr int NS = DEPTH == 8 ? 16 : 8;
	constexpr int SZ = DEPTH > 8 ? 2 : 1;
	using M = LaneMap<NS, BW>;
	constexpr int NR = M::NR;
	constexpr int RPB = 16 / SUBY;                       // rows of this plane per block row
	constexpr int NEF = M::PAIR ? 1 : M::NE;
	constexpr int K = M::SHIFT * SZ / 4;                 // dwords of a lane that lie in the memory unit before the lane's own
	constexpr int LDA = VFGS_LDAUX_ALIGNED, STA = VFGS_STAUX_ALIGNED;
	constexpr int LPB = M::PAIR ? 1 : M::BPL;            // blocks per lane step (PAIR: half a block, see idx0 below)
	constexpr int BPS = M::PAIR ? 32 : 64 * M::BPL;      // grain blocks a segment advances by
	constexpr uint32_t PT_CUR = IMG_BYTES, PT_UP = IMG_BYTES + kParamTableBytes;
	static_assert(IMG_BYTES % 16 == 0, "table image in whole 16-byte units");
	const int pt = comp ? 1 : 0;
	auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };

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

	// ---- in flight together: the table image, the LFSR words of the row's blocks, my first four segments -----------
	// (in this order: a wave's loads return in issue order; fixed instruction stream)
	constexpr int STEP = kWavesPerWG * 64 * 16;
	constexpr int NIT = (IMG_BYTES + STEP - 1) / STEP;
	u32x4 tmp[NIT];
	{
		const __amdgpu_buffer_rsrc_t irs = make_rsrc(a.tables + img_off, IMG_BYTES);
#pragma unroll
		for (int it = 0; it < NIT; it++)      // threads beyond the image re-read (and re-write) its last unit
			tmp[it] = __builtin_amdgcn_raw_buffer_load_b128(irs, min((uint32_t)(threadIdx.x * 16 + it * STEP), (uint32_t)(IMG_BYTES - 16)), 0, 0);
	}
	// parameter table entry e = block e - 1 (clamped into the row): thread t fills entries t, t + 256, ...
	constexpr int NPE = (kParamEntries + kWavesPerWG * 64 - 1) / (kWavesPerWG * 64);
	u32x2 wc[NPE], wu[NPE];
	{
		const __amdgpu_buffer_rsrc_t strs = make_rsrc((const uint8_t*)a.stream, a.stream_bytes);
		const __amdgpu_buffer_rsrc_t strs_up = make_rsrc((const uint8_t*)a.stream, wg_up ? a.stream_bytes : 0);
#pragma unroll
		for (int i = 0; i < NPE; i++)
		{
			const int e = (int)threadIdx.x + i * kWavesPerWG * 64;
			const uint32_t blk = (uint32_t)min(max(e - 1, 0), last);
			const bool need = e < a.nblk + 4 && e < kParamEntries;
			wc[i] = __builtin_amdgcn_raw_buffer_load_b64(strs, need ? ((cur_bit + blk) >> 5) * 4 : kOOB, 0, 0);
			wu[i] = __builtin_amdgcn_raw_buffer_load_b64(strs_up, need ? ((up_bit + blk) >> 5) * 4 : kOOB, 0, 0);
		}
	}
	// A row = rw_segs wave accesses ("positions": its units and the one behind them), walked in groups of four: the four
	// register sets.  One buffer descriptor serves a whole group: base = the first byte of the group, num_records = the
	// bytes the row has left from there (at most the group's 4 KiB), the segment's 1 KiB step sits in the instruction's
	// immediate offset: the hardware range check switches off exactly the lanes behind the row's end -- and every lane of a
	// group that does not exist -- and no access of any lane can leave the row.  (Measured on gfx950: a scalar offset
	// operand IS part of what is checked against num_records, so the row offset has to go into the base.)
	const int tsegs = pd.rw_segs;
	constexpr int NU = 4;                          // positions per group = register sets of the ring
	const int ngroups = (tsegs + NU - 1) / NU;
	const uint8_t* sbase = a.src[comp] + (uint64_t)f * pd.fpitch;
	uint8_t* dbase = a.dst[comp] + (uint64_t)f * pd.dfpitch;
	const uint32_t lane16 = (uint32_t)lane * 16;
	constexpr uint32_t GB = NU * kMaxUnits * 16;         // bytes of a group
	auto row_off = [&](int k) { return (uint32_t)((base + RSTR * k - prow0) * (int)pd.pitch); };
	auto left = [&](int g) { const uint32_t o = (uint32_t)g * GB; return o < pd.rowbytes ? min(pd.rowbytes - o, GB) : 0u; };
	uint32_t w[NU][4];
	if constexpr (NARROW == 0)
	{
		const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(sbase + (k0 < k1 ? row_off(k0) : 0u), k0 < k1 ? left(0) : 0u);
#pragma unroll
		for (int u = 0; u < NU; u++) load_seg<LDA>(rs0, lane16 + u * (kMaxUnits * 16), 0, w[u]);
	}
	else
	{
		// rows of NARROW positions: the four register sets hold 4 / NARROW consecutive rows of the wave
#pragma unroll
		for (int u = 0; u < NU; u++)
		{
			const int k = k0 + u / NARROW;
			const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(sbase + (k < k1 ? row_off(k) : 0u), k < k1 ? pd.rowbytes : 0u);
			load_seg<LDA>(rs0, lane16 + (u % NARROW) * (kMaxUnits * 16), 0, w[u]);
		}
	}
#pragma unroll
	for (int it = 0; it < NIT; it++)
		*(u32x4*)(lds + min((uint32_t)(threadIdx.x * 16 + it * STEP), (uint32_t)(IMG_BYTES - 16))) = tmp[it];

	// ---- block parameters of the row (once per workgroup) ----------------------------------------------------------
	const int fsx = comp == 0 ? 0 : (comp == 1 ? 10 : 20);
	const int fsy = comp == 0 ? 14 : (comp == 1 ? 24 : 4);
	const int fsb = comp == 0 ? 31 : (comp == 1 ? 2 : 15);
#pragma unroll
	for (int i = 0; i < NPE; i++)
	{
		// (rows of up to 252 blocks -- 2160p and narrower -- need the first round only, 4320p two of the three: a wave-uniform
		// branch around arithmetic and LDS writes; the LFSR loads above stay unconditional, switched off by their offsets, so
		// that the waits below can still be counted)
		if (i > 0 && i * kWavesPerWG * 64 >= a.nblk + 4) break;
		const int e = (int)threadIdx.x + i * kWavesPerWG * 64;
		const uint32_t blk = (uint32_t)min(max(e - 1, 0), last);
		bool neg;
		const uint32_t vc = __builtin_amdgcn_alignbit(wc[i].y, wc[i].x, (cur_bit + blk) & 31);
		const uint32_t pc = (block_param<SUBX, SUBY, RS, ONE>(vc, bank_off, fsx, fsy, fsb, &neg) + ((ONE && neg) ? (uint32_t)NEG : 0u)) | (neg ? 0x80000000u : 0u);
		const uint32_t vu = __builtin_amdgcn_alignbit(wu[i].y, wu[i].x, (up_bit + blk) & 31);
		const uint32_t pu = block_param<SUBX, SUBY, RS, ONE>(vu, bank_off, fsx, fsy, fsb, &neg) | (neg ? 0x80000000u : 0u);
		if (e < kParamEntries)
		{
			*(uint32_t*)(lds + PT_CUR + e * 4) = pc;
			*(uint32_t*)(lds + PT_UP + e * 4) = pu;
		}
	}
	__syncthreads();
	if (k0 >= k1)
		return;

	// ---- per lane constants ------------------------------------------------------------------------------------------
	const uint32_t lutb = lut_off * 0x10001u;
	const uint32_t lo2 = a.lo2[pt], hi2 = a.hi2[pt];
	const bool first = M::PAIR && (lane & 1);                              // PAIR: odd lane positions hold the first half of a block
	const uint32_t pairoff = M::PAIR ? (first ? 0u : 8u * (ONE ? 1 : kSlots)) : 0u;
	const uint32_t idx0 = (M::PAIR ? (uint32_t)(lane + 1) >> 1 : (uint32_t)lane * LPB) * 4;   // byte offset of my first entry in segment 0
	const int cl = M::PAIR ? lane - 1 - (lane & 1) : lane * LPB - 1;       // PAIR: left unit of my lane pair; else: block of run 0 (both for segment 0)
	constexpr int SSTEP = M::PAIR ? 64 : BPS;                              // what `cl` advances by per segment
	// DPP moves by one lane; lanes without a source lane (lane 0 / lane 63) keep `old`; the rotations wrap around
	auto lane_up = [](uint32_t old, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x138, 0xf, 0xf, false); };    // wave_shr:1: lane l <- lane l - 1
	auto lane_down = [](uint32_t old, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x130, 0xf, 0xf, false); };  // wave_shl:1: lane l <- lane l + 1
	auto rot_up = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x13c, 0xf, 0xf, false); };      // wave_ror:1: lane 0 <- lane 63
	auto rot_down = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x134, 0xf, 0xf, false); };    // wave_rol:1: lane 63 <- lane 0

	// ---- the walk ----------------------------------------------------------------------------------------------------
	uint32_t carry[4] = {0, 0, 0, 0};      // in lane 0: the last K dwords of lane 63 of the previous segment of the row
	uint32_t outp[4] = {0, 0, 0, 0};       // the previous segment's units: dwords K.. of its lanes (the first 4 - K dwords of a unit)
	uint32_t tp[4] = {0, 0, 0, 0};         // ... and the first K dwords its lanes computed: they belong one lane down
	__amdgpu_buffer_rsrc_t pdst = make_rsrc(dbase, 0);     // where the previous GROUP's last segment goes
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
				const __amdgpu_buffer_rsrc_t cdst = make_rsrc(dbase + (valid ? row_off(k) : 0u), valid ? pd.rowbytes : 0u);
				uint32_t t[4];
#pragma unroll
				for (int d = 0; d < K; d++) t[d] = lane_up(p == 0 ? 0u : carry[d], w[u][4 - K + d]);
#pragma unroll
				for (int d = 0; d < K; d++) carry[d] = rot_up(w[u][4 - K + d]);
#pragma unroll
				for (int d = K; d < 4; d++) asm volatile("v_mov_b32 %0, %1" : "=v"(t[d]) : "v"(w[u][d - K]));
				load_seg<LDA>(nsrc, lane16 + p * (kMaxUnits * 16), 0, w[u]);
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
						grain_unit<DEPTH, BW, true, ONE, ONE && SUBX == 2, NEG>(lds, t, rp, up, lutb, rowoff, uprowoff, wc_, wu_, edge_on, first, lo2, hi2);
					}
					else
					{
#pragma unroll
						for (int rr = 0; rr < NR; rr++) up.pa[rr] = 0u;
						grain_unit<DEPTH, BW, false, ONE, ONE && SUBX == 2, NEG>(lds, t, rp, up, lutb, rowoff, uprowoff, 0, 0, edge_on, first, lo2, hi2);
					}
				}
				// the previous position (of this row, or the last one of the row before: `pdst` is its row) is complete
#pragma unroll
				for (int d = 0; d < K; d++) outp[4 - K + d] = lane_down(rot_down(t[d]), tp[d]);
				store_b128<STA>(pdst, lane16 + ((p + P - 1) % P) * (kMaxUnits * 16), 0, outp);
				pdst = cdst;
#pragma unroll
				for (int d = K; d < 4; d++) outp[d - K] = t[d];
#pragma unroll
				for (int d = 0; d < K; d++) tp[d] = t[d];
#if VFGS_SCHED_FENCE
				__builtin_amdgcn_sched_barrier(0);
#endif
			}
		}
#pragma unroll
		for (int d = 0; d < K; d++) outp[4 - K + d] = lane_down(0u, tp[d]);
		store_b128<STA>(pdst, lane16 + (P - 1) * (kMaxUnits * 16), 0, outp);
		return;
	}
	// one row; `overlap` is a type so that the walk of the (rare) overlap lines is code of its own: the hot loop carries
	// neither their arithmetic nor a branch around it
	auto walk_row = [&](auto overlap, const int k) {
		constexpr bool OV = decltype(overlap)::value;
		const int j = base + RSTR * k - Rabs * RPB;    // row inside the block row
		const int jrow = j * SUBY;
		const uint32_t rowoff = (uint32_t)j * RS, uprowoff = (uint32_t)(RPB + j) * RS;
		const int wc_ = jrow == 0 ? (SUBY > 1 ? 20 : 12) : 24, wu_ = jrow == 0 ? (SUBY > 1 ? 20 : 24) : 12;
		const uint32_t ro = row_off(k);
		for (int g = 0; g < ngroups; g++)
		{
			// the group after this one (this row's next, or the next row's first): what the four refills fetch
			const bool lastg = g + 1 == ngroups;
			const int ng = lastg ? 0 : g + 1;
			const bool nvalid = !lastg || k + 1 < k1;
			const uint32_t nso = nvalid ? (lastg ? row_off(k + 1) : ro) + (uint32_t)ng * GB : 0u;
			const __amdgpu_buffer_rsrc_t nsrc = make_rsrc(sbase + nso, nvalid ? left(ng) : 0u);
			const __amdgpu_buffer_rsrc_t cdst = make_rsrc(dbase + (ro + (uint32_t)g * GB), left(g));
			const uint8_t* pe = lds + idx0 + (uint32_t)(g * NU * BPS * 4);
			const int clg = cl + g * NU * SSTEP;
#pragma unroll
			for (int u = 0; u < NU; u++)
			{
				const bool firsts = u == 0 && g == 0;                    // first segment of the row
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
				// the registers are free: refill them with the segment four steps ahead
				load_seg<LDA>(nsrc, lane16 + u * (kMaxUnits * 16), 0, w[u]);
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
					grain_unit<DEPTH, BW, OV, ONE, ONE && SUBX == 2, NEG>(lds, t, rp, up, lutb, rowoff, uprowoff, OV ? wc_ : 0, OV ? wu_ : 0, edge_on, first, lo2, hi2);
				}
				// the previous segment's units are complete once the K dwords its lanes computed have moved one lane down; its
				// lane 63 takes them from my lane 0 (the previous segment of a row's first one is the last of another row:
				// its lane 63 lies behind that row's end and is never stored)
#pragma unroll
				for (int d = 0; d < K; d++) outp[4 - K + d] = lane_down(rot_down(t[d]), tp[d]);
				if (u == 0) store_b128<STA>(pdst, lane16 + (NU - 1) * (kMaxUnits * 16), 0, outp);
				else store_b128<STA>(cdst, lane16 + (u - 1) * (kMaxUnits * 16), 0, outp);
#pragma unroll
				for (int d = K; d < 4; d++) outp[d - K] = t[d];
#pragma unroll
				for (int d = 0; d < K; d++) tp[d] = t[d];
#if VFGS_SCHED_FENCE
				__builtin_amdgcn_sched_barrier(0);
#endif
			}
			pdst = cdst;
		}
	};
	for (int k = k0; k < k1; k++)
	{
		const int jrow = (base + RSTR * k - Rabs * RPB) * SUBY;
		if (Rabs > 0 && jrow <= 1) walk_row(std::true_type(), k);      // blends in the block above (vfgs_hw.c:173-188, 223-229)
		else walk_row(std::false_type(), k);
	}
	// the last segment of my last row
#pragma unroll
	for (int d = 0; d < K; d++) outp[4 - K + d] = lane_down(0u, tp[d]);
	store_b128<STA>(pdst, lane16 + (NU - 1) * (kMaxUnits * 16), 0, outp);
}

// 8-bit planes with 8-sample blocks hold three block runs and two edges per lane (LaneMap): those kernels get the
// registers of one workgroup less per CU instead of spilling in the row loop
template <int DEPTH, int CSUBX>
constexpr int wg_per_cu() { return (DEPTH == 8 && CSUBX == 2 && VFGS_WG_PER_CU > VFGS_WG_PER_CU_8BIT_SUB) ? VFGS_WG_PER_CU_8BIT_SUB : VFGS_WG_PER_CU; }

template <int DEPTH, int CSUBX, int CSUBY, bool OUT8, bool ONEY, bool ONEC, bool AL>
__global__ __launch_bounds__(kWavesPerWG * 64, (kWavesPerWG * wg_per_cu<DEPTH, CSUBX>() + 3) / 4) void grain_kernel(const KernelArgs a)
{
	constexpr ImageLayout L = image_layout(CSUBX, CSUBY, ONEY, ONEC);
	__shared__ __attribute__((aligned(16))) uint8_t lds[L.lds_bytes];

	const int lane = threadIdx.x & 63;
	// wave-uniform by construction; telling the compiler keeps the decoding, row offsets and buffer
	// descriptors in SGPRs (otherwise every buffer instruction gets a waterfall loop)
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

	const int f = blockIdx.y;            // grid: x = workgroup inside the frame, y = frame of the batch
	int r = blockIdx.x;
	if (r < a.pd[0].wgs)
		run_plane<DEPTH, 16, 1, 1, L.y_rs, OUT8, L.y_bytes, ONEY, AL, L.y_neg>(a, a.pd[0], lds, 0, f, r, L.y_off, L.y_bank, 0, lane, wave);
	else
	{
		r -= a.pd[0].wgs;
		const int comp = 1 + (r >= a.pd[1].wgs);
		if (comp == 2) r -= a.pd[1].wgs;
		run_plane<DEPTH, 16 / CSUBX, CSUBX, CSUBY, L.c_rs, OUT8, L.c_bytes, ONEC, AL, L.c_neg>(a, a.pd[1], lds, comp, f, r, L.c_off[comp - 1], L.c_bank, L.c_lut[comp - 1], lane, wave);
	}
}

// row-walk kernel (run_plane_rw): in place or out of place, same sample size; workgroups numbered frame -> plane -> block row -> part
// Waves per SIMD the row-walk kernels are allocated for.  The 8-bit all-one-pattern kernels need 97..100 registers, one
// allocation granule above the 96 of five waves: asking for five costs one register spilled in the prologue and reloaded
// once per row (not in the group loop) and is worth 3 % (profiles/r03_ab22_lds_probes_and_occupancy.log); the general-form
// kernels are held at four by their LDS image, the others by spills.
template <int DEPTH, bool ONEY, bool ONEC>
constexpr int rw_waves_per_simd() { return (DEPTH == 8 && ONEY && ONEC && VFGS_WG_PER_CU == 4) ? 5 : (kWavesPerWG * VFGS_WG_PER_CU + 3) / 4; }

template <int DEPTH, int CSUBX, int CSUBY, bool ONEY, bool ONEC>
__global__ __launch_bounds__(kWavesPerWG * 64, (rw_waves_per_simd<DEPTH, ONEY, ONEC>())) void grain_rw_kernel(const KernelArgs a)
{
	constexpr ImageLayout L = image_layout(CSUBX, CSUBY, ONEY, ONEC);
	__shared__ __attribute__((aligned(16))) uint8_t lds[L.lds_bytes + kParamBytes];

	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	// grid: x = (workgroup inside the frame, frame of a group of 2^lfronts frames), y = group of frames.  Large frames are swept
	// two at a time: the workgroups of frames 2m and 2m + 1 are dealt out alternately (measured: +1.6 % at 4320p, nothing at 2160p;
	// more than two, or smaller frames: a loss -- profiles/r03_ab18_frame_fronts_in_one_launch.log)
	const int f = (int)(blockIdx.y << a.lfronts) + (int)(blockIdx.x & ((1u << a.lfronts) - 1));
	int r = (int)(blockIdx.x >> a.lfronts);
	if (f >= a.nframes) return;
	if (r < a.pd[0].wgs)
		run_plane_rw<DEPTH, 16, 1, 1, L.y_rs, L.y_bytes, ONEY, L.y_neg, 0>(a, a.pd[0], lds, 0, f, r, L.y_off, L.y_bank, 0, lane, wave);
	else
	{
		r -= a.pd[0].wgs;
		const int comp = 1 + (r >= a.pd[1].wgs);
		if (comp == 2) r -= a.pd[1].wgs;
		// horizontally subsampled chroma rows of one or two positions (2 KiB and less: 1080p at 10 bit, 2160p at 8 bit): several rows per group
		if (CSUBX == 2 && a.pd[1].rw_segs == 2)
			run_plane_rw<DEPTH, 16 / CSUBX, CSUBX, CSUBY, L.c_rs, L.c_bytes, ONEC, L.c_neg, 2>(a, a.pd[1], lds, comp, f, r, L.c_off[comp - 1], L.c_bank, L.c_lut[comp - 1], lane, wave);
		else if (CSUBX == 2 && a.pd[1].rw_segs == 1)
			run_plane_rw<DEPTH, 16 / CSUBX, CSUBX, CSUBY, L.c_rs, L.c_bytes, ONEC, L.c_neg, 1>(a, a.pd[1], lds, comp, f, r, L.c_off[comp - 1], L.c_bank, L.c_lut[comp - 1], lane, wave);
		else
			run_plane_rw<DEPTH, 16 / CSUBX, CSUBX, CSUBY, L.c_rs, L.c_bytes, ONEC, L.c_neg, 0>(a, a.pd[1], lds, comp, f, r, L.c_off[comp - 1], L.c_bank, L.c_lut[comp - 1], lane, wave);
	}
}

// ---------------------------------------------------------------------------------------
// host-side launcher (called from vfgs_host.cpp)

template <int DEPTH, int CSUBX, int CSUBY, bool OUT8, bool ONEY, bool ONEC, bool AL>
static hipError_t launch_t(const KernelArgs& a, int grid, hipStream_t stream)
{
	hipLaunchKernelGGL((grain_kernel<DEPTH, CSUBX, CSUBY, OUT8, ONEY, ONEC, AL>), dim3(grid, a.nframes), dim3(kWavesPerWG * 64), 0, stream, a);
	return hipGetLastError();
}

// every plane type's rows are whole 16-byte units except 8-bit planes with 8-sample blocks and an odd number of blocks:
// only those formats carry the kernels with shifted accesses next to the aligned ones
template <int DEPTH, int CSUBX>
constexpr bool has_shifted() { return !VFGS_ALIGNED || (DEPTH == 8 && CSUBX == 2); }

template <int DEPTH, int CSUBX, int CSUBY, bool ONEY, bool ONEC>
static hipError_t launch_al(const KernelArgs& a, int mode, int grid, hipStream_t stream)
{
	const bool aligned = mode != 0;
	if constexpr (VFGS_ALIGNED != 0)
	{
		if (mode == 2)
		{
			hipLaunchKernelGGL((grain_rw_kernel<DEPTH, CSUBX, CSUBY, ONEY, ONEC>), dim3((unsigned)grid << a.lfronts, ((unsigned)a.nframes + (1u << a.lfronts) - 1) >> a.lfronts),
			                   dim3(kWavesPerWG * 64), 0, stream, a);
			return hipGetLastError();
		}
		if (aligned) return launch_t<DEPTH, CSUBX, CSUBY, false, ONEY, ONEC, true>(a, grid, stream);
	}
	if constexpr (has_shifted<DEPTH, CSUBX>())
		return launch_t<DEPTH, CSUBX, CSUBY, false, ONEY, ONEC, false>(a, grid, stream);
	return hipErrorInvalidValue;
}

template <int DEPTH, int CSUBX, int CSUBY>
static hipError_t launch_one(const KernelArgs& a, bool out8, bool oney, bool onec, int aligned, int grid, hipStream_t stream)
{
	if (DEPTH == 10 && out8) return launch_t<10, CSUBX, CSUBY, true, false, false, false>(a, grid, stream);    // fused 8-bit output: general form, shifted accesses
	if (oney && onec) return launch_al<DEPTH, CSUBX, CSUBY, true, true>(a, aligned, grid, stream);
	if (oney) return launch_al<DEPTH, CSUBX, CSUBY, true, false>(a, aligned, grid, stream);
	if (onec) return launch_al<DEPTH, CSUBX, CSUBY, false, true>(a, aligned, grid, stream);
	return launch_al<DEPTH, CSUBX, CSUBY, false, false>(a, aligned, grid, stream);
}

// oney / onec: the image holds the one-pattern form for luma / chroma (vfgs_layout.h); never with out8.
// aligned: 0 = the plane descriptors were laid out for the kernels with shifted accesses, 1 = for the aligned tiled kernels
// (aligned_ok()), 2 = for the row walk (rowwalk_ok()).
hipError_t launch_grain(const KernelArgs& a, int depth, int csubx, int csuby, bool out8, bool oney, bool onec, int aligned, int grid, hipStream_t stream)
{
	if (out8 && (depth != 10 || oney || onec || aligned)) return hipErrorInvalidValue;
#define VFGS_CASE(D, X, Y) if (depth == D && csubx == X && csuby == Y) return launch_one<D, X, Y>(a, out8, oney, onec, aligned, grid, stream)
	VFGS_CASE(10, 2, 2); VFGS_CASE(10, 2, 1); VFGS_CASE(10, 1, 1); VFGS_CASE(10, 1, 2);
	VFGS_CASE(8, 2, 2);  VFGS_CASE(8, 2, 1);  VFGS_CASE(8, 1, 1);  VFGS_CASE(8, 1, 2);
#undef VFGS_CASE
	return hipErrorInvalidValue;
}

// the name of the instantiation launch_grain() dispatches for these arguments, as the profiler prints it (vfgs_hip_last_launch_info)
void describe_launch(char* out, size_t n, int depth, int csubx, int csuby, bool out8, bool oney, bool onec, int aligned)
{
	auto b = [](bool v) { return v ? "true" : "false"; };
	if (depth == 10 && out8) snprintf(out, n, "grain_kernel<10,%d,%d,true,false,false,false>", csubx, csuby);
	else if (VFGS_ALIGNED && aligned == 2) snprintf(out, n, "grain_rw_kernel<%d,%d,%d,%s,%s>", depth, csubx, csuby, b(oney), b(onec));
	else snprintf(out, n, "grain_kernel<%d,%d,%d,false,%s,%s,%s>", depth, csubx, csuby, b(oney), b(onec), b(VFGS_ALIGNED && aligned == 1));
}

// may a launch use the aligned kernels?  (rows of both plane types are whole 16-byte units; not the narrowed destination)
bool aligned_ok(int depth, int csubx, int nblk, bool out8)
{
	if (!VFGS_ALIGNED || out8) return false;
	return !(depth == 8 && csubx == 2 && (nblk & 1));
}

// ... the row walk?  (additionally: a row's blocks fit the workgroup's parameter table)
bool rowwalk_ok(int depth, int csubx, int nblk, bool out8)
{
#ifdef VFGS_NO_ROWWALK
	return false;
#endif
	// (also 8-bit 4:2:x rows of an odd number of blocks, which end in HALF a unit: a row's descriptor holds exactly the row's
	// bytes and the raw-buffer range check works per dword, so the last lane's access is cut in the middle -- loads return
	// 0 for, stores drop, the dwords behind the row.  Measured: bit-exact incl. the stride padding, 0.50 instead of the
	// tiled kernels' 0.35 at 3856x2160, profiles/r03_ab44_odd_block_counts.log)
	(void)depth; (void)csubx;
	return VFGS_ALIGNED && !out8 && nblk <= kTileBlocks;
}

ImageLayout layout_of(int csubx, int csuby, bool oney, bool onec) { return image_layout(csubx, csuby, oney, onec); }

// the plane type's lane layout, for the host's geometry: samples per lane, samples the unit grid is shifted, lanes per row
void lane_layout(int depth, int bw, int nblk, int* shift_samples, int* lanes)
{
	const int ns = depth == 8 ? 16 : 8;
	const bool pair = (ns == 8 && bw == 16);
	const int shift = pair ? 8 : bw / 2;
	*shift_samples = shift;
	*lanes = (nblk * bw + shift + ns - 1) / ns;
}

}  // namespace vfgs
