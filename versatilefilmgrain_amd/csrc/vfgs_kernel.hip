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
//   * one WAVEFRONT owns one tile = 8 grain blocks (128 luma samples) x one block row
//     (16 luma lines) of Y and the co-located Cb/Cr samples;
//   * every lane moves 16 bytes (10-bit) / 8 bytes (8-bit) = 8 samples per access, 16 lanes
//     cover a 256-byte luma row segment, 4 rows per wave-instruction;
//   * tiles are shifted by HALF A BLOCK (8 luma samples) against the block grid, so every
//     block edge -- the only place where a sample depends on its horizontal neighbours --
//     lies strictly inside a tile: no halo, no inter-wave exchange, in-place is race free;
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

__device__ __forceinline__ uint32_t stream_window(const uint32_t* __restrict__ s, uint32_t bit)
{
	// 32-bit window of the LFSR bit stream = the register after `bit` steps (vfgs_hw.c:74-79)
	const uint32_t* p = s + (bit >> 5);
	uint32_t lo = p[0], hi = p[1];
	return __builtin_amdgcn_alignbit(hi, lo, bit & 31);
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

// One sample's pattern value out of its 8-byte slot group {hi,lo}, slot chosen by the low
// byte of the LUT entry (0..7, or 0x0c = constant 0).
__device__ __forceinline__ int pick_slot(uint32_t hi, uint32_t lo, uint32_t lut_entry)
{
	return (int)(int8_t)__builtin_amdgcn_perm(hi, lo, lut_entry);
}

// 24-bit multiplies are full rate; the 32-bit v_mul_lo_u32 / v_mad_u64_u32 the compiler would
// otherwise pick are quarter rate.  Every product here fits easily (|pattern| < 2^9, scale < 2^8,
// weights < 2^6).
__device__ __forceinline__ int mad24(int a, int b, int c)
{
	return __mul24(a, b) + c;
}

__device__ __forceinline__ int swap_lane_pairs(int v)
{
	// quad_perm:[1,0,3,2]: lane 2m <-> lane 2m+1
	return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);
}

template <int BYTES, int ALIGN>
__device__ __forceinline__ void gload(const uint8_t* p, uint32_t* dst)
{
	p = (const uint8_t*)__builtin_assume_aligned(p, ALIGN);
	__builtin_memcpy(dst, p, BYTES);
}

template <int BYTES, int ALIGN>
__device__ __forceinline__ void gstore(uint8_t* p, const uint32_t* src)
{
	p = (uint8_t*)__builtin_assume_aligned(p, ALIGN);
	__builtin_memcpy(p, src, BYTES);
}

// A lane's 8 consecutive samples of one row, as two independently valid halves of 4.
// DEPTH 10: 16 bytes in memory, kept as 4 dwords of two uint16 each.
// DEPTH  8:  8 bytes in memory, widened to the same 4 x (2 x uint16) form.
template <int DEPTH, int ALIGN>
__device__ __forceinline__ void load_unit(const uint8_t* p, bool v0, bool v1, uint32_t (&w)[4])
{
	w[0] = w[1] = w[2] = w[3] = 0;
	if (DEPTH > 8)
	{
		if (v0 && v1) gload<16, ALIGN>(p, w);
		else if (v0)  gload<8, ALIGN>(p, w);
		else if (v1)  gload<8, ALIGN>(p + 8, w + 2);
	}
	else
	{
		uint32_t r[2] = {0, 0};
		if (v0 && v1) gload<8, ALIGN>(p, r);
		else if (v0)  gload<4, ALIGN>(p, r);
		else if (v1)  gload<4, ALIGN>(p + 4, r + 1);
		w[0] = __builtin_amdgcn_perm(0, r[0], 0x0c010c00);
		w[1] = __builtin_amdgcn_perm(0, r[0], 0x0c030c02);
		w[2] = __builtin_amdgcn_perm(0, r[1], 0x0c010c00);
		w[3] = __builtin_amdgcn_perm(0, r[1], 0x0c030c02);
	}
}

template <int DEPTH, int ALIGN>
__device__ __forceinline__ void store_unit(uint8_t* p, bool v0, bool v1, const uint32_t (&w)[4])
{
	if (DEPTH > 8)
	{
		if (v0 && v1) gstore<16, ALIGN>(p, w);
		else if (v0)  gstore<8, ALIGN>(p, w);
		else if (v1)  gstore<8, ALIGN>(p + 8, w + 2);
	}
	else
	{
		uint32_t r[2];
		r[0] = __builtin_amdgcn_perm(w[1], w[0], 0x06040200);
		r[1] = __builtin_amdgcn_perm(w[3], w[2], 0x06040200);
		if (v0 && v1) gstore<8, ALIGN>(p, r);
		else if (v0)  gstore<4, ALIGN>(p, r);
		else if (v1)  gstore<4, ALIGN>(p + 4, r + 1);
	}
}

// ---------------------------------------------------------------------------------------
// The per-lane grain pipeline for 8 samples of one row.
//
//   w        in/out: samples, 4 x (2 x uint16)
//   lut      LDS byte offset of this component's 256-entry LUT
//   a0,a1    LDS byte offsets of the pattern data of samples 0-3 / 4-7 (current block row)
//   m0,m1    multipliers of those pattern values: sign (x overlap weight if OVERLAP)
//   u0,u1,n0,n1  same for the block row above (OVERLAP only)
//   EDGE16   true : block edge between this lane and its pair lane (16-sample blocks);
//                   `odd` lanes hold the right-hand block's first sample in slot 0,
//                   even lanes the left-hand block's last sample in slot 7
//            false: block edge between samples 3 and 4 of this lane (8-sample blocks)
template <int DEPTH, bool OVERLAP, bool EDGE16>
__device__ __forceinline__ void grain_unit(const uint8_t* lds, uint32_t (&w)[4], uint32_t lut,
                                            uint32_t a0, uint32_t a1, int m0, int m1,
                                            uint32_t u0, uint32_t u1, int n0, int n1,
                                            bool edge_on, bool odd, int scale_shift, uint32_t lo2, uint32_t hi2)
{
	uint32_t e[8];
	int P[8];

	// LUT gather: intensity = sample >> bs, as a uint8 (vfgs_hw.c:157,211); entry address = 4*intensity
#pragma unroll
	for (int k = 0; k < 4; k++)
	{
		uint32_t idx = (DEPTH > 8) ? (w[k] & 0x03fc03fcu) : ((w[k] & 0x00ff00ffu) << 2);
		e[2 * k]     = *(const uint32_t*)(lds + lut + (idx & 0xffffu));
		e[2 * k + 1] = *(const uint32_t*)(lds + lut + (idx >> 16));
	}

	// pattern fetch: 4 samples x 8 slots = 32 bytes per half
	{
		u32x4 c0 = *(const u32x4*)(lds + a0), c1 = *(const u32x4*)(lds + a0 + 16);
		u32x4 c2 = *(const u32x4*)(lds + a1), c3 = *(const u32x4*)(lds + a1 + 16);
		P[0] = pick_slot(c0.y, c0.x, e[0]); P[1] = pick_slot(c0.w, c0.z, e[1]);
		P[2] = pick_slot(c1.y, c1.x, e[2]); P[3] = pick_slot(c1.w, c1.z, e[3]);
		P[4] = pick_slot(c2.y, c2.x, e[4]); P[5] = pick_slot(c2.w, c2.z, e[5]);
		P[6] = pick_slot(c3.y, c3.x, e[6]); P[7] = pick_slot(c3.w, c3.z, e[7]);
	}
	if (OVERLAP)
	{
		// vfgs_hw.c:223-229; lanes outside the two overlap lines carry weights (32, 0): (32 P + 16) >> 5 == P
		u32x4 c0 = *(const u32x4*)(lds + u0), c1 = *(const u32x4*)(lds + u0 + 16);
		u32x4 c2 = *(const u32x4*)(lds + u1), c3 = *(const u32x4*)(lds + u1 + 16);
		int Q[8];
		Q[0] = pick_slot(c0.y, c0.x, e[0]); Q[1] = pick_slot(c0.w, c0.z, e[1]);
		Q[2] = pick_slot(c1.y, c1.x, e[2]); Q[3] = pick_slot(c1.w, c1.z, e[3]);
		Q[4] = pick_slot(c2.y, c2.x, e[4]); Q[5] = pick_slot(c2.w, c2.z, e[5]);
		Q[6] = pick_slot(c3.y, c3.x, e[6]); Q[7] = pick_slot(c3.w, c3.z, e[7]);
#pragma unroll
		for (int k = 0; k < 8; k++)
			P[k] = mad24(Q[k], k < 4 ? n0 : n1, mad24(P[k], k < 4 ? m0 : m1, 16)) >> 5;
	}
	else
	{
#pragma unroll
		for (int k = 0; k < 8; k++)
			P[k] = __mul24(P[k], k < 4 ? m0 : m1);
	}

	// 3-tap filter across the block edge (vfgs_hw.c:250-259), on unfiltered neighbours
	if (EDGE16)
	{
		int mine = odd ? P[0] : P[7];
		int inner = odd ? P[1] : P[6];
		int theirs = swap_lane_pairs(mine);
		int f = (inner + 3 * mine + theirs + 2) >> 2;
		f = edge_on ? f : mine;
		P[0] = odd ? f : P[0];
		P[7] = odd ? P[7] : f;
	}
	else
	{
		int l1 = P[2], l0 = P[3], r0 = P[4], r1 = P[5];
		P[3] = edge_on ? ((l1 + 3 * l0 + r0 + 2) >> 2) : l0;
		P[4] = edge_on ? ((l0 + 3 * r0 + r1 + 2) >> 2) : r0;
	}

	// scale, add, clip (vfgs_hw.c:263-267)
	const int half = 1 << (scale_shift - 1);
#pragma unroll
	for (int k = 0; k < 4; k++)
	{
		int g0 = mad24(P[2 * k], (int)(e[2 * k] >> 16), half) >> scale_shift;
		int g1 = mad24(P[2 * k + 1], (int)(e[2 * k + 1] >> 16), half) >> scale_shift;
		uint32_t gp = __builtin_amdgcn_perm((uint32_t)g1, (uint32_t)g0, 0x05040100);
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
// One plane of one tile.
//
//   BW    block width in samples of this plane (16: luma and 4:4:4 chroma; 8: subsampled chroma)
//   SUBY  vertical subsampling of this plane
//   RS    bank row stride
// Lane geometry: LPR lanes per row, RPL rows per wave-access, NLOAD accesses per block row.
template <int DEPTH, int COMP, int BW, int SUBX, int SUBY, int RS>
struct PlaneTile {
	static constexpr int SZ = DEPTH > 8 ? 2 : 1;
	static constexpr int LPR = (kTilePx / SUBX) / 8;     // 16 or 8
	static constexpr int RPL = 64 / LPR;                 // 4 or 8
	static constexpr int ROWS = 16 / SUBY;               // plane rows per block row
	static constexpr int NLOAD = ROWS / RPL;
	static constexpr int ALIGN = (BW == 16) ? 8 * SZ : 4 * SZ;  // half-block shift: 8 (4) samples
	static_assert(NLOAD >= 1, "");

	uint32_t w[NLOAD][4];
	bool v0[NLOAD], v1[NLOAD];
	uint8_t* ptr[NLOAD];
	// per-lane block data
	BlockParam cur0, cur1, up0, up1;   // half 0 / half 1 (same block for BW == 16)
	uint32_t half_off0, half_off1;     // byte offsets inside the bank row
	bool edge_on, odd;
	int rloc0;                         // plane row inside the block row for access 0

	// issue the global loads of this plane
	__device__ __forceinline__ void issue(const KernelArgs& a, uint8_t* plane, int pstride, int tx, int R, int lane)
	{
		const int u = lane & (LPR - 1);
		rloc0 = lane / LPR;
		int bl, br;        // blocks left / right of the edge this lane (pair) straddles
		bool ok0, ok1;
		int x0;            // first sample of this lane, plane coordinates
		if (BW == 16)
		{
			const int m = u >> 1;
			odd = u & 1;
			bl = 8 * tx - 1 + m;
			br = bl + 1;
			const int blk = odd ? br : bl;
			ok0 = ok1 = (blk >= 0) && (blk < a.nblk);
			x0 = 16 * blk + (odd ? 0 : 8);
		}
		else
		{
			odd = false;
			bl = 8 * tx - 1 + u;
			br = bl + 1;
			ok0 = (bl >= 0) && (bl < a.nblk);
			ok1 = (br >= 0) && (br < a.nblk);
			x0 = 8 * bl + 4;
		}
		edge_on = (bl >= 0) && (br < a.nblk);
#pragma unroll
		for (int q = 0; q < NLOAD; q++)
		{
			const int rloc = q * RPL + rloc0;
			const int yabs = (R * ROWS + rloc) * SUBY;   // luma line this plane row belongs to
			const bool rok = (yabs >= a.y0) && (yabs < a.y0 + a.nlines);
			v0[q] = rok && ok0;
			v1[q] = rok && ok1;
			const int prow = R * ROWS + rloc - a.y0 / SUBY;   // row relative to the stripe pointer
			ptr[q] = plane + ((int64_t)prow * pstride + x0) * SZ;
			load_unit<DEPTH, ALIGN>(ptr[q], v0[q], v1[q], w[q]);
		}
	}

	// derive pattern addresses / signs from the LFSR windows of blocks bl, br
	__device__ __forceinline__ void params(uint32_t vcur_l, uint32_t vcur_r, uint32_t vup_l, uint32_t vup_r, uint32_t bank_off)
	{
		if (BW == 16)
		{
			const uint32_t vc = odd ? vcur_r : vcur_l, vu = odd ? vup_r : vup_l;
			cur0 = cur1 = block_param<COMP, SUBX, SUBY, RS>(vc, bank_off);
			up0 = up1 = block_param<COMP, SUBX, SUBY, RS>(vu, bank_off);
			half_off0 = odd ? 0 : 8 * kSlots;     // even lane: samples 8..15 of the left block
			half_off1 = half_off0 + 4 * kSlots;
		}
		else
		{
			cur0 = block_param<COMP, SUBX, SUBY, RS>(vcur_l, bank_off);
			cur1 = block_param<COMP, SUBX, SUBY, RS>(vcur_r, bank_off);
			up0 = block_param<COMP, SUBX, SUBY, RS>(vup_l, bank_off);
			up1 = block_param<COMP, SUBX, SUBY, RS>(vup_r, bank_off);
			half_off0 = 4 * kSlots;               // samples 4..7 of the left block
			half_off1 = 0;                        // samples 0..3 of the right block
		}
	}

	__device__ __forceinline__ void run(const KernelArgs& a, const uint8_t* lds, uint32_t lut, int R, uint32_t lo2, uint32_t hi2)
	{
		const bool can_overlap = (R > 0);      // y > 15 (vfgs_hw.c:175,180)
#pragma unroll
		for (int q = 0; q < NLOAD; q++)
		{
			if (__builtin_amdgcn_ballot_w64(v0[q] || v1[q]) == 0)
				continue;                      // wave-uniform: nothing of this access lies in the stripe
			const int rloc = q * RPL + rloc0;
			const uint32_t rowoff = __umul24((uint32_t)rloc, (uint32_t)RS);
			const uint32_t a0 = cur0.addr + rowoff + half_off0;
			const uint32_t a1 = cur1.addr + rowoff + half_off1;
			// Only access 0 can contain the two overlap lines (j = 0, 1 <=> plane rows 0 .. 1/SUBY)
			if (q == 0 && can_overlap)
			{
				const int j = rloc * SUBY;     // y & 15
				int wc = 32, wu = 0;
				if (j == 0) { wc = SUBY > 1 ? 20 : 12; wu = SUBY > 1 ? 20 : 24; }
				else if (j == 1) { wc = 24; wu = 12; }
				const uint32_t u0 = up0.addr + ROWS * RS + rowoff + half_off0;
				const uint32_t u1 = up1.addr + ROWS * RS + rowoff + half_off1;
				// rows without overlap must not read past the bank: clamp their (unused) address
				const uint32_t u0s = wu ? u0 : a0, u1s = wu ? u1 : a1;
				grain_unit<DEPTH, true, BW == 16>(lds, w[q], lut, a0, a1,
				                                  __mul24(cur0.sign, wc), __mul24(cur1.sign, wc),
				                                  u0s, u1s, __mul24(up0.sign, wu), __mul24(up1.sign, wu),
				                                  edge_on, odd, a.scale_shift, lo2, hi2);
			}
			else
			{
				grain_unit<DEPTH, false, BW == 16>(lds, w[q], lut, a0, a1, cur0.sign, cur1.sign,
				                                   0, 0, 0, 0, edge_on, odd, a.scale_shift, lo2, hi2);
			}
			store_unit<DEPTH, ALIGN>(ptr[q], v0[q], v1[q], w[q]);
		}
	}
};

// ---------------------------------------------------------------------------------------

template <int DEPTH, int CSUBX, int CSUBY>
__global__ __launch_bounds__(kWavesPerWG * 64) void grain_kernel(const KernelArgs a)
{
	using L = TableLayout<CSUBX, CSUBY>;
	using LumaT = PlaneTile<DEPTH, 0, 16, 1, 1, L::LRS>;
	using CbT = PlaneTile<DEPTH, 1, 16 / CSUBX, CSUBX, CSUBY, L::CRS>;
	using CrT = PlaneTile<DEPTH, 2, 16 / CSUBX, CSUBX, CSUBY, L::CRS>;

	__shared__ __attribute__((aligned(16))) uint8_t lds[L::BYTES];

	// stage banks + LUTs: global (L2 resident) -> LDS, 16 bytes per lane per step
	for (int i = threadIdx.x * 16; i < L::BYTES; i += kWavesPerWG * 64 * 16)
		*(u32x4*)(lds + i) = *(const u32x4*)(a.tables + i);
	__syncthreads();

	const int lane = threadIdx.x & 63;
	const int wave = threadIdx.x >> 6;
	const int tiles_per_frame = a.nbr * a.ntx;
	const int total = tiles_per_frame * a.nframes;
	const int row0 = a.y0 >> 4;

	const uint32_t ylo2 = (uint32_t)a.ylo * 0x10001u, yhi2 = (uint32_t)a.yhi * 0x10001u;
	const uint32_t clo2 = (uint32_t)a.clo * 0x10001u, chi2 = (uint32_t)a.chi * 0x10001u;

	for (int t = blockIdx.x * kWavesPerWG + wave; t < total; t += gridDim.x * kWavesPerWG)
	{
		const int f = t / tiles_per_frame;
		const int rem = t - f * tiles_per_frame;
		const int k = rem / a.ntx;          // block row inside the stripe
		const int tx = rem - k * a.ntx;
		const int R = row0 + k;             // absolute block row (y >> 4)

		uint8_t* Y = a.Y + (uint64_t)f * a.y_frame_pitch;
		uint8_t* U = a.U + (uint64_t)f * a.c_frame_pitch;
		uint8_t* V = a.V + (uint64_t)f * a.c_frame_pitch;

		LumaT ty;
		CbT tu;
		CrT tv;
		ty.issue(a, Y, a.stride, tx, R, lane);
		tu.issue(a, U, a.cstride, tx, R, lane);
		tv.issue(a, V, a.cstride, tx, R, lane);

		// LFSR windows of the blocks this lane touches, for this block row and the one above
		const uint32_t cur_bit = a.cur_bit0 + (uint32_t)f * a.frame_bit_step + (uint32_t)(k * a.nblk);
		const uint32_t up_bit = (k > 0) ? cur_bit - (uint32_t)a.nblk : a.up_bit0 + (uint32_t)f * a.frame_bit_step;
		const int last = a.nblk - 1;
		{
			const int m = (lane & 15) >> 1;
			const int bl = min(max(8 * tx - 1 + m, 0), last), br = min(max(8 * tx + m, 0), last);
			ty.params(stream_window(a.stream, cur_bit + bl), stream_window(a.stream, cur_bit + br),
			          stream_window(a.stream, up_bit + bl), stream_window(a.stream, up_bit + br), L::LUMA_OFF);
		}
		{
			const int m = (CSUBX == 2) ? (lane & 7) : ((lane & 15) >> 1);
			const int bl = min(max(8 * tx - 1 + m, 0), last), br = min(max(8 * tx + m, 0), last);
			const uint32_t cl = stream_window(a.stream, cur_bit + bl), cr = stream_window(a.stream, cur_bit + br);
			const uint32_t ul = stream_window(a.stream, up_bit + bl), ur = stream_window(a.stream, up_bit + br);
			tu.params(cl, cr, ul, ur, L::CHROMA_OFF);
			tv.params(cl, cr, ul, ur, L::CHROMA_OFF);
		}

		ty.run(a, lds, L::LUT_OFF, R, ylo2, yhi2);
		tu.run(a, lds, L::LUT_OFF + 1024, R, clo2, chi2);
		tv.run(a, lds, L::LUT_OFF + 2048, R, clo2, chi2);
	}
}

// ---------------------------------------------------------------------------------------
// host-side launcher (called from vfgs_host.cpp)

template <int DEPTH, int CSUBX, int CSUBY>
static hipError_t launch_t(const KernelArgs& a, int grid, hipStream_t stream)
{
	hipLaunchKernelGGL((grain_kernel<DEPTH, CSUBX, CSUBY>), dim3(grid), dim3(kWavesPerWG * 64), 0, stream, a);
	return hipGetLastError();
}

hipError_t launch_grain(const KernelArgs& a, int depth, int csubx, int csuby, int grid, hipStream_t stream)
{
#define VFGS_CASE(D, X, Y) if (depth == D && csubx == X && csuby == Y) return launch_t<D, X, Y>(a, grid, stream)
	VFGS_CASE(10, 2, 2); VFGS_CASE(10, 2, 1); VFGS_CASE(10, 1, 1); VFGS_CASE(10, 1, 2);
	VFGS_CASE(8, 2, 2);  VFGS_CASE(8, 2, 1);  VFGS_CASE(8, 1, 1);  VFGS_CASE(8, 1, 2);
#undef VFGS_CASE
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
