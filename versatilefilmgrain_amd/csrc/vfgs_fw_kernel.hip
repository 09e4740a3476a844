// Firmware layer on the GPU: grain PATTERN generation for gfx950.
//
// What the reference does on the CPU per configuration (vfgs_fw.c) and what runs here instead:
//
//   frequency-filtered patterns (SEI model 0)         vfgs_fw.c:297-408
//       band-limited Gaussian noise in the low-frequency corner of an NxN block (N = 64 luma,
//       32 chroma), two integer matrix passes with the H.266 DCT-II basis, clip to +-127.
//       -> fw_ff_kernel: one 1024-thread workgroup per pattern, block/basis/intermediate in LDS.
//   auto-regressive patterns (SEI model 1, AFGS1)      vfgs_fw.c:410-502
//       a causal 4x7 filter run in raster order over 82x73 (luma) or 44x38 (chroma) samples,
//       Gaussian noise added to every sample, a 64x64 / 32x32 window cropped out.
//       Sequential on the CPU; sample (y,x) needs row y up to x-1 and rows y-1..y-3 up to
//       x+L, so rows can run skewed by L+1 columns: step t handles x = t - (L+1)y of every row.
//       -> fw_ar_kernel: one wavefront per pattern, lane = row, windows in registers, the row
//          above handed over by DPP; 82 + 4*72 = 370 steps (L = 3) instead of 5986.
//
// Both draw their noise from the firmware's generator, which is the hardware layer's LFSR
// (vfgs_fw.c:284-295): the value used at step n is the 11-bit window at bit n of a per-seed
// bit stream that the host computed once (FwConstants::stream), so no lane iterates the LFSR.
//
// The patterns land in device-resident banks laid out like vfgs_hw.c:49; fw_patch_tables
// copies the generated slots into the slot-interleaved table image the grain kernel reads.
#include <hip/hip_runtime.h>

#include "vfgs_fw_layout.h"
#include "vfgs_layout.h"

namespace vfgs {
namespace {

__device__ inline int noise_index(const uint32_t* w, int n)
{
	// low 11 bits of the generator register after n steps (vfgs_fw.c:375, :486: `n & 2047`)
	const uint64_t two = ((uint64_t)w[(n >> 5) + 1] << 32) | w[n >> 5];
	return (int)((two >> (n & 31)) & 2047u);
}

__device__ inline int clip127(int v) { return v > 127 ? 127 : (v < -127 ? -127 : v); }

// where a finished sample of pattern `jb` goes
__device__ inline void put_sample(const FwLaunch& L, const vfgs_hip_pattern_job& jb, int y, int x, int v)
{
	if (!jb.chroma)
		L.bank[(size_t)jb.index * 4096 + y * 64 + x] = (int8_t)v;                       // vfgs_hw.c:314-318
	else if (L.csubx == 2 && L.csuby == 2)
		L.bank[(size_t)(kSlots + jb.index) * 4096 + y * 64 + x] = (int8_t)v;            // vfgs_hw.c:320-325 with pitch = length = 32
	else
		L.chroma_raw[jb.index * 1024 + y * 32 + x] = (int8_t)v;                         // fw_commit_chroma finishes the copy
}

// ---------------------------------------------------------------------------------------
// frequency-filtered pattern: vfgs_fw.c:362-408 (fill) + :297-360 (two basis passes)
//
// Both passes are small integer matrix products, X = (D^T B + r) >> s and P = clip((X D + 256) >> 9).
// The first has int8 operands on both sides -> v_dot4_i32_i8 over four k at a time; the second
// int16 x int8 -> v_dot2_i32_i16 over two k.  So the operands sit in LDS with k innermost:
//   Bt[i][k]  = B[k][i]          (noise, written transposed by the fill)        row pitch 68 B
//   Dt[j][k]  = D[k*dstep][j]    (basis, transposed while loading)              row pitch 68 B
//   X [j][k]  int16                                                             row pitch 128 B
//   Dp[k/2][i] = {D[k*dstep][i], D[(k+1)*dstep][i]} as int16x2                  row pitch 256 B
// A wavefront shares j (its Dt / X reads are broadcasts) and spreads i over the lanes: the
// 17-dword pitch of Bt and the lane-contiguous Dp make those reads conflict free.
typedef short s16x2_t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(1024) void fw_ff_kernel(FwLaunch L)
{
	const vfgs_hip_pattern_job& jb = L.job[blockIdx.x];
	if (jb.kind != 0) return;
	constexpr int BP = 68;                     // Bt / Dt row pitch in bytes
	__shared__ __attribute__((aligned(16))) int8_t Bt[64 * BP];
	__shared__ __attribute__((aligned(16))) int8_t Dt[64 * BP];
	__shared__ __attribute__((aligned(16))) int16_t X[64 * 64];
	__shared__ __attribute__((aligned(16))) uint32_t Dp[32 * 64];
	__shared__ int8_t G[2048];
	__shared__ uint32_t W[40];
	const int tid = threadIdx.x;
	const int N = jb.chroma ? 32 : 64;       // block size
	const int dstep = 64 / N;                // the 32-point basis is every other row of the 64-point one (vfgs_fw.c:343,354)
	const int gw = N / 16;                   // samples per generator step (vfgs_fw.c:372, :395)
	const int fh = min(gw * (jb.fh + 1), N); // vfgs_fw.c:366-367, :389-390
	const int fv = min(gw * (jb.fv + 1), N);
	const int8_t* D = &L.k->dct[0][0];

	if (tid < 512) ((uint32_t*)G)[tid] = ((const uint32_t*)L.k->gauss)[tid];
	if (tid < 40) W[tid] = L.k->stream[jb.seed_index][tid];
	// basis: one (k, j-quad) dword of D per thread -> four bytes of Dt, one (k-pair, i) entry of Dp per thread pair
	if (tid < N * 16)
	{
		const int k = tid >> 4, j4 = (tid & 15) * 4;        // k < N, j4 < 64
		const uint32_t d = *(const uint32_t*)(D + k * dstep * 64 + j4);
		for (int q = 0; q < 4; q++)
			Dt[(j4 + q) * BP + k] = (int8_t)(d >> (8 * q));
	}
	for (int o = tid; o < (N / 2) * N; o += 1024)
	{
		const int k2 = o / N, i = o % N;
		const int lo = D[(2 * k2) * dstep * 64 + i], hi = D[(2 * k2 + 1) * dstep * 64 + i];
		Dp[k2 * 64 + i] = (uint32_t)(uint16_t)(int16_t)lo | ((uint32_t)(uint16_t)(int16_t)hi << 16);
	}
	__syncthreads();

	// noise in the low-frequency corner, zero elsewhere; one generator step per group of gw samples
	if (tid < N * 16)
	{
		const int l = tid >> 4, k = (tid & 15) * gw;
		const int r = noise_index(W, tid);
		const bool in = k < fh && l < fv;
		for (int j = 0; j < gw; j++)
			Bt[(k + j) * BP + l] = (in && (tid || j)) ? G[(r + j) & 2047] : (int8_t)0;   // (tid || j): no DC (vfgs_fw.c:383, :406)
	}
	__syncthreads();

	// Each thread owns column i and rows j0, j0 + jstep, ... (4 outputs for N = 64, 1 for N = 32).
	const int i = tid % N, j0 = tid / N, jstep = 1024 / N, per = N * N / 1024;
	int acc[4];
	// vertical pass: X[j][i] = (r + sum_k D[k][j] B[k][i]) >> s; rows >= fv of B are zero and are skipped
	for (int m = 0; m < 4; m++) acc[m] = N == 64 ? 256 : 128;   // vfgs_fw.c:307, :340
	for (int k4 = 0; k4 < (fv + 3) / 4; k4++)
	{
		const int b = *(const int*)(Bt + i * BP + 4 * k4);
#pragma unroll
		for (int m = 0; m < 4; m++)
			if (m < per) acc[m] = __builtin_amdgcn_sdot4(*(const int*)(Dt + (j0 + m * jstep) * BP + 4 * k4), b, acc[m], false);
	}
	for (int m = 0; m < per; m++) X[(j0 + m * jstep) * 64 + i] = (int16_t)(acc[m] >> (N == 64 ? 9 : 8));
	__syncthreads();
	// horizontal pass + clip: P[j][i] = (256 + sum_k X[j][k] D[k][i]) >> 9; columns >= fh of X are zero
	for (int m = 0; m < 4; m++) acc[m] = 256;                   // vfgs_fw.c:318, :351
	for (int k2 = 0; k2 < (fh + 1) / 2; k2++)
	{
		const s16x2_t d = __builtin_bit_cast(s16x2_t, Dp[k2 * 64 + i]);
#pragma unroll
		for (int m = 0; m < 4; m++)
			if (m < per)
				acc[m] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2_t, *(const uint32_t*)(X + (j0 + m * jstep) * 64 + 2 * k2)), d, acc[m], false);
	}
	for (int m = 0; m < per; m++) put_sample(L, jb, j0 + m * jstep, i, clip127(acc[m] >> 9));
}

// ---------------------------------------------------------------------------------------
// auto-regressive pattern: vfgs_fw.c:464-501
//
// One wavefront per pattern, lane = row (rows 64..72 of the luma buffer are a second turn of
// lanes 0..8).  With taps reaching L columns to the right in the rows above, row y can run
// K = L+1 columns behind row y-1: at step t lane l works on column t - K*l.  Everything a
// sample needs is then already in registers or long since in LDS:
//   * the row above: the value lane l-1 produced in the previous step is exactly the new right
//     edge (column x+L) of this lane's window -- passed with a wave_ror DPP move, no LDS;
//   * rows y-2, y-3: their column x+L was written at least K steps ago; loaded one step ahead;
//   * the row itself: the lane's own last L results;
//   * the noise term: pre-computed for all samples by all 64 lanes into the buffer itself, the
//     sample later overwrites its own noise.
// The windows are kept in registers and rotate through an unrolled loop of 2L+1 steps, so the
// loop has no barrier, no dependent LDS round trip, and no window copies.
__device__ inline int wave_ror1(int v)
{
	return __builtin_amdgcn_update_dpp(0, v, 0x13C /* wave_ror:1 */, 0xF, 0xF, false);
}

__device__ inline int mad24(int a, int b, int c)
{
	int d;   // int8 samples x int16 taps: the 24-bit multiply-add is a full-rate instruction
	asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "s"(a), "v"(b), "v"(c));   // taps are wave-uniform: SGPR operand
	return d;
}

template <int L>
__device__ __forceinline__ void ar_rows(int8_t* buf, const short* coef, int width, int height, int scale, int lane)
{
	constexpr int K = L + 1, WN = 2 * L + 1;
	int ca[L][WN], co[L];
#pragma unroll
	for (int r = 0; r < L; r++)
#pragma unroll
		for (int p = 0; p < WN; p++)
			ca[r][p] = coef[(3 - (r + 1)) * 7 + 3 + (p - L)];     // row y-(r+1), column x+(p-L)
#pragma unroll
	for (int i = 0; i < L; i++)
		co[i] = coef[3 * 7 + 3 - (i + 1)];                         // row y, column x-(i+1)

	int win[L][WN], own[L];
#pragma unroll
	for (int r = 0; r < L; r++)
#pragma unroll
		for (int p = 0; p < WN; p++) win[r][p] = 0;
#pragma unroll
	for (int i = 0; i < L; i++) own[i] = 0;

	const int rnd = 1 << (scale - 1);
	const int second_at = 64 * K - L;          // lanes 0.. start feeding their second row (y = lane + 64) here
	const int steps = width + K * (height - 1);
	// operands that are actually used lie inside the picture buffer; the positions before a row starts
	// and after it ends read whatever the guard zones around the buffer hold (kArGuardLo / kArGuardHi)
	auto cell = [&](int i) { return (int)buf[i]; };

	// per lane: column x of row y; the row's first cell is `row0`
	int u = -K * lane;                                   // = t - K * lane
	int x = u, y = lane, row0 = lane * width;
	int nz = cell(row0 + x);
	int pre[L];
#pragma unroll
	for (int r = 1; r < L; r++) pre[r] = cell(row0 - (r + 1) * width + x + L);
	int out = 0;

	for (int t0 = 0; t0 < steps; t0 += WN)
	{
#pragma unroll
		for (int s = 0; s < WN; s++)
		{
			// next step's position and operands first: they were written long ago (or never: the
			// noise), so their LDS latency hides behind this step's arithmetic
			const int un = u + 1;
			const bool turn = un == second_at && lane + 64 < height;
			const int xn = turn ? -L : x + 1;
			const int yn = turn ? lane + 64 : y;
			const int row0n = turn ? row0 + 64 * width : row0;
			const int nz_next = cell(row0n + xn);
			int pre_next[L];
#pragma unroll
			for (int r = 1; r < L; r++) pre_next[r] = cell(row0n - (r + 1) * width + xn + L);

			// new right edge of every window (logical position WN-1 lives at physical (WN-1+s) % WN)
			win[0][(WN - 1 + s) % WN] = wave_ror1(out);
#pragma unroll
			for (int r = 1; r < L; r++) win[r][(WN - 1 + s) % WN] = pre[r];
			// one accumulator per row keeps the dependent chains short; the tap that needs the DPP result last
			int acc[L + 1];
#pragma unroll
			for (int r = L - 1; r >= 0; r--)
			{
				acc[r] = 0;
#pragma unroll
				for (int p = 0; p < WN; p++)
					acc[r] = mad24(ca[r][p], win[r][(p + s) % WN], acc[r]);
			}
			acc[L] = 0;
#pragma unroll
			for (int i = L - 1; i >= 0; i--) acc[L] = mad24(co[i], own[i], acc[L]);
			int g = rnd;
#pragma unroll
			for (int r = 0; r <= L; r++) g += acc[r];
			const bool interior = (unsigned)(y - 3) < (unsigned)(height - 3) && (unsigned)(x - 3) < (unsigned)(width - 6);   // vfgs_fw.c:470
			g = interior ? g >> scale : 0;                                                                            // vfgs_fw.c:488
			out = clip127(g + nz);
#pragma unroll
			for (int i = L - 1; i > 0; i--) own[i] = own[i - 1];
			own[0] = out;
			if ((unsigned)x < (unsigned)width && y < height) buf[row0 + x] = (int8_t)out;

			u = un; x = xn; y = yn; row0 = row0n; nz = nz_next;
#pragma unroll
			for (int r = 1; r < L; r++) pre[r] = pre_next[r];
			__builtin_amdgcn_wave_barrier();
		}
	}
}

__global__ __launch_bounds__(256) void fw_ar_kernel(FwLaunch L)
{
	const vfgs_hip_pattern_job& jb = L.job[blockIdx.x];
	if (jb.kind != 1) return;
	// reads of not-yet / no-longer needed positions reach at most 3 rows + 3 columns in front of the buffer
	// and 128 cells behind it
	constexpr int kArGuardLo = 256, kArGuardHi = 128;
	__shared__ int8_t arena[kArGuardLo + 82 * 73 + kArGuardHi];
	__shared__ int8_t G[2048];
	__shared__ uint32_t W[kFwStreamWords];
	int8_t* const buf = arena + kArGuardLo;
	const int tid = threadIdx.x;
	const int sub = jb.chroma ? 2 : 1;
	const int width = sub > 1 ? 44 : 82, height = sub > 1 ? 38 : 73;   // vfgs_fw.c:423-424
	const int shift = jb.shift;

	// four wavefronts prepare (and later store); the recursion itself is one wavefront's work
	for (int i = tid; i < 512; i += 256) ((uint32_t*)G)[i] = ((const uint32_t*)L.k->gauss)[i];
	for (int i = tid; i < kFwStreamWords; i += 256) W[i] = L.k->stream[jb.seed_index][i];
	__syncthreads();
	// the noise term of every sample (vfgs_fw.c:492-493): one generator step per sample in raster order
	for (int n = tid; n < width * height; n += 256)
		buf[n] = (int8_t)(((int)G[noise_index(W, n)] + (1 << (shift - 1))) >> shift);
	__syncthreads();

	// lag = how far the taps reach (rows above / columns to either side)
	int lag = 1;
	for (int j = 0; j < 4; j++)
		for (int i = 0; i < 7; i++)
			if (jb.coef[j * 7 + i] && !(j == 3 && i >= 3))
				lag = max(lag, max(3 - j, abs(i - 3)));
	if (tid < 64)
	{
		if (lag == 1) ar_rows<1>(buf, jb.coef, width, height, jb.scale, tid);
		else if (lag == 2) ar_rows<2>(buf, jb.coef, width, height, jb.scale, tid);
		else ar_rows<3>(buf, jb.coef, width, height, jb.scale, tid);
	}
	__syncthreads();

	// crop (vfgs_fw.c:498-501)
	const int n = 64 / sub, off = 3 + 6 / sub;
	for (int o = tid; o < n * n; o += 256)
	{
		const int y = o / n, x = o % n;
		put_sample(L, jb, y, x, buf[width * (off + y) + off + x]);
	}
}

// ---------------------------------------------------------------------------------------
// Chroma bank copy for layouts other than 4:2:0.  The reference firmware hands the hardware
// layer a 64x64 scratch buffer whose first 1024 bytes are the 32x32 chroma pattern and whose
// tail still holds the last LUMA pattern (vfgs_fw.c:519,603-623); vfgs_set_chroma_pattern reads
// it with a pitch of 64/csuby and a length of 64/csubx (vfgs_hw.c:323-324).  Same bytes here.
__global__ __launch_bounds__(256) void fw_commit_chroma(FwLaunch L)
{
	const vfgs_hip_pattern_job& jb = L.job[blockIdx.x];
	if (!jb.chroma) return;
	const int rows = 64 / L.csuby, cols = 64 / L.csubx, pitch = 64 / L.csuby;
	for (int o = threadIdx.x; o < rows * cols; o += 256)
	{
		const int i = o / cols, x = o % cols;
		const int src = pitch * i + x;
		int8_t v = 0;
		if (src < 1024) v = L.chroma_raw[jb.index * 1024 + src];
		else if (L.last_luma >= 0) v = L.bank[(size_t)L.last_luma * 4096 + src];
		L.bank[(size_t)(kSlots + jb.index) * 4096 + i * 64 + x] = v;
	}
}

// ---------------------------------------------------------------------------------------
// Copy device-generated slots into a device image (image_layout of vfgs_layout.h) that the host
// has just uploaded with the host-set slots and the LUTs.  one_*: that plane type is stored in the
// one-pattern form, holding slot slot_y / slot_cb / slot_cr only (8 = the all-zero pattern: nothing to copy).
struct PatchArgs {
	uint8_t* img;
	const int8_t* bank;
	uint32_t mask_luma, mask_chroma;
	ImageLayout L;
	int one_y, one_c, slot_y, slot_cb, slot_cr;
	int pk16;             // 8 bit: one-pattern banks hold int16 values (vfgs_layout.h "packed 16-bit form")
};

__global__ __launch_bounds__(256) void fw_patch_tables(const PatchArgs p)
{
	const int o = blockIdx.x * 256 + threadIdx.x;
	const ImageLayout& L = p.L;
	if (o < 64 * 64)
	{
		const int r = o >> 6, x = o & 63;
		uint8_t* dst = p.img + L.y_off + L.y_bank;
		if (p.one_y)
		{
			if (p.slot_y < kSlots && (p.mask_luma >> p.slot_y & 1))
			{
				const int v = p.bank[(size_t)p.slot_y * 4096 + o];       // (generated values are clipped to +-127)
				if (p.pk16)
				{
					*(int16_t*)(dst + r * L.y_rs + 2 * x) = (int16_t)v;
					*(int16_t*)(dst + L.y_neg + r * L.y_rs + 2 * x) = (int16_t)-v;
				}
				else
				{
					dst[r * L.y_rs + x] = (uint8_t)v;
					dst[L.y_neg + r * L.y_rs + x] = (uint8_t)-v;
				}
			}
		}
		else
			for (int k = 0; k < kSlots; k++)
				if (p.mask_luma >> k & 1)
					dst[r * L.y_rs + x * kSlots + k] = (uint8_t)p.bank[(size_t)k * 4096 + o];
	}
	else if (o - 4096 < L.cw * L.ch)
	{
		const int q = o - 4096;
		const int r = q / L.cw, x = q % L.cw;
		if (p.one_c)
		{
			for (int c = 0; c < 2; c++)
			{
				const int k = c ? p.slot_cr : p.slot_cb;
				if (k < kSlots && (p.mask_chroma >> k & 1))
				{
					const int v = p.bank[(size_t)(kSlots + k) * 4096 + r * 64 + x];
					if (p.pk16)
					{
						*(int16_t*)(p.img + L.c_off[c] + L.c_bank + r * L.c_rs + 2 * x) = (int16_t)v;
						*(int16_t*)(p.img + L.c_off[c] + L.c_bank + L.c_neg + r * L.c_rs + 2 * x) = (int16_t)-v;
					}
					else
					{
						p.img[L.c_off[c] + L.c_bank + r * L.c_rs + x] = (uint8_t)v;
						p.img[L.c_off[c] + L.c_bank + L.c_neg + r * L.c_rs + x] = (uint8_t)-v;
					}
				}
			}
		}
		else
			for (int k = 0; k < kSlots; k++)
				if (p.mask_chroma >> k & 1)
					for (int c = 0; c < 2; c++)      // (the general form holds the chroma bank once per component's sub-image)
						p.img[L.c_off[c] + L.c_bank + r * L.c_rs + x * kSlots + k] = (uint8_t)p.bank[(size_t)(kSlots + k) * 4096 + r * 64 + x];
	}
}

}  // namespace

hipError_t launch_fw_generate(const FwLaunch& L, hipStream_t stream)
{
	bool ff = false, ar = false, chroma = false;
	for (int i = 0; i < L.njobs; i++)
	{
		(L.job[i].kind ? ar : ff) = true;
		chroma = chroma || L.job[i].chroma;
	}
	if (ff) hipLaunchKernelGGL(fw_ff_kernel, dim3(L.njobs), dim3(1024), 0, stream, L);
	if (ar) hipLaunchKernelGGL(fw_ar_kernel, dim3(L.njobs), dim3(256), 0, stream, L);
	if (chroma && !(L.csubx == 2 && L.csuby == 2))
		hipLaunchKernelGGL(fw_commit_chroma, dim3(L.njobs), dim3(256), 0, stream, L);
	return hipGetLastError();
}

hipError_t launch_fw_patch(uint8_t* img, const int8_t* bank, uint32_t mask_luma, uint32_t mask_chroma, int csubx, int csuby,
                           bool one_y, bool one_c, int slot_y, int slot_cb, int slot_cr, bool depth8, hipStream_t stream)
{
	PatchArgs p{};
	p.img = img; p.bank = bank; p.mask_luma = mask_luma; p.mask_chroma = mask_chroma;
	p.L = image_layout(csubx, csuby, one_y, one_c, depth8);
	p.pk16 = depth8 && kPk16;
	p.one_y = one_y; p.one_c = one_c; p.slot_y = slot_y; p.slot_cb = slot_cb; p.slot_cr = slot_cr;
	const int n = 4096 + p.L.cw * p.L.ch;
	hipLaunchKernelGGL(fw_patch_tables, dim3((n + 255) / 256), dim3(256), 0, stream, p);
	return hipGetLastError();
}

}  // namespace vfgs
