// Firmware layer on the GPU: grain PATTERN generation for gfx950.
//
// What the reference does on the CPU per configuration (vfgs_fw.c) and what runs here instead:
//
//   frequency-filtered patterns (SEI model 0)         vfgs_fw.c:297-408
//       band-limited Gaussian noise in the low-frequency corner of an NxN block (N = 64 luma,
//       32 chroma), two integer matrix passes with the H.266 DCT-II basis, clip to +-127.
//       -> fw_ff_kernel: one 256-thread workgroup per pattern, block/basis/intermediate in LDS.
//   auto-regressive patterns (SEI model 1, AFGS1)      vfgs_fw.c:410-502
//       a causal 4x7 filter run in raster order over 82x73 (luma) or 44x38 (chroma) samples,
//       Gaussian noise added to every sample, a 64x64 / 32x32 window cropped out.
//       Sequential on the CPU; sample (y,x) needs row y up to x-1 and rows y-1..y-3 up to
//       x+3, so rows can run skewed by 4 columns: step t handles x = t - 4y of every row.
//       -> fw_ar_kernel: one wavefront per pattern, lane l owns row (t>>2) - l, <= 21 lanes
//          busy, 82 + 4*72 = 370 steps instead of 5986.
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
__global__ __launch_bounds__(256) void fw_ff_kernel(FwLaunch L)
{
	const vfgs_hip_pattern_job& jb = L.job[blockIdx.x];
	if (jb.kind != 0) return;
	__shared__ int8_t B[64 * 64];
	__shared__ int16_t X[64 * 64];
	__shared__ int8_t D[64 * 64];
	__shared__ int8_t G[2048];
	__shared__ uint32_t W[40];
	const int tid = threadIdx.x;
	const int N = jb.chroma ? 32 : 64;       // block size
	const int dstep = 64 / N;                // the 32-point basis is every other row of the 64-point one (vfgs_fw.c:343,354)
	const int gw = N / 16;                   // samples per generator step (vfgs_fw.c:372, :395)
	const int fh = min(gw * (jb.fh + 1), N); // vfgs_fw.c:366-367, :389-390
	const int fv = min(gw * (jb.fv + 1), N);

	for (int i = tid; i < 1024; i += 256) ((uint32_t*)D)[i] = ((const uint32_t*)L.k->dct)[i];
	for (int i = tid; i < 512; i += 256) ((uint32_t*)G)[i] = ((const uint32_t*)L.k->gauss)[i];
	if (tid < 40) W[tid] = L.k->stream[jb.seed_index][tid];
	__syncthreads();

	// noise in the low-frequency corner, zero elsewhere; one generator step per group of gw samples
	for (int g = tid; g < N * 16; g += 256)
	{
		const int l = g >> 4, k = (g & 15) * gw;
		const int r = noise_index(W, g);
		const bool in = k < fh && l < fv;
		for (int j = 0; j < gw; j++)
			B[l * N + k + j] = in ? G[(r + j) & 2047] : (int8_t)0;
	}
	__syncthreads();
	if (tid == 0) B[0] = 0;                  // no DC (vfgs_fw.c:383, :406)
	__syncthreads();

	// vertical pass; rows >= fv of B are zero and are skipped
	for (int o = tid; o < N * N; o += 256)
	{
		const int j = o / N, i = o % N;
		int acc = N == 64 ? 256 : 128;       // vfgs_fw.c:307, :340
		for (int k = 0; k < fv; k++)
			acc += (int)D[k * dstep * 64 + j] * (int)B[k * N + i];
		X[o] = (int16_t)(acc >> (N == 64 ? 9 : 8));
	}
	__syncthreads();
	// horizontal pass + clip; columns >= fh of X are zero
	for (int o = tid; o < N * N; o += 256)
	{
		const int j = o / N, i = o % N;
		int acc = 256;                       // vfgs_fw.c:318, :351
		for (int k = 0; k < fh; k++)
			acc += (int)X[j * N + k] * (int)D[k * dstep * 64 + i];
		put_sample(L, jb, j, i, clip127(acc >> 9));
	}
}

// ---------------------------------------------------------------------------------------
// auto-regressive pattern: vfgs_fw.c:464-501, rows skewed by four columns
__global__ __launch_bounds__(64) void fw_ar_kernel(FwLaunch L)
{
	const vfgs_hip_pattern_job& jb = L.job[blockIdx.x];
	if (jb.kind != 1) return;
	__shared__ int8_t buf[82 * 73];
	__shared__ int8_t G[2048];
	__shared__ uint32_t W[kFwStreamWords];
	const int lane = threadIdx.x;
	const int sub = jb.chroma ? 2 : 1;
	const int width = sub > 1 ? 44 : 82, height = sub > 1 ? 38 : 73;   // vfgs_fw.c:423-424
	const int scale = jb.scale, shift = jb.shift;

	for (int i = lane; i < 512; i += 64) ((uint32_t*)G)[i] = ((const uint32_t*)L.k->gauss)[i];
	for (int i = lane; i < kFwStreamWords; i += 64) W[i] = L.k->stream[jb.seed_index][i];
	int c[28];
	for (int i = 0; i < 28; i++) c[i] = jb.coef[i];
	__syncthreads();

	const int steps = width + 4 * (height - 1);
	for (int t = 0; t < steps; t++)
	{
		const int y = (t >> 2) - lane;
		const int x = t - 4 * y;             // = (t & 3) + 4 * lane
		if (y >= 0 && y < height && x < width)
		{
			int g = 0;
			if (y >= 3 && x >= 3 && x < width - 3)                     // vfgs_fw.c:470
			{
				const int8_t* p = buf + width * y + x;
#pragma unroll
				for (int j = -3; j <= 0; j++)
#pragma unroll
					for (int i = -3; i <= 3; i++)
						if (i < 0 || j < 0)
							g += c[(3 + j) * 7 + 3 + i] * (int)p[width * j + i];
				g = (g + (1 << (scale - 1))) >> scale;                 // vfgs_fw.c:488
			}
			const int n = y * width + x;                               // generator steps taken before this sample
			g += ((int)G[noise_index(W, n)] + (1 << (shift - 1))) >> shift;   // vfgs_fw.c:492
			buf[n] = (int8_t)clip127(g);
		}
		__syncthreads();   // one wavefront: orders this step's LDS writes before the next step's reads
	}

	// crop (vfgs_fw.c:498-501)
	const int n = 64 / sub, off = 3 + 6 / sub;
	for (int o = lane; o < n * n; o += 64)
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
// Copy device-generated slots into a table image (TableLayout of vfgs_layout.h) that the host
// has just uploaded with the host-set slots and the LUTs.
__global__ __launch_bounds__(256) void fw_patch_tables(uint8_t* img, const int8_t* bank, uint32_t mask_luma, uint32_t mask_chroma,
                                                       int lrs, int chroma_off, int cw, int ch, int crs)
{
	const int o = blockIdx.x * 256 + threadIdx.x;
	if (o < 64 * 64)
	{
		const int r = o >> 6, x = o & 63;
		for (int k = 0; k < kSlots; k++)
			if (mask_luma >> k & 1)
				img[r * lrs + x * kSlots + k] = (uint8_t)bank[(size_t)k * 4096 + o];
	}
	else if (o - 4096 < cw * ch)
	{
		const int q = o - 4096;
		const int r = q / cw, x = q % cw;
		for (int k = 0; k < kSlots; k++)
			if (mask_chroma >> k & 1)
				img[chroma_off + r * crs + x * kSlots + k] = (uint8_t)bank[(size_t)(kSlots + k) * 4096 + r * 64 + x];
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
	if (ff) hipLaunchKernelGGL(fw_ff_kernel, dim3(L.njobs), dim3(256), 0, stream, L);
	if (ar) hipLaunchKernelGGL(fw_ar_kernel, dim3(L.njobs), dim3(64), 0, stream, L);
	if (chroma && !(L.csubx == 2 && L.csuby == 2))
		hipLaunchKernelGGL(fw_commit_chroma, dim3(L.njobs), dim3(256), 0, stream, L);
	return hipGetLastError();
}

hipError_t launch_fw_patch(uint8_t* img, const int8_t* bank, uint32_t mask_luma, uint32_t mask_chroma, int csubx, int csuby, hipStream_t stream)
{
	const int cw = 64 / csubx, ch = 64 / csuby;
	const int lrs = 64 * kSlots + 16, crs = cw * kSlots + 16;
	const int n = 4096 + cw * ch;
	hipLaunchKernelGGL(fw_patch_tables, dim3((n + 255) / 256), dim3(256), 0, stream, img, bank, mask_luma, mask_chroma, lrs, 64 * lrs, cw, ch, crs);
	return hipGetLastError();
}

}  // namespace vfgs
