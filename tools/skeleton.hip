// Developer microbenchmark (not product): which MEMORY STRUCTURE can carry the grain kernel's
// arithmetic on 4320p 10-bit 4:2:0 at 8 frames per launch?  All kernels move the same bytes
// (every sample read once, written once, in place unless noted) and run a synthetic VALU + LDS
// load of configurable size per 16-byte unit, so that the overlap behaviour of each structure
// shows; results are wrong by design.
//
//   flat_copy    out-of-place uint4 copy, grid-stride                       (ceiling A)
//   flat_rmw     in-place, 4 KiB contiguous per wave step                   (ceiling B)
//   rowitem      round-1 structure: item = 1 row x 4 segments of one plane, a workgroup takes
//                consecutive items, load 4 -> compute 4 -> store 4, per-item overhead EXTRA
//   strip        item rows are walked tile-by-tile down a block row (strip = tile x 8/16 rows);
//                every wave owns one contiguous range of row items, per-strip overhead EXTRA,
//                the next row's segment is fetched into the registers just stored (rolling
//                prefetch, no second register set)
//
// hipcc --offload-arch=gfx950 -O3 -o skeleton skeleton.hip && ./skeleton
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>
#include <functional>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kOOB = 0x80000000u;
constexpr int kLdsBytes = 48640;    // the product's table image (4:2:0)

struct Geo {
	uint8_t* base[3];       // plane pointers of frame 0
	uint64_t fpitch[3];     // bytes between frames
	uint32_t extent[3];     // bytes of one frame's plane
	int pitch[3];           // row pitch, bytes
	int rows[3];            // rows per frame
	int rpb[3];             // rows per block row (16 luma, 8 chroma 4:2:0)
	int tiles[3];           // tiles per row
	int upt[3];             // lanes used per segment
	int shift[3];           // bytes the segment grid is shifted left
	int nframes;
	int work;               // synthetic compute iterations per segment (8 VALU + 1 LDS gather each)
	int extra;              // synthetic iterations per item (rowitem) / per strip (strip)
};

__device__ __forceinline__ void fake_compute(u32x4& v, const uint8_t* lds, int iters, uint32_t salt)
{
	uint32_t a = v.x, b = v.y, c = v.z, d = v.w;
#pragma unroll 4
	for (int i = 0; i < iters; i++)
	{
		const uint32_t t = *(const uint32_t*)(lds + ((a ^ salt) & 0x3ffcu));   // LUT-like gather
		a = __builtin_amdgcn_perm(a, b, 0x06050403u) + t;
		b = (b ^ c) + d;
		c = __builtin_amdgcn_alignbit(c, d, 7) ^ a;
		d = (d + b) ^ (c >> 3);
	}
	v.x = a; v.y = b; v.z = c; v.w = d;
}

__device__ __forceinline__ void stage_lds(uint8_t* lds, const uint8_t* tables)
{
	for (int i = threadIdx.x * 16; i < kLdsBytes; i += blockDim.x * 16)
		*(u32x4*)(lds + i) = *(const u32x4*)(tables + i);
	__syncthreads();
}

__global__ void k_flat_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i] + 1u;
}

template <int UNROLL>
__global__ void k_flat_rmw(u32x4* __restrict__ buf, size_t n)
{
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
	const int lane = threadIdx.x & 63;
	for (size_t base = wave * 64 * UNROLL; base + 64 * UNROLL <= n; base += nwaves * 64 * UNROLL)
	{
		u32x4 v[UNROLL];
#pragma unroll
		for (int u = 0; u < UNROLL; u++) v[u] = buf[base + u * 64 + lane];
#pragma unroll
		for (int u = 0; u < UNROLL; u++) buf[base + u * 64 + lane] = v[u] + 1u;
	}
}


// non-persistent in-place variants: KB KiB per wave (1 KiB per access), optional XCD-contiguous remap and cache policy
template <int KB, int REMAP, int LDAUX, int STAUX>
__global__ void k_flat_rmw_np2(uint8_t* __restrict__ buf, size_t nbytes)
{
	size_t wg = blockIdx.x;
	if (REMAP) { const size_t per = gridDim.x / 8; wg = (blockIdx.x % 8) * per + blockIdx.x / 8; if (blockIdx.x >= per * 8) wg = blockIdx.x; }
	const size_t wave = wg * (blockDim.x >> 6) + (threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	const size_t base = wave * (size_t)KB * 1024;
	if (base + (size_t)KB * 1024 > nbytes) return;
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(buf + base), 0, KB * 1024, 0x00020000);
	u32x4 v[KB];
#pragma unroll
	for (int u = 0; u < KB; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (u * 64 + lane) * 16, 0, LDAUX);
#pragma unroll
	for (int u = 0; u < KB; u++) __builtin_amdgcn_raw_buffer_store_b128(v[u] + 1u, rs, (u * 64 + lane) * 16, 0, STAUX);
}


// np2 with the grain kernel's occupancy and row walk: LDSKB KiB of LDS per workgroup (36 -> 4 workgroups = 16 waves per CU),
// optional staging of that LDS from a table, ROWS consecutive 4 KiB tiles per wave (stride = the workgroup's 16 KiB) with the
// registers of an access refilled right after its store
template <int LDSKB, int STAGE, int ROWS, int LDAUX, int STAUX, int MODE = 0>
__global__ __launch_bounds__(256) void k_flat_rmw_np4(uint8_t* __restrict__ buf, size_t nbytes, const uint8_t* tables)
{
	// MODE 0: store + refill per access; 1: per row load all, store all; 2: all rows loaded up front; 3: as 0, the wave's rows contiguous
	__shared__ __attribute__((aligned(16))) uint8_t lds[LDSKB * 1024 + 16];
	if (STAGE) { for (int i = threadIdx.x * 16; i < LDSKB * 1024; i += 256 * 16) *(u32x4*)(lds + i) = *(const u32x4*)(tables + i); __syncthreads(); }
	else if (LDSKB) lds[threadIdx.x] = 1;
	const int lane = threadIdx.x & 63;
	const uint32_t rstride = MODE == 3 ? 4096 : 16384;
	const size_t base = MODE == 3 ? ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS * 4096 : ((size_t)blockIdx.x * ROWS * 4 + (threadIdx.x >> 6)) * 4096;
	if ((size_t)(blockIdx.x + 1) * ROWS * 16384 > nbytes) return;
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(buf + base), 0, ROWS * 16384, 0x00020000);
	if (MODE == 2)
	{
		u32x4 w[ROWS][4];
#pragma unroll
		for (int r = 0; r < ROWS; r++)
#pragma unroll
			for (int u = 0; u < 4; u++) w[r][u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (u * 64 + lane) * 16 + r * rstride, 0, LDAUX);
#pragma unroll
		for (int r = 0; r < ROWS; r++)
#pragma unroll
			for (int u = 0; u < 4; u++) __builtin_amdgcn_raw_buffer_store_b128(w[r][u] + (uint32_t)lds[LDSKB ? (w[r][u].x & 1023) : 0], rs, (u * 64 + lane) * 16 + r * rstride, 0, STAUX);
		return;
	}
	u32x4 v[4];
	if (MODE == 1)
	{
		for (int r = 0; r < ROWS; r++)
		{
#pragma unroll
			for (int u = 0; u < 4; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (u * 64 + lane) * 16, r * rstride, LDAUX);
#pragma unroll
			for (int u = 0; u < 4; u++) __builtin_amdgcn_raw_buffer_store_b128(v[u] + (uint32_t)lds[LDSKB ? (v[u].x & 1023) : 0], rs, (u * 64 + lane) * 16, r * rstride, STAUX);
		}
		return;
	}
#pragma unroll
	for (int u = 0; u < 4; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (u * 64 + lane) * 16, 0, LDAUX);
	for (int r = 0; r < ROWS; r++)
	{
#pragma unroll
		for (int u = 0; u < 4; u++)
		{
			__builtin_amdgcn_raw_buffer_store_b128(v[u] + (uint32_t)lds[LDSKB ? (v[u].x & 1023) : 0], rs, (u * 64 + lane) * 16, r * rstride, STAUX);
			v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, r + 1 < ROWS ? (uint32_t)((u * 64 + lane) * 16) : kOOB, (r + 1) * rstride, LDAUX);
			__builtin_amdgcn_sched_barrier(0);
		}
	}
}

// window test: as np4 MODE 0, but row r of workgroup g is the 16 KiB chunk r * gridDim.x + g: at any time the chip works on
// one dense window of the buffer although every wave lives for ROWS chunks (what a persistent grid-stride kernel does)
template <int LDSKB, int ROWS, int LDAUX, int STAUX>
__global__ __launch_bounds__(256) void k_flat_rmw_np5(uint8_t* __restrict__ buf, size_t nbytes)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds[LDSKB * 1024 + 16];
	if (LDSKB) lds[threadIdx.x] = 1;
	const int lane = threadIdx.x & 63;
	const size_t rstride = (size_t)gridDim.x * 16384;
	const uint8_t* b0 = buf + (size_t)blockIdx.x * 16384 + (threadIdx.x >> 6) * 4096;
	if (rstride * ROWS > nbytes) return;
	u32x4 v[4];
	__amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)b0, 0, 4096, 0x00020000);
#pragma unroll
	for (int u = 0; u < 4; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (u * 64 + lane) * 16, 0, LDAUX);
	for (int r = 0; r < ROWS; r++)
	{
		const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc((void*)(b0 + (r + 1) * rstride), 0, r + 1 < ROWS ? 4096 : 0, 0x00020000);
#pragma unroll
		for (int u = 0; u < 4; u++)
		{
			__builtin_amdgcn_raw_buffer_store_b128(v[u] + (uint32_t)lds[LDSKB ? (v[u].x & 1023) : 0], rs, (u * 64 + lane) * 16, 0, STAUX);
			v[u] = __builtin_amdgcn_raw_buffer_load_b128(rn, (u * 64 + lane) * 16, 0, LDAUX);
			__builtin_amdgcn_sched_barrier(0);
		}
		rs = rn;
	}
}

template <int UNROLL, int LDAUX, int STAUX>
__global__ void k_flat_rmw_pers_nt(uint8_t* __restrict__ buf, size_t nbytes)
{
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
	const int lane = threadIdx.x & 63;
	for (size_t base = wave * 1024 * UNROLL; base + 1024 * UNROLL <= nbytes; base += nwaves * 1024 * UNROLL)
	{
		const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(buf + base), 0, 1024 * UNROLL, 0x00020000);
		u32x4 v[UNROLL];
#pragma unroll
		for (int u = 0; u < UNROLL; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (u * 64 + lane) * 16, 0, LDAUX);
#pragma unroll
		for (int u = 0; u < UNROLL; u++) __builtin_amdgcn_raw_buffer_store_b128(v[u] + 1u, rs, (u * 64 + lane) * 16, 0, STAUX);
	}
}

// persistent waves that draw 4 KiB (x ROWS) items from a device counter, in memory order: long-lived waves (tables could stay
// in LDS) whose accesses nevertheless form one dense window.  The next ticket is drawn before the current item is moved.
template <int LDSKB, int ROWS, int LDAUX, int STAUX>
__global__ __launch_bounds__(256) void k_flat_rmw_ticket(uint8_t* __restrict__ buf, size_t nbytes, unsigned* counter)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds[LDSKB * 1024 + 16];
	if (LDSKB) lds[threadIdx.x] = 1;
	const int lane = threadIdx.x & 63;
	const unsigned nitems = (unsigned)(nbytes / ((size_t)ROWS * 4096));
	const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc((void*)counter, 0, 4, 0x00020000);
	auto draw = [&]() { return (unsigned)__builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, crs, lane == 0 ? 0u : kOOB, 0, 0); };
	unsigned mine = (unsigned)__builtin_amdgcn_readfirstlane((int)draw());
	while (mine < nitems)
	{
		const unsigned drawn = draw();          // in flight while this item moves
		const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(buf + (size_t)mine * ROWS * 4096), 0, ROWS * 4096, 0x00020000);
		u32x4 v[4];
#pragma unroll
		for (int u = 0; u < 4; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (u * 64 + lane) * 16, 0, LDAUX);
		for (int r = 0; r < ROWS; r++)
		{
#pragma unroll
			for (int u = 0; u < 4; u++)
			{
				__builtin_amdgcn_raw_buffer_store_b128(v[u] + (uint32_t)lds[LDSKB ? (v[u].x & 1023) : 0], rs, (u * 64 + lane) * 16, r * 4096, STAUX);
				v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, r + 1 < ROWS ? (uint32_t)((u * 64 + lane) * 16) : kOOB, (r + 1) * 4096, LDAUX);
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		mine = (unsigned)__builtin_amdgcn_readfirstlane((int)drawn);
	}
}

// np4 with SEGS x 1 KiB per wave and row instead of 4 KiB, THREADS per workgroup: smaller per-wave footprint (shorter wave lifetime,
// smaller window) at more waves per CU.  Workgroup = (THREADS / 64) waves side by side, each walks ROWS rows of the workgroup's strip.
template <int LDSKB, int SEGS, int ROWS, int THREADS, int LDAUX, int STAUX>
__global__ __launch_bounds__(THREADS) void k_flat_rmw_np6(uint8_t* __restrict__ buf, size_t nbytes)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds[LDSKB * 1024 + 16];
	if (LDSKB) lds[threadIdx.x] = 1;
	const int lane = threadIdx.x & 63;
	constexpr int WAVES = THREADS / 64;
	constexpr uint32_t rstride = WAVES * SEGS * 1024;           // one row of the workgroup
	const size_t base = (size_t)blockIdx.x * ROWS * rstride + (threadIdx.x >> 6) * SEGS * 1024;
	if ((size_t)(blockIdx.x + 1) * ROWS * rstride > nbytes) return;
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(buf + base), 0, ROWS * rstride, 0x00020000);
	u32x4 v[SEGS];
#pragma unroll
	for (int u = 0; u < SEGS; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (u * 64 + lane) * 16, 0, LDAUX);
	for (int r = 0; r < ROWS; r++)
	{
#pragma unroll
		for (int u = 0; u < SEGS; u++)
		{
			__builtin_amdgcn_raw_buffer_store_b128(v[u] + (uint32_t)lds[LDSKB ? (v[u].x & 1023) : 0], rs, (u * 64 + lane) * 16, r * rstride, STAUX);
			v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, r + 1 < ROWS ? (uint32_t)((u * 64 + lane) * 16) : kOOB, (r + 1) * rstride, LDAUX);
			__builtin_amdgcn_sched_barrier(0);
		}
	}
}

// closer to the grain kernel: workgroup of 4 waves, 36 KB table image staged first (loads before the sample loads), barrier, PRO
// iterations of dependent VALU + LDS work once per wave (the block parameters), WORK iterations per segment (the grain), ROWS rows
template <int ROWS, int WORK, int PRO, int LDAUX, int STAUX>
__global__ __launch_bounds__(256) void k_flat_rmw_np8(uint8_t* __restrict__ buf, size_t nbytes, const uint8_t* tables)
{
	constexpr int LDSB = 36 * 1024;
	__shared__ __attribute__((aligned(16))) uint8_t lds[LDSB];
	const int lane = threadIdx.x & 63;
	u32x4 tmp[9];
#pragma unroll
	for (int i = 0; i < 9; i++) tmp[i] = *(const u32x4*)(tables + threadIdx.x * 16 + i * 4096);
	const size_t base = ((size_t)blockIdx.x * ROWS * 4 + (threadIdx.x >> 6)) * 4096;
	if ((size_t)(blockIdx.x + 1) * ROWS * 16384 > nbytes) return;
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(buf + base), 0, ROWS * 16384, 0x00020000);
	u32x4 v[4];
#pragma unroll
	for (int u = 0; u < 4; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (u * 64 + lane) * 16, 0, LDAUX);
#pragma unroll
	for (int i = 0; i < 9; i++) *(u32x4*)(lds + threadIdx.x * 16 + i * 4096) = tmp[i];
	__syncthreads();
	u32x4 par = {(uint32_t)lane, (uint32_t)blockIdx.x, 3u, 4u};
	if (PRO) fake_compute(par, lds, PRO, 5u);
	for (int r = 0; r < ROWS; r++)
	{
#pragma unroll
		for (int u = 0; u < 4; u++)
		{
			if (WORK) fake_compute(v[u], lds, WORK, par.x & 0xff);
			__builtin_amdgcn_raw_buffer_store_b128(v[u], rs, (u * 64 + lane) * 16, r * 16384, STAUX);
			v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, r + 1 < ROWS ? (uint32_t)((u * 64 + lane) * 16) : kOOB, (r + 1) * 16384, LDAUX);
			__builtin_amdgcn_sched_barrier(0);
		}
	}
}

// workgroup = TW tiles x TR rows, one wave per (tile, row): every wave moves ONE 4 KiB row item, the table image is staged once per
// workgroup of TW * TR waves; LDSKB sets how many workgroups fit a CU
template <int TW, int TR, int LDSKB, int WORK, int PRO>
__global__ __launch_bounds__(TW * TR * 64) void k_flat_rmw_np9(uint8_t* __restrict__ buf, size_t nbytes, const uint8_t* tables)
{
	constexpr int THREADS = TW * TR * 64;
	__shared__ __attribute__((aligned(16))) uint8_t lds[LDSKB * 1024];
	const int lane = threadIdx.x & 63;
	constexpr int NIT = (36 * 1024 + THREADS * 16 - 1) / (THREADS * 16);
	u32x4 tmp[NIT];
#pragma unroll
	for (int i = 0; i < NIT; i++) tmp[i] = *(const u32x4*)(tables + min((int)(threadIdx.x * 16 + i * THREADS * 16), 36 * 1024 - 16));
	const size_t base = ((size_t)blockIdx.x * TW * TR + (threadIdx.x >> 6)) * 4096;
	if ((size_t)(blockIdx.x + 1) * TW * TR * 4096 > nbytes) return;
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(buf + base), 0, 4096, 0x00020000);
	u32x4 v[4];
#pragma unroll
	for (int u = 0; u < 4; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (u * 64 + lane) * 16, 0, 2);
#pragma unroll
	for (int i = 0; i < NIT; i++) *(u32x4*)(lds + min((int)(threadIdx.x * 16 + i * THREADS * 16), 36 * 1024 - 16)) = tmp[i];
	__syncthreads();
	u32x4 par = {(uint32_t)lane, (uint32_t)blockIdx.x, 3u, 4u};
	if (PRO) fake_compute(par, lds, PRO, 5u);
#pragma unroll
	for (int u = 0; u < 4; u++)
	{
		if (WORK) fake_compute(v[u], lds, WORK, par.x & 0xff);
		__builtin_amdgcn_raw_buffer_store_b128(v[u], rs, (u * 64 + lane) * 16, 0, 2);
		__builtin_amdgcn_sched_barrier(0);
	}
}

// the same with every access shifted by SHIFT bytes (what the grain kernel's half-block shift does to its 1 KiB accesses),
// UPT lanes per access (the grain kernel: 62), and optionally only the 4 KiB tile shifted while its accesses stay aligned:
// TILEMODE 1: aligned 1 KiB accesses, the tile's first 16 bytes are not stored and the 16 bytes behind it are moved by one lane
template <int SHIFT, int UPT, int TILEMODE, int LDAUX, int STAUX>
__global__ void k_flat_rmw_np3(uint8_t* __restrict__ buf, size_t nbytes)
{
	const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	const size_t tile = (size_t)UPT * 16 * 4;
	const size_t base = wave * tile;
	if (base + tile + 64 > nbytes || base < 64) return;
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(buf + base), 0, (uint32_t)tile + 32, 0x00020000);
	const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(buf + base - 64), 0, (uint32_t)tile + 128, 0x00020000);
	u32x4 v[4], x;
	if (TILEMODE == 0)
	{
#pragma unroll
		for (int u = 0; u < 4; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs2, lane < UPT ? (uint32_t)(64 - SHIFT + (u * UPT + lane) * 16) : kOOB, 0, LDAUX);
#pragma unroll
		for (int u = 0; u < 4; u++) __builtin_amdgcn_raw_buffer_store_b128(v[u] + 1u, rs2, lane < UPT ? (uint32_t)(64 - SHIFT + (u * UPT + lane) * 16) : kOOB, 0, STAUX);
	}
	else
	{
#pragma unroll
		for (int u = 0; u < 4; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (uint32_t)((u * 64 + lane) * 16), 0, LDAUX);
		x = __builtin_amdgcn_raw_buffer_load_b128(rs, lane == 0 ? (uint32_t)tile : kOOB, 0, LDAUX);
#pragma unroll
		for (int u = 0; u < 4; u++) __builtin_amdgcn_raw_buffer_store_b128(v[u] + 1u, rs, (u == 0 && lane == 0) ? kOOB : (uint32_t)((u * 64 + lane) * 16), 0, STAUX);
		__builtin_amdgcn_raw_buffer_store_b128(x + 1u, rs, lane == 0 ? (uint32_t)tile : kOOB, 0, STAUX);
	}
}

// ---- round-1 structure -----------------------------------------------------------------------
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_rowitem(const Geo g, const uint8_t* tables)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds[kLdsBytes];
	stage_lds(lds, tables);
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	int per_plane[3], per_frame = 0;
	for (int p = 0; p < 3; p++) { per_plane[p] = g.rows[p] * g.tiles[p]; per_frame += per_plane[p]; }
	const int nitems = per_frame * g.nframes;
	const int step = gridDim.x * WAVES;
	for (int item = blockIdx.x * WAVES + wave; item < nitems; item += step)
	{
		const int f = item / per_frame;
		int r = item - f * per_frame, p = 0;
		while (r >= per_plane[p]) { r -= per_plane[p]; p++; }
		const int row = r / g.tiles[p], tile = r % g.tiles[p];
		const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(g.base[p] + (uint64_t)f * g.fpitch[p]), 0, g.extent[p], 0x00020000);
		uint32_t off[4];
		u32x4 v[4];
#pragma unroll
		for (int s = 0; s < 4; s++)
		{
			const int x = ((tile * 4 + s) * g.upt[p] + lane) * 16 - g.shift[p];
			const bool ok = lane < g.upt[p] && x >= 0 && x + 16 <= g.pitch[p];
			off[s] = ok ? (uint32_t)(row * g.pitch[p] + x) : kOOB;
			v[s] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[s], 0, 0);
		}
		if (g.extra) { u32x4 t = {off[0], off[1], off[2], off[3]}; fake_compute(t, lds, g.extra, 77u); asm volatile("" :: "v"(t.x), "v"(t.y)); }
#pragma unroll
		for (int s = 0; s < 4; s++)
		{
			fake_compute(v[s], lds, g.work, (uint32_t)s);
			__builtin_amdgcn_raw_buffer_store_b128(v[s], rs, off[s], 0, 0);
		}
	}
}


// non-persistent in-place: one 4 KiB chunk per wave, grid covers the buffer
__global__ void k_flat_rmw_np(u32x4* __restrict__ buf, size_t n)
{
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const int lane = threadIdx.x & 63;
	const size_t base = wave * 256;
	if (base + 256 > n) return;
	u32x4 v[4];
#pragma unroll
	for (int u = 0; u < 4; u++) v[u] = buf[base + u * 64 + lane];
#pragma unroll
	for (int u = 0; u < 4; u++) buf[base + u * 64 + lane] = v[u] + 1u;
}

// persistent in-place with rolling prefetch: chunk k+1 is requested before chunk k is stored
template <int UNROLL>
__global__ void k_flat_rmw_pf(u32x4* __restrict__ buf, size_t n)
{
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
	const int lane = threadIdx.x & 63;
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, 0x7fffffff, 0x00020000);
	const size_t nchunks = n / (64 * UNROLL);
	u32x4 v[UNROLL];
	size_t c = wave;
	if (c >= nchunks) return;
	// chunks beyond 2 GiB: rebase the descriptor per chunk
	auto desc = [&](size_t chunk) { return __builtin_amdgcn_make_buffer_rsrc((void*)(buf + chunk * 64 * UNROLL), 0, chunk < nchunks ? 1024 * UNROLL : 0, 0x00020000); };
	__amdgpu_buffer_rsrc_t cur = desc(c);
#pragma unroll
	for (int u = 0; u < UNROLL; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(cur, (u * 64 + lane) * 16, 0, 0);
	for (; c < nchunks; c += nwaves)
	{
		const __amdgpu_buffer_rsrc_t nxt = desc(c + nwaves);
#pragma unroll
		for (int u = 0; u < UNROLL; u++)
		{
			__builtin_amdgcn_raw_buffer_store_b128(v[u] + 1u, cur, (u * 64 + lane) * 16, 0, 0);
			v[u] = __builtin_amdgcn_raw_buffer_load_b128(nxt, (u * 64 + lane) * 16, 0, 0);
		}
		cur = nxt;
	}
	(void)rs;
}

// ---- round-1 item order with rolling prefetch ------------------------------------------------------
// MODE 0: item = blockIdx * WAVES + wave + k * gridDim * WAVES (round 1);  the next item of a wave is the same
//         tile `step / tiles` rows further down, so the lane geometry is constant inside a plane
// MODE 1: every workgroup owns a contiguous range of rows and sweeps it WAVES items (WAVES / tiles rows) at a time
template <int WAVES, int MODE>
__global__ __launch_bounds__(WAVES * 64) void k_rowitem_pf(const Geo g, const uint8_t* tables)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds[kLdsBytes];
	stage_lds(lds, tables);
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	int per_plane[3], per_frame = 0;
	for (int p = 0; p < 3; p++) { per_plane[p] = g.rows[p] * g.tiles[p]; per_frame += per_plane[p]; }
	const long nitems = (long)per_frame * g.nframes;
	long item, end, step;
	if (MODE == 0) { item = blockIdx.x * WAVES + wave; end = nitems; step = (long)gridDim.x * WAVES; }
	else
	{
		const long lo = (long)blockIdx.x * nitems / gridDim.x / WAVES * WAVES, hi = blockIdx.x + 1 == gridDim.x ? nitems : (long)(blockIdx.x + 1) * nitems / gridDim.x / WAVES * WAVES;
		item = lo + wave; end = hi; step = WAVES;
	}
	struct It { __amdgpu_buffer_rsrc_t rs; uint32_t off[4]; uint32_t rowb; };
	auto decode = [&](long it) {
		It d;
		const bool valid = it < end;
		const long i2 = valid ? it : 0;
		const int f = (int)(i2 / per_frame);
		int r = (int)(i2 - (long)f * per_frame), p = 0;
		while (r >= per_plane[p]) { r -= per_plane[p]; p++; }
		const int row = r / g.tiles[p], tile = r % g.tiles[p];
		d.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(g.base[p] + (uint64_t)f * g.fpitch[p]), 0, valid ? g.extent[p] : 0, 0x00020000);
#pragma unroll
		for (int s = 0; s < 4; s++)
		{
			const int x = ((tile * 4 + s) * g.upt[p] + lane) * 16 - g.shift[p];
			const bool ok = lane < g.upt[p] && x >= 0 && x + 16 <= g.pitch[p];
			d.off[s] = ok ? (uint32_t)x : kOOB;
		}
		d.rowb = (uint32_t)(row * g.pitch[p]);
		return d;
	};
	if (item >= end) return;
	It cur = decode(item);
	u32x4 v[4];
#pragma unroll
	for (int s = 0; s < 4; s++) v[s] = __builtin_amdgcn_raw_buffer_load_b128(cur.rs, cur.off[s], cur.rowb, 0);
	for (; item < end; item += step)
	{
		const It nxt = decode(item + step);
		if (g.extra) { u32x4 t = {cur.off[0], cur.off[1], cur.off[2], cur.off[3]}; fake_compute(t, lds, g.extra, 77u); asm volatile("" :: "v"(t.x), "v"(t.y)); }
#pragma unroll
		for (int s = 0; s < 4; s++)
		{
			u32x4 t = v[s];
			fake_compute(t, lds, g.work, (uint32_t)s);
			__builtin_amdgcn_raw_buffer_store_b128(t, cur.rs, cur.off[s], cur.rowb, 0);
			v[s] = __builtin_amdgcn_raw_buffer_load_b128(nxt.rs, nxt.off[s], nxt.rowb, 0);
		}
		cur = nxt;
	}
}


// ---- round-1 item order, NOT persistent: a workgroup of WAVES waves owns STEPS x WAVES consecutive items and exits;
// the hardware dispatcher hands out workgroups as CUs free up (dynamic balance, compact in-order window)
template <int WAVES, int PF>
__global__ __launch_bounds__(WAVES * 64) void k_rowitem_np(const Geo g, const uint8_t* tables, int steps)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds[kLdsBytes];
	stage_lds(lds, tables);
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	int per_plane[3], per_frame = 0;
	for (int p = 0; p < 3; p++) { per_plane[p] = g.rows[p] * g.tiles[p]; per_frame += per_plane[p]; }
	const long nitems = (long)per_frame * g.nframes;
	long item = (long)blockIdx.x * steps * WAVES + wave;
	const long end = min(nitems, (long)(blockIdx.x + 1) * steps * WAVES);
	struct It { __amdgpu_buffer_rsrc_t rs; uint32_t off[4]; uint32_t rowb; };
	auto decode = [&](long it) {
		It d;
		const bool valid = it < end;
		const long i2 = valid ? it : 0;
		const int f = (int)(i2 / per_frame);
		int r = (int)(i2 - (long)f * per_frame), p = 0;
		while (r >= per_plane[p]) { r -= per_plane[p]; p++; }
		const int row = r / g.tiles[p], tile = r % g.tiles[p];
		d.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(g.base[p] + (uint64_t)f * g.fpitch[p]), 0, valid ? g.extent[p] : 0, 0x00020000);
#pragma unroll
		for (int s = 0; s < 4; s++)
		{
			const int x = ((tile * 4 + s) * g.upt[p] + lane) * 16 - g.shift[p];
			const bool ok = lane < g.upt[p] && x >= 0 && x + 16 <= g.pitch[p];
			d.off[s] = ok ? (uint32_t)x : kOOB;
		}
		d.rowb = (uint32_t)(row * g.pitch[p]);
		return d;
	};
	if (item >= end) return;
	It cur = decode(item);
	u32x4 v[4];
	if (PF)
	{
#pragma unroll
		for (int s = 0; s < 4; s++) v[s] = __builtin_amdgcn_raw_buffer_load_b128(cur.rs, cur.off[s], cur.rowb, 0);
	}
	for (; item < end; item += WAVES)
	{
		const It nxt = decode(item + WAVES);
		if (!PF)
		{
#pragma unroll
			for (int s = 0; s < 4; s++) v[s] = __builtin_amdgcn_raw_buffer_load_b128(cur.rs, cur.off[s], cur.rowb, 0);
		}
		if (g.extra) { u32x4 t = {cur.off[0], cur.off[1], cur.off[2], cur.off[3]}; fake_compute(t, lds, g.extra, 77u); asm volatile("" :: "v"(t.x), "v"(t.y)); }
#pragma unroll
		for (int s = 0; s < 4; s++)
		{
			u32x4 t = v[s];
			fake_compute(t, lds, g.work, (uint32_t)s);
			__builtin_amdgcn_raw_buffer_store_b128(t, cur.rs, cur.off[s], cur.rowb, 0);
			if (PF) v[s] = __builtin_amdgcn_raw_buffer_load_b128(nxt.rs, nxt.off[s], nxt.rowb, 0);
		}
		cur = nxt;
	}
}


// ---- round-1 item order, persistent workgroups that draw chunks of STEPS x WAVES consecutive items from ONE device
// counter (dynamic balance, compact in-order window, no dispatch / staging dead time); the ticket for the chunk after
// next is drawn while the current one is processed, the rolling prefetch runs across chunk boundaries
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_rowitem_dyn(const Geo g, const uint8_t* tables, int steps, unsigned* counter)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds[kLdsBytes];
	__shared__ unsigned s_ticket[2];
	stage_lds(lds, tables);
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	int per_plane[3], per_frame = 0;
	for (int p = 0; p < 3; p++) { per_plane[p] = g.rows[p] * g.tiles[p]; per_frame += per_plane[p]; }
	const long nitems = (long)per_frame * g.nframes;
	const unsigned chunk = steps * WAVES;
	const unsigned nchunks = (unsigned)((nitems + chunk - 1) / chunk);
	struct It { __amdgpu_buffer_rsrc_t rs; uint32_t off[4]; uint32_t rowb; };
	auto decode = [&](long it, bool valid) {
		It d;
		valid = valid && it < nitems;
		const long i2 = valid ? it : 0;
		const int f = (int)(i2 / per_frame);
		int r = (int)(i2 - (long)f * per_frame), p = 0;
		while (r >= per_plane[p]) { r -= per_plane[p]; p++; }
		const int row = r / g.tiles[p], tile = r % g.tiles[p];
		d.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(g.base[p] + (uint64_t)f * g.fpitch[p]), 0, valid ? g.extent[p] : 0, 0x00020000);
#pragma unroll
		for (int s = 0; s < 4; s++)
		{
			const int x = ((tile * 4 + s) * g.upt[p] + lane) * 16 - g.shift[p];
			const bool ok = lane < g.upt[p] && x >= 0 && x + 16 <= g.pitch[p];
			d.off[s] = ok ? (uint32_t)x : kOOB;
		}
		d.rowb = (uint32_t)(row * g.pitch[p]);
		return d;
	};
	// tickets: cur = chunk being processed, nxt = the one after it (already drawn)
	if (threadIdx.x == 0) { s_ticket[0] = atomicAdd(counter, 1u); s_ticket[1] = atomicAdd(counter, 1u); }
	__syncthreads();
	unsigned cur = s_ticket[0], nxt = s_ticket[1];
	__syncthreads();
	if (cur >= nchunks) return;
	It ci = decode((long)cur * chunk + wave, true);
	u32x4 v[4];
#pragma unroll
	for (int s = 0; s < 4; s++) v[s] = __builtin_amdgcn_raw_buffer_load_b128(ci.rs, ci.off[s], ci.rowb, 0);
	while (cur < nchunks)
	{
		// the chunk after next; consumed at the end of this chunk.  Branch-free (one lane has a valid offset): a branch
		// around a vector-memory instruction would make the compiler wait for ALL outstanding loads at every wait
		const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc((void*)counter, 0, 4, 0x00020000);
		const unsigned drawn = (unsigned)__builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, crs, threadIdx.x == 0 ? 0u : kOOB, 0, 0);
		for (int st = 0; st < steps; st++)
		{
			const bool lastst = st + 1 == steps;
			const long nit = lastst ? (long)nxt * chunk + wave : (long)cur * chunk + (st + 1) * WAVES + wave;
			const It ni = decode(nit, !lastst || nxt < nchunks);
			if (g.extra) { u32x4 t = {ci.off[0], ci.off[1], ci.off[2], ci.off[3]}; fake_compute(t, lds, g.extra, 77u); asm volatile("" :: "v"(t.x), "v"(t.y)); }
#pragma unroll
			for (int s = 0; s < 4; s++)
			{
				u32x4 t = v[s];
				fake_compute(t, lds, g.work, (uint32_t)s);
				__builtin_amdgcn_raw_buffer_store_b128(t, ci.rs, ci.off[s], ci.rowb, 0);
				v[s] = __builtin_amdgcn_raw_buffer_load_b128(ni.rs, ni.off[s], ni.rowb, 0);
			}
			ci = ni;
		}
		if (threadIdx.x == 0) s_ticket[0] = drawn;
		__builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0) only
		__builtin_amdgcn_s_barrier();
		cur = nxt;
		nxt = s_ticket[0];
		__builtin_amdgcn_s_barrier();
	}
}

// ---- band order --------------------------------------------------------------------------------
// A band = one block row of one plane (rpb rows x tiles tiles).  A group of GW waves of a workgroup sweeps a band
// GW / tiles rows at a time (each step = that many FULL rows, contiguous); wave role inside the group: tile = i % tiles,
// row phase = i / tiles.  Bands are dealt to groups round-robin.  Per-band overhead EXTRA, rolling prefetch optional.
template <int WAVES, int GW, int PREFETCH>
__global__ __launch_bounds__(WAVES * 64) void k_band(const Geo g, const uint8_t* tables)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds[kLdsBytes];
	stage_lds(lds, tables);
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int grp = wave / GW, wi = wave % GW, ngrp = gridDim.x * (WAVES / GW);
	int bands_plane[3], bands_frame = 0;
	for (int p = 0; p < 3; p++) { bands_plane[p] = (g.rows[p] + g.rpb[p] - 1) / g.rpb[p]; bands_frame += bands_plane[p]; }
	const int nbands = bands_frame * g.nframes;
	for (int band = blockIdx.x * (WAVES / GW) + grp; band < nbands; band += ngrp)
	{
		const int f = band / bands_frame;
		int R = band - f * bands_frame, p = 0;
		while (R >= bands_plane[p]) { R -= bands_plane[p]; p++; }
		const int tiles = g.tiles[p], phases = GW / tiles;
		const int tile = wi % tiles, ph = wi / tiles;
		const int rows_here = min(g.rpb[p], g.rows[p] - R * g.rpb[p]);
		const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(g.base[p] + (uint64_t)f * g.fpitch[p]), 0, g.extent[p], 0x00020000);
		uint32_t off[4];
#pragma unroll
		for (int s = 0; s < 4; s++)
		{
			const int x = ((tile * 4 + s) * g.upt[p] + lane) * 16 - g.shift[p];
			const bool ok = lane < g.upt[p] && x >= 0 && x + 16 <= g.pitch[p];
			off[s] = ok ? (uint32_t)x : kOOB;
		}
		if (g.extra) { u32x4 t = {off[0], off[1], off[2], off[3]}; fake_compute(t, lds, g.extra, 77u); asm volatile("" :: "v"(t.x), "v"(t.y)); }
		const uint32_t rstep = (uint32_t)(phases * g.pitch[p]);
		uint32_t rowb = (uint32_t)((R * g.rpb[p] + ph) * g.pitch[p]);
		u32x4 v[4];
		if (PREFETCH)
		{
			const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)(g.base[p] + (uint64_t)f * g.fpitch[p]), 0, ph < rows_here ? g.extent[p] : 0, 0x00020000);
#pragma unroll
			for (int s = 0; s < 4; s++) v[s] = __builtin_amdgcn_raw_buffer_load_b128(r0, off[s], rowb, 0);
			for (int j = ph; j < rows_here; j += phases)
			{
				const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc((void*)(g.base[p] + (uint64_t)f * g.fpitch[p]), 0, j + phases < rows_here ? g.extent[p] : 0, 0x00020000);
#pragma unroll
				for (int s = 0; s < 4; s++)
				{
					u32x4 t = v[s];
					fake_compute(t, lds, g.work, (uint32_t)s);
					__builtin_amdgcn_raw_buffer_store_b128(t, rs, off[s], rowb, 0);
					v[s] = __builtin_amdgcn_raw_buffer_load_b128(rn, off[s], rowb + rstep, 0);
				}
				rowb += rstep;
			}
		}
		else
		{
			for (int j = ph; j < rows_here; j += phases)
			{
#pragma unroll
				for (int s = 0; s < 4; s++) v[s] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[s], rowb, 0);
#pragma unroll
				for (int s = 0; s < 4; s++)
				{
					fake_compute(v[s], lds, g.work, (uint32_t)s);
					__builtin_amdgcn_raw_buffer_store_b128(v[s], rs, off[s], rowb, 0);
				}
				rowb += rstep;
			}
		}
	}
}

// ---- strips with rolling prefetch ----------------------------------------------------------------
// Row items in the order frame -> plane -> block row -> tile -> row; wave w owns the contiguous
// range [w * n / nwaves, (w + 1) * n / nwaves).
template <int WAVES, int PREFETCH>
__global__ __launch_bounds__(WAVES * 64) void k_strip(const Geo g, const uint8_t* tables)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds[kLdsBytes];
	stage_lds(lds, tables);
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	int per_plane[3], per_frame = 0;
	for (int p = 0; p < 3; p++) { per_plane[p] = g.rows[p] * g.tiles[p]; per_frame += per_plane[p]; }
	const long nitems = (long)per_frame * g.nframes;
	const long nw = (long)gridDim.x * WAVES, w = (long)blockIdx.x * WAVES + wave;
	long item = w * nitems / nw;
	const long end = (w + 1) * nitems / nw;
	while (item < end)
	{
		// decode the strip that `item` lies in
		const int f = (int)(item / per_frame);
		int r = (int)(item - (long)f * per_frame), p = 0;
		while (r >= per_plane[p]) { r -= per_plane[p]; p++; }
		const int per_brow = g.rpb[p] * g.tiles[p];
		const int R = r / per_brow;
		const int rows_here = min(g.rpb[p], g.rows[p] - R * g.rpb[p]);
		const int r2 = r - R * per_brow;
		const int tile = r2 / rows_here, j0 = r2 % rows_here;
		const int nrows = (int)min((long)(rows_here - j0), end - item);     // rows of this strip that are mine
		const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(g.base[p] + (uint64_t)f * g.fpitch[p]), 0, g.extent[p], 0x00020000);
		uint32_t off[4];
#pragma unroll
		for (int s = 0; s < 4; s++)
		{
			const int x = ((tile * 4 + s) * g.upt[p] + lane) * 16 - g.shift[p];
			const bool ok = lane < g.upt[p] && x >= 0 && x + 16 <= g.pitch[p];
			off[s] = ok ? (uint32_t)x : kOOB;
		}
		if (g.extra) { u32x4 t = {off[0], off[1], off[2], off[3]}; fake_compute(t, lds, g.extra, 77u); asm volatile("" :: "v"(t.x), "v"(t.y)); }
		uint32_t rowb = (uint32_t)((R * g.rpb[p] + j0) * g.pitch[p]);
		u32x4 v[4];
		if (PREFETCH)
		{
#pragma unroll
			for (int s = 0; s < 4; s++) v[s] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[s], rowb, 0);
			for (int j = 0; j < nrows; j++)
			{
				// the prefetch of the row after the last one goes through a descriptor with zero records:
				// the hardware drops it, the instruction stream (and the compiler's vmcnt counting) stays fixed
				const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc((void*)(g.base[p] + (uint64_t)f * g.fpitch[p]), 0, j + 1 < nrows ? g.extent[p] : 0, 0x00020000);
#pragma unroll
				for (int s = 0; s < 4; s++)
				{
					u32x4 t = v[s];
					fake_compute(t, lds, g.work, (uint32_t)s);
					__builtin_amdgcn_raw_buffer_store_b128(t, rs, off[s], rowb, 0);
					v[s] = __builtin_amdgcn_raw_buffer_load_b128(rn, off[s], rowb + g.pitch[p], 0);
				}
				rowb += g.pitch[p];
			}
		}
		else
		{
			for (int j = 0; j < nrows; j++)
			{
#pragma unroll
				for (int s = 0; s < 4; s++) v[s] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[s], rowb, 0);
#pragma unroll
				for (int s = 0; s < 4; s++)
				{
					fake_compute(v[s], lds, g.work, (uint32_t)s);
					__builtin_amdgcn_raw_buffer_store_b128(v[s], rs, off[s], rowb, 0);
				}
				rowb += g.pitch[p];
			}
		}
		item += nrows;
	}
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv)
{
	const int W = 7680, H = 4320, NF = 8, POOL = 3;
	const size_t ybytes = (size_t)W * H * 2, cbytes = ybytes / 4, fbytes = ybytes + 2 * cbytes, set = fbytes * NF;
	uint8_t* pool[POOL];
	uint8_t* alt;
	for (int i = 0; i < POOL; i++) { CK(hipMalloc(&pool[i], set)); CK(hipMemset(pool[i], 0x5a + i, set)); }
	CK(hipMalloc(&alt, set));
	unsigned* counter;
	CK(hipMalloc(&counter, 64));
	uint8_t* tables;
	CK(hipMalloc(&tables, kLdsBytes)); CK(hipMemset(tables, 3, kLdsBytes));
	hipDeviceProp_t prop;
	CK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

	auto geo = [&](int s, int work, int extra) {
		Geo g{};
		uint8_t* b = pool[s];
		g.base[0] = b; g.base[1] = b + ybytes * NF; g.base[2] = b + ybytes * NF + cbytes * NF;
		g.fpitch[0] = ybytes; g.fpitch[1] = g.fpitch[2] = cbytes;
		g.extent[0] = (uint32_t)ybytes; g.extent[1] = g.extent[2] = (uint32_t)cbytes;
		g.pitch[0] = W * 2; g.pitch[1] = g.pitch[2] = W;
		g.rows[0] = H; g.rows[1] = g.rows[2] = H / 2;
		g.rpb[0] = 16; g.rpb[1] = g.rpb[2] = 8;
		g.tiles[0] = 4; g.tiles[1] = g.tiles[2] = 2;
		g.upt[0] = 62; g.upt[1] = g.upt[2] = 61;
		g.shift[0] = 16; g.shift[1] = g.shift[2] = 8;
		g.nframes = NF; g.work = work; g.extra = extra;
		return g;
	};

	struct Variant { std::string name; std::function<void(int)> launch; std::vector<float> t; };
	std::vector<Variant> vs;
	const size_t n16 = set / 16;
	const size_t set_b = set;
#define NP2(NAME, KB, REMAP, LA, SA, THREADS) vs.push_back({NAME, [&](int s) { k_flat_rmw_np2<KB, REMAP, LA, SA><<<(unsigned)(set_b / ((size_t)KB * 1024 * (THREADS / 64))), THREADS>>>(pool[s], set_b); }, {}})
#define NP3(NAME, SHIFT, UPT, TM, LA, SA) vs.push_back({NAME, [&](int s) { k_flat_rmw_np3<SHIFT, UPT, TM, LA, SA><<<(unsigned)(set_b / ((size_t)UPT * 64 * 4)), 256>>>(pool[s], set_b); }, {}})
#define NP4(NAME, LDSKB, STAGE, ROWS, LA, SA, MODE) vs.push_back({NAME, [&](int s) { k_flat_rmw_np4<LDSKB, STAGE, ROWS, LA, SA, MODE><<<(unsigned)(set_b / ((size_t)ROWS * 16384)), 256>>>(pool[s], set_b, tables); }, {}})
#define NP5(NAME, LDSKB, ROWS, LA, SA) vs.push_back({NAME, [&](int s) { k_flat_rmw_np5<LDSKB, ROWS, LA, SA><<<(unsigned)(set_b / ((size_t)ROWS * 16384)), 256>>>(pool[s], set_b); }, {}})
#define PNT(NAME, WGCU, LA, SA) vs.push_back({NAME, [&](int s) { k_flat_rmw_pers_nt<4, LA, SA><<<WGCU * cus, 256>>>(pool[s], set_b); }, {}})
#define TKT(NAME, LDSKB, ROWS, WGCU, LA, SA) vs.push_back({NAME, [&](int s) { CK(hipMemsetAsync(counter, 0, 4)); k_flat_rmw_ticket<LDSKB, ROWS, LA, SA><<<WGCU * cus, 256>>>(pool[s], set_b, counter); }, {}})
#define NP8(NAME, ROWS, WORK, PRO) vs.push_back({NAME, [&](int s) { k_flat_rmw_np8<ROWS, WORK, PRO, 2, 2><<<(unsigned)(set_b / ((size_t)ROWS * 16384)), 256>>>(pool[s], set_b, tables); }, {}})
#define NP9(NAME, TW, TR, LDSKB, WORK, PRO) vs.push_back({NAME, [&](int s) { k_flat_rmw_np9<TW, TR, LDSKB, WORK, PRO><<<(unsigned)(set_b / ((size_t)TW * TR * 4096)), TW * TR * 64>>>(pool[s], set_b, tables); }, {}})
	NP4("np4 36K LDS rows 1 both nt", 36, 0, 1, 2, 2, 0);
	NP8("np8 staged rows 4, work 12, prologue 40", 4, 12, 40);
	NP8("np8 staged rows 2, work 12, prologue 40", 2, 12, 40);
	NP9("np9 16 waves/wg (4x4) 1 row each, 36K: 2 wg/cu", 4, 4, 36, 0, 0);
	NP9("np9 16 waves/wg (4x4) 1 row each, 100K: 1 wg/cu", 4, 4, 100, 0, 0);
	NP9("np9 16 waves/wg, 100K, work 12 prologue 40", 4, 4, 100, 12, 40);
	NP9("np9 16 waves/wg, 36K, work 12 prologue 40", 4, 4, 36, 12, 40);
	NP9("np9 8 waves/wg (4x2) 1 row each, 72K: 2 wg/cu", 4, 2, 72, 0, 0);
	NP9("np9 8 waves/wg, 72K, work 12 prologue 40", 4, 2, 72, 12, 40);
	NP9("np9 8 waves/wg, 36K (4 wg/cu), work 12 prologue 40", 4, 2, 36, 12, 40);
	NP9("np9 4 waves/wg (4x1), 36K (4 wg/cu), work 12 pro 40", 4, 1, 36, 12, 40);
	NP2("np2 4KiB/wave aligned plain", 4, 0, 0, 0, 256);
	NP2("np2 4KiB/wave aligned both nt", 4, 0, 2, 2, 256);
	NP2("np2 4KiB/wave aligned both nt XCD-contig", 4, 1, 2, 2, 256);
	NP3("np3 tile-shift aligned accesses plain", 0, 64, 1, 0, 0);
	NP3("np3 tile-shift aligned accesses both nt", 0, 64, 1, 2, 2);
	NP3("np3 tile-shift aligned accesses nt loads", 0, 64, 1, 2, 0);

	const int rounds = argc > 1 ? atoi(argv[1]) : 5, reps = 6;
	for (int r = 0; r <= rounds; r++)
		for (auto& v : vs)
		{
			CK(hipEventRecord(e0));
			for (int k = 0; k < reps; k++) v.launch(k % POOL);
			CK(hipEventRecord(e1));
			CK(hipEventSynchronize(e1));
			CK(hipGetLastError());
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			if (r) v.t.push_back(ms / reps * 1e3f);
		}
	printf("%-44s %10s %10s %8s %8s\n", "variant (8 frames 4320p 10b 4:2:0 / launch)", "med us", "min us", "GB/s", "of 8TB/s");
	for (auto& v : vs)
	{
		std::sort(v.t.begin(), v.t.end());
		const float med = v.t[v.t.size() / 2], mn = v.t[0];
		const double gbs = 2.0 * set / (med * 1e-6) / 1e9;
		printf("%-44s %10.1f %10.1f %8.0f %8.3f\n", v.name.c_str(), med, mn, gbs, gbs / 8000.0);
	}
	return 0;
}
