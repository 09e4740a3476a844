// Developer microbenchmark (not product), round 4.  Two questions the round-3 verdict asks to be priced before building:
//
//  (A) "one front per workgroup": the four waves of a workgroup split ONE contiguous run of memory into interleaved 1 KiB
//      segments (wave w: segments w, w + 4, ...), so that a workgroup's traffic is a single contiguous 16 KiB window instead
//      of four windows 15 KB apart (the shipped row walk: one row per wave).  Memory pattern only: the lane 63 -> lane 0
//      hand-over between consecutive segments would cross waves in a real kernel and is NOT modelled (an upper bound).
//
//  (B) "statically persistent row walk" for SMALL launches (1080p x 8: 100 MB): grid = resident workgroup slots, each
//      workgroup stages its table image ONCE and walks a contiguous run of chunks (a chunk = the rows of the shipped
//      workgroup: 4 waves x RPW rows of one block row); per chunk only the block-parameter prologue (EP loads issued one
//      chunk ahead, PRO work, one barrier, double-buffered table).  Against the shipped non-persistent structure on the
//      same bytes, back-to-back launches on one stream (so fill, drain and the gap between launches are all in the figure).
//
// Same conventions as skeleton2.hip: every 16-byte unit is read once and written once in place (nontemporal), a synthetic
// VALU + LDS load per segment (WORK) and per prologue (PRO); results are wrong by design except with --verify
// (WORK = PRO = 0, every dword must come out incremented exactly once).
//
// hipcc --offload-arch=gfx950 -O3 -o skeleton3 skeleton3.hip && ./skeleton3 [rounds] [--verify]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <string>
#include <functional>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
constexpr uint32_t kOOB = 0x80000000u;
constexpr int LDSB = 36 * 1024;     // the general-form luma image

__device__ __forceinline__ void fake_compute(u32x4& v, const uint8_t* lds, int iters, uint32_t salt)
{
	uint32_t a = v.x, b = v.y, c = v.z, d = v.w;
#pragma unroll 4
	for (int i = 0; i < iters; i++)
	{
		const uint32_t t = *(const uint32_t*)(lds + ((a ^ salt) & 0x3ffcu));
		a = __builtin_amdgcn_perm(a, b, 0x06050403u) + t;
		b = (b ^ c) + d;
		c = __builtin_amdgcn_alignbit(c, d, 7) ^ a;
		d = (d + b) ^ (c >> 3);
	}
	v.x = a; v.y = b; v.z = c; v.w = d;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, uint32_t bytes)
{
	return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}

template <int IMG>
__device__ __forceinline__ void stage_issue(u32x4 (&tmp)[9], const uint8_t* tables)
{
#pragma unroll
	for (int i = 0; i < 9; i++) tmp[i] = (i * 4096 < IMG) ? *(const u32x4*)(tables + threadIdx.x * 16 + i * 4096) : u32x4{0, 0, 0, 0};
}
template <int IMG>
__device__ __forceinline__ void stage_commit(const u32x4 (&tmp)[9], uint8_t* lds)
{
#pragma unroll
	for (int i = 0; i < 9; i++) if (i * 4096 < IMG) *(u32x4*)(lds + threadIdx.x * 16 + i * 4096) = tmp[i];
}

// ---- the shipped structure, runtime row geometry: a wave streams rows of `segs` KiB through a ring of four register sets;
// a workgroup = 4 waves x rpw rows (wave w: rows w, w + 4, ...) = one chunk; image of IMG bytes staged per workgroup
template <int IMG, int WORK, int PRO, int EP, int PARLDS, bool VERIFY>
__global__ __launch_bounds__(256) void k_np(uint8_t* __restrict__ buf, uint32_t nchunks, int segs, int rpw, const uint8_t* tables)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds[LDSB + 4096];
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	if (blockIdx.x >= nchunks) return;
	const uint32_t rowb = (uint32_t)segs * 1024;
	const size_t base = ((size_t)blockIdx.x * rpw * 4 + wave) * rowb;
	u32x4 tmp[9];
	stage_issue<IMG>(tmp, tables);
	const __amdgpu_buffer_rsrc_t trs = rsrc(tables, LDSB);
	u32x2 ex[EP ? EP : 1];
#pragma unroll
	for (int i = 0; i < EP; i++) ex[i] = __builtin_amdgcn_raw_buffer_load_b64(trs, (uint32_t)(((blockIdx.x * 37 + i * 11 + lane) & 1023) * 8), 0, 0);
	const __amdgpu_buffer_rsrc_t rs = rsrc(buf + base, (uint32_t)(rpw - 1) * 4 * rowb + rowb);
	u32x4 v[4];
#pragma unroll
	for (int u = 0; u < 4; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, u < segs ? (uint32_t)((u * 64 + lane) * 16) : kOOB, 0, 2);
	stage_commit<IMG>(tmp, lds);
	u32x4 par = {(uint32_t)lane, (uint32_t)blockIdx.x, 3u, 4u};
#pragma unroll
	for (int i = 0; i < EP; i++) { par.z ^= ex[i].x; par.w += ex[i].y; }
	if (PRO) fake_compute(par, lds, PRO, 5u);
	*(uint32_t*)(lds + LDSB + threadIdx.x * 16) = par.x;
	__syncthreads();
	for (int r = 0; r < rpw; r++)
	{
		const uint32_t ro = (uint32_t)r * 4 * rowb;
		for (int s0 = 0; s0 < segs; s0 += 4)
		{
#pragma unroll
			for (int u = 0; u < 4; u++)
			{
				const int sg = s0 + u;
				u32x4 t;
				asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
				             : "=&v"(t.x), "=&v"(t.y), "=&v"(t.z), "=&v"(t.w) : "v"(v[u].x), "v"(v[u].y), "v"(v[u].z), "v"(v[u].w));
				const bool wrap = sg + 4 >= segs;
				const int nsg = wrap ? u : sg + 4;
				const uint32_t noff = (wrap ? ro + 4 * rowb : ro) + (uint32_t)nsg * 1024;
				const bool nvalid = (!wrap || r + 1 < rpw) && nsg < segs;
				v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, nvalid ? (uint32_t)(lane * 16) : kOOB, noff, 2);
				if (sg < segs)
				{
					uint32_t salt = par.x & 0xff;
#pragma unroll
					for (int k = 0; k < PARLDS; k++) salt ^= *(const uint32_t*)(lds + LDSB + ((sg * 64 + lane + k * 17) & 1023) * 4);
					if (VERIFY) t = t + 1u;
					else if (WORK) fake_compute(t, lds, WORK, salt & 0xff);
				}
				__builtin_amdgcn_raw_buffer_store_b128(t, rs, sg < segs ? (uint32_t)(lane * 16) : kOOB, ro + (uint32_t)sg * 1024, 2);
				__builtin_amdgcn_sched_barrier(0);
			}
		}
	}
}

// ---- (B) statically persistent: workgroup g owns chunks [g * cpw, (g + 1) * cpw); image staged once; per chunk: the EP
// prologue loads were issued one chunk earlier, PRO work, table into the chunk's half of a double buffer, ONE barrier; the
// ring of register sets runs on across chunk boundaries (the refill behind a chunk's last row is the next chunk's first)
template <int IMG, int WORK, int PRO, int EP, int PARLDS, bool VERIFY>
__global__ __launch_bounds__(256) void k_pers(uint8_t* __restrict__ buf, uint32_t nchunks, int segs, int rpw, int cpw, int interleave, const uint8_t* tables)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds[LDSB + 2 * 4096];
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	// my chunks: contiguous run [g cpw, (g + 1) cpw), or every gridDim.x-th (the chip sweeps one dense window)
	auto chunk_of = [&](int k) -> uint32_t { return interleave ? blockIdx.x + (uint32_t)k * gridDim.x : blockIdx.x * (uint32_t)cpw + (uint32_t)k; };
	int nmine = 0;
	while (nmine < cpw && chunk_of(nmine) < nchunks) nmine++;
	if (nmine == 0) return;
	const uint32_t rowb = (uint32_t)segs * 1024;
	const uint32_t chunkb = (uint32_t)rpw * 4 * rowb;
	u32x4 tmp[9];
	stage_issue<IMG>(tmp, tables);
	const __amdgpu_buffer_rsrc_t trs = rsrc(tables, LDSB);
	u32x2 ex[EP ? EP : 1];
	auto issue_ex = [&](uint32_t c) {
#pragma unroll
		for (int i = 0; i < EP; i++) ex[i] = __builtin_amdgcn_raw_buffer_load_b64(trs, (uint32_t)(((c * 37 + i * 11 + lane) & 1023) * 8), 0, 0);
	};
	issue_ex(chunk_of(0));
	const __amdgpu_buffer_rsrc_t rs = rsrc(buf, nchunks * chunkb);      // (the whole set: < 2 GiB, checked by the host)
	const uint32_t wbase = (uint32_t)wave * rowb;
	u32x4 v[4];
#pragma unroll
	for (int u = 0; u < 4; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, u < segs ? (uint32_t)((u * 64 + lane) * 16) : kOOB, chunk_of(0) * chunkb + wbase, 2);
	stage_commit<IMG>(tmp, lds);
	for (int k = 0; k < nmine; k++)
	{
		const uint32_t c = chunk_of(k);
		u32x4 par = {(uint32_t)lane, c, 3u, 4u};
#pragma unroll
		for (int i = 0; i < EP; i++) { par.z ^= ex[i].x; par.w += ex[i].y; }
		if (k + 1 < nmine) issue_ex(chunk_of(k + 1));
		if (PRO) fake_compute(par, lds, PRO, 5u);
		uint8_t* ptab = lds + LDSB + (k & 1) * 4096;
		*(uint32_t*)(ptab + threadIdx.x * 16) = par.x;
		__syncthreads();
		const uint32_t co = c * chunkb + wbase;
		const uint32_t nco = (k + 1 < nmine ? chunk_of(k + 1) : 0u) * chunkb + wbase;
		for (int r = 0; r < rpw; r++)
		{
			const uint32_t ro = co + (uint32_t)r * 4 * rowb;
			const bool lastrow = r + 1 == rpw;
			for (int s0 = 0; s0 < segs; s0 += 4)
			{
#pragma unroll
				for (int u = 0; u < 4; u++)
				{
					const int sg = s0 + u;
					u32x4 t;
					asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
					             : "=&v"(t.x), "=&v"(t.y), "=&v"(t.z), "=&v"(t.w) : "v"(v[u].x), "v"(v[u].y), "v"(v[u].z), "v"(v[u].w));
					const bool wrap = sg + 4 >= segs;
					const int nsg = wrap ? u : sg + 4;
					// the row after this one: the wave's next row of the chunk, or its first row of the next chunk
					const uint32_t nro = lastrow ? nco : ro + 4 * rowb;
					const uint32_t noff = (wrap ? nro : ro) + (uint32_t)nsg * 1024;
					const bool nvalid = (!wrap || !lastrow || k + 1 < nmine) && nsg < segs;
					v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, nvalid ? (uint32_t)(lane * 16) : kOOB, noff, 2);
					if (sg < segs)
					{
						uint32_t salt = par.x & 0xff;
#pragma unroll
						for (int q = 0; q < PARLDS; q++) salt ^= *(const uint32_t*)(ptab + ((sg * 64 + lane + q * 17) & 1023) * 4);
						if (VERIFY) t = t + 1u;
						else if (WORK) fake_compute(t, lds, WORK, salt & 0xff);
					}
					__builtin_amdgcn_raw_buffer_store_b128(t, rs, sg < segs ? (uint32_t)(lane * 16) : kOOB, ro + (uint32_t)sg * 1024, 2);
					__builtin_amdgcn_sched_barrier(0);
				}
			}
		}
	}
}

// ---- (A) one front per workgroup: a chunk = 4 * rpw rows = 4 * rpw * segs contiguous KiB; wave w moves segments
// w, w + 4, w + 8, ... of the chunk (ring of four: 16 KiB ahead in the chunk's stream)
template <int IMG, int WORK, int PRO, int EP, int PARLDS, bool VERIFY>
__global__ __launch_bounds__(256) void k_front(uint8_t* __restrict__ buf, uint32_t nchunks, int segs, int rpw, const uint8_t* tables)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds[LDSB + 4096];
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	if (blockIdx.x >= nchunks) return;
	const int tsegs = segs * rpw * 4;                 // KiB of the chunk
	const int mine = (tsegs - wave + 3) / 4;          // my segments: wave + 4 i, i < mine
	u32x4 tmp[9];
	stage_issue<IMG>(tmp, tables);
	const __amdgpu_buffer_rsrc_t trs = rsrc(tables, LDSB);
	u32x2 ex[EP ? EP : 1];
#pragma unroll
	for (int i = 0; i < EP; i++) ex[i] = __builtin_amdgcn_raw_buffer_load_b64(trs, (uint32_t)(((blockIdx.x * 37 + i * 11 + lane) & 1023) * 8), 0, 0);
	const __amdgpu_buffer_rsrc_t rs = rsrc(buf + (size_t)blockIdx.x * tsegs * 1024, (uint32_t)tsegs * 1024);
	const uint32_t wb = (uint32_t)wave * 1024;
	u32x4 v[4];
#pragma unroll
	for (int u = 0; u < 4; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, u < mine ? (uint32_t)(lane * 16) : kOOB, wb + u * 4096, 2);
	stage_commit<IMG>(tmp, lds);
	u32x4 par = {(uint32_t)lane, (uint32_t)blockIdx.x, 3u, 4u};
#pragma unroll
	for (int i = 0; i < EP; i++) { par.z ^= ex[i].x; par.w += ex[i].y; }
	if (PRO) fake_compute(par, lds, PRO, 5u);
	*(uint32_t*)(lds + LDSB + threadIdx.x * 16) = par.x;
	__syncthreads();
	for (int i0 = 0; i0 < mine; i0 += 4)
	{
#pragma unroll
		for (int u = 0; u < 4; u++)
		{
			const int i = i0 + u;
			u32x4 t;
			asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
			             : "=&v"(t.x), "=&v"(t.y), "=&v"(t.z), "=&v"(t.w) : "v"(v[u].x), "v"(v[u].y), "v"(v[u].z), "v"(v[u].w));
			v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, i + 4 < mine ? (uint32_t)(lane * 16) : kOOB, wb + (uint32_t)(i + 4) * 4096, 2);
			if (i < mine)
			{
				uint32_t salt = par.x & 0xff;
#pragma unroll
				for (int k = 0; k < PARLDS; k++) salt ^= *(const uint32_t*)(lds + LDSB + ((i * 64 + lane + k * 17) & 1023) * 4);
				if (VERIFY) t = t + 1u;
				else if (WORK) fake_compute(t, lds, WORK, salt & 0xff);
			}
			__builtin_amdgcn_raw_buffer_store_b128(t, rs, i < mine ? (uint32_t)(lane * 16) : kOOB, wb + (uint32_t)i * 4096, 2);
			__builtin_amdgcn_sched_barrier(0);
		}
	}
}

__global__ __launch_bounds__(256) void k_np1(uint8_t* __restrict__ buf, size_t nbytes)
{
	const int lane = threadIdx.x & 63;
	const size_t base = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4096;
	if (base + 4096 > nbytes) return;
	const __amdgpu_buffer_rsrc_t rs = rsrc(buf + base, 4096);
	u32x4 v[4];
#pragma unroll
	for (int u = 0; u < 4; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (u * 64 + lane) * 16, 0, 2);
#pragma unroll
	for (int u = 0; u < 4; u++) __builtin_amdgcn_raw_buffer_store_b128(v[u] + 1u, rs, (u * 64 + lane) * 16, 0, 2);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Variant { std::string name; std::function<void(int)> launch; std::vector<float> t; };

int main(int argc, char** argv)
{
	bool verify = false;
	int rounds = 5;
	for (int i = 1; i < argc; i++) { if (!strcmp(argv[i], "--verify")) verify = true; else rounds = atoi(argv[i]); }
	setvbuf(stdout, nullptr, _IOLBF, 0);
	hipDeviceProp_t prop;
	CK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	uint8_t* tables;
	CK(hipMalloc(&tables, 48 * 1024)); CK(hipMemset(tables, 3, 48 * 1024));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

	// shapes: (name, bytes per launch, KiB per row, rows per wave) -- rows are rounded to whole KiB, chunks to whole rows
	struct Shape { const char* name; size_t bytes; int segs, rpw; };
	const Shape shapes[] = {
		{"4320p x 8 (1.59 GB)", (size_t)7680 * 4320 * 3 * 8, 16, 1},
		{"1080p x 8 (100 MB)", (size_t)1920 * 1080 * 3 * 8, 4, 2},
		{"1080p x 32 (398 MB)", (size_t)1920 * 1080 * 3 * 32, 4, 2},
		{"2160p x 8 (398 MB)", (size_t)3840 * 2160 * 3 * 8, 8, 1},
	};
	const int POOL = 3;
	const size_t maxset = shapes[0].bytes;
	uint8_t* pool[POOL];
	for (int i = 0; i < POOL; i++) { CK(hipMalloc(&pool[i], maxset)); CK(hipMemset(pool[i], 0x11, maxset)); }

	if (verify)
	{
		int bad_total = 0;
		std::vector<uint32_t> hostbuf(shapes[1].bytes / 4);
		auto check = [&](const char* name, uint32_t nchunks, size_t chunkb, std::function<void()> run) {
			CK(hipMemset(pool[0], 0x11, shapes[1].bytes));
			run();
			CK(hipDeviceSynchronize());
			CK(hipGetLastError());
			CK(hipMemcpy(hostbuf.data(), pool[0], shapes[1].bytes, hipMemcpyDeviceToHost));
			const size_t lim = (size_t)nchunks * chunkb / 4;
			size_t bad = 0, first = 0;
			for (size_t i = 0; i < hostbuf.size(); i++)
			{
				const uint32_t want = i < lim ? 0x11111112u : 0x11111111u;
				if (hostbuf[i] != want) { if (!bad) first = i; bad++; }
			}
			printf("verify %-40s %s (%zu wrong dwords, first at %zu)\n", name, bad ? "FAIL" : "ok", bad, first);
			bad_total += bad != 0;
		};
		for (int segs : {4, 3, 16})
			for (int rpw : {1, 2})
			{
				const size_t chunkb = (size_t)segs * 1024 * rpw * 4;
				const uint32_t nchunks = (uint32_t)(shapes[1].bytes / chunkb) - 1;
				char nm[96];
				snprintf(nm, sizeof nm, "np segs %d rpw %d", segs, rpw);
				check(nm, nchunks, chunkb, [&]() { k_np<LDSB, 0, 0, 0, 0, true><<<nchunks, 256>>>(pool[0], nchunks, segs, rpw, tables); });
				snprintf(nm, sizeof nm, "front segs %d rpw %d", segs, rpw);
				check(nm, nchunks, chunkb, [&]() { k_front<LDSB, 0, 0, 0, 0, true><<<nchunks, 256>>>(pool[0], nchunks, segs, rpw, tables); });
				for (int grid : {4 * cus, 1000, 7})
				{
					const int cpw = (int)((nchunks + grid - 1) / grid);
					snprintf(nm, sizeof nm, "pers segs %d rpw %d grid %d cpw %d", segs, rpw, grid, cpw);
					for (int il : {0, 1}) check(nm, nchunks, chunkb, [&]() { k_pers<LDSB, 0, 0, 2, 0, true><<<grid, 256>>>(pool[0], nchunks, segs, rpw, cpw, il, tables); });
				}
			}
		return bad_total ? 1 : 0;
	}

	for (const Shape& sh : shapes)
	{
		const size_t chunkb = (size_t)sh.segs * 1024 * sh.rpw * 4;
		const uint32_t nchunks = (uint32_t)(sh.bytes / chunkb);
		const size_t set = (size_t)nchunks * chunkb;
		std::vector<Variant> vs;
		const int segs = sh.segs, rpw = sh.rpw;
		vs.push_back({"np1 one 4 KiB item per wave, nothing staged (ceiling)", [&, set](int s) { k_np1<<<(unsigned)(set / 16384), 256>>>(pool[s], set); }, {}});
#define NP(NAME, IMG, WORK, PRO, EP, PL) vs.push_back({NAME, [&, nchunks, segs, rpw](int s) { k_np<IMG, WORK, PRO, EP, PL, false><<<nchunks, 256>>>(pool[s], nchunks, segs, rpw, tables); }, {}})
#define FRONT(NAME, IMG, WORK, PRO, EP, PL) vs.push_back({NAME, [&, nchunks, segs, rpw](int s) { k_front<IMG, WORK, PRO, EP, PL, false><<<nchunks, 256>>>(pool[s], nchunks, segs, rpw, tables); }, {}})
#define PERS(NAME, IMG, WORK, PRO, EP, PL, WGCU, IL) vs.push_back({NAME, [&, nchunks, segs, rpw](int s) { const int grid = std::min<int>(WGCU * cus, nchunks); const int cpw = (nchunks + grid - 1) / grid; \
		k_pers<IMG, WORK, PRO, EP, PL, false><<<(nchunks + cpw - 1) / cpw, 256>>>(pool[s], nchunks, segs, rpw, cpw, IL, tables); }, {}})
		NP("np   shipped structure: image 36 KB, work 15, pro 40, +8", LDSB, 15, 40, 8, 2);
		NP("np   image 12 KB (one-pattern), work 15, pro 40, +8", 12 * 1024, 15, 40, 8, 2);
		NP("np   image 36 KB, no work", LDSB, 0, 0, 0, 0);
		FRONT("front image 36 KB, work 15, pro 40, +8", LDSB, 15, 40, 8, 2);
		FRONT("front image 36 KB, no work", LDSB, 0, 0, 0, 0);
		PERS("pers contiguous 4 wg/cu, image 36 KB, work 15, pro 40, +8", LDSB, 15, 40, 8, 2, 4, 0);
		PERS("pers interleaved 4 wg/cu, image 36 KB, work 15, pro 40, +8", LDSB, 15, 40, 8, 2, 4, 1);
		PERS("pers interleaved 4 wg/cu, image 12 KB, work 15, pro 40, +8", 12 * 1024, 15, 40, 8, 2, 4, 1);
		NP("np   image 36 KB, work 15, pro 40, +0", LDSB, 15, 40, 0, 2);
		NP("np   image 36 KB, work 15, pro 0, +0", LDSB, 15, 0, 0, 2);
		PERS("pers interleaved 4 wg/cu, image 36 KB, work 15, pro 40, +0", LDSB, 15, 40, 0, 2, 4, 1);
		PERS("pers interleaved 4 wg/cu, image 36 KB, work 15, pro 0, +0", LDSB, 15, 0, 0, 2, 4, 1);
		PERS("pers interleaved 4 wg/cu, image 36 KB, work 0, pro 40, +8", LDSB, 0, 40, 8, 2, 4, 1);
		PERS("pers contiguous 4 wg/cu, image 36 KB, no work", LDSB, 0, 0, 0, 0, 4, 0);
		PERS("pers interleaved 4 wg/cu, image 36 KB, no work", LDSB, 0, 0, 0, 0, 4, 1);
		PERS("pers interleaved 3 wg/cu, image 36 KB, work 15, pro 40, +8", LDSB, 15, 40, 8, 2, 3, 1);
		PERS("pers interleaved 2 wg/cu, image 36 KB, work 15, pro 40, +8", LDSB, 15, 40, 8, 2, 2, 1);
		const int reps = set > (500u << 20) ? 6 : 24;
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
		printf("\n== %s: %u chunks of %zu KiB (rows of %d KiB, %d per wave), %d launches back to back per sample ==\n", sh.name, nchunks, chunkb / 1024, segs, rpw, reps);
		printf("%-62s %10s %10s %8s %8s\n", "variant", "med us", "min us", "GB/s", "of 8TB/s");
		for (auto& v : vs)
		{
			std::sort(v.t.begin(), v.t.end());
			const float med = v.t[v.t.size() / 2], mn = v.t[0];
			const double gbs = 2.0 * set / (med * 1e-6) / 1e9;
			printf("%-62s %10.1f %10.1f %8.0f %8.3f\n", v.name.c_str(), med, mn, gbs, gbs / 8000.0);
		}
	}
	return 0;
}
