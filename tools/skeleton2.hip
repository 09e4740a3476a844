// Developer microbenchmark (not product), round 3: can PERSISTENT workgroups -- table image staged once per
// launch -- keep the chip on one dense window of memory when their work is handed out by tickets?
//
// All kernels move the same bytes as the grain kernel at 8 frames of 4320p 10-bit 4:2:0 per launch (every 16-byte
// unit read once and written once, in place, nontemporal, 4 KiB "row items" = 4 wave accesses of 1 KiB) and run a
// synthetic VALU + LDS load per segment (WORK) and per item (PRO: the block parameters); results are wrong by
// design except in --verify mode (WORK = PRO = 0: every dword must come out incremented exactly once).
//
//   ring     one ticket = one CHUNK of WAVES consecutive items, drawn by wave 0 of a workgroup from the ticket head of
//            its XCD (8 heads, chunk = 8 n + head; exhausted heads are passed on to the next XCD's) D chunks ahead and
//            published to the other waves through a ring of R slots in LDS; wave w always takes item w of a chunk; a
//            slot is only overwritten when all WAVES waves have consumed it.  No barrier after the prologue.
//   wtix     every wave draws its own items from NH global heads (item = NH n + head), no stealing
//   static   persistent grid-stride (item = wave + k * waves)
//   np8      NOT persistent: the shipped structure (4 waves, table image staged per workgroup, 4 rows per wave)
//   np1      NOT persistent, nothing staged, one item per wave: the ceiling
//
// hipcc --offload-arch=gfx950 -O3 -o skeleton2 skeleton2.hip && ./skeleton2 [rounds] [--verify]
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
constexpr uint32_t kOOB = 0x80000000u;
constexpr int kLdsBytes = 48 * 1024;    // both plane types' table images
constexpr uint32_t kSkip = 0xfffffffeu, kEnd = 0xffffffffu;
constexpr uint32_t kMaxSpins = 1u << 21;   // x ~100 cycles: a fraction of a second

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

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, uint32_t bytes)
{
	return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}

__device__ __forceinline__ int xcc_id()
{
	int x;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
	return x & 7;
}


// LDS words used for synchronisation are accessed with explicit instructions: a `volatile` access makes hipcc drain ALL
// counters (s_waitcnt vmcnt(0)) around it, which would serialise the rolling prefetch at every poll
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)p; }
__device__ __forceinline__ unsigned long long lds_ld64(const void* p)
{
	unsigned long long v;
	asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr(p)) : "memory");
	return v;
}
__device__ __forceinline__ uint32_t lds_ld32(const void* p)
{
	uint32_t v;
	asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr(p)) : "memory");
	return v;
}
__device__ __forceinline__ void lds_st64(void* p, unsigned long long v) { asm volatile("ds_write_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(lds_addr(p)), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_st32(void* p, uint32_t v) { asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(lds_addr(p)), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_inc(void* p) { asm volatile("ds_add_u32 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(lds_addr(p)), "v"(1u) : "memory"); }
__device__ __forceinline__ uint32_t lds_inc_rtn(void* p)
{
	uint32_t v;
	asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr(p)), "v"(1u) : "memory");
	return v;
}

template <int THREADS>
__device__ __forceinline__ void stage(uint8_t* lds, const uint8_t* tables)
{
	for (int i = threadIdx.x * 16; i < kLdsBytes; i += THREADS * 16) *(u32x4*)(lds + i) = *(const u32x4*)(tables + i);
}

// the item loop body shared by the persistent variants: v holds the current item's 4 segments (in flight), the refills
// for the NEXT item are issued as each segment is consumed
template <int WORK, int PRO, bool VERIFY>
__device__ __forceinline__ void do_item(u32x4 (&v)[4], const uint8_t* lds, __amdgpu_buffer_rsrc_t cur, __amdgpu_buffer_rsrc_t nxt, int lane, uint32_t item)
{
	u32x4 par = {(uint32_t)lane, item, 3u, 4u};
	if (PRO) fake_compute(par, lds, PRO, 5u);
#pragma unroll
	for (int u = 0; u < 4; u++)
	{
		// (an explicit copy -- the grain kernel's lane rotation -- so that the refill lands in the SAME registers every
		// iteration; a renamed destination would have to be copied back at the loop end, behind a wait for the data)
		u32x4 t;
		asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
		             : "=&v"(t.x), "=&v"(t.y), "=&v"(t.z), "=&v"(t.w) : "v"(v[u].x), "v"(v[u].y), "v"(v[u].z), "v"(v[u].w));
		v[u] = __builtin_amdgcn_raw_buffer_load_b128(nxt, (u * 64 + lane) * 16, 0, 2);
		if (VERIFY) t = t + 1u;
		else if (WORK) fake_compute(t, lds, WORK, par.x & 0xff);
		__builtin_amdgcn_raw_buffer_store_b128(t, cur, (u * 64 + lane) * 16, 0, 2);
		__builtin_amdgcn_sched_barrier(0);
	}
}


// ---- vector memory under explicit control ------------------------------------------------------------------------------
// hipcc derives s_waitcnt vmcnt(N) per loop from the MERGE of the loop's entry and back-edge states, so a rolling-prefetch
// loop whose first iteration starts with fewer operations in flight than a steady-state one gets the first iteration's
// (small) N forever -- the shipped grain kernel drains to vmcnt(0) at the top of every row.  The X variants below issue
// their loads, stores and ticket atomics from inline asm (invisible to that pass) and wait with hand-counted N; the
// prologue issues dummy (out-of-range, dropped) stores so that the queue has its steady-state shape from the start.
// s_nop 4: an SGPR operand written by a VALU instruction (v_readfirstlane, v_readlane of a spilled SGPR) needs 5 wait
// states before a VMEM instruction reads it, and the hazard recognizer does not look into inline asm.
typedef uint32_t desc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ desc_t mkdesc(const void* p, uint32_t bytes)
{
	const uint64_t a = (uint64_t)p;
	desc_t d = {(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, bytes, 0x00020000u};
	d.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.x); d.y = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.y);
	d.z = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.z); d.w = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.w);
	return d;
}
__device__ __forceinline__ void xload(u32x4& d, uint32_t voff, desc_t rs)
{
	asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen nt" : "=v"(d) : "v"(voff), "s"(rs) : "memory");
}
__device__ __forceinline__ void xstore(const u32x4& d, uint32_t voff, desc_t rs)
{
	asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, 0 offen nt\n\ts_nop 1" :: "v"(d), "v"(voff), "s"(rs) : "memory");
}
__device__ __forceinline__ void xatomic_inc(uint32_t& d, uint32_t voff, desc_t rs)
{
	d = 1;
	asm volatile("s_nop 4\n\tbuffer_atomic_add %0, %1, %2, 0 offen sc0 sc1" : "+v"(d) : "v"(voff), "s"(rs) : "memory");
}
template <int N> __device__ __forceinline__ void xwait(u32x4& d) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(d) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void xwait1(uint32_t& d) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(d) : "n"(N) : "memory"); }

// item body with explicit waits.  Queue per item, in issue order: [A] L0' S0 L1' S1 L2' S2 L3' S3  (A: the ticket atomic of
// the variants that draw one per item, issued before the body; primes: the NEXT item's segments).  Waiting for segment u:
// the younger operations are the rest of the previous item's body, this item's A, and this item's first u pairs = 7 + ATOM.
template <int WORK, int PRO, bool VERIFY, int ATOM>
__device__ __forceinline__ void do_item_x(u32x4 (&v)[4], const uint8_t* lds, desc_t cur, desc_t nxt, int lane, uint32_t item)
{
	u32x4 par = {(uint32_t)lane, item, 3u, 4u};
	if (PRO) fake_compute(par, lds, PRO, 5u);
#pragma unroll
	for (int u = 0; u < 4; u++)
	{
		xwait<7 + ATOM>(v[u]);
		u32x4 t;
		asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
		             : "=&v"(t.x), "=&v"(t.y), "=&v"(t.z), "=&v"(t.w) : "v"(v[u].x), "v"(v[u].y), "v"(v[u].z), "v"(v[u].w));
		xload(v[u], (u * 64 + lane) * 16, nxt);
		if (VERIFY) t = t + 1u;
		else if (WORK) fake_compute(t, lds, WORK, par.x & 0xff);
		xstore(t, (u * 64 + lane) * 16, cur);
	}
}
// the first item's loads, interleaved with dropped stores: the steady-state queue shape from the start
__device__ __forceinline__ void first_item_x(u32x4 (&v)[4], desc_t cur, int lane)
{
	const u32x4 z = {0, 0, 0, 0};
#pragma unroll
	for (int u = 0; u < 4; u++) { xload(v[u], (u * 64 + lane) * 16, cur); xstore(z, kOOB, cur); }
}

// ---- ring: rigid (wave w <-> item w of every chunk) ----------------------------------------------------------------
template <int WAVES, int WORK, int PRO, int D, int R, bool VERIFY, bool X = false>
__global__ __launch_bounds__(WAVES * 64) void k_ring(uint8_t* __restrict__ buf, uint32_t nitems, const uint8_t* tables,
                                                       unsigned long long* heads, unsigned long long* heads_next, uint32_t* errflag)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds[kLdsBytes];
	__shared__ unsigned long long s_slot[R];
	__shared__ uint32_t s_cons[R];
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	stage<WAVES * 64>(lds, tables);
	if (threadIdx.x < R) { s_slot[threadIdx.x] = 0; s_cons[threadIdx.x] = WAVES; }
	if (blockIdx.x == 0 && threadIdx.x < 256) heads_next[threadIdx.x * 16] = 0;     // the next launch's ticket heads (all 256 of the block)
	__syncthreads();
	const uint32_t nchunks = (nitems + WAVES - 1) / WAVES;
	// wave 0: the workgroup's ticket clerk
	int h = xcc_id(), tried = 0;
	bool ended = false;
	auto draw = [&](int n_lanes) {   // n_lanes tickets from head h (lanes 0 .. n_lanes-1); other waves: nothing
		const __amdgpu_buffer_rsrc_t crs = rsrc(heads + h * 16, 8);
		return (uint32_t)__builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, crs, (wave == 0 && lane < n_lanes) ? 0u : kOOB, 0, 16);   // sc1: device scope (the XCDs' L2s are not coherent)
	};
	auto publish = [&](uint32_t seq, uint32_t n, int hh) {
		uint32_t chunk;
		if (ended) chunk = kEnd;
		else if (n * 8 + hh < nchunks) chunk = n * 8 + hh;
		else
		{
			if (hh == h) { h = (h + 1) & 7; tried++; if (tried == 8) ended = true; }
			chunk = ended ? kEnd : kSkip;
		}
		for (uint32_t spins = 0; lds_ld32(&s_cons[seq % R]) != WAVES; spins++)
		{
			if (spins > kMaxSpins) { *errflag = 1; ended = true; chunk = kEnd; break; }    // never hang the GPU: give up loudly, once
			__builtin_amdgcn_s_sleep(1);
		}
		lds_st32(&s_cons[seq % R], 0);
		lds_st64(&s_slot[seq % R], ((unsigned long long)(seq + 1) << 32) | chunk);
	};
	if (wave == 0)
	{
		const int h0 = h;
		const uint32_t t = draw(D);
		for (int i = 0; i < D; i++) publish(i, (uint32_t)__builtin_amdgcn_readlane((int)t, i), h0);
	}
	auto take = [&](uint32_t seq) {   // the chunk of sequence number seq (waits until it has been published)
		unsigned long long s;
		for (uint32_t spins = 0;; spins++)
		{
			s = lds_ld64(&s_slot[seq % R]);
			if ((uint32_t)(s >> 32) == seq + 1) break;
			if (spins > kMaxSpins) { *errflag = 2; s = kEnd; break; }
			__builtin_amdgcn_s_sleep(1);
		}
		if (lane == 0) lds_inc(&s_cons[seq % R]);
		return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)s);
	};
	auto item_of = [&](uint32_t chunk) { return chunk >= kSkip ? kOOB : chunk * WAVES + wave; };
	auto desc = [&](uint32_t item) { return rsrc(buf + (size_t)(item < nitems ? item : 0) * 4096, item < nitems ? 4096 : 0); };

	uint32_t c0 = take(0);
	if (c0 == kEnd) return;
	uint32_t item = item_of(c0);
	if constexpr (X)
	{
		auto xdesc = [&](uint32_t it) { return mkdesc(buf + (size_t)(it < nitems ? it : 0) * 4096, it < nitems ? 4096 : 0); };
		desc_t cur = xdesc(item);
		u32x4 v[4];
		first_item_x(v, cur, lane);
		int pend_h = h;
		uint32_t pend = 0;
		for (uint32_t k = 0;; k++)
		{
			if (wave == 0 && k > 0)
			{
				xwait1<8>(pend);       // the atomic of one item ago: 8 younger operations
				publish(k + D - 1, (uint32_t)__builtin_amdgcn_readfirstlane((int)pend), pend_h);
			}
			pend_h = h;
			xatomic_inc(pend, (wave == 0 && lane == 0 && !ended) ? 0u : kOOB, mkdesc(heads + h * 16, 8));
			const uint32_t c1 = take(k + 1);
			const uint32_t nitem = c1 == kEnd ? kOOB : item_of(c1);
			const desc_t nxt = xdesc(nitem);
			do_item_x<WORK, PRO, VERIFY, 1>(v, lds, cur, nxt, lane, item);
			if (c1 == kEnd) break;
			cur = nxt;
			item = nitem;
		}
		asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) :: "memory");   // (keeps the last refills' registers reserved until they have landed)
		return;
	}
	__amdgpu_buffer_rsrc_t cur = desc(item);
	u32x4 v[4];
#pragma unroll
	for (int u = 0; u < 4; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(cur, (u * 64 + lane) * 16, 0, 2);
	int pend_h = h;
	uint32_t pend = 0;
	for (uint32_t k = 0;; k++)
	{
		// wave 0: publish the ticket drawn one item ago, draw the next (fixed instruction stream: the atomic is issued by
		// every wave, with an out-of-range offset in all lanes but lane 0 of wave 0)
		if (wave == 0 && k > 0) publish(k + D - 1, (uint32_t)__builtin_amdgcn_readfirstlane((int)pend), pend_h);
		pend_h = h;
		pend = draw(ended ? 0 : 1);
		const uint32_t c1 = take(k + 1);
		const uint32_t nitem = c1 == kEnd ? kOOB : item_of(c1);
		const __amdgpu_buffer_rsrc_t nxt = desc(nitem);
		do_item<WORK, PRO, VERIFY>(v, lds, cur, nxt, lane, item);
		if (c1 == kEnd) break;
		cur = nxt;
		item = nitem;
	}
}

// ---- wtix: every wave draws single items from NH global heads --------------------------------------------------------
template <int WAVES, int WORK, int PRO, bool VERIFY, bool X = false>
__global__ __launch_bounds__(WAVES * 64) void k_wtix(uint8_t* __restrict__ buf, uint32_t nitems, const uint8_t* tables,
                                                       unsigned long long* heads, unsigned long long* heads_next, int nh)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds[kLdsBytes];
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	stage<WAVES * 64>(lds, tables);
	if (blockIdx.x == 0 && threadIdx.x < 256) heads_next[threadIdx.x * 16] = 0;
	__syncthreads();
	// spread the waves that share a head over the XCDs (workgroups are dealt round-robin to the XCDs)
	const int hd = (int)(((blockIdx.x >> 3) * WAVES + wave + (blockIdx.x & 7) * 5) % (unsigned)nh);
	const __amdgpu_buffer_rsrc_t crs = rsrc(heads + hd * 16, 8);
	auto draw = [&]() { return (uint32_t)__builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, crs, lane == 0 ? 0u : kOOB, 0, 16); };
	auto desc = [&](uint32_t item) { return rsrc(buf + (size_t)(item < nitems ? item : 0) * 4096, item < nitems ? 4096 : 0); };
	uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane((int)draw()) * nh + hd;
	if (item >= nitems) return;
	if constexpr (X)
	{
		auto xdesc = [&](uint32_t it) { return mkdesc(buf + (size_t)(it < nitems ? it : 0) * 4096, it < nitems ? 4096 : 0); };
		const desc_t hrs = mkdesc(heads + hd * 16, 8);
		uint32_t t1;
		xatomic_inc(t1, lane == 0 ? 0u : kOOB, hrs);
		desc_t cur = xdesc(item);
		u32x4 v[4];
		first_item_x(v, cur, lane);
		while (true)
		{
			xwait1<8>(t1);
			const uint32_t nitem = (uint32_t)__builtin_amdgcn_readfirstlane((int)t1) * nh + hd;
			xatomic_inc(t1, lane == 0 ? 0u : kOOB, hrs);
			const desc_t nxt = xdesc(nitem);
			do_item_x<WORK, PRO, VERIFY, 1>(v, lds, cur, nxt, lane, item);
			if (nitem >= nitems) break;
			cur = nxt;
			item = nitem;
		}
		asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) :: "memory");   // (keeps the last refills' registers reserved until they have landed)
		return;
	}
	uint32_t t1 = draw();
	__amdgpu_buffer_rsrc_t cur = desc(item);
	u32x4 v[4];
#pragma unroll
	for (int u = 0; u < 4; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(cur, (u * 64 + lane) * 16, 0, 2);
	while (true)
	{
		const uint32_t nitem = (uint32_t)__builtin_amdgcn_readfirstlane((int)t1) * nh + hd;
		t1 = draw();
		const __amdgpu_buffer_rsrc_t nxt = desc(nitem);
		do_item<WORK, PRO, VERIFY>(v, lds, cur, nxt, lane, item);
		if (nitem >= nitems) break;
		cur = nxt;
		item = nitem;
	}
}

// ---- static grid-stride ------------------------------------------------------------------------------------------------
template <int WAVES, int WORK, int PRO, bool X = false>
__global__ __launch_bounds__(WAVES * 64) void k_static(uint8_t* __restrict__ buf, uint32_t nitems, const uint8_t* tables)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds[kLdsBytes];
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	stage<WAVES * 64>(lds, tables);
	__syncthreads();
	const uint32_t step = gridDim.x * WAVES;
	auto desc = [&](uint32_t item) { return rsrc(buf + (size_t)(item < nitems ? item : 0) * 4096, item < nitems ? 4096 : 0); };
	uint32_t item = blockIdx.x * WAVES + wave;
	if (item >= nitems) return;
	if constexpr (X)
	{
		auto xdesc = [&](uint32_t it) { return mkdesc(buf + (size_t)(it < nitems ? it : 0) * 4096, it < nitems ? 4096 : 0); };
		desc_t cur = xdesc(item);
		u32x4 v[4];
		first_item_x(v, cur, lane);
		for (; item < nitems; item += step)
		{
			const desc_t nxt = xdesc(item + step);
			do_item_x<WORK, PRO, false, 0>(v, lds, cur, nxt, lane, item);
			cur = nxt;
		}
		asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) :: "memory");   // (keeps the last refills' registers reserved until they have landed)
		return;
	}
	__amdgpu_buffer_rsrc_t cur = desc(item);
	u32x4 v[4];
#pragma unroll
	for (int u = 0; u < 4; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(cur, (u * 64 + lane) * 16, 0, 2);
	for (; item < nitems; item += step)
	{
		const __amdgpu_buffer_rsrc_t nxt = desc(item + step);
		do_item<WORK, PRO, false>(v, lds, cur, nxt, lane, item);
		cur = nxt;
	}
}

// ---- not persistent: the shipped structure (36 KB image per 4-wave workgroup, ROWS rows per wave) ------------------------
template <int ROWS, int WORK, int PRO, bool X = false>
__global__ __launch_bounds__(256) void k_np8(uint8_t* __restrict__ buf, size_t nbytes, const uint8_t* tables)
{
	constexpr int LDSB = 36 * 1024;
	__shared__ __attribute__((aligned(16))) uint8_t lds[LDSB];
	const int lane = threadIdx.x & 63;
	const size_t base = ((size_t)blockIdx.x * ROWS * 4 + (threadIdx.x >> 6)) * 4096;
	if ((size_t)(blockIdx.x + 1) * ROWS * 16384 > nbytes) return;
	if constexpr (X)
	{
		// every vector-memory operation of the kernel is issued from asm: a wait the compiler inserted for a load of its
		// own would count only the operations it knows of and drain the ones it does not
		u32x4 ti[9];
		const desc_t trs = mkdesc(tables, LDSB);
#pragma unroll
		for (int i = 0; i < 9; i++) asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ti[i]) : "v"((uint32_t)(threadIdx.x * 16 + i * 4096)), "s"(trs) : "memory");
		u32x4 v[4];
		desc_t cur = mkdesc(buf + base, 4096);
		first_item_x(v, cur, lane);
#pragma unroll
		for (int i = 0; i < 9; i++) { xwait<8>(ti[i]); *(u32x4*)(lds + threadIdx.x * 16 + i * 4096) = ti[i]; }   // the image, not the samples behind it
		__syncthreads();
		u32x4 par = {(uint32_t)lane, (uint32_t)blockIdx.x, 3u, 4u};
		if (PRO) fake_compute(par, lds, PRO, 5u);
		for (int r = 0; r < ROWS; r++)
		{
			const desc_t nxt = mkdesc(buf + base + (size_t)(r + 1) * 16384, r + 1 < ROWS ? 4096 : 0);
#pragma unroll
			for (int u = 0; u < 4; u++)
			{
				xwait<7>(v[u]);
				u32x4 t;
				asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
				             : "=&v"(t.x), "=&v"(t.y), "=&v"(t.z), "=&v"(t.w) : "v"(v[u].x), "v"(v[u].y), "v"(v[u].z), "v"(v[u].w));
				xload(v[u], (u * 64 + lane) * 16, nxt);
				if (WORK) fake_compute(t, lds, WORK, par.x & 0xff);
				xstore(t, (u * 64 + lane) * 16, cur);
			}
			cur = nxt;
		}
		asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) :: "memory");   // (keeps the last refills' registers reserved until they have landed)
		return;
	}
	u32x4 tmp[9];
#pragma unroll
	for (int i = 0; i < 9; i++) tmp[i] = *(const u32x4*)(tables + threadIdx.x * 16 + i * 4096);
	const __amdgpu_buffer_rsrc_t rs = rsrc(buf + base, ROWS * 16384);
	u32x4 v[4];
#pragma unroll
	for (int u = 0; u < 4; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (u * 64 + lane) * 16, 0, 2);
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
			u32x4 t;
			asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
			             : "=&v"(t.x), "=&v"(t.y), "=&v"(t.z), "=&v"(t.w) : "v"(v[u].x), "v"(v[u].y), "v"(v[u].z), "v"(v[u].w));
			v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, r + 1 < ROWS ? (uint32_t)((u * 64 + lane) * 16) : kOOB, (r + 1) * 16384, 2);
			if (WORK) fake_compute(t, lds, WORK, par.x & 0xff);
			__builtin_amdgcn_raw_buffer_store_b128(t, rs, (u * 64 + lane) * 16, r * 16384, 2);
			__builtin_amdgcn_sched_barrier(0);
		}
	}
}

// np8 with the grain kernel's extra vector-memory INSTRUCTIONS: EP extra 8-byte loads from a small (L2-resident) table in the
// prologue (the LFSR words: 8 in the shipped kernel), and per row 1 narrow load + 2 narrow stores (ER = 1: the shipped aligned
// kernel's "pre" and "tail" accesses, one active lane each).  Same bytes to HBM; how much do the instructions cost?
template <int ROWS, int WORK, int PRO, int EP, int ER>
__global__ __launch_bounds__(256) void k_np8e(uint8_t* __restrict__ buf, size_t nbytes, const uint8_t* tables)
{
	constexpr int LDSB = 36 * 1024;
	__shared__ __attribute__((aligned(16))) uint8_t lds[LDSB];
	const int lane = threadIdx.x & 63;
	const size_t base = ((size_t)blockIdx.x * ROWS * 4 + (threadIdx.x >> 6)) * 4096;
	if ((size_t)(blockIdx.x + 1) * ROWS * 16384 > nbytes) return;
	u32x4 tmp[9];
#pragma unroll
	for (int i = 0; i < 9; i++) tmp[i] = *(const u32x4*)(tables + threadIdx.x * 16 + i * 4096);
	typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
	const __amdgpu_buffer_rsrc_t trs = rsrc(tables, LDSB);
	u32x2 ex[EP ? EP : 1];
#pragma unroll
	for (int i = 0; i < EP; i++) ex[i] = __builtin_amdgcn_raw_buffer_load_b64(trs, (uint32_t)(((blockIdx.x * 37 + i * 11 + lane) & 1023) * 8), 0, 0);
	const __amdgpu_buffer_rsrc_t rs = rsrc(buf + base, ROWS * 16384);
	u32x4 v[4];
	uint32_t pre = 0;
#pragma unroll
	for (int u = 0; u < 4; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (u * 64 + lane) * 16, 0, 2);
	if (ER) pre = __builtin_amdgcn_raw_buffer_load_b32(rs, lane == 0 ? 4092u : kOOB, 0, 2);
#pragma unroll
	for (int i = 0; i < 9; i++) *(u32x4*)(lds + threadIdx.x * 16 + i * 4096) = tmp[i];
	__syncthreads();
	u32x4 par = {(uint32_t)lane, (uint32_t)blockIdx.x, 3u, 4u};
#pragma unroll
	for (int i = 0; i < EP; i++) { par.z ^= ex[i].x; par.w += ex[i].y; }
	if (PRO) fake_compute(par, lds, PRO, 5u);
	for (int r = 0; r < ROWS; r++)
	{
#pragma unroll
		for (int u = 0; u < 4; u++)
		{
			u32x4 t;
			asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
			             : "=&v"(t.x), "=&v"(t.y), "=&v"(t.z), "=&v"(t.w) : "v"(v[u].x), "v"(v[u].y), "v"(v[u].z), "v"(v[u].w));
			if (ER && u == 0) t.x ^= pre;
			v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, r + 1 < ROWS ? (uint32_t)((u * 64 + lane) * 16) : kOOB, (r + 1) * 16384, 2);
			if (ER && u == 0) pre = __builtin_amdgcn_raw_buffer_load_b32(rs, (r + 1 < ROWS && lane == 0) ? 4092u : kOOB, (r + 1) * 16384, 2);
			if (WORK) fake_compute(t, lds, WORK, par.x & 0xff);
			__builtin_amdgcn_raw_buffer_store_b128(t, rs, (u * 64 + lane) * 16, r * 16384, 2);
			if (ER && u == 0) __builtin_amdgcn_raw_buffer_store_b32(t.y, rs, lane == 0 ? 4092u : kOOB, r * 16384, 2);
			if (ER && u == 3) __builtin_amdgcn_raw_buffer_store_b32(t.z, rs, lane == 63 ? 4088u : kOOB, r * 16384, 2);
			__builtin_amdgcn_sched_barrier(0);
		}
	}
}


// np10 "row walk": a wave streams ONE row (SEGS x 1 KiB, contiguous) through a ring of 4 register sets; a workgroup = 4 waves =
// 4 consecutive rows (the same 64 KiB footprint and table image as np8).  No tile boundaries inside a row -> none of the shipped
// kernel's narrow per-tile accesses; per-segment parameters come from LDS (PARLDS extra ds_reads per segment), computed by the
// workgroup in the prologue (PRO per wave, as np8).  ROWS rows per wave (stride 4 rows), EP extra prologue loads.
template <int SEGS, int ROWS, int WORK, int PRO, int EP, int PARLDS>
__global__ __launch_bounds__(256) void k_np10(uint8_t* __restrict__ buf, size_t nbytes, const uint8_t* tables)
{
	constexpr int LDSB = 36 * 1024;
	__shared__ __attribute__((aligned(16))) uint8_t lds[LDSB + 4096];
	const int lane = threadIdx.x & 63;
	const int wave = threadIdx.x >> 6;
	constexpr uint32_t rowb = SEGS * 1024;
	const size_t base = ((size_t)blockIdx.x * ROWS * 4 + wave) * rowb;
	if ((size_t)(blockIdx.x + 1) * ROWS * 4 * rowb > nbytes) return;
	u32x4 tmp[9];
#pragma unroll
	for (int i = 0; i < 9; i++) tmp[i] = *(const u32x4*)(tables + threadIdx.x * 16 + i * 4096);
	typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
	const __amdgpu_buffer_rsrc_t trs = rsrc(tables, LDSB);
	u32x2 ex[EP ? EP : 1];
#pragma unroll
	for (int i = 0; i < EP; i++) ex[i] = __builtin_amdgcn_raw_buffer_load_b64(trs, (uint32_t)(((blockIdx.x * 37 + i * 11 + lane) & 1023) * 8), 0, 0);
	const __amdgpu_buffer_rsrc_t rs = rsrc(buf + base, (ROWS - 1) * 4 * rowb + rowb);
	u32x4 v[4];
#pragma unroll
	for (int u = 0; u < 4; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (u * 64 + lane) * 16, 0, 2);
#pragma unroll
	for (int i = 0; i < 9; i++) *(u32x4*)(lds + threadIdx.x * 16 + i * 4096) = tmp[i];
	u32x4 par = {(uint32_t)lane, (uint32_t)blockIdx.x, 3u, 4u};
#pragma unroll
	for (int i = 0; i < EP; i++) { par.z ^= ex[i].x; par.w += ex[i].y; }
	if (PRO) fake_compute(par, lds, PRO, 5u);       // (reads the previous contents of LDS: timing only)
	*(uint32_t*)(lds + LDSB + threadIdx.x * 16) = par.x;
	__syncthreads();
	// segment s of row r lives in register set s % 4; the refill of a set is segment s + 4 (or the next row's first segments)
	for (int r = 0; r < ROWS; r++)
	{
		const uint32_t ro = (uint32_t)r * 4 * rowb;
		for (int s0 = 0; s0 < SEGS; s0 += 4)
		{
#pragma unroll
			for (int u = 0; u < 4; u++)
			{
				const int sg = s0 + u;
				u32x4 t;
				asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
				             : "=&v"(t.x), "=&v"(t.y), "=&v"(t.z), "=&v"(t.w) : "v"(v[u].x), "v"(v[u].y), "v"(v[u].z), "v"(v[u].w));
				// next: segment sg + 4 of this row, or segment (sg + 4 - SEGS) of the next row
				const bool wrap = sg + 4 >= SEGS;
				const uint32_t noff = wrap ? ro + 4 * rowb + (uint32_t)(sg + 4 - SEGS) * 1024 : ro + (uint32_t)(sg + 4) * 1024;
				const bool nvalid = !wrap || r + 1 < ROWS;
				v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, nvalid ? (uint32_t)(lane * 16) : kOOB, noff, 2);
				uint32_t salt = par.x & 0xff;
#pragma unroll
				for (int k = 0; k < PARLDS; k++) salt ^= *(const uint32_t*)(lds + LDSB + ((sg * 64 + lane + k * 17) & 1023) * 4);
				if (WORK) fake_compute(t, lds, WORK, salt & 0xff);
				__builtin_amdgcn_raw_buffer_store_b128(t, rs, (uint32_t)(lane * 16), ro + (uint32_t)sg * 1024, 2);
				__builtin_amdgcn_sched_barrier(0);
			}
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

int main(int argc, char** argv)
{
	bool verify = false;
	int rounds = 5;
	for (int i = 1; i < argc; i++) { if (!strcmp(argv[i], "--verify")) verify = true; else rounds = atoi(argv[i]); }
	const int W = 7680, H = 4320, NF = 8, POOL = 3;
	const size_t ybytes = (size_t)W * H * 2, fbytes = ybytes + ybytes / 2, set = fbytes * NF;
	const uint32_t nitems = (uint32_t)(set / 4096);
	uint8_t* pool[POOL];
	for (int i = 0; i < POOL; i++) { CK(hipMalloc(&pool[i], set)); CK(hipMemset(pool[i], 0x5a + i, set)); }
	unsigned long long* heads;      // two blocks of 256 heads, 128 bytes apart
	const size_t head_block = 256 * 16;
	CK(hipMalloc(&heads, 2 * head_block * 8)); CK(hipMemset(heads, 0, 2 * head_block * 8));
	uint32_t* errflag;
	CK(hipMalloc(&errflag, 4)); CK(hipMemset(errflag, 0, 4));
	setvbuf(stdout, nullptr, _IOLBF, 0);
	uint8_t* tables;
	CK(hipMalloc(&tables, kLdsBytes)); CK(hipMemset(tables, 3, kLdsBytes));
	hipDeviceProp_t prop;
	CK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	int flip = 0;
	auto hcur = [&]() { return heads + (flip & 1) * head_block; };
	auto hnext = [&]() { return heads + ((flip + 1) & 1) * head_block; };

	struct Variant { std::string name; std::function<void(int)> launch; std::vector<float> t; };
	std::vector<Variant> vs;
#define RINGX(NAME, WAVES, WGCU, WORK, PRO, D, R) vs.push_back({NAME, [&](int s) { k_ring<WAVES, WORK, PRO, D, R, false, true><<<WGCU * cus, WAVES * 64>>>(pool[s], nitems, tables, hcur(), hnext(), errflag); flip++; }, {}})
#define WTIXX(NAME, WAVES, WGCU, WORK, PRO, NH) vs.push_back({NAME, [&](int s) { k_wtix<WAVES, WORK, PRO, false, true><<<WGCU * cus, WAVES * 64>>>(pool[s], nitems, tables, hcur(), hnext(), NH); flip++; }, {}})
#define STATX(NAME, WAVES, WGCU, WORK, PRO) vs.push_back({NAME, [&](int s) { k_static<WAVES, WORK, PRO, true><<<WGCU * cus, WAVES * 64>>>(pool[s], nitems, tables); }, {}})
#define NP8X(NAME, ROWS, WORK, PRO) vs.push_back({NAME, [&](int s) { k_np8<ROWS, WORK, PRO, true><<<(unsigned)(set / ((size_t)ROWS * 16384)), 256>>>(pool[s], set, tables); }, {}})
#define NP8E(NAME, ROWS, WORK, PRO, EP, ER) vs.push_back({NAME, [&](int s) { k_np8e<ROWS, WORK, PRO, EP, ER><<<(unsigned)(set / ((size_t)ROWS * 16384)), 256>>>(pool[s], set, tables); }, {}})
#define NP10(NAME, SEGS, ROWS, WORK, PRO, EP, PARLDS) vs.push_back({NAME, [&](int s) { k_np10<SEGS, ROWS, WORK, PRO, EP, PARLDS><<<(unsigned)(set / ((size_t)ROWS * 4 * SEGS * 1024)), 256>>>(pool[s], set, tables); }, {}})
#define RING(NAME, WAVES, WGCU, WORK, PRO, D, R) vs.push_back({NAME, [&](int s) { k_ring<WAVES, WORK, PRO, D, R, false><<<WGCU * cus, WAVES * 64>>>(pool[s], nitems, tables, hcur(), hnext(), errflag); flip++; }, {}})
#define WTIX(NAME, WAVES, WGCU, WORK, PRO, NH) vs.push_back({NAME, [&](int s) { k_wtix<WAVES, WORK, PRO, false><<<WGCU * cus, WAVES * 64>>>(pool[s], nitems, tables, hcur(), hnext(), NH); flip++; }, {}})
#define STAT(NAME, WAVES, WGCU, WORK, PRO) vs.push_back({NAME, [&](int s) { k_static<WAVES, WORK, PRO><<<WGCU * cus, WAVES * 64>>>(pool[s], nitems, tables); }, {}})
#define NP8(NAME, ROWS, WORK, PRO) vs.push_back({NAME, [&](int s) { k_np8<ROWS, WORK, PRO><<<(unsigned)(set / ((size_t)ROWS * 16384)), 256>>>(pool[s], set, tables); }, {}})

	if (verify)
	{
		// every ticket variant once on a buffer of known content: each dword must come out as value + 1
		struct V { const char* name; std::function<void()> run; };
		std::vector<V> tests = {
			{"ring 16 waves", [&]() { k_ring<16, 0, 0, 4, 12, true><<<cus, 1024>>>(pool[0], nitems, tables, hcur(), hnext(), errflag); flip++; }},
			{"ring 8 waves x2", [&]() { k_ring<8, 0, 0, 4, 12, true><<<2 * cus, 512>>>(pool[0], nitems, tables, hcur(), hnext(), errflag); flip++; }},
			{"ring 4 waves x4", [&]() { k_ring<4, 0, 0, 4, 12, true><<<4 * cus, 256>>>(pool[0], nitems, tables, hcur(), hnext(), errflag); flip++; }},
			{"ring 16 waves, 3 workgroups", [&]() { k_ring<16, 0, 0, 4, 12, true><<<3, 1024>>>(pool[0], nitems / 64, tables, hcur(), hnext(), errflag); flip++; }},

			{"ring X 16 waves", [&]() { k_ring<16, 0, 0, 4, 12, true, true><<<cus, 1024>>>(pool[0], nitems, tables, hcur(), hnext(), errflag); flip++; }},
			{"ring X 4 waves x4", [&]() { k_ring<4, 0, 0, 4, 12, true, true><<<4 * cus, 256>>>(pool[0], nitems, tables, hcur(), hnext(), errflag); flip++; }},
			{"wtix X 16 waves 64 heads", [&]() { k_wtix<16, 0, 0, true, true><<<cus, 1024>>>(pool[0], nitems, tables, hcur(), hnext(), 64); flip++; }},
			{"wtix 16 waves 64 heads", [&]() { k_wtix<16, 0, 0, true><<<cus, 1024>>>(pool[0], nitems, tables, hcur(), hnext(), 64); flip++; }},
		};
		std::vector<uint32_t> hostbuf(set / 4);
		int bad_total = 0;
		for (auto& t : tests)
		{
			CK(hipMemset(pool[0], 0x11, set));
			t.run();
			CK(hipDeviceSynchronize());
			CK(hipGetLastError());
			CK(hipMemcpy(hostbuf.data(), pool[0], set, hipMemcpyDeviceToHost));
			const bool small = !strcmp(t.name, "ring 16 waves, 3 workgroups");
			const size_t lim = small ? (size_t)(nitems / 64) * 1024 : set / 4;
			size_t bad = 0, first = 0;
			for (size_t i = 0; i < set / 4; i++)
			{
				const uint32_t want = i < lim ? 0x11111112u : 0x11111111u;
				if (hostbuf[i] != want) { if (!bad) first = i; bad++; }
			}
			const bool wtix = !strncmp(t.name, "wtix", 4);     // wtix does not steal: a few items at the end stay undone by design
			uint32_t ef = 0;
			CK(hipMemcpy(&ef, errflag, 4, hipMemcpyDeviceToHost));
			if (ef) { printf("verify %-32s SPIN LIMIT HIT (flag %u)\n", t.name, ef); CK(hipMemset(errflag, 0, 4)); bad_total++; }
			printf("verify %-32s %s (%zu wrong dwords, first at %zu)%s\n", t.name, bad ? (wtix ? "INCOMPLETE" : "FAIL") : "ok", bad, first, wtix ? " [no stealing: informational]" : "");
			if (bad && !wtix) bad_total++;
		}
		return bad_total ? 1 : 0;
	}

	const char* only = getenv("SKEL_ONLY");   // "ring": the persistent variants of the first experiment; default: instruction-count experiment
	vs.push_back({"np1 one item per wave, nothing staged (ceiling)", [&](int s) { k_np1<<<(unsigned)(set / 16384), 256>>>(pool[s], set); }, {}});
	NP8("np8 staged rows 4, work 15, pro 40 (shipped structure)", 4, 15, 40);
	if (only && !strcmp(only, "ring"))
	{
		NP8X("np8 X rows 4, work 15, pro 40", 4, 15, 40);
		NP8X("np8 X rows 4, no work", 4, 0, 0);
		NP8X("np8 X rows 2, work 15, pro 40", 2, 15, 40);
		RING("ring 16 waves x1/cu, work 15, pro 6", 16, 1, 15, 6, 4, 12);
		RINGX("ring X 16 waves x1/cu, work 15, pro 6", 16, 1, 15, 6, 4, 12);
		RINGX("ring X 16 waves x1/cu, no work", 16, 1, 0, 0, 4, 12);
		RINGX("ring X 8 waves x2/cu, work 15, pro 6", 8, 2, 15, 6, 4, 12);
		RINGX("ring X 4 waves x4/cu, work 15, pro 6", 4, 4, 15, 6, 4, 12);
		WTIX("wtix 16 waves x1/cu, 64 heads, work 15, pro 6", 16, 1, 15, 6, 64);
		WTIXX("wtix X 16 waves x1/cu, 64 heads, work 15, pro 6", 16, 1, 15, 6, 64);
		WTIXX("wtix X 16 waves x1/cu, 64 heads, no work", 16, 1, 0, 0, 64);
		STAT("static 16 waves x1/cu, work 15, pro 6", 16, 1, 15, 6);
		STATX("static X 16 waves x1/cu, work 15, pro 6", 16, 1, 15, 6);
		STATX("static X 16 waves x1/cu, no work", 16, 1, 0, 0);
	}
	else
	{
		NP8E("np8e rows 4, work 15, pro 40, +0 prologue loads, +0 narrow", 4, 15, 40, 0, 0);
		NP8E("np8e rows 4, work 15, pro 40, +8 prologue, +3 narrow (= shipped counts)", 4, 15, 40, 8, 1);
		NP10("np10 row walk 16 segs x 1 row, work 15, pro 40, +8 prologue, 2 param reads", 16, 1, 15, 40, 8, 2);
		NP10("np10 row walk 12 segs x 1 row (12 KiB rows), no work", 12, 1, 0, 0, 0, 0);
		NP10("np10 row walk 20 segs x 1 row (20 KiB rows), no work", 20, 1, 0, 0, 0, 0);
		NP10("np10 row walk 12 segs x 1 row, work 15, pro 40, +8, 2 param reads", 12, 1, 15, 40, 8, 2);
		NP10("np10 row walk 16 segs x 1 row, work 15, pro 40, +0 prologue, 2 param reads", 16, 1, 15, 40, 0, 2);
		NP10("np10 row walk 16 segs x 1 row, work 15, pro 40, +8, no param reads", 16, 1, 15, 40, 8, 0);
		NP10("np10 row walk 16 segs x 1 row, no work", 16, 1, 0, 0, 0, 0);
		NP10("np10 row walk 8 segs x 1 row, work 15, pro 40, +8, 2 param reads", 8, 1, 15, 40, 8, 2);
		NP10("np10 row walk 8 segs x 2 rows, work 15, pro 40, +8, 2 param reads", 8, 2, 15, 40, 8, 2);
		NP10("np10 row walk 16 segs x 2 rows, work 15, pro 40, +8, 2 param reads", 16, 2, 15, 40, 8, 2);
		NP10("np10 row walk 4 segs x 4 rows (= np8 order per wave), work 15, pro 40, +8, 2", 4, 4, 15, 40, 8, 2);
	}

	const int reps = 6;
	for (int r = 0; r <= rounds; r++)
		for (auto& v : vs)
		{
			if (r == 0) printf("# warm-up: %s\n", v.name.c_str());
			CK(hipEventRecord(e0));
			for (int k = 0; k < reps; k++) v.launch(k % POOL);
			CK(hipEventRecord(e1));
			CK(hipEventSynchronize(e1));
			CK(hipGetLastError());
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			if (r) v.t.push_back(ms / reps * 1e3f);
		}
	{ uint32_t ef = 0; CK(hipMemcpy(&ef, errflag, 4, hipMemcpyDeviceToHost)); if (ef) printf("SPIN LIMIT HIT during timing (flag %u): ring results invalid\n", ef); }
	printf("%-56s %10s %10s %8s %8s\n", "variant (8 frames 4320p 10b 4:2:0 / launch)", "med us", "min us", "GB/s", "of 8TB/s");
	for (auto& v : vs)
	{
		std::sort(v.t.begin(), v.t.end());
		const float med = v.t[v.t.size() / 2], mn = v.t[0];
		const double gbs = 2.0 * set / (med * 1e-6) / 1e9;
		printf("%-56s %10.1f %10.1f %8.0f %8.3f\n", v.name.c_str(), med, mn, gbs, gbs / 8000.0);
	}
	return 0;
}
