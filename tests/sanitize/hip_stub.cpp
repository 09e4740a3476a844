// TEST INFRASTRUCTURE -- never linked into the product, never calls the oracle.
//
// A stand-in for the ~35 HIP runtime entry points the host layer uses (vfgs_host.cpp, vfgs_fw_host.cpp) and for the five
// kernel-side functions it calls, so that the host layer -- the line call with its look-ahead, the stripe rings, the host
// pipelines, the replica worker threads, the frame lists -- runs on a machine without a GPU under AddressSanitizer,
// UndefinedBehaviorSanitizer and ThreadSanitizer (GPU sanitizers are not available on the pool; the reference declares the
// switch at /root/reference/CMakeLists.txt:25-29 and never wires it).  Values are not checked here (the GPU parity tests do
// that); ADDRESSES are: "device memory" is malloc'ed, every copy and every launch touches exactly the bytes the real one
// would, so a wrong extent, pitch, ring slot or lifetime is a sanitizer report.
//
// Streams are modelled, not ignored: an asynchronous operation is QUEUED and runs only when something the host waits on
// depends on it (hipStreamSynchronize, hipEventSynchronize, a synchronous copy, hipFree ...), in stream order and honouring
// hipStreamWaitEvent.  So memory that is freed, reused or overwritten while a queued copy or kernel still needs it is seen
// (as the real runtime would see it, late), and so is a result the host reads before the operation that produces it was
// waited for (the bytes are simply not there yet -- the walks notice through their own byte markers).  Pageable host memory
// follows the runtime's rules: an upload from it is staged before the call returns, a download to it blocks the caller.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <vector>

#include "../../versatilefilmgrain_amd/csrc/vfgs_fw_layout.h"
#include "../../versatilefilmgrain_amd/csrc/vfgs_layout.h"

namespace {

struct Stream { int id; int device; };
struct Event { Stream* stream = nullptr; uint64_t seq = 0; bool recorded = false; };
struct Op {
	uint64_t seq;
	Stream* stream;
	Event* wait_for_ev_stream_owner = nullptr;      // unused
	Stream* dep_stream = nullptr;                   // hipStreamWaitEvent: what must have run first
	uint64_t dep_seq = 0;
	std::function<void()> fn;
};

std::recursive_mutex g_mu;
std::deque<Op> g_pending;                 // in submission order
uint64_t g_seq = 0;
Stream g_null_stream{0, 0};
int g_next_stream = 1;
thread_local int t_device = 0;
int g_ndevices = 2;                       // (two "devices": the replica paths need more than one)
std::map<const void*, size_t> g_pinned;   // hipHostMalloc allocations
std::map<const void*, size_t> g_devmem;   // hipMalloc allocations
uint64_t g_stats_deferred = 0, g_stats_ops = 0;

Stream* S(hipStream_t s) { return s ? (Stream*)s : &g_null_stream; }

bool is_pinned(const void* p)
{
	auto it = g_pinned.upper_bound(p);
	if (it == g_pinned.begin()) return false;
	--it;
	return (const char*)p < (const char*)it->first + it->second;
}

// run everything `stream` has queued up to `seq`, and first whatever those operations wait for
void run_stream(Stream* stream, uint64_t seq)
{
	for (;;)
	{
		size_t i = 0;
		for (; i < g_pending.size(); i++)
			if (g_pending[i].stream == stream && g_pending[i].seq <= seq) break;
		if (i == g_pending.size()) return;
		Op op = std::move(g_pending[i]);
		g_pending.erase(g_pending.begin() + (long)i);
		if (op.dep_stream) run_stream(op.dep_stream, op.dep_seq);
		if (op.fn) op.fn();
	}
}

void run_all()
{
	while (!g_pending.empty())
	{
		Op op = std::move(g_pending.front());
		g_pending.pop_front();
		if (op.dep_stream) run_stream(op.dep_stream, op.dep_seq);
		if (op.fn) op.fn();
	}
}

void enqueue(hipStream_t s, std::function<void()> fn, Stream* dep_stream = nullptr, uint64_t dep_seq = 0)
{
	Op op;
	op.seq = ++g_seq;
	op.stream = S(s);
	op.dep_stream = dep_stream;
	op.dep_seq = dep_seq;
	op.fn = std::move(fn);
	g_pending.push_back(std::move(op));
	g_stats_ops++;
}

void copy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height)
{
	for (size_t r = 0; r < height; r++) memmove((char*)dst + r * dpitch, (const char*)src + r * spitch, width);
}

}  // namespace

extern "C" {

// what the walks ask the stub (not part of HIP)
void vfgs_stub_stats(uint64_t out[3])
{
	std::lock_guard<std::recursive_mutex> g(g_mu);
	out[0] = g_stats_ops; out[1] = g_stats_deferred; out[2] = g_pending.size();
}
void vfgs_stub_set_devices(int n) { g_ndevices = n; }
// fault injection: the next `n` pinned allocations fail (hipErrorOutOfMemory), as on a host that has run out of lockable memory
static int g_fail_pinned = 0;
void vfgs_stub_fail_pinned_allocs(int n) { g_fail_pinned = n; }

hipError_t hipGetDeviceCount(int* n) { *n = g_ndevices; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = t_device; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= g_ndevices) return hipErrorInvalidDevice; t_device = d; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d)
{
	if (d < 0 || d >= g_ndevices) return hipErrorInvalidDevice;
	memset(p, 0, sizeof *p);
	snprintf(p->name, sizeof p->name, "stub");
	snprintf(p->gcnArchName, sizeof p->gcnArchName, "gfx950:sramecc+:xnack-");
	p->multiProcessorCount = 256;
	p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
	p->clockRate = 2400000;
	return hipSuccess;
}
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "stub error"; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = -1; return hipSuccess; }

hipError_t hipMalloc(void** p, size_t n)
{
	std::lock_guard<std::recursive_mutex> g(g_mu);
	*p = malloc(n ? n : 1);
	if (!*p) return hipErrorOutOfMemory;
	memset(*p, 0xd5, n);                     // device memory starts out as garbage
	g_devmem[*p] = n;
	return hipSuccess;
}
hipError_t hipFree(void* p)
{
	std::lock_guard<std::recursive_mutex> g(g_mu);
	if (!p) return hipSuccess;
	run_all();                               // (hipFree synchronises the device)
	if (!g_devmem.erase(p)) { fprintf(stderr, "hip_stub: hipFree of %p, which hipMalloc did not return\n", p); abort(); }
	free(p);
	return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned)
{
	std::lock_guard<std::recursive_mutex> g(g_mu);
	if (g_fail_pinned > 0) { g_fail_pinned--; *p = nullptr; return hipErrorOutOfMemory; }
	*p = malloc(n ? n : 1);
	if (!*p) return hipErrorOutOfMemory;
	g_pinned[*p] = n;
	return hipSuccess;
}
hipError_t hipHostFree(void* p)
{
	std::lock_guard<std::recursive_mutex> g(g_mu);
	if (!p) return hipSuccess;
	run_all();
	if (!g_pinned.erase(p)) { fprintf(stderr, "hip_stub: hipHostFree of %p, which hipHostMalloc did not return\n", p); abort(); }
	free(p);
	return hipSuccess;
}
hipError_t hipMemset(void* p, int v, size_t n)
{
	std::lock_guard<std::recursive_mutex> g(g_mu);
	run_all();
	memset(p, v, n);
	return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind)
{
	std::lock_guard<std::recursive_mutex> g(g_mu);
	run_all();
	memmove(dst, src, n);
	return hipSuccess;
}

hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind, hipStream_t s)
{
	std::lock_guard<std::recursive_mutex> g(g_mu);
	if (width > dpitch || width > spitch) return hipErrorInvalidPitchValue;
	const bool h2d = kind == hipMemcpyHostToDevice, d2h = kind == hipMemcpyDeviceToHost;
	if (h2d && !is_pinned(src))
	{
		// pageable source: the runtime stages it before the call returns
		auto stage = std::make_shared<std::vector<char>>(width * height);
		copy2d(stage->data(), width, src, spitch, width, height);
		enqueue(s, [=] { copy2d(dst, dpitch, stage->data(), width, width, height); });
		return hipSuccess;
	}
	if (d2h && !is_pinned(dst))
	{
		// pageable destination: the caller blocks until the bytes are there
		enqueue(s, [=] { copy2d(dst, dpitch, src, spitch, width, height); });
		run_stream(S(s), g_seq);
		return hipSuccess;
	}
	g_stats_deferred++;
	enqueue(s, [=] { copy2d(dst, dpitch, src, spitch, width, height); });
	return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t s)
{
	return hipMemcpy2DAsync(dst, n, src, n, n, 1, kind == hipMemcpyDefault ? hipMemcpyDeviceToDevice : kind, s);
}

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned)
{
	std::lock_guard<std::recursive_mutex> g(g_mu);
	*s = (hipStream_t) new Stream{g_next_stream++, t_device};
	return hipSuccess;
}
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned flags, int) { return hipStreamCreateWithFlags(s, flags); }
hipError_t hipStreamDestroy(hipStream_t s)
{
	std::lock_guard<std::recursive_mutex> g(g_mu);
	if (!s) return hipErrorInvalidHandle;
	run_stream(S(s), g_seq);                 // (the runtime lets queued work finish; nothing may refer to the stream afterwards)
	for (const Op& op : g_pending)
		if (op.dep_stream == S(s)) { fprintf(stderr, "hip_stub: a queued operation waits for an event of a destroyed stream\n"); abort(); }
	delete (Stream*)s;
	return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s)
{
	std::lock_guard<std::recursive_mutex> g(g_mu);
	run_stream(S(s), g_seq);
	return hipSuccess;
}
hipError_t hipDeviceSynchronize(void)
{
	std::lock_guard<std::recursive_mutex> g(g_mu);
	run_all();
	return hipSuccess;
}

hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t) new Event; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e)
{
	std::lock_guard<std::recursive_mutex> g(g_mu);
	if (!e) return hipErrorInvalidHandle;
	delete (Event*)e;                        // (a wait already queued on it has captured what it waits for, as in the runtime)
	return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
	std::lock_guard<std::recursive_mutex> g(g_mu);
	Event* ev = (Event*)e;
	enqueue(s, nullptr);                     // a marker: "everything queued on s so far"
	ev->stream = S(s); ev->seq = g_seq; ev->recorded = true;
	return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned)
{
	std::lock_guard<std::recursive_mutex> g(g_mu);
	Event* ev = (Event*)e;
	if (!ev->recorded) return hipSuccess;    // (an event never recorded is complete)
	enqueue(s, nullptr, ev->stream, ev->seq);
	return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e)
{
	std::lock_guard<std::recursive_mutex> g(g_mu);
	Event* ev = (Event*)e;
	if (ev->recorded) run_stream(ev->stream, ev->seq);
	return hipSuccess;
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b)
{
	std::lock_guard<std::recursive_mutex> g(g_mu);
	Event *ea = (Event*)a, *eb = (Event*)b;
	if (ea->recorded) run_stream(ea->stream, ea->seq);
	if (eb->recorded) run_stream(eb->stream, eb->seq);
	*ms = 0.001f;
	return hipSuccess;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------------
// The kernel side (vfgs_kernel.hip, vfgs_fw_kernel.hip) as byte movers: every launch reads and writes exactly the bytes the
// real kernels address -- the table image, the LFSR window, every row of every plane of every frame -- and copies source rows
// to destination rows unchanged ("grain" of zero), so that whatever the host layer got wrong about an extent is an access the
// sanitizer sees.
namespace vfgs {

ImageLayout layout_of(int csubx, int csuby, bool oney, bool onec, bool depth8) { return image_layout(csubx, csuby, oney, onec, depth8); }

void describe_launch(char* out, size_t n, int depth, int csubx, int csuby, bool out8, bool oney, bool onec, bool wide, bool persist)
{
	snprintf(out, n, "stub<%d,%d,%d,%d,%d,%d,%d,%d>", depth, csubx, csuby, out8, oney, onec, wide, persist);
}

static unsigned char g_sink;
static void touch(const void* p, size_t n)
{
	const unsigned char* b = (const unsigned char*)p;
	unsigned char x = 0;
	for (size_t i = 0; i < n; i += 64) x ^= b[i];
	if (n) x ^= b[n - 1];
	g_sink ^= x;
}

hipError_t launch_grain(const KernelArgs& a, const FrameTable* list, int depth, int csubx, int csuby, bool out8, bool oney, bool onec, bool wide, bool persist,
                        int grid, hipStream_t stream)
{
	if ((out8 && depth != 10) || wide != (a.nblk > kTileBlocks) || grid <= 0) return hipErrorInvalidValue;
	if ((a.listed != 0) != (list != nullptr) || (list && a.nframes > kListFrames)) return hipErrorInvalidValue;
	if (persist && (oney || wide || depth != 10)) return hipErrorInvalidValue;
	const KernelArgs k = a;
	FrameTable ft{};
	if (list) ft = *list;
	std::lock_guard<std::recursive_mutex> g(g_mu);
	g_stats_deferred++;
	enqueue(stream, [=] {
		const ImageLayout L = image_layout(csubx, csuby, oney, onec, depth == 8);
		touch(k.tables, (size_t)L.bytes);
		touch(k.stream, k.stream_bytes);
		// the windows a launch reads: 64 bits behind the last block of the last block row of the last frame
		const uint64_t last_bit = (uint64_t)k.cur_bit0 + (uint64_t)(k.nframes - 1) * k.frame_bit_step + (uint64_t)(k.nbrows - 1) * k.nblk + (uint64_t)k.nblk - 1;
		if ((last_bit >> 5) * 4 + 8 > k.stream_bytes) { fprintf(stderr, "hip_stub: the launch reads LFSR bits behind its window\n"); abort(); }
		if (k.y0 >= 16 || k.nbrows > 1) { /* up_bit0 addresses the row above the stripe's first */ }
		if (((uint64_t)k.up_bit0 >> 5) * 4 + 8 > k.stream_bytes) { fprintf(stderr, "hip_stub: up_bit0 behind the window\n"); abort(); }
		for (int f = 0; f < k.nframes; f++)
			for (int c = 0; c < 3; c++)
			{
				const PlaneDesc& pd = k.pd[c ? 1 : 0];
				const int suby = c ? csuby : 1;
				const int row_first = (k.y0 + suby - 1) / suby, prow0 = k.y0 / suby;
				const uint8_t* sb = k.listed ? ft.src[c][f] : k.src[c] + (uint64_t)f * pd.fpitch;
				uint8_t* db = k.listed ? ft.dst[c][f] : k.dst[c] + (uint64_t)f * pd.dfpitch;
				for (int r = row_first; r < row_first + pd.nrows; r++)
				{
					const uint8_t* srow = sb + (size_t)(r - prow0) * pd.pitch;
					uint8_t* drow = db + (size_t)(r - prow0) * (out8 ? pd.dpitch : pd.pitch);
					if (out8)
						for (uint32_t i = 0; i < pd.drowbytes; i++) drow[i] = (uint8_t)((((const uint16_t*)srow)[i] + 2) >> 2);
					else
						memmove(drow, srow, pd.rowbytes);
				}
			}
	});
	return hipSuccess;
}

hipError_t launch_fw_generate(const FwLaunch& L, hipStream_t stream)
{
	const FwLaunch k = L;
	std::lock_guard<std::recursive_mutex> g(g_mu);
	enqueue(stream, [=] {
		touch(k.k, sizeof(FwConstants));
		for (int i = 0; i < k.njobs; i++)
		{
			const bool chroma = k.job[i].chroma != 0;
			memset(k.bank + (size_t)((chroma ? kSlots : 0) + k.job[i].index) * 4096, (i * 7 + 1) & 0x7f, 4096);
			if (chroma && !(k.csubx == 2 && k.csuby == 2)) memset(k.chroma_raw + (size_t)k.job[i].index * 1024, 1, 1024);
		}
	});
	return hipSuccess;
}

hipError_t launch_fw_patch(uint8_t* img, const int8_t* bank, uint32_t, uint32_t, int csubx, int csuby, bool one_y, bool one_c, int, int, int, bool depth8,
                           hipStream_t stream)
{
	std::lock_guard<std::recursive_mutex> g(g_mu);
	enqueue(stream, [=] {
		const ImageLayout L = image_layout(csubx, csuby, one_y, one_c, depth8);
		touch(bank, 2 * kSlots * 4096);
		touch(img, (size_t)L.bytes);
		img[L.bytes - 1] ^= 0;               // (a write at the far end)
	});
	return hipSuccess;
}

}  // namespace vfgs
