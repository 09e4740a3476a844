// Host side of libvfgs_hip: the process-global hardware-layer state of the reference
// (vfgs_hw.c:49-63) kept on the host, its device images, the seed state machine
// (vfgs_hw.c:288-298,309-310) expressed as positions in one LFSR bit stream, and the
// C ABI of include/vfgs_hip.h.  All sample arithmetic happens in vfgs_kernel.hip; there is
// no CPU implementation of the grain path in this library.
#include <hip/hip_runtime.h>

#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vfgs_hip.h"
#include "../../include/vfgs_hip_fw.h"
#include "vfgs_fw_layout.h"
#include "vfgs_layout.h"

namespace vfgs {
hipError_t launch_grain(const KernelArgs& a, const FrameTable* list, int depth, int csubx, int csuby, bool out8, bool oney, bool onec, bool wide, bool persist, int grid, hipStream_t stream);
ImageLayout layout_of(int csubx, int csuby, bool oney, bool onec, bool depth8);
void describe_launch(char* out, size_t n, int depth, int csubx, int csuby, bool out8, bool oney, bool onec, bool wide, bool persist);
hipError_t launch_fw_generate(const FwLaunch& L, hipStream_t stream);
hipError_t launch_fw_patch(uint8_t* img, const int8_t* bank, uint32_t mask_luma, uint32_t mask_chroma, int csubx, int csuby,
                           bool one_y, bool one_c, int slot_y, int slot_cb, int slot_cr, bool depth8, hipStream_t stream);
}

// The constant tables of the grain models (oracle/dump_fw_tables.c documents origin and layout),
// linked in as data.
#if !defined(__HIP_DEVICE_COMPILE__)
__asm__(".section .rodata\n.balign 16\n.hidden vfgs_fw_blob\n.globl vfgs_fw_blob\nvfgs_fw_blob:\n.incbin \"" VFGS_FW_TABLES_PATH "\"\n"
        ".hidden vfgs_fw_blob_end\n.globl vfgs_fw_blob_end\nvfgs_fw_blob_end:\n.previous\n");
#endif
extern "C" const unsigned char vfgs_fw_blob[], vfgs_fw_blob_end[];

#ifndef VFGS_RW_MIN_FILL_PCT
#define VFGS_RW_MIN_FILL_PCT 100  // a launch should fill this share of the chip's wave slots, else its workgroups get half the rows (25 -> 100: single
#endif                            // frames +2..8 %, 8-frame launches unchanged; 300 loses 10 % at 1080p x 8: profiles/r03_ab40_min_fill.log)
#ifndef VFGS_PERSIST_MIN_TASKS
#define VFGS_PERSIST_MIN_TASKS 2  // general-form luma of small pictures: persistent workgroups (one staging of the 36 KB table image for several
#endif                            // tasks) when every one of them gets at least this many tasks; 0 = never (2 instead of 3: 1080p x 8 +3 %,
                                  // profiles/r04_ab3_persistence_at_4320p_and_two_tasks.log)
#ifndef VFGS_PERSIST_MAX_WG_KB
#define VFGS_PERSIST_MAX_WG_KB 32 // ... and only for luma workgroups of at most this many KB of samples (4320p, 60 KB: no gain at 8, -1 % at 16 frames)
#endif
#ifndef VFGS_RW_WG_BYTES
#define VFGS_RW_WG_BYTES 24576 // a workgroup's rows should hold at least this many bytes (where its block row allows; 16 KiB: the same, 48 KiB: -3..-13 %
#endif                         // at 8 and at 32 frames per launch, profiles/r04_ab1_wg_bytes_vs_batch_and_tiny_launch_floor.log)

namespace {

using vfgs::KernelArgs;

// ------------------------------------------------------------------------------------
// errors

// per thread: vfgs_hip_last_error_string() hands out a pointer into the string, which only this thread's next failing call changes
thread_local int g_err = 0;
thread_local std::string g_errstr;

int fail(int code, const char* fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	g_err = code;
	g_errstr = buf;
	return code;
}

[[noreturn]] void die(const char* what)
{
	// the drop-in calls return void (vfgs_hw.h:51-62): like the reference's asserts, abort
	fprintf(stderr, "libvfgs_hip: fatal: %s (%s)\n", what, g_errstr.c_str());
	abort();
}

#define HIP_TRY(expr)                                                                          \
	do {                                                                                       \
		hipError_t e_ = (expr);                                                                \
		if (e_ != hipSuccess)                                                                  \
			return fail((int)e_, "%s -> %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
	} while (0)

// ------------------------------------------------------------------------------------
// LFSR bit stream (vfgs_hw.c:74-79 as a random-access sequence)
//
// t[0..31] are the bits of the register, t[m] = t[m-31] ^ t[m-3] for m >= 32; the register
// after n steps is the window t[n..n+31].  Raising the characteristic polynomial to the
// 32nd power over GF(2) gives the same recurrence on whole 32-bit words,
// W[n] = W[n-31] ^ W[n-3], valid from word 32 on (every referenced bit index >= 1), so a
// megabyte of stream costs a few hundred microseconds on the host.

uint32_t lfsr_step(uint32_t r)
{
	return (r >> 1) | ((((r >> 1) ^ (r >> 29)) & 1u) << 31);
}

// ---- jump-ahead ----
// The step of vfgs_hw.c:74-79 is linear over GF(2), so "the register n steps later" is a 32 x 32 bit matrix applied to the
// register (column 0 is zero: bit 0 only shifts out), and n steps are log2(n) squarings of the one-step matrix.  A rank of a
// stripe split owns a few block rows of every frame of a batch: with the matrix for one FRAME's worth of steps it visits the
// windows it needs and never generates the rows in between (StripeStream below).
struct LfsrMatrix {
	uint32_t col[32];        // col[j] = image of the register with only bit j set
	uint32_t apply(uint32_t r) const
	{
		uint32_t v = 0;
		for (int j = 0; r; j++, r >>= 1) if (r & 1) v ^= col[j];
		return v;
	}
	static LfsrMatrix one_step()
	{
		LfsrMatrix m;
		for (int j = 0; j < 32; j++) m.col[j] = lfsr_step(1u << j);
		return m;
	}
	static LfsrMatrix identity()
	{
		LfsrMatrix m;
		for (int j = 0; j < 32; j++) m.col[j] = 1u << j;
		return m;
	}
	LfsrMatrix then(const LfsrMatrix& b) const      // first this, then b
	{
		LfsrMatrix c;
		for (int j = 0; j < 32; j++) c.col[j] = b.apply(col[j]);
		return c;
	}
	static LfsrMatrix steps(uint64_t n)
	{
		LfsrMatrix r = identity(), p = one_step();
		for (; n; n >>= 1)
		{
			if (n & 1) r = r.then(p);
			p = p.then(p);
		}
		return r;
	}
};

// a matrix as four 256-entry tables: one application = 4 loads + 3 XORs
struct LfsrJump {
	uint64_t nbits = ~0ull;
	uint32_t t[4][256];
	void build(uint64_t n)
	{
		if (n == nbits) return;
		const LfsrMatrix m = LfsrMatrix::steps(n);
		for (int k = 0; k < 4; k++)
		{
			t[k][0] = 0;
			for (int v = 1; v < 256; v++) t[k][v] = t[k][v & (v - 1)] ^ m.col[8 * k + __builtin_ctz(v)];
		}
		nbits = n;
	}
	uint32_t operator()(uint32_t r) const { return t[0][r & 255] ^ t[1][(r >> 8) & 255] ^ t[2][(r >> 16) & 255] ^ t[3][r >> 24]; }
};

// words [32, n) of the stream whose first 32 words are w[0..31]: W[n] = W[n-31] ^ W[n-3], and the same recurrence squared
// three times (W[n] = W[n-62] ^ W[n-6] from word 63 on, W[n] = W[n-124] ^ W[n-12] from word 125 on, W[n] = W[n-248] ^ W[n-24] from
// word 249 on; StreamCache::fill goes on to the 32nd power for its megabyte windows).  Every phase moves chunks as long as its
// short lag and keeps the chunk before in REGISTERS: the chain of dependent steps then costs one XOR per step instead of a store ->
// load round trip per step, which is what bounded the plain loops (145 -> 94 ns per 531-word segment on the build host,
// tools/dev/lfsr_extend_bench.cpp; 16-byte pieces in the long phases).
static inline uint64_t ld64(const uint32_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline void st64(uint32_t* p, uint64_t v) { memcpy(p, &v, 8); }
void lfsr_extend(uint32_t* w, size_t n)
{
	if (n <= 32) return;       // (a segment shorter than its head: nothing to extend, and w[29..31] may not be the caller's)
	size_t i = 32;
	{
		uint32_t p0 = w[29], p1 = w[30], p2 = w[31];
		for (; i + 3 <= n && i <= 62; i += 3)      // words 32 .. 64 (the base recurrence holds from word 32 on, also beyond 62)
		{
			p0 ^= w[i - 31]; p1 ^= w[i - 30]; p2 ^= w[i - 29];
			w[i] = p0; w[i + 1] = p1; w[i + 2] = p2;
		}
	}
	if (i == 65 && i + 6 <= n)
	{
		uint64_t q0 = ld64(w + i - 6), q1 = ld64(w + i - 4), q2 = ld64(w + i - 2);
		for (; i + 6 <= n && i <= 125; i += 6)     // words 65 .. 130
		{
			q0 ^= ld64(w + i - 62); q1 ^= ld64(w + i - 60); q2 ^= ld64(w + i - 58);
			st64(w + i, q0); st64(w + i + 2, q1); st64(w + i + 4, q2);
		}
	}
#if defined(__SSE2__)
	if (i == 131 && i + 12 <= n)
	{
		__m128i r0 = _mm_loadu_si128((const __m128i*)(w + i - 12)), r1 = _mm_loadu_si128((const __m128i*)(w + i - 8)), r2 = _mm_loadu_si128((const __m128i*)(w + i - 4));
		for (; i + 12 <= n && i <= 249; i += 12)   // words 131 .. 250
		{
			r0 = _mm_xor_si128(r0, _mm_loadu_si128((const __m128i*)(w + i - 124)));
			r1 = _mm_xor_si128(r1, _mm_loadu_si128((const __m128i*)(w + i - 120)));
			r2 = _mm_xor_si128(r2, _mm_loadu_si128((const __m128i*)(w + i - 116)));
			_mm_storeu_si128((__m128i*)(w + i), r0); _mm_storeu_si128((__m128i*)(w + i + 4), r1); _mm_storeu_si128((__m128i*)(w + i + 8), r2);
		}
	}
	if (i == 251 && i + 24 <= n)
	{
		__m128i r[6];
		for (int k = 0; k < 6; k++) r[k] = _mm_loadu_si128((const __m128i*)(w + i - 24 + 4 * k));
		for (; i + 24 <= n; i += 24)
			for (int k = 0; k < 6; k++)
			{
				r[k] = _mm_xor_si128(r[k], _mm_loadu_si128((const __m128i*)(w + i - 248 + 4 * k)));
				_mm_storeu_si128((__m128i*)(w + i + 4 * k), r[k]);
			}
	}
#endif
	// whatever the chunks left (the tail of a segment; everything behind word 130 without SSE2), word by word with the longest valid lag
	for (; i < n; i++) w[i] = i >= 249 ? w[i - 248] ^ w[i - 24] : i >= 125 ? w[i - 124] ^ w[i - 12] : i >= 63 ? w[i - 62] ^ w[i - 6] : w[i - 31] ^ w[i - 3];
}

// the first 32 words of the stream that begins with register `reg` (bit by bit: once per chain of segments, ~1 us)
void lfsr_head(uint32_t reg, uint32_t (&w)[32])
{
	for (int i = 0; i < 32; i++)
	{
		w[i] = reg;
		for (int k = 0; k < 32; k++) reg = lfsr_step(reg);
	}
}

// Lifetime of one image slot (LFSR stream window or table image) of a small ring.  The image is uploaded on the stream of
// the call that needs it first and then read by kernels on whatever streams the caller uses; the slot may be overwritten
// (pinned source and device copy) only after all of that has finished.  Steady-state launches must not pay for this --
// an event record between two kernels costs microseconds -- so nothing is recorded per launch: the guard remembers the
// (few) streams that used the slot, other streams wait once for the upload, and when the slot is LEFT (a refill, a new
// table image, a new seed) one event is recorded behind the last use on each of those streams.
struct SlotGuard {
	static constexpr int kMaxUsers = 4;
	hipEvent_t upload_ev = nullptr, leave_ev[kMaxUsers] = {nullptr, nullptr, nullptr, nullptr};
	hipStream_t users[kMaxUsers] = {nullptr, nullptr, nullptr, nullptr};   // distinct streams that used the slot since its upload
	int nusers = 0, nleft = 0;
	bool live = false;              // uploaded and not yet left

	hipError_t uploaded(hipStream_t stream)
	{
		hipError_t e;
		if (!upload_ev && (e = hipEventCreateWithFlags(&upload_ev, hipEventDisableTiming)) != hipSuccess) return e;
		if ((e = hipEventRecord(upload_ev, stream)) != hipSuccess) return e;
		users[0] = stream; nusers = 1; nleft = 0; live = true;
		return hipSuccess;
	}
	// before a launch on `stream` that reads the slot
	hipError_t use(hipStream_t stream)
	{
		for (int i = 0; i < nusers; i++) if (users[i] == stream) return hipSuccess;
		hipError_t e;
		if ((e = hipStreamWaitEvent(stream, upload_ev, 0)) != hipSuccess) return e;       // (users[0] uploaded it)
		if (nusers == kMaxUsers)
		{   // more streams than we track: let the oldest one drain now, then forget it
			if ((e = hipStreamSynchronize(users[1])) != hipSuccess) return e;
			for (int i = 1; i + 1 < nusers; i++) users[i] = users[i + 1];
			nusers--;
		}
		users[nusers++] = stream;
		return hipSuccess;
	}
	// the slot stops being the current one: remember when its readers are done
	hipError_t leave()
	{
		if (!live) return hipSuccess;
		hipError_t e;
		for (int i = 0; i < nusers; i++)
		{
			if (!leave_ev[i] && (e = hipEventCreateWithFlags(&leave_ev[i], hipEventDisableTiming)) != hipSuccess) return e;
			if ((e = hipEventRecord(leave_ev[i], users[i])) != hipSuccess) return e;
		}
		nleft = nusers; nusers = 0; live = false;
		return hipSuccess;
	}
	// before the slot is overwritten
	hipError_t wait_free()
	{
		hipError_t e;
		if (live && (e = leave()) != hipSuccess) return e;
		for (int i = 0; i < nleft; i++)
			if ((e = hipEventSynchronize(leave_ev[i])) != hipSuccess) return e;   // normally long complete
		nleft = 0;
		return hipSuccess;
	}
	void destroy()
	{
		if (upload_ev) (void)hipEventDestroy(upload_ev);
		for (hipEvent_t& ev : leave_ev) { if (ev) (void)hipEventDestroy(ev); ev = nullptr; }
		upload_ev = nullptr; nusers = nleft = 0; live = false;
	}
};

// Host + device image of a window of the stream, in a small ring of slots (pinned host
// words + device words) so that a refill never overwrites what queued kernels still read.
class StreamCache {
public:
	static constexpr int kSlots = 4;
	static constexpr uint64_t kMaxRefill = 1u << 18; // 1 MiB of stream per refill = 64 frames of 4320p
	static constexpr uint64_t kFirstRefill = 1u << 12;

	void reseed(uint32_t reg)
	{
		epoch_++;
		seed_reg_ = reg;
		nck_ = 0;
		checkpoint(0, reg);
		if (cur_ >= 0) (void)slot_[cur_].guard.leave();    // kernels queued so far still read it
		if (next_ >= 0) (void)slot_[next_].guard.leave();  // a window prepared ahead belongs to the old seed (its upload may still run)
		cur_ = next_ = -1;   // nothing valid; slots keep their allocations
		// A new seed per frame is the normal case for AFGS1 (vfgs_fw.c:672), so the first window after a
		// reseed is only as large as the call needs; a stream that keeps being consumed grows its refills.
		refill_ = kFirstRefill;
	}

	// register after `bit` steps
	uint32_t window(uint64_t bit)
	{
		const uint64_t w = bit >> 5;
		const unsigned sh = bit & 31;
		if (cur_ >= 0 && w >= slot_[cur_].wbase && w + 1 < slot_[cur_].wbase + slot_[cur_].nwords)
		{
			const uint32_t* p = slot_[cur_].host + (w - slot_[cur_].wbase);
			return sh ? (p[0] >> sh) | (p[1] << (32 - sh)) : p[0];
		}
		// outside the window: from a known (word, register) point, 32 words bit by bit, then whole words
		// by the word recurrence with a 32-word ring (3 ns per 32 steps instead of per step)
		uint64_t from = 0;
		uint32_t reg = seed_reg_;
		for (int i = 0; i < nck_; i++)          // the nearest known point at or in front of the word
			if (ck_word_[i] <= w && ck_word_[i] >= from) { from = ck_word_[i]; reg = ck_reg_[i]; }
		if (w - from < 40)
		{
			for (uint64_t n = (bit - (from << 5)); n; n--) reg = lfsr_step(reg);
			return reg;
		}
		uint32_t ring[32];
		for (int i = 0; i < 32; i++)
		{
			ring[i] = reg;
			for (int k = 0; k < 32; k++) reg = lfsr_step(reg);
		}
		for (uint64_t i = from + 32; i <= w + 1; i++)            // word i lives in ring[(i - from) & 31]
			ring[(i - from) & 31] = ring[(i - from - 31) & 31] ^ ring[(i - from - 3) & 31];
		const uint32_t lo = ring[(w - from) & 31], hi = ring[(w + 1 - from) & 31];
		return sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
	}

	// make the device image cover absolute bits [lo, hi + 64); returns 0 or a HIP error
	//
	// A refill in the caller's stream is a bubble between two of its kernels (1 MiB host -> device: ~35 us per 64 frames of
	// 4320p, 1-2 % of a saturated stream: profiles/r03_refill_bubbles.log).  So when a call reaches the second half of the
	// current window, the NEXT window is built and uploaded on an internal copy stream; by the time a call needs it the
	// upload is long complete and the caller's stream only waits for its event (SlotGuard::use).
	hipError_t ensure(uint64_t lo, uint64_t hi, hipStream_t stream)
	{
		// + 80: every wave reads a 64-dword slice starting at the dword of its row's first window
		const uint64_t wlo = lo >> 5, whi = (hi >> 5) + 80;
		hipError_t e;
		if (!(cur_ >= 0 && wlo >= slot_[cur_].wbase && whi <= slot_[cur_].wbase + slot_[cur_].nwords) &&
		    next_ >= 0 && wlo >= slot_[next_].wbase && whi <= slot_[next_].wbase + slot_[next_].nwords)
		{
			// the window prepared ahead becomes the current one
			if (cur_ >= 0 && (e = slot_[cur_].guard.leave()) != hipSuccess) return e;
			cur_ = next_;
			next_ = -1;
			stats_[2]++;
			checkpoint(slot_[cur_].wbase, slot_[cur_].host[0]);
		}
		if (cur_ >= 0 && wlo >= slot_[cur_].wbase && whi <= slot_[cur_].wbase + slot_[cur_].nwords)
		{
			// (also right after a switch: a launch that uses up more than half a window -- 64 frames of 4320p at 8 ranks -- needs
			// the next one built during every call)
			if (next_ < 0 && lookahead_ && whi > slot_[cur_].wbase + slot_[cur_].nwords / 2 && slot_[cur_].nwords >= kMaxRefill)
			{
				if (!copy_stream_ && (e = hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking)) != hipSuccess) return e;
				// from this call's first word on: whatever follows in stream order is inside it
				if ((e = fill((last_ + 1) % kSlots, wlo, std::max<uint64_t>(refill_, 2 * (whi - wlo)), copy_stream_)) != hipSuccess) return e;
				next_ = last_;
				stats_[1]++;
			}
			return hipSuccess;
		}
		if (next_ >= 0)
		{   // prepared for a continuation that did not come (a jump in the stream): give the slot back
			if ((e = slot_[next_].guard.leave()) != hipSuccess) return e;
			next_ = -1;
		}
		const uint64_t n = std::max<uint64_t>(refill_, 2 * (whi - wlo));
		refill_ = std::min<uint64_t>(refill_ * 4, kMaxRefill);
		if (cur_ >= 0 && (e = slot_[cur_].guard.leave()) != hipSuccess) return e;
		if ((e = fill((last_ + 1) % kSlots, wlo, n, stream)) != hipSuccess) { cur_ = -1; return e; }   // (fill reads the old window: never its own slot)
		cur_ = last_;
		stats_[0]++;
		checkpoint(wlo, slot_[cur_].host[0]);
		return hipSuccess;
	}

	// a (position, register) pair someone else has computed (StripeStream's chain of jumps): a nearer point to step from
	// than the one this cache knows, so that the next ordinary window behind a long run of stripe batches does not walk there
	void note(uint64_t bit, uint32_t reg)
	{
		while (bit & 31) { reg = lfsr_step(reg); bit++; }
		checkpoint(bit >> 5, reg);
	}

	// {refills in a caller's stream, windows built ahead on the copy stream, switches to a window built ahead, words of the current window}
	void stats(uint64_t out[4]) const { out[0] = stats_[0]; out[1] = stats_[1]; out[2] = stats_[2]; out[3] = dev_words(); }

	// before every kernel launch on `stream` that reads the current slot
	hipError_t use(hipStream_t stream) { return cur_ < 0 ? hipSuccess : slot_[cur_].guard.use(stream); }

	uint32_t seed_reg() const { return seed_reg_; }
	uint64_t epoch() const { return epoch_; }      // counts the reloads of the register: what was built for another one is stale
	const uint32_t* dev() const { return cur_ < 0 ? nullptr : slot_[cur_].dev; }
	uint64_t base_bit() const { return cur_ < 0 ? 0 : slot_[cur_].wbase << 5; }
	uint64_t dev_words() const { return cur_ < 0 ? 0 : slot_[cur_].nwords; }

	void release()
	{
		for (Slot& s : slot_)
		{
			if (s.host) (void)hipHostFree(s.host);
			if (s.dev) (void)hipFree(s.dev);
			s.guard.destroy();
			s = Slot{};
		}
		if (copy_stream_) (void)hipStreamDestroy(copy_stream_);
		copy_stream_ = nullptr;
		cur_ = next_ = -1;
	}

private:
	struct Slot {
		uint32_t* host = nullptr;
		uint32_t* dev = nullptr;
		uint64_t cap = 0, wbase = 0, nwords = 0;
		SlotGuard guard;
	};

	// build words [wlo, wlo + n) of the stream in slot `idx` and upload them on `stream`; the slot becomes last_
	hipError_t fill(int idx, uint64_t wlo, uint64_t n, hipStream_t stream)
	{
		const uint32_t reg0 = window(wlo << 5);
		Slot& s = slot_[idx];
		hipError_t e;
		// its upload and every kernel that read it, on any stream, must be done before its pinned source and its device
		// words are overwritten (SlotGuard); normally long complete
		if ((e = s.guard.wait_free()) != hipSuccess) return e;
		if (s.cap < n)
		{
			if (s.host) (void)hipHostFree(s.host);
			if (s.dev) (void)hipFree(s.dev);
			s.host = nullptr; s.dev = nullptr; s.cap = 0;
			const uint64_t cap = std::max<uint64_t>(n, kMaxRefill);   // allocate once, whatever the refill size
			if ((e = hipHostMalloc((void**)&s.host, cap * 4, hipHostMallocDefault)) != hipSuccess) return e;
			if ((e = hipMalloc((void**)&s.dev, cap * 4)) != hipSuccess) return e;
			s.cap = cap;
		}
		// 32 words bit by bit, then W[n] = W[n-31] ^ W[n-3] (valid from word 32 of any base)
		uint32_t reg = reg0;
		uint64_t i = 0;
		for (; i < n && i < 32; i++)
		{
			s.host[i] = reg;
			for (int k = 0; k < 32; k++) reg = lfsr_step(reg);
		}
		// ... and from word 1024 on by the same recurrence raised to the 32nd power once more,
		// W[n] = W[n-992] ^ W[n-96]: 96 independent words per step, which the compiler vectorises
		// (2 MiB of stream: 80 us instead of 1.9 ms -- at 8 frames x 8 stripes per launch the plain
		// recurrence alone took longer than the kernel it feeds)
		for (; i < n && i < 1024; i++) s.host[i] = s.host[i - 31] ^ s.host[i - 3];
		for (; i + 96 <= n; i += 96)
		{
			uint32_t* __restrict d = s.host + i;
			const uint32_t* __restrict a = s.host + i - 992;
			const uint32_t* __restrict b = s.host + i - 96;
			for (int k = 0; k < 96; k++) d[k] = a[k] ^ b[k];
		}
		for (; i < n; i++) s.host[i] = s.host[i - 992] ^ s.host[i - 96];
		s.wbase = wlo;
		s.nwords = n;
		if ((e = hipMemcpyAsync(s.dev, s.host, n * 4, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
		last_ = idx;
		return s.guard.uploaded(stream);
	}

	Slot slot_[kSlots];
	int cur_ = -1, last_ = -1;
	int next_ = -1;                     // a window built ahead on copy_stream_, not yet the current one
	uint64_t stats_[3] = {0, 0, 0};
	hipStream_t copy_stream_ = nullptr;
#ifdef VFGS_NO_LOOKAHEAD            // developer A/B (tools/dev/build_variant.sh): every refill in the caller's stream
	bool lookahead_ = false;
#else
	bool lookahead_ = true;
#endif
	uint64_t refill_ = kFirstRefill;
	uint32_t seed_reg_ = 0xdeadbeefu;   // register at bit 0 (vfgs_hw.c:52-55 power-on value)
	uint64_t epoch_ = 0;
	// Known (word, register) points to step from: the last few, because the newest may lie AHEAD of what is asked for next -- a window or a
	// stripe image built ahead for calls that then do not come (another entry point takes over at the current registers); with one point
	// only, such a request would walk from bit 0 (seconds, after a few hundred thousand frames)
	static constexpr int kCheckpoints = 6;
	uint64_t ck_word_[kCheckpoints] = {0, 0, 0, 0, 0, 0};
	uint32_t ck_reg_[kCheckpoints] = {0xdeadbeefu, 0, 0, 0, 0, 0};
	int nck_ = 1, ck_next_ = 1;
	void checkpoint(uint64_t word, uint32_t reg)
	{
		for (int i = 0; i < nck_; i++) if (ck_word_[i] == word) return;
		if (nck_ == 0) ck_next_ = 0;
		ck_word_[ck_next_] = word; ck_reg_[ck_next_] = reg;
		ck_next_ = (ck_next_ + 1) % kCheckpoints;
		nck_ = std::min(nck_ + 1, (int)kCheckpoints);
	}
};

// The stream of a BATCH OF STRIPES (vfgs_hip_add_grain_frames_part_dev and the frame-list form: what one rank of a stripe split
// runs per step).  Frame f of the batch reads the windows of the block rows [G0 - 1, G1) it owns: nseg segments of
// seg_words words that lie a whole frame's worth of bits apart (vfgs_hw.c:291-298,309-310: (nbr - 1) x nblk steps per frame).
// StreamCache would build everything in between -- at eight ranks 64 whole frames = 1 MiB per call, 47-56 us of host time
// (profiles/r05_host_overhead_weak_and_strong_shapes.log).  Here segment f + 1 is the JUMP of segment f: its first 32 words are
// the jump matrix applied to the first 32 words of segment f (the recurrence is shift invariant, so every word jumps alike),
// the rest follows by the word recurrence; the segments are stored back to back and the kernel simply gets seg_words x 32 as
// its frame_bit_step.  The image of the NEXT calls (the batches behind this one, same shape) is built and uploaded on the copy
// stream once the calls have reached the second half of the current one, as StreamCache does for its windows; a call that does
// not continue the chain starts a new one from StreamCache::window().
class StripeStream {
public:
	struct Image { const uint32_t* dev; uint32_t words; };

	// segment f = stream bits [first_bit + f * step_bits, ... + 32 * seg_words), f < nseg, of the register `lfsr` was loaded with last.
	// An image holds the segments of SEVERAL consecutive calls where a call is small (kImageSegments: a rank of a strong split
	// takes 8 frames per call; one upload per 8 calls instead of one per call), so a call may find its segments in the middle of
	// the current image.
	hipError_t ensure(StreamCache& lfsr, uint64_t first_bit, uint64_t step_bits, unsigned nseg, unsigned seg_words, hipStream_t stream, Image* out)
	{
		hipError_t e;
		const Key want{lfsr.epoch(), first_bit, step_bits, nseg, seg_words};
		long at = cur_ >= 0 ? slot_[cur_].key.find(want) : -1;
		if (at < 0 && next_ >= 0 && slot_[next_].key.find(want) >= 0)
		{
			if (cur_ >= 0 && (e = slot_[cur_].guard.leave()) != hipSuccess) return e;
			cur_ = next_;
			next_ = -1;
			stats_[2]++;
			at = slot_[cur_].key.find(want);
		}
		if (at < 0)
		{
			if (next_ >= 0) { if ((e = slot_[next_].guard.leave()) != hipSuccess) return e; next_ = -1; }
			if (cur_ >= 0 && (e = slot_[cur_].guard.leave()) != hipSuccess) return e;
			Key k = want;
			k.nseg = nseg * std::max(1u, kImageSegments / nseg);
			if ((e = build((last_ + 1) % kSlots, lfsr, k, stream)) != hipSuccess) { cur_ = -1; return e; }
			cur_ = last_;
			stats_[0]++;
			at = 0;
		}
		if ((e = slot_[cur_].guard.use(stream)) != hipSuccess) return e;
		out->dev = slot_[cur_].dev + (size_t)at * seg_words;
		out->words = nseg * seg_words;
		used_end_ = (unsigned)at + nseg;
		return hipSuccess;
	}

	// after the launch: once the calls have reached the second half of the current image, the image behind it, on the copy stream
	hipError_t prepare_next(StreamCache& lfsr)
	{
		if (cur_ < 0 || next_ >= 0 || !lookahead_ || slot_[cur_].key.epoch != lfsr.epoch() || used_end_ * 2 <= slot_[cur_].key.nseg) return hipSuccess;
		Key k = slot_[cur_].key;
		k.first_bit += (uint64_t)k.nseg * k.step_bits;
		hipError_t e;
		if (!copy_stream_ && (e = hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking)) != hipSuccess) return e;
		if ((e = build((last_ + 1) % kSlots, lfsr, k, copy_stream_)) != hipSuccess) return e;
		next_ = last_;
		stats_[1]++;
		return hipSuccess;
	}

	// {images built in a caller's stream, built ahead on the copy stream, switches to one built ahead}
	void stats(uint64_t out[3]) const { out[0] = stats_[0]; out[1] = stats_[1]; out[2] = stats_[2]; }

	void release()
	{
		for (Slot& s : slot_)
		{
			if (s.host) (void)hipHostFree(s.host);
			if (s.dev) (void)hipFree(s.dev);
			s.guard.destroy();
			s = Slot{};
		}
		if (copy_stream_) (void)hipStreamDestroy(copy_stream_);
		copy_stream_ = nullptr;
		cur_ = next_ = -1;
		chain_valid_ = false;
	}

	// the segments themselves, into host memory (vfgs_hip_lfsr_segments: what the tests compare with the reference's stepping)
	void generate(StreamCache& lfsr, uint64_t first_bit, uint64_t step_bits, unsigned nseg, unsigned seg_words, uint32_t* out)
	{
		fill(lfsr, Key{lfsr.epoch(), first_bit, step_bits, nseg, seg_words}, out);
	}

private:
	static constexpr int kSlots = 4;
	static constexpr unsigned kImageSegments = 64;      // an image holds whole calls' worth of segments up to about this many
	struct Key {
		uint64_t epoch = 0, first_bit = 0, step_bits = 0;
		unsigned nseg = 0, seg_words = 0;
		// the segment of this image at which the segments `w` asks for begin, or -1
		long find(const Key& w) const
		{
			if (epoch != w.epoch || step_bits != w.step_bits || seg_words != w.seg_words || w.first_bit < first_bit || !step_bits) return -1;
			const uint64_t d = w.first_bit - first_bit;
			if (d % step_bits || d / step_bits + w.nseg > nseg) return -1;
			return (long)(d / step_bits);
		}
	};
	struct Slot {
		uint32_t* host = nullptr;
		uint32_t* dev = nullptr;
		uint64_t cap = 0;
		Key key;
		SlotGuard guard;
	};

	void fill(StreamCache& lfsr, const Key& k, uint32_t* out)
	{
		jump_.build(k.step_bits);
		uint32_t head[32];
		if (chain_valid_ && chain_epoch_ == k.epoch && chain_step_ == k.step_bits && chain_bit_ == k.first_bit)
			memcpy(head, chain_head_, sizeof head);        // continues the last image: its last segment, jumped once more
		else
			lfsr_head(lfsr.window(k.first_bit), head);
		for (unsigned f = 0; f < k.nseg; f++)
		{
			uint32_t* w = out + (size_t)f * k.seg_words;
			memcpy(w, head, sizeof(uint32_t) * std::min<unsigned>(32, k.seg_words));
			lfsr_extend(w, k.seg_words);
			for (int i = 0; i < 32; i++) head[i] = jump_(head[i]);
		}
		memcpy(chain_head_, head, sizeof head);
		chain_valid_ = true; chain_epoch_ = k.epoch; chain_step_ = k.step_bits;
		chain_bit_ = k.first_bit + (uint64_t)k.nseg * k.step_bits;
		lfsr.note(chain_bit_, head[0]);
	}

	hipError_t build(int idx, StreamCache& lfsr, const Key& k, hipStream_t stream)
	{
		Slot& s = slot_[idx];
		hipError_t e;
		if ((e = s.guard.wait_free()) != hipSuccess) return e;
		const uint64_t n = (uint64_t)k.nseg * k.seg_words;
		if (s.cap < n)
		{
			if (s.host) (void)hipHostFree(s.host);
			if (s.dev) (void)hipFree(s.dev);
			s.host = nullptr; s.dev = nullptr; s.cap = 0;
			const uint64_t cap = std::max<uint64_t>(n, 1u << 15);
			if ((e = hipHostMalloc((void**)&s.host, cap * 4, hipHostMallocDefault)) != hipSuccess) return e;
			if ((e = hipMalloc((void**)&s.dev, cap * 4)) != hipSuccess) return e;
			s.cap = cap;
		}
		fill(lfsr, k, s.host);
		s.key = k;
		if ((e = hipMemcpyAsync(s.dev, s.host, n * 4, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
		last_ = idx;
		return s.guard.uploaded(stream);
	}

	Slot slot_[kSlots];
	int cur_ = -1, next_ = -1, last_ = -1;
	unsigned used_end_ = 0;             // segments of the current image the calls have used up
	hipStream_t copy_stream_ = nullptr;
	LfsrJump jump_;
	bool chain_valid_ = false;
	uint64_t chain_epoch_ = 0, chain_step_ = 0, chain_bit_ = 0;
	uint32_t chain_head_[32];
	uint64_t stats_[3] = {0, 0, 0};
#ifdef VFGS_NO_LOOKAHEAD
	bool lookahead_ = false;
#else
	bool lookahead_ = true;
#endif
};

// ------------------------------------------------------------------------------------
// device buffer ring: an image that is being replaced may still be read by kernels that
// were queued earlier on another stream, so replacements go to the next slot.

struct DevRing {
	static constexpr int N = 4;
	void* buf[N] = {nullptr, nullptr, nullptr, nullptr};      // device images
	uint8_t* host[N] = {nullptr, nullptr, nullptr, nullptr};  // their pinned host sources (async H2D reads them later)
	SlotGuard guard[N];
	size_t cap[N] = {0, 0, 0, 0};
	int cur = -1;

	// the next slot, safe to overwrite (host and device side)
	hipError_t next(size_t bytes, void** dev, uint8_t** src)
	{
		hipError_t e;
		if (cur >= 0 && (e = guard[cur].leave()) != hipSuccess) return e;
		const int nxt = (cur + 1) % N;
		if ((e = guard[nxt].wait_free()) != hipSuccess) return e;
		if (cap[nxt] < bytes)
		{
			if (buf[nxt]) (void)hipFree(buf[nxt]);
			if (host[nxt]) (void)hipHostFree(host[nxt]);
			buf[nxt] = nullptr; host[nxt] = nullptr; cap[nxt] = 0;
			if ((e = hipMalloc(&buf[nxt], bytes)) != hipSuccess) return e;
			if ((e = hipHostMalloc((void**)&host[nxt], bytes, hipHostMallocDefault)) != hipSuccess) return e;
			cap[nxt] = bytes;
		}
		cur = nxt;
		*dev = buf[cur];
		*src = host[cur];
		return hipSuccess;
	}
	hipError_t uploaded(hipStream_t stream) { return cur < 0 ? hipSuccess : guard[cur].uploaded(stream); }
	hipError_t use(hipStream_t stream) { return cur < 0 ? hipSuccess : guard[cur].use(stream); }
	void* current() const { return cur < 0 ? nullptr : buf[cur]; }
	void release()
	{
		for (int i = 0; i < N; i++)
		{
			if (buf[i]) (void)hipFree(buf[i]);
			if (host[i]) (void)hipHostFree(host[i]);
			guard[i].destroy();
			buf[i] = nullptr; host[i] = nullptr; cap[i] = 0;
		}
		cur = -1;
	}
};

// ------------------------------------------------------------------------------------
// the singleton

struct State {
	// mirror of the reference's statics (vfgs_hw.c:49-63)
	int8_t bank[2][vfgs::kSlots + 1][64][64];
	uint8_t slut[3][256];
	uint8_t plut[3][256];
	int scale_shift = 5 + 6;
	int bs = 0;
	int ymin = 0, ymax = 255, cmin = 0, cmax = 255;
	int csubx = 2, csuby = 2;

	// seed registers as positions in the stream: {rnd, rnd_up, line_rnd, line_rnd_up}
	StreamCache lfsr;
	StripeStream stripes;               // the stream of batches of stripes (jump-ahead; run_device)
	bool stripe_stream_last = false;    // the last launch read its LFSR windows from `stripes`
	uint64_t rnd = 0, rnd_up = 0, line_rnd = 0, line_rnd_up = 0;

	// device
	bool inited = false;
	int device = -1;
	int cu_count = 0;
	bool tables_dirty = true;
	bool img_one_y = false, img_one_c = false;   // form of the current table image (vfgs_layout.h: one-pattern form)
	bool img_wide = false;                       // ... and the width class it was built for (image_form's second argument)
	DevRing tables_ring;
	// staging for the host-pointer entry points
	void* stage[3] = {nullptr, nullptr, nullptr};
	size_t stage_cap[3] = {0, 0, 0};
	uint8_t* bounce[3] = {nullptr, nullptr, nullptr};   // pinned: small host stripes (single lines) travel through these, never
	size_t bounce_cap[3] = {0, 0, 0};                   // through the caller's own pages (run_host)
	hipStream_t own_stream = nullptr;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	// frames in host memory, pipelined (vfgs_hip_add_grain_frames_host): a ring of device frames, one stream per stage
	struct HostPipe {
		static constexpr int kDepth = 3;
		hipStream_t up = nullptr, run = nullptr, down = nullptr;
		void* dev[kDepth][3] = {};
		size_t cap[kDepth][3] = {};
		hipEvent_t up_done[kDepth] = {}, run_done[kDepth] = {}, down_done[kDepth] = {};
		bool busy[kDepth] = {};
		void release()
		{
			for (int k = 0; k < kDepth; k++)
			{
				for (int i = 0; i < 3; i++) { if (dev[k][i]) (void)hipFree(dev[k][i]); dev[k][i] = nullptr; cap[k][i] = 0; }
				if (up_done[k]) (void)hipEventDestroy(up_done[k]);
				if (run_done[k]) (void)hipEventDestroy(run_done[k]);
				if (down_done[k]) (void)hipEventDestroy(down_done[k]);
				up_done[k] = run_done[k] = down_done[k] = nullptr;
				busy[k] = false;
			}
			if (up) (void)hipStreamDestroy(up);
			if (run) (void)hipStreamDestroy(run);
			if (down) (void)hipStreamDestroy(down);
			up = run = down = nullptr;
		}
	} pipe;

	// firmware layer on the device (include/vfgs_hip_fw.h): slots whose pattern was generated on
	// the device live in dev_bank; the host mirror above holds the slots set through the setters
	uint32_t dev_origin[2] = {0, 0};          // bit k: slot k of bank c is device-generated
	int8_t* dev_bank = nullptr;               // [2][kSlots][64][64]
	int8_t* dev_raw = nullptr;                // [kSlots][32*32]
	vfgs::FwConstants* fw_const = nullptr;    // device copy of the model constants + noise streams
	std::vector<vfgs::FwLaunch> fw_pending;   // generation requests not yet launched (they run on the next grain call's stream)
	vfgs::FwLaunch fw_last{};                 // the most recent request: re-sending it unchanged (a new seed per frame with the
	bool fw_last_valid = false;               // same model, the usual AFGS1 stream) generates nothing
	hipStream_t bank_stream = nullptr;        // stream of the last kernels that touched dev_bank
	hipEvent_t bank_ev = nullptr;             // ... and their completion, for the (rare) change of stream
	bool bank_used = false;

	// what every processing call needs to know about the pattern LUTs (768 entries), worked out when they change, not per call
	bool plut_seen = false;           // the three fields below are valid
	int plut_bad_c = -1, plut_bad_i = 0;      // an entry that selects a slot > 8 (undefined in the reference), or -1
	int plut_slot[3] = {0, 0, 0};     // per component: the slot every intensity selects, or -1

	// ---- overlap region (vfgs_hip_overlap_begin / _end): device-pointer calls on `user` run alternately on two internal streams
	struct Overlap {
		bool active = false;
		hipStream_t user = nullptr, s[2] = {nullptr, nullptr};
		hipEvent_t fork = nullptr, join[2] = {nullptr, nullptr};
		unsigned n = 0;
	} ov;

	// ---- several devices in one process (vfgs_hip_init_devices): states 1.. are replicas of state 0 --------------
	uint64_t prog_gen = 0;            // bumped whenever the programmed state (banks, LUTs, parameters, patterns) may have changed
	uint64_t seed_epoch = 0;          // bumped whenever the LFSR is reloaded
	uint64_t synced_prog = ~0ull, synced_seed = ~0ull;   // replica: what of the primary it mirrors

	// ---- look-ahead of the line API (see line_call()) -----------------------------------
	uint64_t gen = 0;                 // bumped by every call that changes state other than a line call
	struct LineAhead {
		bool enabled = true;
		// call pattern learned from consecutive line calls
		bool have_prev = false;
		const uint8_t *pY = nullptr, *pU = nullptr, *pV = nullptr;
		unsigned py = 0, pwidth = 0;
		ptrdiff_t ypitch = 0, cpitch = 0;     // host row pitches in bytes, 0 = not yet known
		unsigned frame_h = 0;                 // lines of the buffer the caller has proven to own (see line_call), 0 = unknown
		const uint8_t *bY = nullptr, *bU = nullptr, *bV = nullptr;   // plane pointers of line 0 of the walk in progress
		bool declared = false;                // frame_h and the pitches come from vfgs_hip_declare_frame
		bool from_zero = false;               // the calls since the last line 0 were consecutive lines of one walk
		// the stripes computed ahead of the caller's walk: a ring of slots, each with pinned snapshots of the caller's lines
		// (`in`, also the upload source), pinned results (`out`) and a device stripe; upload, kernel and download of a stripe
		// run on three streams, so the stripes further down the frame travel while the caller consumes this one
		static constexpr int kRing = 3;
		struct Slot {
			bool used = false, waited = false;
			unsigned y0 = 0, n = 0, crow0 = 0;
			uint8_t* in[3] = {nullptr, nullptr, nullptr};      // (parts of the ring's one pinned block / one device block, below)
			uint8_t* out[3] = {nullptr, nullptr, nullptr};
			void* dev[3] = {nullptr, nullptr, nullptr};
			size_t cap[3] = {0, 0, 0};                         // bytes each of in / out / dev [i] may hold
			hipEvent_t up_done = nullptr, run_done = nullptr, done = nullptr;
		} slot[kRing];
		// ONE pinned and ONE device allocation for the whole ring: the 27 allocations this used to take (3 slots x 3 planes x
		// in / out / device) cost the first stripe of a process 32-40 ms (profiles/r06_promise_probe_*.log)
		uint8_t* pin = nullptr;
		void* devblk = nullptr;
		size_t pin_cap = 0, dev_cap = 0;
		hipStream_t up = nullptr, run = nullptr, down = nullptr;
		bool own_streams = false;             // up / run / down are streams of the look-ahead's own (else: the state's own_stream)
		bool valid = false;
		uint64_t gen = 0;
		int head = 0;                         // slot of the stripe that holds line `next`
		unsigned next = 0, width = 0;         // the line the next call is expected to bring
		unsigned base_y = 0;                  // Y0/U0/V0 address this line
		unsigned issued_end = 0, frame_end = 0, stripe_lines = 0;   // first line not yet computed ahead; end of the proven rows
		const uint8_t *Y0 = nullptr, *U0 = nullptr, *V0 = nullptr;
		unsigned rowlen[3] = {0, 0, 0}, dpitch[3] = {0, 0, 0};
		uint64_t spec[4] = {0, 0, 0, 0};      // the seed registers behind the last stripe computed ahead {rnd, rnd_up, line_rnd, line_rnd_up}
		uint64_t stripes_issued = 0, stripes_dropped = 0;
		void release()
		{
			for (Slot& sl : slot)
			{
				for (int i = 0; i < 3; i++) { sl.in[i] = sl.out[i] = nullptr; sl.dev[i] = nullptr; sl.cap[i] = 0; }
				if (sl.up_done) (void)hipEventDestroy(sl.up_done);
				if (sl.run_done) (void)hipEventDestroy(sl.run_done);
				if (sl.done) (void)hipEventDestroy(sl.done);
				sl.up_done = sl.run_done = sl.done = nullptr;
				sl.used = sl.waited = false;
			}
			if (own_streams)
			{
				if (up) (void)hipStreamDestroy(up);
				if (run) (void)hipStreamDestroy(run);
				if (down) (void)hipStreamDestroy(down);
			}
			up = run = down = nullptr;
			own_streams = false;
			if (pin) (void)hipHostFree(pin);
			if (devblk) (void)hipFree(devblk);
			pin = nullptr; devblk = nullptr; pin_cap = dev_cap = 0;
			valid = false;
		}
	} la;

	State()
	{
		memset(bank, 0, sizeof bank);
		memset(slut, 0, sizeof slut);
		memset(plut, 0, sizeof plut);
		lfsr.reseed(0xdeadbeefu);   // vfgs_hw.c:52-55
	}
};

// State 0 is the reference's process-global hardware layer.  After vfgs_hip_init_devices() states 1..n-1 are replicas bound to
// the other devices: the host-memory entry points give every device a stripe of each frame and run them concurrently, one
// worker thread per extra device, each with ITS state current (g_cur).  Everything else only ever sees state 0.
constexpr int kMaxDevices = 8;
State g_states[kMaxDevices];
thread_local State* g_cur = nullptr;
int g_ndev = 1;

State& S()
{
	return g_cur ? *g_cur : g_states[0];
}

std::mutex g_mu;

vfgs_hip_launch_info g_last_launch{};      // what the primary state's most recent grain launch dispatched
bool g_last_launch_valid = false;
// set while a host-memory entry point (the line call and its look-ahead stripes, vfgs_add_grain_stripe, vfgs_hip_add_grain_frames_host)
// launches on the library's own staging buffers: vfgs_hip_last_launch_info().internal
thread_local bool g_internal_launch = false;     // (per thread: the replicas' worker threads stage too; only the primary's run_device records launch info)
struct InternalLaunch {
	InternalLaunch() { g_internal_launch = true; }
	~InternalLaunch() { g_internal_launch = false; }
};

int ensure_init(int device)
{
	State& s = S();
	if (s.inited && (device < 0 || device == s.device))
	{
		// every allocation, event and launch below belongs to the library's device, whatever device the calling
		// thread has made current in the meantime
		int cur = -1;
		if (hipGetDevice(&cur) != hipSuccess || cur != s.device) HIP_TRY(hipSetDevice(s.device));
		return 0;
	}
	if (s.inited)
		return fail(1, "vfgs_hip_init: already initialised on device %d", s.device);
	int n = 0;
	HIP_TRY(hipGetDeviceCount(&n));
	if (n <= 0)
		return fail(2, "no HIP device visible");
	if (device < 0)
		HIP_TRY(hipGetDevice(&device));
	HIP_TRY(hipSetDevice(device));
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		return fail(3, "device %d is %s; this library contains gfx950 code only", device, prop.gcnArchName);
	s.cu_count = prop.multiProcessorCount;
	s.device = device;
	HIP_TRY(hipStreamCreateWithFlags(&s.own_stream, hipStreamNonBlocking));
	HIP_TRY(hipEventCreate(&s.ev0));
	HIP_TRY(hipEventCreate(&s.ev1));
	s.inited = true;
	return 0;
}

// ------------------------------------------------------------------------------------
// pattern generation on the device

int fw_prepare(State& s)
{
	if (s.fw_const) return 0;
	if ((size_t)(vfgs_fw_blob_end - vfgs_fw_blob) != 7168)
		return fail(30, "embedded model tables have the wrong size");
	std::vector<uint8_t> img(sizeof(vfgs::FwConstants));
	memcpy(img.data(), vfgs_fw_blob, 7168);
	vfgs::FwConstants* k = (vfgs::FwConstants*)img.data();
	for (int i = 0; i < vfgs::kFwSeeds; i++)
	{
		// the firmware's generator (vfgs_fw.c:284-295) is the hardware layer's LFSR: stream word w
		// is the register after 32*w steps, from word 32 on W[n] = W[n-31] ^ W[n-3]
		uint32_t reg = k->seed[i];
		for (int w = 0; w < vfgs::kFwStreamWords; w++)
		{
			if (w < 32) { k->stream[i][w] = reg; for (int b = 0; b < 32; b++) reg = lfsr_step(reg); }
			else k->stream[i][w] = k->stream[i][w - 31] ^ k->stream[i][w - 3];
		}
	}
	HIP_TRY(hipMalloc((void**)&s.fw_const, sizeof(vfgs::FwConstants)));
	HIP_TRY(hipMemcpy(s.fw_const, img.data(), img.size(), hipMemcpyHostToDevice));
	HIP_TRY(hipMalloc((void**)&s.dev_bank, 2 * vfgs::kSlots * 4096));
	HIP_TRY(hipMemset(s.dev_bank, 0, 2 * vfgs::kSlots * 4096));
	HIP_TRY(hipMalloc((void**)&s.dev_raw, vfgs::kSlots * 1024));
	HIP_TRY(hipEventCreateWithFlags(&s.bank_ev, hipEventDisableTiming));
	return 0;
}

// dev_bank is about to be used by kernels on `stream`.  All of its users normally sit on one
// stream and are ordered by it; only when the caller moves to another stream does the new one
// wait for the old one's last use.
int fw_bank_stream(State& s, hipStream_t stream)
{
	if (s.bank_used && s.bank_stream != stream)
	{
		HIP_TRY(hipEventRecord(s.bank_ev, s.bank_stream));
		HIP_TRY(hipStreamWaitEvent(stream, s.bank_ev, 0));
	}
	s.bank_stream = stream;
	s.bank_used = true;
	return 0;
}

// Generation requests are only recorded when they are made (the firmware interface has no
// stream, and needs no device until grain is actually added); they are launched here, on the
// stream of the grain call that first needs the patterns, so plain stream order makes them
// visible and a configuration switch costs no cross-stream synchronisation.
int fw_flush(State& s, hipStream_t stream)
{
	if (s.fw_pending.empty()) return 0;
	if (int e = fw_prepare(s)) return e;
	if (int e = fw_bank_stream(s, stream)) return e;
	for (vfgs::FwLaunch& L : s.fw_pending)
	{
		L.k = s.fw_const; L.bank = s.dev_bank; L.chroma_raw = s.dev_raw;
		HIP_TRY(vfgs::launch_fw_generate(L, stream));
	}
	s.fw_pending.clear();
	return 0;
}

int fw_generate(const vfgs_hip_pattern_job* jobs, int n)
{
	State& s = S();
	if (n <= 0) return 0;
	if (n > vfgs::kFwMaxJobs) return fail(31, "vfgs_hip_generate_patterns: %d jobs (at most %d)", n, vfgs::kFwMaxJobs);
	vfgs::FwLaunch L{};
	L.last_luma = -1;
	bool seen_chroma = false;
	for (int i = 0; i < n; i++)
	{
		const vfgs_hip_pattern_job& j = jobs[i];
		if (j.index < 0 || j.index >= vfgs::kSlots) return fail(20, "pattern job %d: slot %d", i, j.index);
		if (j.seed_index < 0 || j.seed_index >= vfgs::kFwSeeds) return fail(32, "pattern job %d: seed index %d", i, j.seed_index);
		if (j.kind == 0) { if (j.fh > 32767 || j.fv > 32767 || j.fh < -32768 || j.fv < -32768) return fail(33, "pattern job %d: cut-off out of int16 range", i); }
		else if (j.kind == 1) { if (j.scale < 1 || j.scale > 15 || j.shift < 1 || j.shift > 7) return fail(34, "pattern job %d: scale %d / shift %d", i, j.scale, j.shift); }
		else return fail(35, "pattern job %d: kind %d", i, j.kind);
		if (j.chroma) seen_chroma = true;
		else { if (seen_chroma) return fail(36, "pattern jobs: luma jobs must come first"); L.last_luma = j.index; }
		L.job[i] = j;
	}
	L.njobs = n; L.csubx = s.csubx; L.csuby = s.csuby;   // the layout at the time of the call, as vfgs_hw.c:320-325
	if (s.fw_last_valid && !memcmp(&L, &s.fw_last, sizeof L))
	{
		// identical to the previous request; still current unless a setter replaced one of its slots since
		bool intact = true;
		for (int i = 0; i < n; i++) intact = intact && (s.dev_origin[jobs[i].chroma ? 1 : 0] >> jobs[i].index & 1);
		if (intact) return 0;
	}
	s.fw_last = L;
	s.fw_last_valid = true;
	s.fw_pending.push_back(L);
	if (s.fw_pending.size() >= 32)
	{
		// nobody added grain for many configurations: requests whose every slot is rewritten by a
		// later request will never be seen -- drop them (at most one request per slot survives)
		uint32_t covered[2] = {0, 0};
		std::vector<vfgs::FwLaunch> keep;
		for (size_t i = s.fw_pending.size(); i-- > 0;)
		{
			const vfgs::FwLaunch& q = s.fw_pending[i];
			bool visible = false;
			for (int k = 0; k < q.njobs; k++)
				visible = visible || !(covered[q.job[k].chroma ? 1 : 0] >> q.job[k].index & 1);
			if (!visible) continue;
			for (int k = 0; k < q.njobs; k++) covered[q.job[k].chroma ? 1 : 0] |= 1u << q.job[k].index;
			keep.push_back(q);
		}
		s.fw_pending.assign(keep.rbegin(), keep.rend());
	}
	for (int i = 0; i < n; i++) s.dev_origin[jobs[i].chroma ? 1 : 0] |= 1u << jobs[i].index;
	s.tables_dirty = true;
	s.prog_gen++;
	return 0;
}

// The slot a component's pattern LUT selects for EVERY intensity (0..8), or -1 if it depends on the intensity.
int uniform_slot(const uint8_t (&plut)[256])
{
	const int k = plut[0] >> 4;
	for (int i = 1; i < 256; i++)
		if ((plut[i] >> 4) != k) return -1;
	return k;
}

// Build the device image of vfgs_layout.h (luma sub-image, chroma sub-image(s); general or one-pattern form) from the mirror.
void build_tables(const State& s, uint8_t* img, const vfgs::ImageLayout& L, bool one_y, bool one_c, const int (&slot)[3])
{
	memset(img, 0, L.bytes);
	const bool pk16 = s.bs == 0 && vfgs::kPk16;     // 8 bit: one-pattern components in the packed 16-bit form (vfgs_layout.h)
	// banks
	uint8_t* yb = img + L.y_off + L.y_bank;
	for (int r = 0; r < 64; r++)
		for (int x = 0; x < 64; x++)
		{
			if (one_y)
			{
				const int v = slot[0] < vfgs::kSlots ? s.bank[0][slot[0]][r][x] : 0;
				if (pk16)
				{
					const int16_t p = (int16_t)v, n = (int16_t)-v;
					memcpy(yb + r * L.y_rs + 2 * x, &p, 2);
					memcpy(yb + L.y_neg + r * L.y_rs + 2 * x, &n, 2);
				}
				else
				{
					yb[r * L.y_rs + x] = (uint8_t)v;
					yb[L.y_neg + r * L.y_rs + x] = (uint8_t)-v;      // the negated copy (never -128: image_form)
				}
			}
			else
				for (int k = 0; k < vfgs::kSlots; k++) yb[r * L.y_rs + x * vfgs::kSlots + k] = (uint8_t)s.bank[0][k][r][x];
		}
	for (int c = 0; c < 2; c++)       // (one sub-image per chroma component in both forms, vfgs_layout.h)
	{
		uint8_t* cb = img + L.c_off[c] + L.c_bank;
		for (int r = 0; r < L.ch; r++)
			for (int x = 0; x < L.cw; x++)
			{
				if (one_c)
				{
					const int v = slot[1 + c] < vfgs::kSlots ? s.bank[1][slot[1 + c]][r][x] : 0;
					if (pk16)
					{
						const int16_t p = (int16_t)v, n = (int16_t)-v;
						memcpy(cb + r * L.c_rs + 2 * x, &p, 2);
						memcpy(cb + L.c_neg + r * L.c_rs + 2 * x, &n, 2);
					}
					else
					{
						cb[r * L.c_rs + x] = (uint8_t)v;
						cb[L.c_neg + r * L.c_rs + x] = (uint8_t)-v;
					}
				}
				else
					for (int k = 0; k < vfgs::kSlots; k++) cb[r * L.c_rs + x * vfgs::kSlots + k] = (uint8_t)s.bank[1][k][r][x];
			}
	}
	// LUTs
	for (int c = 0; c < 3; c++)
	{
		uint32_t* lut = (uint32_t*)(c == 0 ? img + L.y_off : img + L.c_off[c - 1] + L.c_lut[c - 1]);
		if (pk16 && (c == 0 ? one_y : one_c))
		{
			memcpy(lut, s.slut[c], 256);     // the scale bytes themselves (vfgs_hw.c:50); the shift travels in KernelArgs::pk_shift
			continue;
		}
		for (int i = 0; i < 256; i++)
		{
			const int sl = s.plut[c][i] >> 4;   // vfgs_hw.c:212
			const uint32_t sel = sl < vfgs::kSlots ? (uint32_t)sl : 0x0cu;  // slot 8: the reference's all-zero bank
			// scale pre-shifted so that (scale' * P + 2^15) >> 16 == round(scale * P, scale_shift) (vfgs_hw.c:263):
			// the kernel reads the result's high half instead of shifting; <= 255 << 10 fits the 24-bit field
			const int sc = s.slut[c][i] << (16 - s.scale_shift);
			lut[i] = (sel << 24) | ((uint32_t)sc & 0xffffffu);             // +scale table
			if (!(c == 0 ? one_y : one_c)) lut[256 + i] = (sel << 24) | ((uint32_t)(-sc) & 0xffffffu);    // -scale table (general form only)
		}
	}
}

void digest_pluts(State& s)
{
	if (s.plut_seen) return;
	s.plut_bad_c = -1;
	for (int c = 0; c < 3; c++)
	{
		s.plut_slot[c] = uniform_slot(s.plut[c]);
		for (int i = 0; i < 256 && s.plut_bad_c < 0; i++)
			if ((s.plut[c][i] >> 4) > vfgs::kSlots) { s.plut_bad_c = c; s.plut_bad_i = i; }
	}
	s.plut_seen = true;
}

int check_luts(State& s)
{
	digest_pluts(s);
	if (s.plut_bad_c >= 0)
		return fail(4, "pattern LUT %d[%d] selects slot %d > 8 (undefined in the reference, vfgs_hw.c:49,212)", s.plut_bad_c, s.plut_bad_i,
		            s.plut[s.plut_bad_c][s.plut_bad_i] >> 4);
	return 0;
}

// Which form of the table image (vfgs_layout.h) the current state gets.  wide: the picture is wider than 8192 samples (rows walked
// in parts); those kernels exist for the combinations that matter there -- everything general, one-pattern chroma under general
// or one-pattern luma, at 4:2:0 and 4:4:4 (vfgs_kernel.hip launch_form) -- and every other case gets the general form even where
// one pattern would do.  The pattern LUTs must have been digested.
void image_form(const State& s, bool wide, bool* one_y, bool* one_c)
{
	bool want_general = false;
#ifdef VFGS_NO_ONE_PATTERN      // tools/gpu_variants.sh: always the general form
	want_general = true;
#endif
	const int slot[3] = {s.plut_slot[0], s.plut_slot[1], s.plut_slot[2]};
	// (the one-pattern form stores a negated copy of the pattern, vfgs_layout.h: not for a pattern that holds -128.  Slots the
	// firmware generated on the device are clipped to +-127; slot 8 is the all-zero pattern)
	auto negatable = [&](int pt, int k) {
		if (k >= vfgs::kSlots || (s.dev_origin[pt] >> k & 1)) return true;
		const int rows = pt ? 64 / s.csuby : 64, cols = pt ? 64 / s.csubx : 64;
		for (int r = 0; r < rows; r++)
			if (memchr(s.bank[pt][k][r], 0x80, cols)) return false;
		return true;
	};
	// 8 bit: the one-pattern form multiplies pattern and scale in 16 bits, two samples per instruction (vfgs_layout.h "packed 16-bit
	// form"; vfgs_hw.c:263): |P| <= 127 (no -128, above), so the form is exact while max(scale) * 127 + 2^(shift-1) fits an int16
	// -- scale <= 249 at the usual shift of 11; a LUT beyond that keeps the general form
	auto fits16 = [&](int c) {
		if (s.bs != 0 || !vfgs::kPk16) return true;
		int mx = 0;
		for (int i = 0; i < 256; i++) mx = std::max(mx, (int)s.slut[c][i]);
		return mx * 127 + (1 << (s.scale_shift - 1)) <= 32767;
	};
	*one_y = !want_general && slot[0] >= 0 && negatable(0, slot[0]) && fits16(0);
	*one_c = !want_general && slot[1] >= 0 && slot[2] >= 0 && negatable(1, slot[1]) && negatable(1, slot[2]) && fits16(1) && fits16(2);
	if (wide && !(s.csubx == s.csuby && *one_c)) *one_y = *one_c = false;
}

// the form is a function of the programmed state and of the width class: an image that is not dirty keeps its own
bool image_is_current(const State& s, bool wide)
{
	return !s.tables_dirty && s.tables_ring.current() && s.img_wide == wide;
}

int upload_tables(State& s, hipStream_t stream, bool wide)
{
	if (int e = check_luts(s)) return e;
	if (image_is_current(s, wide)) return 0;     // (the steady state: nothing is looked at per call)
	const int slot[3] = {s.plut_slot[0], s.plut_slot[1], s.plut_slot[2]};     // (check_luts has just digested them)
	bool one_y, one_c;
	image_form(s, wide, &one_y, &one_c);
	if (!s.tables_dirty && s.tables_ring.current() && one_y == s.img_one_y && one_c == s.img_one_c)
	{
		s.img_wide = wide;     // the same image serves this request too
		return 0;
	}
	if (int e = fw_flush(s, stream)) return e;
	void* dst = nullptr;
	const vfgs::ImageLayout L = vfgs::layout_of(s.csubx, s.csuby, one_y, one_c, s.bs == 0);
	uint8_t* img = nullptr;
	HIP_TRY(s.tables_ring.next(L.bytes, &dst, &img));
	build_tables(s, img, L, one_y, one_c, slot);
	HIP_TRY(hipMemcpyAsync(dst, img, L.bytes, hipMemcpyHostToDevice, stream));
	if (s.dev_origin[0] | s.dev_origin[1])
	{
		// device-generated slots never visit the host: copy them bank -> image on the device
		if (int e = fw_bank_stream(s, stream)) return e;
		HIP_TRY(vfgs::launch_fw_patch((uint8_t*)dst, s.dev_bank, s.dev_origin[0], s.dev_origin[1], s.csubx, s.csuby, one_y, one_c,
		                              slot[0], slot[1], slot[2], s.bs == 0, stream));
	}
	HIP_TRY(s.tables_ring.uploaded(stream));
	s.tables_dirty = false;
	s.img_one_y = one_y; s.img_one_c = one_c;
	s.img_wide = wide;
	return 0;
}

int upload_stream(State& s, uint64_t lo, uint64_t hi, hipStream_t stream)
{
	HIP_TRY(s.lfsr.ensure(lo, hi, stream));
	return 0;
}

struct StripePlan {
	uint64_t cur0, up0;   // positions for the first block row of the processed part
};

// The seed state machine of vfgs_add_grain_line (vfgs_hw.c:291-298, 309-310) run over lines
// [y, y+n) on stream *positions*; records the registers of line `mark`.
StripePlan advance_seeds(State& s, unsigned y, unsigned n, unsigned nblk, unsigned mark)
{
	// All lines of one block row see the same registers at entry and leave the same registers
	// behind, so the per-line machine of the reference collapses to one step per block row.
	StripePlan p{0, 0};
	const unsigned end = y + n;
	for (unsigned a = y; a < end;)
	{
		const unsigned b = std::min(end, (a | 15u) + 1);      // lines [a, b) lie in one block row
		if (a && (a & 15) == 0)                               // vfgs_hw.c:291-296
		{
			s.line_rnd_up = s.line_rnd;
			s.line_rnd = s.rnd;
		}
		if (mark >= a && mark < b)
		{
			p.cur0 = s.line_rnd;                              // vfgs_hw.c:297-298
			p.up0 = s.line_rnd_up;
		}
		s.rnd = s.line_rnd + nblk;                            // after any line of this block row (vfgs_hw.c:309-310)
		s.rnd_up = s.line_rnd_up + nblk;
		a = b;
	}
	return p;
}

int check_geometry(const State& s, const void* dY, const void* dU, const void* dV, unsigned width, unsigned stride, unsigned cstride)
{
	const unsigned sz = s.bs ? 2 : 1;
	const unsigned nblk = (width + 15) / 16;
	if (width <= 128)   // vfgs_hw.c:168
		return fail(5, "width %u: the hardware layer requires width > 128 (vfgs_hw.c:168)", width);
	if (nblk > 1984)    // a row's LFSR windows must fit the 64-dword slice a wave keeps in LDS
		return fail(17, "width %u exceeds the supported 31744 samples", width);
	if (stride < nblk * 16 || cstride < nblk * 16 / s.csubx)
		return fail(6, "stride %u/%u too small: whole 16-sample blocks are written (need >= %u/%u)", stride, cstride, nblk * 16, nblk * 16 / s.csubx);
	if (((uintptr_t)dY | (uintptr_t)dU | (uintptr_t)dV) & 15)
		return fail(7, "plane pointers must be 16-byte aligned");
	if ((stride * sz) % 16 || (cstride * sz) % 16)
		return fail(8, "row pitch must be a multiple of 16 bytes");
	if (s.scale_shift + s.bs < 8 || s.scale_shift + s.bs > 13)   // vfgs_hw.c:170
		return fail(9, "scale_shift out of range (vfgs_hw.c:170)");
	return 0;
}

// Core: launch the kernel over `nframes` frames, lines [part_y, part_y+part_h) of each.
struct DstGeom {          // destination geometry when it differs from the source's (8-bit output of a 10-bit path)
	bool out8 = false;
	unsigned stride = 0, cstride = 0;
	uint64_t ypitch = 0, cpitch = 0;
};

int run_device(const void* sY, const void* sU, const void* sV, void* dY, void* dU, void* dV, unsigned width,
               unsigned frame_y, unsigned frame_h, unsigned part_y, unsigned part_h, unsigned stride, unsigned cstride,
               unsigned nframes, uint64_t ypitch, uint64_t cpitch, hipStream_t stream, DstGeom dg = DstGeom(),
               const vfgs::FrameTable* list = nullptr)     // list: the planes of the frames (sY.. / dY.. = those of frame 0, the pitches unused)
{
	State& s = S();
	if (list && nframes > (unsigned)vfgs::kListFrames) return fail(19, "internal: a listed launch holds at most %d frames", vfgs::kListFrames);
	if (int e = ensure_init(-1)) return e;
	if (int e = check_geometry(s, sY, sU, sV, width, stride, cstride)) return e;
	if (!dg.out8)
	{
		if (int e = check_geometry(s, dY, dU, dV, width, stride, cstride)) return e;
	}
	else
	{
		const unsigned nb = (width + 15) / 16;
		if (s.bs != 2) return fail(16, "8-bit output needs a 10-bit path (vfgs_set_depth(10))");
		if (dg.stride < nb * 16 || dg.cstride < nb * 16 / s.csubx) return fail(6, "destination stride too small");
		if ((((uintptr_t)dY | (uintptr_t)dU | (uintptr_t)dV) & 15) || dg.stride % 16 || dg.cstride % 16 || (dg.ypitch | dg.cpitch) % 16)
			return fail(8, "destination planes, pitches and frame pitches must be multiples of 16 bytes");
	}
	if (part_h == 0 || nframes == 0) return 0;
	if (int e = check_luts(s)) return e;     // before any state moves: a refused call leaves the seed registers alone

	const unsigned nblk = (width + 15) / 16;
	const unsigned sz = s.bs ? 2 : 1;
	const uint64_t crows = (uint64_t)(part_y + part_h - 1) / s.csuby - part_y / s.csuby + 1;
	const uint64_t yext = (uint64_t)part_h * stride * sz, cext = crows * cstride * sz;
	if (yext >= 0x80000000ull || cext >= 0x80000000ull)
		return fail(15, "a plane stripe of %llu bytes exceeds the 2 GiB buffer window", (unsigned long long)yext);
	KernelArgs a{};
	a.src[0] = (const uint8_t*)sY; a.src[1] = (const uint8_t*)sU; a.src[2] = (const uint8_t*)sV;
	a.dst[0] = (uint8_t*)dY; a.dst[1] = (uint8_t*)dU; a.dst[2] = (uint8_t*)dV;
	a.y0 = (int)part_y;
	a.nblk = (int)nblk;
	const int nbr_stripe = (int)(((part_y + part_h - 1) >> 4) - (part_y >> 4) + 1);
	a.nbrows = nbr_stripe;
	a.nframes = (int)nframes;
	a.listed = list ? 1 : 0;
	a.lo2[0] = (uint32_t)(s.ymin << s.bs) * 0x10001u; a.hi2[0] = (uint32_t)(s.ymax << s.bs) * 0x10001u;
	a.lo2[1] = (uint32_t)(s.cmin << s.bs) * 0x10001u; a.hi2[1] = (uint32_t)(s.cmax << s.bs) * 0x10001u;
	a.pk_shift = s.scale_shift;     // (read by the 8-bit one-pattern forms only: vfgs_layout.h "packed 16-bit form"; 8..13, check_state)
	// Geometry: a wave streams whole rows (positions = the row's units + the one behind them: the lanes compute bytes shifted
	// by part of a unit); a workgroup = kWavesPerWG x rw_rpw rows of one block row, VFGS_RW_WG_BYTES where the block row allows.
	// Launches that leave wave slots empty get workgroups of half the rows (single frames up to 2160p).
	const unsigned nparts = (nblk + vfgs::kTileBlocks - 1) / vfgs::kTileBlocks;     // passes over the parameter table a row needs
	int rw_shrink = 0;                      // halvings of the rows per wave (small launches)
	bool form_one_y = false, form_one_c = false;   // the form the table image will have (upload_tables below)
	digest_pluts(s);
	const bool wide = nparts > 1;           // rows walked in parts: not every form of the table image has such a kernel (image_form)
	if (image_is_current(s, wide)) form_one_y = s.img_one_y;
	else if (s.plut_bad_c < 0) image_form(s, wide, &form_one_y, &form_one_c);
	(void)form_one_c;
	for (int pass = 0; pass < 3; pass++)
	{
		long waves = 0;
		for (int pt = 0; pt < 2; pt++)
		{
			vfgs::PlaneDesc& d = a.pd[pt];
			const unsigned subx = pt ? s.csubx : 1, suby = pt ? s.csuby : 1;
			const unsigned bw = 16 / subx, rpb = 16 / suby;
			d.pitch = (pt ? cstride : stride) * sz;
			d.dpitch = dg.out8 ? (pt ? dg.cstride : dg.stride) : d.pitch;
			d.fpitch = pt ? cpitch : ypitch;
			d.dfpitch = dg.out8 ? (pt ? dg.cpitch : dg.ypitch) : d.fpitch;
			d.rowbytes = nblk * bw * sz;
			d.drowbytes = dg.out8 ? nblk * bw : d.rowbytes;
			d.nrows = pt ? (int)((part_y + part_h + suby - 1) / suby) - (int)((part_y + suby - 1) / suby) : (int)part_h;
			auto lg = [](int v) { int l = 0; while ((1 << l) < v) l++; return l; };
			const int units = (int)((d.rowbytes + 15) / 16);       // (8-bit 4:2:x rows of an odd number of blocks end in half a unit)
			d.rw_segs = (units + 1 + vfgs::kMaxUnits - 1) / vfgs::kMaxUnits;
			int rpw = 1;
			// (8-bit luma in the general form -- per-sample pattern selection, 24 LDS instructions per position -- does better with
			// waves of one row: +4 % at 2160p, profiles/r03_ab39_workgroup_bytes_8bit.log; every other form loses 4-7 % with them)
			const size_t wg_bytes = (pt == 0 && s.bs == 0 && !form_one_y) ? VFGS_RW_WG_BYTES / 2 : VFGS_RW_WG_BYTES;
			while (vfgs::kWavesPerWG * rpw * 2 <= (int)rpb && (size_t)vfgs::kWavesPerWG * rpw * d.rowbytes < wg_bytes) rpw *= 2;
			for (int i = 0; i < rw_shrink && rpw > 1; i++) rpw /= 2;
			if (nparts > 1) rpw = 1;        // rows walked in parts (more than 512 blocks): the kernel's part loop assumes one row per wave
			d.rw_rpw = rpw;
			d.rw_splits = std::max<int>(1, (int)rpb / (vfgs::kWavesPerWG * rpw));
			d.rw_lsplits = lg(d.rw_splits);
			d.wgs = d.nrows > 0 ? nbr_stripe * d.rw_splits : 0;
			waves += (long)(pt ? 2 : 1) * d.wgs * vfgs::kWavesPerWG * nframes;
		}
		const long slots = (long)s.cu_count * 16;          // wave slots of the chip at the kernels' occupancy
		if (pass < 2 && waves * 100 < VFGS_RW_MIN_FILL_PCT * slots && (a.pd[0].rw_rpw > 1 || a.pd[1].rw_rpw > 1)) { rw_shrink++; continue; }
		break;
	}

	// seeds: frames 0 and 1 run the state machine (vfgs_hw.c:291-298, 309-310; one step per block row); from the end of frame 0 on
	// all four registers move by the same amount per frame -- a frame of nbr block rows rotates nbr - 1 times (not at line 0) and
	// every rotation moves line_rnd, and line_rnd_up behind it, by one row of blocks: G = f (nbr - 1) + r, SURVEY 8a -- so the
	// frames behind them are one addition each register, whatever the batch size
	uint64_t first_cur = 0, first_up = 0, second_cur = 0, lo = ~0ull, hi = 0;
	for (unsigned f = 0; f < nframes && f < 2; f++)
	{
		StripePlan p = advance_seeds(s, frame_y, frame_h, nblk, part_y);
		if (f == 0) { first_cur = p.cur0; first_up = p.up0; }
		if (f == 1) second_cur = p.cur0;
		lo = std::min(lo, std::min(p.cur0, p.up0));
		hi = std::max(hi, p.cur0 + (uint64_t)nbr_stripe * nblk);
		hi = std::max(hi, p.up0 + nblk);
	}
	if (nframes > 2)
	{
		if (frame_y != 0) return fail(10, "internal: a batch of stripes that do not begin at line 0");
		const uint64_t step = second_cur - first_cur, more = (uint64_t)(nframes - 2) * step;
		s.rnd += more; s.rnd_up += more; s.line_rnd += more; s.line_rnd_up += more;
		hi = std::max(hi, second_cur + more + (uint64_t)nbr_stripe * nblk);
	}
	// Images are uploaded on the stream of the call that needs them first; a call on ANOTHER stream waits (once) for that
	// upload, and a slot is only overwritten after all its readers (SlotGuard)
	if (int e = upload_tables(s, stream, wide)) return e;
	HIP_TRY(s.tables_ring.use(stream));
	a.tables = (const uint8_t*)s.tables_ring.current();
	// A batch of stripes that are a small part of their frames (a rank of a stripe split) reads short runs of the
	// stream that lie a whole frame apart: those runs alone, each the jump of the one before (StripeStream).  One segment =
	// [a word in front of the row above the stripe's first, 64 bits behind its last block].
	const uint64_t frame_step = nframes > 1 ? second_cur - first_cur : 0;
	const uint64_t seg_bits = 32 + (uint64_t)nblk + (uint64_t)nbr_stripe * nblk + 64;
	const unsigned seg_words = (unsigned)((seg_bits + 31) / 32) + 1;
	static const bool jump_on = [] { const char* e = getenv("VFGS_HIP_STRIPE_JUMP"); return !(e && e[0] == '0'); }();
	const bool stripes = jump_on && nframes >= 2 && first_cur >= (uint64_t)nblk + 32 && first_up + nblk + 32 >= first_cur &&
	                     (uint64_t)seg_words * 32 * 4 <= frame_step * 3;      // (the segments are at most three quarters of what lies between them)
	s.stripe_stream_last = stripes;
	if (stripes)
	{
		const uint64_t seg0 = first_cur - nblk - 32;
		StripeStream::Image img{};
		HIP_TRY(s.stripes.ensure(s.lfsr, seg0, frame_step, nframes, seg_words, stream, &img));
		a.stream = img.dev;
		a.stream_bytes = img.words * 4;
		a.cur_bit0 = (uint32_t)(first_cur - seg0);
		a.up_bit0 = (uint32_t)(std::max(first_up, seg0) - seg0);     // (only read where the stripe begins below the frame's first block row: the row above)
		a.frame_bit_step = seg_words * 32;
	}
	else
	{
		if (int e = upload_stream(s, lo, hi, stream)) return e;
		HIP_TRY(s.lfsr.use(stream));
		a.stream = s.lfsr.dev();
		a.stream_bytes = (uint32_t)(s.lfsr.dev_words() * 4);
		a.cur_bit0 = (uint32_t)(first_cur - s.lfsr.base_bit());
		a.up_bit0 = (uint32_t)(first_up - s.lfsr.base_bit());
		a.frame_bit_step = (uint32_t)frame_step;
	}

	// one workgroup per (frame, plane, block row, part of it), numbered in memory order
	const long per_frame = (long)a.pd[0].wgs + 2L * a.pd[1].wgs;
	if (per_frame > 0x3fffffffL || nframes > 65535) return fail(14, "launch too large");
	if (per_frame == 0) return 0;
	// General-form luma of small pictures: a workgroup stages 36 KB of tables for 15-30 KB of samples.  Where a launch holds several
	// rounds of luma workgroups, P persistent ones share the luma tasks instead (task t -> workgroup t % P, so they sweep the frames
	// in memory order together); chroma keeps one workgroup per task behind them in the grid.
	bool persist = false;
	long grid = per_frame;
	// (10 bit only: at 8 bit the general-form kernels are bound by their LDS instructions, and confining luma to P < all workgroup
	// slots costs them 7 %; at 10 bit 1080p gains 6 % at 32 and 64 frames per launch, 2160p nothing: profiles/r04_ab2_persistent_luma.log)
	if (VFGS_PERSIST_MIN_TASKS > 0 && s.bs == 2 && !wide && !s.img_one_y && a.pd[0].wgs > 0 &&
	    (size_t)vfgs::kWavesPerWG * a.pd[0].rw_rpw * a.pd[0].rowbytes <= ((size_t)VFGS_PERSIST_MAX_WG_KB << 10))
	{
		const long tasks = (long)a.pd[0].wgs * nframes, slots = (long)s.cu_count * 4;     // (general form: four workgroups per CU)
		const long k = (tasks + slots - 1) / slots;
		if (k >= VFGS_PERSIST_MIN_TASKS)
		{
			const long P = (tasks + k - 1) / k;
			persist = true;
			a.persist_wgs = (int)P;
			a.persist_step_f = (int)(P / a.pd[0].wgs);
			a.persist_step_r = (int)(P % a.pd[0].wgs);
			grid = P + 2L * a.pd[1].wgs * nframes;
			if (grid > 0x7fffffffL) return fail(14, "launch too large");
		}
	}
	// batches of large frames: two frames are swept at the same time (vfgs_kernel.hip grain_rw_kernel)
#ifdef VFGS_NO_FRONTS
	a.lfronts = 0;
#else
	// (not inside an overlap region: there the second sweep is the launch on the other stream, and four fronts lose 15 %)
	const bool in_region = g_states[0].ov.active && (stream == g_states[0].ov.s[0] || stream == g_states[0].ov.s[1]);
	a.lfronts = (!persist && nframes >= 2 && !in_region && 2 * (yext + 2 * cext) >= (64u << 20)) ? 1 : 0;
#endif
	HIP_TRY(vfgs::launch_grain(a, list, 8 + s.bs, s.csubx, s.csuby, dg.out8, s.img_one_y, s.img_one_c, wide, persist, (int)grid, stream));
	// the stream of the batches behind this one, while the GPU works on this one.  (Failing to work ahead is not a failure of THIS call, which is
	// queued and whose registers have moved: the next call then builds its image in its own stream.)
	if (stripes && s.stripes.prepare_next(s.lfsr) != hipSuccess) (void)hipGetLastError();
	if (&s == &g_states[0])
	{
		vfgs_hip_launch_info& li = g_last_launch;
		const unsigned long long n = li.launches + 1;
		li = vfgs_hip_launch_info{};
		li.launches = n;
		li.depth = 8 + s.bs; li.csubx = s.csubx; li.csuby = s.csuby;
		li.out8 = dg.out8; li.one_y = s.img_one_y; li.one_c = s.img_one_c;
		li.in_place = (sY == dY && sU == dU && sV == dV);
		li.listed = a.listed;
		li.internal = g_internal_launch ? 1 : 0;
		li.nframes = (int)nframes;
		li.workgroups_per_frame = (int)per_frame;
		li.frames_per_front = 1 << a.lfronts;
		for (int pt = 0; pt < 2; pt++)
		{
			li.rows_per_wave[pt] = a.pd[pt].rw_rpw;
			li.positions_per_row[pt] = a.pd[pt].rw_segs;
		}
		li.parts_per_row = (int)nparts;
		li.persistent_luma_workgroups = persist ? a.persist_wgs : 0;
		li.waves_per_workgroup = vfgs::kWavesPerWG;
		const vfgs::ImageLayout L = vfgs::layout_of(s.csubx, s.csuby, s.img_one_y, s.img_one_c, s.bs == 0);
		li.lds_bytes_per_workgroup = vfgs::lds_allocation(s.bs != 0, s.img_one_y, s.img_one_c, wide, L.lds_bytes + vfgs::kParamBytes);     // (what the kernel allocates)
		vfgs::describe_launch(li.kernel, sizeof li.kernel, 8 + s.bs, s.csubx, s.csuby, dg.out8, s.img_one_y, s.img_one_c, wide, persist);
		g_last_launch_valid = true;
	}
	return 0;
}

// host-memory stripe: stage through device buffers (compatibility path of the line API)
// Lines [y, y + height) are the call's stripe (the seed registers advance over all of them); lines [py, py + ph) inside it are
// the part THIS state's device processes (the whole stripe unless several devices share it).  Y/U/V address line y.
int run_host(void* Y, void* U, void* V, unsigned y, unsigned width, unsigned height, unsigned stride, unsigned cstride, unsigned py, unsigned ph)
{
	State& s = S();
	if (int e = ensure_init(-1)) return e;
	if (height == 0) return 0;
	const unsigned sz = s.bs ? 2 : 1;
	const unsigned nblk = (width + 15) / 16;
	if (ph == 0)
	{
		// nothing of this stripe is mine: the registers still move (vfgs_hw.c:291-298, 309-310)
		if (int e = check_luts(s)) return e;
		advance_seeds(s, y, height, nblk, y);
		return 0;
	}
	// rows of each plane touched by lines [py, py+ph)
	const unsigned crow0 = py / s.csuby;
	const unsigned crows = (py + ph - 1) / s.csuby - crow0 + 1;
	const unsigned rows[3] = {ph, crows, crows};
	const unsigned rowlen[3] = {nblk * 16 * sz, nblk * 16 / s.csubx * sz, nblk * 16 / s.csubx * sz};  // bytes the reference touches per row
	const unsigned dpitch[3] = {(rowlen[0] + 255) & ~255u, (rowlen[1] + 255) & ~255u, (rowlen[2] + 255) & ~255u};
	// (a 1-line stripe from vfgs_add_grain_line carries no pitch; it needs none)
	const size_t spitch[3] = {std::max<size_t>((size_t)stride * sz, rowlen[0]), std::max<size_t>((size_t)cstride * sz, rowlen[1]),
	                          std::max<size_t>((size_t)cstride * sz, rowlen[2])};
	uint8_t* host[3] = {(uint8_t*)Y + (size_t)(py - y) * spitch[0], (uint8_t*)U + (size_t)(crow0 - y / s.csuby) * spitch[1],
	                    (uint8_t*)V + (size_t)(crow0 - y / s.csuby) * spitch[2]};
	// Small stripes -- above all the single lines of the drop-in call -- go through pinned bounce buffers with two memcpys on
	// this thread.  Handing the caller's own (pageable) pages to the runtime makes it register them with the GPU, and a
	// registered frame buffer turns the caller's NEXT fread into / fwrite from it into a crawl: the unchanged reference CLI spent
	// 1.5 s of system time per three 4320p frames in its own file I/O after a first, line-by-line walk (tools/dev/line_time_shim.c,
	// profiles/r04_cli_shim_probe.log).  Large stripes keep the direct copies: there the runtime's pin-and-DMA path is the faster one.
	size_t total = 0;
	for (int i = 0; i < 3; i++) total += (size_t)dpitch[i] * rows[i];
	const bool via_bounce = total <= (1u << 20);
	for (int i = 0; i < 3; i++)
	{
		const size_t need = (size_t)dpitch[i] * rows[i] + 256;
		if (s.stage_cap[i] < need)
		{
			if (s.stage[i]) HIP_TRY(hipFree(s.stage[i]));
			s.stage[i] = nullptr; s.stage_cap[i] = 0;
			HIP_TRY(hipMalloc(&s.stage[i], need));
			s.stage_cap[i] = need;
		}
		if (via_bounce)
		{
			if (s.bounce_cap[i] < need)
			{
				if (s.bounce[i]) HIP_TRY(hipHostFree(s.bounce[i]));
				s.bounce[i] = nullptr; s.bounce_cap[i] = 0;
				HIP_TRY(hipHostMalloc((void**)&s.bounce[i], std::max<size_t>(need, 64u << 10), hipHostMallocDefault));
				s.bounce_cap[i] = std::max<size_t>(need, 64u << 10);
			}
			for (unsigned r = 0; r < rows[i]; r++)
				memcpy(s.bounce[i] + (size_t)r * dpitch[i], host[i] + (size_t)r * spitch[i], rowlen[i]);
			HIP_TRY(hipMemcpyAsync(s.stage[i], s.bounce[i], (size_t)dpitch[i] * rows[i], hipMemcpyHostToDevice, s.own_stream));
		}
		else
			HIP_TRY(hipMemcpy2DAsync(s.stage[i], dpitch[i], host[i], spitch[i], rowlen[i], rows[i], hipMemcpyHostToDevice, s.own_stream));
	}
	{
		InternalLaunch mark;
		if (int e = run_device(s.stage[0], s.stage[1], s.stage[2], s.stage[0], s.stage[1], s.stage[2], width, y, height, py, ph,
		                       dpitch[0] / sz, dpitch[1] / sz, 1, 0, 0, s.own_stream))
			return e;
	}
	for (int i = 0; i < 3; i++)
	{
		if (via_bounce) HIP_TRY(hipMemcpyAsync(s.bounce[i], s.stage[i], (size_t)dpitch[i] * rows[i], hipMemcpyDeviceToHost, s.own_stream));
		else HIP_TRY(hipMemcpy2DAsync(host[i], spitch[i], s.stage[i], dpitch[i], rowlen[i], rows[i], hipMemcpyDeviceToHost, s.own_stream));
	}
	HIP_TRY(hipStreamSynchronize(s.own_stream));
	if (via_bounce)
		for (int i = 0; i < 3; i++)
			for (unsigned r = 0; r < rows[i]; r++)
				memcpy(host[i] + (size_t)r * spitch[i], s.bounce[i] + (size_t)r * dpitch[i], rowlen[i]);
	return 0;
}

int run_host(void* Y, void* U, void* V, unsigned y, unsigned width, unsigned height, unsigned stride, unsigned cstride)
{
	return run_host(Y, U, V, y, width, height, stride, cstride, y, height);
}

// ------------------------------------------------------------------------------------
// Frames in host memory, pipelined (the data path around yuv_read / yuv_write, yuv.c:162-214).
// Frame i is uploaded on one stream while frame i-1 runs on a second and frame i-2 is downloaded on a third; a ring of
// three device frames, events between the stages, the caller's thread only waits when it wants a ring slot back.  Only
// the bytes the reference touches (whole blocks of every row) travel, so stride padding in host memory is never written.
// With pinned host memory (vfgs_hip_host_alloc) the copies are asynchronous; pageable memory works, the runtime then
// stages every copy itself and the calling thread blocks for it.  The seed registers advance frame by frame exactly as
// with nframes calls of the frame entry point.
// (py, ph: the lines of every frame THIS state's device processes -- the whole frame unless several devices share the frames)
int run_host_frames(void* const* Y, void* const* U, void* const* V, unsigned nframes, unsigned width, unsigned height,
                    unsigned stride, unsigned cstride, unsigned py, unsigned ph)
{
	State& s = S();
	if (int e = ensure_init(-1)) return e;
	if (nframes == 0 || height == 0) return 0;
	if (!Y || !U || !V) return fail(4, "vfgs_hip_add_grain_frames_host: null pointer array");
	if (width <= 128) return fail(5, "width %u: the hardware layer requires width > 128 (vfgs_hw.c:168)", width);
	if ((width + 15) / 16 > 1984) return fail(17, "width %u exceeds the supported 31744 samples", width);
	if (int e = check_luts(s)) return e;     // before anything is queued
	State::HostPipe& P = s.pipe;
	constexpr int D = State::HostPipe::kDepth;
	const unsigned sz = s.bs ? 2 : 1;
	const unsigned nblk = (width + 15) / 16;
	if (ph == 0)
	{
		for (unsigned f = 0; f < nframes; f++) advance_seeds(s, 0, height, nblk, 0);   // nothing of the frames is mine: the registers still move
		return 0;
	}
	const unsigned crow0 = py / s.csuby;
	const unsigned crows = (py + ph - 1) / s.csuby - crow0 + 1;
	const unsigned rows[3] = {ph, crows, crows};
	const unsigned rowlen[3] = {nblk * 16 * sz, nblk * 16 / s.csubx * sz, nblk * 16 / s.csubx * sz};
	if ((size_t)stride * sz < rowlen[0] || (size_t)cstride * sz < rowlen[1])
		return fail(6, "stride too small: every row must hold whole 16-sample blocks (vfgs_hw.c:301)");
	const unsigned dpitch[3] = {(rowlen[0] + 255) & ~255u, (rowlen[1] + 255) & ~255u, (rowlen[2] + 255) & ~255u};
	const size_t spitch[3] = {(size_t)stride * sz, (size_t)cstride * sz, (size_t)cstride * sz};
	if (!P.up)
	{
		HIP_TRY(hipStreamCreateWithFlags(&P.up, hipStreamNonBlocking));
		HIP_TRY(hipStreamCreateWithFlags(&P.run, hipStreamNonBlocking));
		HIP_TRY(hipStreamCreateWithFlags(&P.down, hipStreamNonBlocking));
		for (int k = 0; k < D; k++)
		{
			HIP_TRY(hipEventCreateWithFlags(&P.up_done[k], hipEventDisableTiming));
			HIP_TRY(hipEventCreateWithFlags(&P.run_done[k], hipEventDisableTiming));
			HIP_TRY(hipEventCreateWithFlags(&P.down_done[k], hipEventDisableTiming));
		}
	}
	int rc = 0;
	// (no early return below: whatever happens, the three streams are drained before the call returns -- copies into the
	// caller's memory may be in flight)
	auto hip = [&](hipError_t e, const char* what) {
		if (e != hipSuccess && !rc) rc = fail((int)e, "%s -> %s (vfgs_hip_add_grain_frames_host)", what, hipGetErrorString(e));
		return e == hipSuccess;
	};
	for (unsigned f = 0; f < nframes && !rc; f++)
	{
		const int k = (int)(f % D);
		if (!Y[f] || !U[f] || !V[f]) { rc = fail(4, "vfgs_hip_add_grain_frames_host: null plane pointer in frame %u", f); break; }
		uint8_t* host[3] = {(uint8_t*)Y[f] + (size_t)py * spitch[0], (uint8_t*)U[f] + (size_t)crow0 * spitch[1], (uint8_t*)V[f] + (size_t)crow0 * spitch[2]};
		if (P.busy[k]) { if (!hip(hipEventSynchronize(P.down_done[k]), "hipEventSynchronize")) break; P.busy[k] = false; }   // the slot's previous frame is back in host memory
		bool ok = true;
		for (int i = 0; i < 3 && ok; i++)
		{
			const size_t need = (size_t)dpitch[i] * rows[i] + 256;
			if (P.cap[k][i] < need)
			{
				if (P.dev[k][i]) (void)hipFree(P.dev[k][i]);
				P.dev[k][i] = nullptr; P.cap[k][i] = 0;
				ok = hip(hipMalloc(&P.dev[k][i], need), "hipMalloc");
				if (ok) P.cap[k][i] = need;
			}
			ok = ok && hip(hipMemcpy2DAsync(P.dev[k][i], dpitch[i], host[i], spitch[i], rowlen[i], rows[i], hipMemcpyHostToDevice, P.up), "hipMemcpy2DAsync (upload)");
		}
		if (!ok) break;
		if (!hip(hipEventRecord(P.up_done[k], P.up), "hipEventRecord") || !hip(hipStreamWaitEvent(P.run, P.up_done[k], 0), "hipStreamWaitEvent")) break;
		{
			InternalLaunch mark;
			rc = run_device(P.dev[k][0], P.dev[k][1], P.dev[k][2], P.dev[k][0], P.dev[k][1], P.dev[k][2], width, 0, height, py, ph,
			                dpitch[0] / sz, dpitch[1] / sz, 1, 0, 0, P.run);
		}
		if (rc) break;
		if (!hip(hipEventRecord(P.run_done[k], P.run), "hipEventRecord") || !hip(hipStreamWaitEvent(P.down, P.run_done[k], 0), "hipStreamWaitEvent")) break;
		for (int i = 0; i < 3 && ok; i++)
			ok = hip(hipMemcpy2DAsync(host[i], spitch[i], P.dev[k][i], dpitch[i], rowlen[i], rows[i], hipMemcpyDeviceToHost, P.down), "hipMemcpy2DAsync (download)");
		if (!ok) break;
		if (!hip(hipEventRecord(P.down_done[k], P.down), "hipEventRecord")) break;
		P.busy[k] = true;
	}
	// everything queued so far comes home before the call returns, also after an error
	(void)hipStreamSynchronize(P.up);
	(void)hipStreamSynchronize(P.run);
	const hipError_t e = hipStreamSynchronize(P.down);
	for (int k = 0; k < D; k++) P.busy[k] = false;
	if (!rc && e != hipSuccess) rc = fail(1, "hipStreamSynchronize: %s", hipGetErrorString(e));
	return rc;
}

// ------------------------------------------------------------------------------------
// vfgs_add_grain_line with look-ahead.
//
// The drop-in call hands over ONE line and must be complete on return, which costs a full
// H2D + launch + D2H + sync (~90 us) per line.  The reference's frame loop (vfgs_main.c:664-682)
// however walks a frame that is already complete in host memory, top to bottom, with fixed
// row pitches.  So once the pitches are known from two consecutive calls, a miss at line y starts
// computing the REST of the frame from the lines the caller has not handed over yet, in stripes of
// ~2 MB that travel through a ring of three slots: snapshot of the caller's lines into pinned
// memory, upload, kernel, download into pinned memory, on three streams.  The call returns as soon
// as the first stripe is back; while the caller walks through a stripe the next two are in flight,
// and entering a stripe queues the one after those (round 3 did one synchronous round trip per 256
// lines: 17 per 4320p frame).  A later call that is exactly the predicted next line (same y,
// pointers, width, no state change in between) and whose input bytes still equal the snapshot is
// served by a memcpy; anything else drops the stripes and starts over from the caller's registers.
// Lines the caller has not handed over are never written, the seed registers advance per call
// exactly as before (the stripes ahead run from a speculative copy), so the observable behaviour
// is unchanged.
// Lines are only ever read ahead inside rows the caller has PROVEN to own: either it has walked this very buffer
// (same three line-0 pointers, same width) top to bottom once before -- then the rows up to the height of that walk
// are known -- or it has told us with vfgs_hip_declare_frame().  A first frame, or a frame in a buffer not seen in the
// previous walk, is computed line by line.  vfgs_hip_line_lookahead(0) or VFGS_HIP_LINE_LOOKAHEAD=0 turn it off.

// Lines of a stripe computed ahead: about 2 MB of the caller's frame, whole block rows.  Small enough that a stripe's
// snapshot is still in the host's caches when its lines are compared and handed back, large enough for the link
// (profiles/r04_line_lookahead_stripe_sweep.log: 0.5 .. 8 MB x 1 or 2 stripes in flight; the calling thread's three passes
// over every byte -- snapshot, compare, hand back -- are what bounds the call, not the link).
unsigned lookahead_stripe_lines(const State& s, unsigned nblk)
{
	const unsigned sz = s.bs ? 2 : 1;
	const size_t line_bytes = (size_t)nblk * 16 * sz + 2 * ((size_t)nblk * 16 / s.csubx * sz) / s.csuby;
	size_t target = 2u << 20;
#ifdef VFGS_DEV_BUILD      // A/B of the stripe size (tools/dev): VFGS_LA_STRIPE_KB
	if (const char* e = getenv("VFGS_LA_STRIPE_KB")) target = (size_t)atoi(e) << 10;
#endif
	const unsigned n = (unsigned)(target / line_bytes) & ~15u;
	return std::min(512u, std::max(32u, n));
}

int lookahead_depth()
{
#ifdef VFGS_DEV_BUILD      // A/B of the stripes in flight ahead of the one being consumed (tools/dev): VFGS_LA_DEPTH=1..kRing-1
	if (const char* e = getenv("VFGS_LA_DEPTH")) return std::min(State::LineAhead::kRing - 1, std::max(1, atoi(e)));
#endif
	return State::LineAhead::kRing - 1;
}

// the caller's pointers for line y of the walk in progress (vfgs_main.c:672-681)
void lookahead_line_ptrs(const State& s, unsigned y, const uint8_t* (&p)[3])
{
	const State::LineAhead& la = s.la;
	const ptrdiff_t crow = (ptrdiff_t)(y / s.csuby) - (ptrdiff_t)(la.base_y / s.csuby);
	p[0] = la.Y0 + (ptrdiff_t)(y - la.base_y) * la.ypitch;
	p[1] = la.U0 + crow * la.cpitch;
	p[2] = la.V0 + crow * la.cpitch;
}

// every stripe still in flight lands before its buffers are reused (or released)
int lookahead_drain(State& s)
{
	for (State::LineAhead::Slot& sl : s.la.slot)
	{
		if (sl.used && !sl.waited) { HIP_TRY(hipEventSynchronize(sl.done)); s.la.stripes_dropped++; }
		sl.used = sl.waited = false;
	}
	return 0;
}

// Compute lines [y0, y0 + n) ahead of the caller into ring slot k: snapshot of the caller's lines (which it has NOT handed
// over yet, but has proven to own), upload, kernel from the speculative seed registers, download -- all queued, nothing waited for.
int lookahead_issue_impl(State& s, int k, unsigned y0, unsigned n);

int lookahead_issue(State& s, int k, unsigned y0, unsigned n)
{
	const int e = lookahead_issue_impl(s, k, y0, n);
	if (e)
	{
		// part of the stripe may be queued: nothing of it may still be moving when the slot's buffers are used again
		State::LineAhead& la = s.la;
		if (la.up) (void)hipStreamSynchronize(la.up);
		if (la.run) (void)hipStreamSynchronize(la.run);
		if (la.down) (void)hipStreamSynchronize(la.down);
		la.slot[k].used = false;
		la.valid = false;
	}
	return e;
}

int lookahead_issue_impl(State& s, int k, unsigned y0, unsigned n)
{
	State::LineAhead& la = s.la;
	State::LineAhead::Slot& sl = la.slot[k];
	const unsigned sz = s.bs ? 2 : 1;
	if (sl.used && !sl.waited) HIP_TRY(hipEventSynchronize(sl.done));
	sl.used = false;
	if (!la.up)
		la.up = la.run = la.down = s.own_stream;      // (lookahead_streams: streams of its own once the process has shown that it stays)
	if (!sl.done)
	{
		HIP_TRY(hipEventCreateWithFlags(&sl.up_done, hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&sl.run_done, hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
	}
	const unsigned crow0 = y0 / s.csuby;
	const unsigned crows = (y0 + n - 1) / s.csuby - crow0 + 1;
	const unsigned rows[3] = {n, crows, crows};
	const size_t spitch[3] = {(size_t)la.ypitch, (size_t)la.cpitch, (size_t)la.cpitch};
	const uint8_t* host[3];
	lookahead_line_ptrs(s, y0, host);
	for (int i = 0; i < 3; i++)
	{
		if ((size_t)la.dpitch[i] * rows[i] > sl.cap[i])      // (lookahead_buffers sized the slots for the largest stripe of this walk)
			return fail(31, "look-ahead: a stripe of %u rows does not fit its ring slot", rows[i]);
		for (unsigned r = 0; r < rows[i]; r++)   // snapshot of the caller's lines (also the H2D source)
			memcpy(sl.in[i] + (size_t)r * la.dpitch[i], host[i] + (size_t)r * spitch[i], la.rowlen[i]);
		HIP_TRY(hipMemcpyAsync(sl.dev[i], sl.in[i], (size_t)la.dpitch[i] * rows[i], hipMemcpyHostToDevice, la.up));
	}
	HIP_TRY(hipEventRecord(sl.up_done, la.up));
	HIP_TRY(hipStreamWaitEvent(la.run, sl.up_done, 0));
	// the kernel runs from the registers behind the previous stripe; the caller's registers only move when it hands lines over
	const uint64_t keep[4] = {s.rnd, s.rnd_up, s.line_rnd, s.line_rnd_up};
	s.rnd = la.spec[0]; s.rnd_up = la.spec[1]; s.line_rnd = la.spec[2]; s.line_rnd_up = la.spec[3];
	int e;
	{
		InternalLaunch mark;
		e = run_device(sl.dev[0], sl.dev[1], sl.dev[2], sl.dev[0], sl.dev[1], sl.dev[2], la.width, y0, n, y0, n,
		               la.dpitch[0] / sz, la.dpitch[1] / sz, 1, 0, 0, la.run);
	}
	la.spec[0] = s.rnd; la.spec[1] = s.rnd_up; la.spec[2] = s.line_rnd; la.spec[3] = s.line_rnd_up;
	s.rnd = keep[0]; s.rnd_up = keep[1]; s.line_rnd = keep[2]; s.line_rnd_up = keep[3];
	if (e) return e;
	HIP_TRY(hipEventRecord(sl.run_done, la.run));
	HIP_TRY(hipStreamWaitEvent(la.down, sl.run_done, 0));
	for (int i = 0; i < 3; i++)
		HIP_TRY(hipMemcpyAsync(sl.out[i], sl.dev[i], (size_t)la.dpitch[i] * rows[i], hipMemcpyDeviceToHost, la.down));
	HIP_TRY(hipEventRecord(sl.done, la.down));
	sl.used = true; sl.waited = false;
	sl.y0 = y0; sl.n = n; sl.crow0 = crow0;
	la.issued_end = y0 + n;
	la.stripes_issued++;
	return 0;
}

// Upload, kernel and download of the stripes run on three streams of their own, so that a stripe's download, the next one's kernel
// and the upload of the one behind it overlap -- in a process that stays.  Three fresh streams cost a process 23-27 ms to create and
// another 7-8 ms for the first download on one of them (profiles/r06_stream_first_use_probe.log): a tenth of a short run of the
// unchanged CLI.  So the first kStripesOnOwnStream stripes of a process travel on the stream the single lines use (one after the
// other: the calling thread's three passes over every byte bound the walk anyway), and the streams are created at
// the first miss behind them.  Nothing of the ring is in flight (line_speculate has drained it).
int lookahead_streams(State& s)
{
	State::LineAhead& la = s.la;
	constexpr uint64_t kStripesOnOwnStream = 64;
	if (la.own_streams || la.stripes_issued < kStripesOnOwnStream) return 0;
	hipStream_t st[3] = {nullptr, nullptr, nullptr};
	for (hipStream_t& x : st)
		if (hipStreamCreateWithFlags(&x, hipStreamNonBlocking) != hipSuccess)
		{
			(void)hipGetLastError();
			for (hipStream_t y : st) if (y) (void)hipStreamDestroy(y);
			return 0;      // (the stripes stay where they are)
		}
	la.up = st[0]; la.run = st[1]; la.down = st[2];
	la.own_streams = true;
	return 0;
}

// The ring's buffers for stripes of up to la.stripe_lines lines at the pitches of la.dpitch: one pinned block (snapshots and
// results of every slot) and one device block, grown when a walk needs more; nothing of the ring is in flight (line_speculate
// has drained it).
int lookahead_buffers(State& s)
{
	State::LineAhead& la = s.la;
	size_t need[3], per_slot = 0;
	for (int i = 0; i < 3; i++)
	{
		const size_t rows = i ? la.stripe_lines / s.csuby + 2 : la.stripe_lines + 1;
		need[i] = ((size_t)la.dpitch[i] * rows + 255) & ~(size_t)255;
		per_slot += need[i];
	}
	const size_t pin_need = 2 * per_slot * State::LineAhead::kRing, dev_need = per_slot * State::LineAhead::kRing;
	if (la.pin_cap < pin_need)
	{
		if (la.pin) HIP_TRY(hipHostFree(la.pin));
		la.pin = nullptr; la.pin_cap = 0;
		HIP_TRY(hipHostMalloc((void**)&la.pin, pin_need, hipHostMallocDefault));
		la.pin_cap = pin_need;
	}
	if (la.dev_cap < dev_need)
	{
		if (la.devblk) HIP_TRY(hipFree(la.devblk));
		la.devblk = nullptr; la.dev_cap = 0;
		HIP_TRY(hipMalloc(&la.devblk, dev_need));
		la.dev_cap = dev_need;
	}
	for (int k = 0; k < State::LineAhead::kRing; k++)
	{
		size_t off = 0;
		for (int i = 0; i < 3; i++)
		{
			State::LineAhead::Slot& sl = la.slot[k];
			sl.in[i] = la.pin + (size_t)k * 2 * per_slot + off;
			sl.out[i] = la.pin + (size_t)k * 2 * per_slot + per_slot + off;
			sl.dev[i] = (uint8_t*)la.devblk + (size_t)k * per_slot + off;
			sl.cap[i] = need[i];
			off += need[i];
		}
	}
	return 0;
}

// the next stripe(s) down the frame, up to kRing - 1 ahead of the one being consumed
int lookahead_extend(State& s)
{
	State::LineAhead& la = s.la;
	for (int d = 1; d <= lookahead_depth(); d++)
	{
		State::LineAhead::Slot& sl = la.slot[(la.head + d) % State::LineAhead::kRing];
		if (sl.used && sl.y0 >= la.slot[la.head].y0 + la.slot[la.head].n) continue;     // already ahead of the head
		if (la.issued_end >= la.frame_end) break;
		if (int e = lookahead_issue(s, (la.head + d) % State::LineAhead::kRing, la.issued_end, std::min(la.stripe_lines, la.frame_end - la.issued_end)))
			return e;
	}
	return 0;
}

// hand line y (inside the head stripe, which has landed) to the caller
void lookahead_serve(State& s, void* Y, void* U, void* V, unsigned y)
{
	const State::LineAhead& la = s.la;
	const State::LineAhead::Slot& sl = la.slot[la.head];
	const size_t k = y - sl.y0, crow = (size_t)(y / s.csuby - sl.crow0);
	memcpy(Y, sl.out[0] + k * la.dpitch[0], la.rowlen[0]);
	if (y % s.csuby == 0)
	{
		memcpy(U, sl.out[1] + crow * la.dpitch[1], la.rowlen[1]);
		memcpy(V, sl.out[2] + crow * la.dpitch[2], la.rowlen[2]);
	}
}

// A miss at line y with rows [y, frame_end) proven: start over from the caller's registers.  The first stripe ends on a block
// row boundary; the next ones are queued behind it right away, so their transfers overlap the caller's walk through this one.
int line_speculate(State& s, void* Y, void* U, void* V, unsigned y, unsigned width, unsigned frame_end)
{
	State::LineAhead& la = s.la;
	const unsigned sz = s.bs ? 2 : 1;
	const unsigned nblk = (width + 15) / 16;
	la.valid = false;
	if (int e = lookahead_drain(s)) return e;
	for (int i = 0; i < 3; i++)
	{
		la.rowlen[i] = (i ? nblk * 16 / s.csubx : nblk * 16) * sz;
		la.dpitch[i] = (la.rowlen[i] + 255) & ~255u;
	}
	la.width = width;
	la.base_y = y;
	la.Y0 = (const uint8_t*)Y; la.U0 = (const uint8_t*)U; la.V0 = (const uint8_t*)V;
	la.frame_end = frame_end;
	la.stripe_lines = lookahead_stripe_lines(s, nblk);
	if (int e = lookahead_buffers(s)) return e;
	if (int e = lookahead_streams(s)) return e;
	la.spec[0] = s.rnd; la.spec[1] = s.rnd_up; la.spec[2] = s.line_rnd; la.spec[3] = s.line_rnd_up;
	la.head = 0;
	la.issued_end = y;
	const unsigned first_end = std::min(frame_end, (y & ~15u) + la.stripe_lines);
	if (int e = lookahead_issue(s, 0, y, first_end - y)) return e;
	if (int e = lookahead_extend(s)) return e;
	HIP_TRY(hipEventSynchronize(la.slot[0].done));
	la.slot[0].waited = true;
	lookahead_serve(s, Y, U, V, y);
	advance_seeds(s, y, 1, nblk, y);
	la.gen = s.gen;
	la.next = y + 1;
	la.valid = la.next < la.frame_end;
	return 0;
}

int line_call(void* Y, void* U, void* V, unsigned y, unsigned width)
{
	State& s = S();
	if (int e = ensure_init(-1)) return e;
	State::LineAhead& la = s.la;
	static const bool env_on = [] { const char* e = getenv("VFGS_HIP_LINE_LOOKAHEAD"); return !(e && e[0] == '0'); }();
	if (!la.enabled || !env_on)
		return run_host(Y, U, V, y, width, 1, 0, 0);

	const uint8_t *cY = (const uint8_t*)Y, *cU = (const uint8_t*)U, *cV = (const uint8_t*)V;
	const unsigned nblk = (width + 15) / 16;
	int rc = -1;

	// 1. served from a stripe computed ahead?
	if (la.valid)
	{
		bool hit = la.gen == s.gen && y == la.next && width == la.width;
		if (hit)
		{
			const uint8_t* want[3];
			lookahead_line_ptrs(s, y, want);
			hit = cY == want[0] && cU == want[1] && cV == want[2];
		}
		if (hit && y == la.slot[la.head].y0 + la.slot[la.head].n)
		{
			// the walk enters the next stripe: the slot behind it is free for the stripe after the ones in flight
			const int nh = (la.head + 1) % State::LineAhead::kRing;
			hit = la.slot[nh].used && la.slot[nh].y0 == y;
			if (hit)
			{
				la.slot[la.head].used = false;
				la.head = nh;
				// (a failure to queue stripes FURTHER ahead is about later lines: this line is then computed alone below -- a miss --
				// so that the caller, who gets no error code from the void drop-in call, still receives it and the registers stay in step)
				if (lookahead_extend(s)) hit = false;
			}
		}
		if (hit)
		{
			State::LineAhead::Slot& sl = la.slot[la.head];
			if (!sl.waited)
			{
				if (hipEventSynchronize(sl.done) != hipSuccess) { (void)hipGetLastError(); hit = false; }
				else sl.waited = true;
			}
		}
		if (hit)
		{
			State::LineAhead::Slot& sl = la.slot[la.head];
			const size_t k = y - sl.y0, crow = (size_t)(y / s.csuby - sl.crow0);
			const bool chroma = (y % s.csuby) == 0;
			hit = !memcmp(cY, sl.in[0] + k * la.dpitch[0], la.rowlen[0]) &&
			      (!chroma || (!memcmp(cU, sl.in[1] + crow * la.dpitch[1], la.rowlen[1]) && !memcmp(cV, sl.in[2] + crow * la.dpitch[2], la.rowlen[2])));
		}
		if (hit)
		{
			lookahead_serve(s, Y, U, V, y);
			advance_seeds(s, y, 1, nblk, y);
			la.next = y + 1;
			if (la.next >= la.frame_end)
				la.valid = false;
			rc = 0;
		}
		else
			la.valid = false;
	}

	if (rc != 0)
	{
		// 2. learn the caller's walk from consecutive calls (vfgs_main.c:675-680)
		if (la.have_prev && y == la.py + 1 && width == la.pwidth)
		{
			la.ypitch = cY - la.pY;
			if (s.csuby == 1 || (la.py & 1))
			{
				const ptrdiff_t du = cU - la.pU, dv = cV - la.pV;
				la.cpitch = (du == dv) ? du : 0;
			}
			else if (cU != la.pU || cV != la.pV)
				la.cpitch = 0, la.ypitch = 0;     // not the reference's walk: no look-ahead
		}
		else if (la.have_prev && y == 0 && la.py > 0 && width == la.pwidth && !la.declared)
		{
			// wrapped around.  The walk that just ended proves py + 1 rows -- of the buffer it went through: it carries
			// over only if this walk starts in the same buffer
			la.frame_h = (cY == la.bY && cU == la.bU && cV == la.bV) ? la.py + 1 : 0;
		}
		else if (la.have_prev && !la.declared)
			la.ypitch = la.cpitch = 0, la.frame_h = 0, la.from_zero = false;
		if (y == 0)
		{
			if (la.declared && !(cY == la.bY && cU == la.bU && cV == la.bV && width == la.pwidth))
				la.declared = false, la.ypitch = la.cpitch = 0, la.frame_h = 0;      // not the declared frame
			else if (!la.declared && !(cY == la.bY && cU == la.bU && cV == la.bV))
				la.ypitch = la.cpitch = 0;      // another buffer: the pitches of the previous walk prove nothing about this one (they are learned again from its first lines)
			la.bY = cY; la.bU = cU; la.bV = cV;
			la.from_zero = true;
		}
		// A promise from outside the program, for binaries that cannot be rebuilt with vfgs_hip_declare_frame():
		// VFGS_HIP_FRAME_HEIGHT=<lines> says that every walk that starts at line 0 goes through planes of at least that many
		// lines (at the pitches its first lines show).  Looked at on misses only.
		if (!la.declared && la.frame_h == 0 && la.from_zero && la.ypitch > 0 && la.cpitch > 0)
			if (const char* e = getenv("VFGS_HIP_FRAME_HEIGHT"))
			{
				const long hgt = atol(e);
				if (hgt > 0 && hgt <= 65536) la.frame_h = (unsigned)hgt;
			}

		// 3. compute: this line alone, or this line plus the lines the caller is about to hand over (never beyond
		// the rows it has proven to own)
		const unsigned sz = s.bs ? 2 : 1;
		const size_t ylen = (size_t)nblk * 16 * sz, clen = (size_t)nblk * 16 / s.csubx * sz;
		const bool ahead = la.ypitch >= (ptrdiff_t)ylen && la.cpitch >= (ptrdiff_t)clen && width > 128 && la.frame_h > y + 1;
		rc = ahead ? line_speculate(s, Y, U, V, y, width, la.frame_h) : run_host(Y, U, V, y, width, 1, 0, 0);
		if (ahead && rc)
		{
			// working ahead failed (no pinned memory for the ring, a stripe that could not be queued): the caller's registers have
			// not moved -- stripes run from a copy of them -- so the line is computed alone, as without the look-ahead
			la.valid = false;
			rc = run_host(Y, U, V, y, width, 1, 0, 0);
		}
	}
	la.have_prev = true;
	la.pY = cY; la.pU = cU; la.pV = cV; la.py = y; la.pwidth = width;
	return rc;
}

// ------------------------------------------------------------------------------------
// Several devices in one process (vfgs_hip_init_devices): the reference's API is one process-global hardware layer driven by
// one thread (vfgs_main.c:664-682), so a frame that lives in HOST memory can only use one PCIe link -- unless the library
// splits it.  Replicas of the programmed state are kept for the other devices; the host-memory entry points hand every
// device a stripe of whole 16-line block rows of each frame (no halo, no exchange: SURVEY 8e) and run the devices
// concurrently, one worker thread per extra device.  Every state advances the seed registers over the WHOLE stripe or
// frame, exactly like the ranks of bench.py, so all of them stay in step without talking to each other.

// bring replica r (current state of the calling thread, its device current) in line with primary p
int sync_replica(State& r, State& p)
{
	if (r.synced_prog != p.prog_gen)
	{
		bool changed = memcmp(r.bank, p.bank, sizeof r.bank) || memcmp(r.slut, p.slut, sizeof r.slut) || memcmp(r.plut, p.plut, sizeof r.plut) ||
		               r.scale_shift != p.scale_shift || r.bs != p.bs || r.ymin != p.ymin || r.ymax != p.ymax || r.cmin != p.cmin ||
		               r.cmax != p.cmax || r.csubx != p.csubx || r.csuby != p.csuby || r.dev_origin[0] != p.dev_origin[0] ||
		               r.dev_origin[1] != p.dev_origin[1];
		memcpy(r.bank, p.bank, sizeof r.bank); memcpy(r.slut, p.slut, sizeof r.slut); memcpy(r.plut, p.plut, sizeof r.plut);
		r.plut_seen = false;
		r.scale_shift = p.scale_shift; r.bs = p.bs; r.ymin = p.ymin; r.ymax = p.ymax; r.cmin = p.cmin; r.cmax = p.cmax;
		r.csubx = p.csubx; r.csuby = p.csuby;
		r.dev_origin[0] = p.dev_origin[0]; r.dev_origin[1] = p.dev_origin[1];
		r.fw_pending.clear(); r.fw_last_valid = false;
		if (p.dev_origin[0] | p.dev_origin[1])
		{
			// patterns generated on the primary's device never visit the host: copy its banks (64 KiB; the dispatcher has
			// flushed and drained the primary's generation kernels before the workers start)
			if (int e = fw_prepare(r)) return e;
			if (int e = fw_bank_stream(r, r.own_stream)) return e;
			HIP_TRY(hipMemcpyAsync(r.dev_bank, p.dev_bank, 2 * vfgs::kSlots * 4096, hipMemcpyDefault, r.own_stream));
			HIP_TRY(hipStreamSynchronize(r.own_stream));
			changed = true;
		}
		if (changed) r.tables_dirty = true;
		r.synced_prog = p.prog_gen;
	}
	if (r.synced_seed != p.seed_epoch)
	{
		r.lfsr.reseed(p.lfsr.seed_reg());
		r.synced_seed = p.seed_epoch;
	}
	r.rnd = p.rnd; r.rnd_up = p.rnd_up; r.line_rnd = p.line_rnd; r.line_rnd_up = p.line_rnd_up;
	return 0;
}

// lines [y, y + height) -> n parts of whole block rows, sizes differing by at most one block row
void split_lines(unsigned y, unsigned height, int n, int d, unsigned* py, unsigned* ph)
{
	const unsigned br0 = y >> 4, nbr = ((y + height - 1) >> 4) - br0 + 1;
	const unsigned base = nbr / n, extra = nbr % n;
	const unsigned first = br0 + d * base + std::min<unsigned>(d, extra), count = base + ((unsigned)d < extra ? 1 : 0);
	const unsigned lo = std::max(y, first * 16), hi = std::min(y + height, (first + count) * 16);
	*py = count ? lo : y;
	*ph = (count && hi > lo) ? hi - lo : 0;
}

// run fn(d) for every device, state d current on the thread that runs it; device 0 on the calling thread
template <class F>
int on_all_devices(F&& fn)
{
	State& p = g_states[0];
	if (int e = ensure_init(-1)) return e;
	if (p.dev_origin[0] | p.dev_origin[1])
	{
		// device-generated patterns: have them finished before the replicas copy them
		bool stale = false;
		for (int d = 1; d < g_ndev; d++) stale = stale || g_states[d].synced_prog != p.prog_gen;
		if (stale)
		{
			if (int e = fw_flush(p, p.own_stream)) return e;
			if (int e = fw_bank_stream(p, p.own_stream)) return e;
			HIP_TRY(hipStreamSynchronize(p.own_stream));
		}
	}
	// the replicas take over the primary's programming and seed registers BEFORE anything runs: the primary's own part
	// advances those registers
	for (int d = 1; d < g_ndev; d++)
	{
		g_cur = &g_states[d];
		int e = ensure_init(-1);
		if (!e) e = sync_replica(g_states[d], p);
		g_cur = nullptr;
		if (e) { (void)hipSetDevice(p.device); return e; }
	}
	HIP_TRY(hipSetDevice(p.device));
	int rc[kMaxDevices] = {};
	std::string msg[kMaxDevices];
	auto body = [&](int d) {
		g_cur = &g_states[d];
		int e = ensure_init(-1);
		if (!e) e = fn(d);
		rc[d] = e;
		if (e) msg[d] = g_errstr;
		g_cur = nullptr;
	};
	std::vector<std::thread> workers;
	for (int d = 1; d < g_ndev; d++) workers.emplace_back(body, d);
	body(0);
	for (std::thread& t : workers) t.join();
	for (int d = 0; d < g_ndev; d++)
		if (rc[d]) return fail(rc[d], "device %d of %d: %s", d, g_ndev, msg[d].c_str());
	return 0;
}

int run_host_multi(void* Y, void* U, void* V, unsigned y, unsigned width, unsigned height, unsigned stride, unsigned cstride)
{
	// a stripe of less than one block row per device is not worth the threads (and a 1-line call has no pitch to split by)
	if (g_ndev == 1 || height < 16u * g_ndev)
		return run_host(Y, U, V, y, width, height, stride, cstride);
	return on_all_devices([&](int d) {
		unsigned py, ph;
		split_lines(y, height, g_ndev, d, &py, &ph);
		return run_host(Y, U, V, y, width, height, stride, cstride, py, ph);
	});
}

int run_host_frames_multi(void* const* Y, void* const* U, void* const* V, unsigned nframes, unsigned width, unsigned height,
                          unsigned stride, unsigned cstride)
{
	if (g_ndev == 1 || height < 16u * g_ndev)
		return run_host_frames(Y, U, V, nframes, width, height, stride, cstride, 0, height);
	return on_all_devices([&](int d) {
		unsigned py, ph;
		split_lines(0, height, g_ndev, d, &py, &ph);
		return run_host_frames(Y, U, V, nframes, width, height, stride, cstride, py, ph);
	});
}

void release_state(State& s);

// inside an overlap region the device-pointer calls made on the region's stream alternate between two internal streams
hipStream_t pick_stream(void* stream)
{
	State& s = g_states[0];
	if (s.ov.active && (hipStream_t)stream == s.ov.user) return s.ov.s[s.ov.n++ & 1];
	return (hipStream_t)stream;
}

}  // namespace

namespace vfgs {
// for the other host files of the library (vfgs_cfg_host.cpp): record an error like fail()
int set_error(int code, const char* msg) { return fail(code, "%s", msg); }
}

namespace {
void release_state_impl(State& s)
{
	if (!s.inited) return;
	(void)fw_flush(s, s.own_stream);
	(void)hipDeviceSynchronize();
	// device-generated patterns move to the host mirror so the programmed state survives
	for (int c = 0; c < 2; c++)
		for (int k = 0; k < vfgs::kSlots; k++)
			if (s.dev_origin[c] >> k & 1)
				(void)hipMemcpy(s.bank[c][k], s.dev_bank + (size_t)(c * vfgs::kSlots + k) * 4096, 4096, hipMemcpyDeviceToHost);
	s.tables_ring.release();
	s.lfsr.release();
	s.stripes.release();
	for (int i = 0; i < 3; i++) { if (s.stage[i]) (void)hipFree(s.stage[i]); s.stage[i] = nullptr; s.stage_cap[i] = 0; }
	for (int i = 0; i < 3; i++) { if (s.bounce[i]) (void)hipHostFree(s.bounce[i]); s.bounce[i] = nullptr; s.bounce_cap[i] = 0; }
	s.pipe.release();
	s.la.release();
	if (s.fw_const) (void)hipFree(s.fw_const);
	if (s.dev_bank) (void)hipFree(s.dev_bank);
	if (s.dev_raw) (void)hipFree(s.dev_raw);
	if (s.bank_ev) (void)hipEventDestroy(s.bank_ev);
	s.fw_const = nullptr; s.dev_bank = nullptr; s.dev_raw = nullptr; s.bank_ev = nullptr;
	s.bank_used = false; s.bank_stream = nullptr;
	s.dev_origin[0] = s.dev_origin[1] = 0;
	s.fw_last_valid = false;
	for (int i = 0; i < 2; i++)
	{
		if (s.ov.s[i]) (void)hipStreamDestroy(s.ov.s[i]);
		if (s.ov.join[i]) (void)hipEventDestroy(s.ov.join[i]);
		s.ov.s[i] = nullptr; s.ov.join[i] = nullptr;
	}
	if (s.ov.fork) (void)hipEventDestroy(s.ov.fork);
	s.ov.fork = nullptr; s.ov.active = false;
	if (s.own_stream) (void)hipStreamDestroy(s.own_stream);
	if (s.ev0) (void)hipEventDestroy(s.ev0);
	if (s.ev1) (void)hipEventDestroy(s.ev1);
	s.own_stream = nullptr; s.ev0 = s.ev1 = nullptr;
	s.tables_dirty = true;
	s.inited = false;
}
void release_state(State& s) { release_state_impl(s); }
}  // namespace

// ========================================================================================
// C ABI

extern "C" {

void vfgs_set_luma_pattern(int index, signed char* P)
{
	std::lock_guard<std::mutex> g(g_mu);
	S().gen++;
	S().prog_gen++;
	if (index < 0 || index >= vfgs::kSlots) { fail(20, "vfgs_set_luma_pattern: index %d", index); die("bad pattern index (vfgs_hw.c:316)"); }
	State& s = S();
	const bool host_slot = !(s.dev_origin[0] >> index & 1);
	if (host_slot && !memcmp(s.bank[0][index], P, 64 * 64)) return;   // unchanged: the device image stays valid
	memcpy(s.bank[0][index], P, 64 * 64);   // vfgs_hw.c:317
	s.dev_origin[0] &= ~(1u << index);
	s.tables_dirty = true;
}

void vfgs_set_chroma_pattern(int index, signed char* P)
{
	std::lock_guard<std::mutex> g(g_mu);
	S().gen++;
	S().prog_gen++;
	State& s = S();
	if (index < 0 || index >= vfgs::kSlots) { fail(20, "vfgs_set_chroma_pattern: index %d", index); die("bad pattern index (vfgs_hw.c:322)"); }
	bool same = !(s.dev_origin[1] >> index & 1);
	for (int i = 0; i < 64 / s.csuby && same; i++)
		same = !memcmp(s.bank[1][index][i], P + (64 / s.csuby) * i, 64 / s.csubx);
	if (same) return;                        // unchanged: the device image stays valid
	for (int i = 0; i < 64 / s.csuby; i++)   // vfgs_hw.c:323-324: pitch from csuby, length from csubx
		memcpy(s.bank[1][index][i], P + (64 / s.csuby) * i, 64 / s.csubx);
	s.dev_origin[1] &= ~(1u << index);
	s.tables_dirty = true;
}

void vfgs_set_scale_lut(int c, unsigned char lut[])
{
	std::lock_guard<std::mutex> g(g_mu);
	S().gen++;
	S().prog_gen++;
	if (c < 0 || c > 2) { fail(21, "vfgs_set_scale_lut: component %d", c); die("bad component (vfgs_hw.c:329)"); }
	if (!memcmp(S().slut[c], lut, 256)) return;   // unchanged (e.g. the same model re-sent with a new seed): nothing to upload
	memcpy(S().slut[c], lut, 256);
	S().tables_dirty = true;
}

void vfgs_set_pattern_lut(int c, unsigned char lut[])
{
	std::lock_guard<std::mutex> g(g_mu);
	S().gen++;
	S().prog_gen++;
	if (c < 0 || c > 2) { fail(21, "vfgs_set_pattern_lut: component %d", c); die("bad component (vfgs_hw.c:335)"); }
	if (!memcmp(S().plut[c], lut, 256)) return;
	memcpy(S().plut[c], lut, 256);
	S().plut_seen = false;
	S().tables_dirty = true;
}

void vfgs_set_seed(unsigned int seed)
{
	std::lock_guard<std::mutex> g(g_mu);
	S().gen++;
	State& s = S();
	s.lfsr.reseed(seed << 1);   // vfgs_hw.c:343
	s.seed_epoch++;
	s.rnd = s.rnd_up = s.line_rnd = s.line_rnd_up = 0;
}

void vfgs_set_scale_shift(int shift)
{
	std::lock_guard<std::mutex> g(g_mu);
	S().gen++;
	S().prog_gen++;
	if (shift < 2 || shift >= 8) { fail(22, "vfgs_set_scale_shift: %d", shift); die("shift out of 2..7 (vfgs_hw.c:348)"); }
	if (S().scale_shift != shift + 6 - S().bs) S().tables_dirty = true;   // the LUT image holds pre-shifted scales
	S().scale_shift = shift + 6 - S().bs;   // vfgs_hw.c:349
}

void vfgs_set_depth(int depth)
{
	std::lock_guard<std::mutex> g(g_mu);
	S().gen++;
	S().prog_gen++;
	State& s = S();
	if (depth != 8 && depth != 10) { fail(23, "vfgs_set_depth: %d", depth); die("depth must be 8 or 10 (vfgs_hw.c:354)"); }
	if (s.bs != depth - 8) s.tables_dirty = true;
	s.scale_shift += s.bs - (depth - 8);     // vfgs_hw.c:356-359
	s.bs = depth - 8;
}

void vfgs_set_legal_range(int legal)
{
	std::lock_guard<std::mutex> g(g_mu);
	S().gen++;
	S().prog_gen++;
	State& s = S();
	s.ymin = s.cmin = legal ? 16 : 0;        // vfgs_hw.c:366-378
	s.ymax = legal ? 235 : 255;
	s.cmax = legal ? 240 : 255;
}

void vfgs_set_chroma_subsampling(int subx, int suby)
{
	std::lock_guard<std::mutex> g(g_mu);
	S().gen++;
	S().prog_gen++;
	if ((subx != 1 && subx != 2) || (suby != 1 && suby != 2)) { fail(24, "vfgs_set_chroma_subsampling: %d,%d", subx, suby); die("subsampling must be 1 or 2 (vfgs_hw.c:384-385)"); }
	if (S().csubx == subx && S().csuby == suby) return;
	S().csubx = subx;
	S().csuby = suby;
	S().tables_dirty = true;
}

void vfgs_add_grain_line(void* Y, void* U, void* V, int y, int width)
{
	std::lock_guard<std::mutex> g(g_mu);
	if (line_call(Y, U, V, (unsigned)y, (unsigned)width))
		die("vfgs_add_grain_line");
}

void vfgs_add_grain_stripe(void* Y, void* U, void* V, unsigned y, unsigned width, unsigned height, unsigned stride, unsigned cstride)
{
	std::lock_guard<std::mutex> g(g_mu);
	S().gen++;
	if (run_host_multi(Y, U, V, y, width, height, stride, cstride))
		die("vfgs_add_grain_stripe");
}

void vfgs_hip_line_lookahead(int enable)
{
	std::lock_guard<std::mutex> g(g_mu);
	S().la.enabled = enable != 0;
	S().la.valid = false;
}

int vfgs_hip_declare_frame(const void* Y, const void* U, const void* V, unsigned width, unsigned height, unsigned stride, unsigned cstride)
{
	std::lock_guard<std::mutex> g(g_mu);
	State& s = S();
	State::LineAhead& la = s.la;
	const unsigned sz = s.bs ? 2 : 1, nblk = (width + 15) / 16;
	if (!Y || !U || !V || height == 0) { la.declared = false; la.frame_h = 0; la.ypitch = la.cpitch = 0; la.valid = false; return 0; }
	if (stride < nblk * 16 || cstride < nblk * 16 / s.csubx)
		return fail(6, "vfgs_hip_declare_frame: stride %u/%u too small: whole 16-sample blocks are written (need >= %u/%u)", stride, cstride, nblk * 16, nblk * 16 / s.csubx);
	la.declared = true;
	la.bY = (const uint8_t*)Y; la.bU = (const uint8_t*)U; la.bV = (const uint8_t*)V;
	la.pwidth = width;
	la.frame_h = height;
	la.ypitch = (ptrdiff_t)stride * sz; la.cpitch = (ptrdiff_t)cstride * sz;
	la.have_prev = false;
	la.valid = false;
	return 0;
}

void vfgs_hip_reset_state(void)
{
	std::lock_guard<std::mutex> g(g_mu);
	S().gen++;
	S().prog_gen++;
	State& s = S();
	memset(s.bank, 0, sizeof s.bank);
	memset(s.slut, 0, sizeof s.slut);
	memset(s.plut, 0, sizeof s.plut);
	s.plut_seen = false;
	s.dev_origin[0] = s.dev_origin[1] = 0;
	s.fw_pending.clear();
	s.fw_last_valid = false;
	s.scale_shift = 5 + 6;
	s.bs = 0;
	s.ymin = s.cmin = 0;
	s.ymax = s.cmax = 255;
	s.csubx = s.csuby = 2;
	s.lfsr.reseed(0xdeadbeefu);
	s.seed_epoch++;
	s.rnd = s.rnd_up = s.line_rnd = s.line_rnd_up = 0;
	s.tables_dirty = true;
	if (s.ov.active)   // a region left open (a caller that failed between _begin and _end): drain its internal streams on the host --
	{                  // the stream it was opened on is the caller's and may be gone by now, so it is not touched
		for (int i = 0; i < 2; i++)
			if (s.ov.s[i]) (void)hipStreamSynchronize(s.ov.s[i]);
		s.ov.active = false;
		s.ov.user = nullptr;
	}
}

int vfgs_hip_init(int device)
{
	std::lock_guard<std::mutex> g(g_mu);
	return ensure_init(device);
}

int vfgs_hip_overlap_begin(void* stream)
{
	std::lock_guard<std::mutex> g(g_mu);
	if (int e = ensure_init(-1)) return e;
	State& s = S();
	if (s.ov.active) return fail(27, "vfgs_hip_overlap_begin: a region is already open");
	// The two streams must sit on two hardware queues or nothing overlaps.  The runtime deals a fixed number of queues per
	// priority class to the process's streams in creation order and doubles up beyond it (measured: with one more application
	// stream alive the two landed on one queue and the region gained nothing); the high-priority class is a pool of its own
	// that the library is normally alone in, so its first two streams get a queue each.
	int prio_least = 0, prio_greatest = 0;
	HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
#ifdef VFGS_DEV_BUILD      // A/B of the priority class (tools/dev): VFGS_OV_PRIO=normal|low
	if (const char* e = getenv("VFGS_OV_PRIO")) prio_greatest = e[0] == 'n' ? 0 : (e[0] == 'l' ? prio_least : prio_greatest);
#endif
	for (int i = 0; i < 2; i++)
	{
		if (!s.ov.s[i]) HIP_TRY(hipStreamCreateWithPriority(&s.ov.s[i], hipStreamNonBlocking, prio_greatest));
		if (!s.ov.join[i]) HIP_TRY(hipEventCreateWithFlags(&s.ov.join[i], hipEventDisableTiming));
	}
	if (!s.ov.fork) HIP_TRY(hipEventCreateWithFlags(&s.ov.fork, hipEventDisableTiming));
	HIP_TRY(hipEventRecord(s.ov.fork, (hipStream_t)stream));
	for (int i = 0; i < 2; i++) HIP_TRY(hipStreamWaitEvent(s.ov.s[i], s.ov.fork, 0));
	s.ov.user = (hipStream_t)stream;
	s.ov.n = 0;
	s.ov.active = true;
	return 0;
}

int vfgs_hip_overlap_end(void* stream)
{
	std::lock_guard<std::mutex> g(g_mu);
	if (int e = ensure_init(-1)) return e;
	State& s = S();
	if (!s.ov.active || s.ov.user != (hipStream_t)stream) return fail(27, "vfgs_hip_overlap_end: no region open on this stream");
	// both joins are attempted whatever the first one returns; a join that cannot be queued is replaced by waiting for that
	// internal stream on the host, so that work queued on `stream` after this call never runs ahead of the region's launches;
	// only then is the region closed, and the first error (if any) reported
	hipError_t first = hipSuccess;
	for (int i = 0; i < 2; i++)
	{
		hipError_t e = hipEventRecord(s.ov.join[i], s.ov.s[i]);
		if (e == hipSuccess) e = hipStreamWaitEvent(s.ov.user, s.ov.join[i], 0);
		if (e != hipSuccess)
		{
			(void)hipStreamSynchronize(s.ov.s[i]);
			if (first == hipSuccess) first = e;
		}
	}
	s.ov.active = false;
	s.ov.user = nullptr;
	HIP_TRY(first);
	return 0;
}

int vfgs_hip_init_devices(const int* devices, int n)
{
	std::lock_guard<std::mutex> g(g_mu);
	if (!devices || n < 1 || n > kMaxDevices) return fail(26, "vfgs_hip_init_devices: %d devices (1..%d)", n, kMaxDevices);
	int have = 0;
	HIP_TRY(hipGetDeviceCount(&have));
	for (int i = 0; i < n; i++)
		if (devices[i] < 0 || devices[i] >= have) return fail(2, "vfgs_hip_init_devices: device %d (%d visible)", devices[i], have);
	for (int i = 1; i < kMaxDevices; i++)
		if (g_states[i].inited && (i >= n || g_states[i].device != devices[i]))
		{
			g_cur = &g_states[i];
			(void)hipSetDevice(g_states[i].device);
			release_state(g_states[i]);
			g_cur = nullptr;
		}
	if (int e = ensure_init(devices[0])) return e;
	for (int i = 1; i < n; i++)
	{
		g_cur = &g_states[i];
		const int e = ensure_init(devices[i]);
		g_cur = nullptr;
		if (e) return e;
		g_states[i].synced_prog = g_states[i].synced_seed = ~0ull;
		g_states[i].la.enabled = false;
	}
	g_ndev = n;
	(void)hipSetDevice(devices[0]);
	return 0;
}

void vfgs_hip_shutdown(void)
{
	std::lock_guard<std::mutex> g(g_mu);
	for (int i = kMaxDevices - 1; i >= 0; i--)
	{
		if (!g_states[i].inited) continue;
		g_cur = &g_states[i];
		(void)hipSetDevice(g_states[i].device);
		release_state(g_states[i]);
		g_cur = nullptr;
	}
	g_ndev = 1;
}

int vfgs_hip_add_grain_stripe_dev(void* dY, void* dU, void* dV, unsigned y, unsigned width, unsigned height,
                                  unsigned stride, unsigned cstride, void* stream)
{
	std::lock_guard<std::mutex> g(g_mu);
	S().gen++;
	return run_device(dY, dU, dV, dY, dU, dV, width, y, height, y, height, stride, cstride, 1, 0, 0, pick_stream(stream));
}

int vfgs_hip_add_grain_frame_dev(void* dY, void* dU, void* dV, unsigned width, unsigned height,
                                 unsigned stride, unsigned cstride, void* stream)
{
	std::lock_guard<std::mutex> g(g_mu);
	S().gen++;
	return run_device(dY, dU, dV, dY, dU, dV, width, 0, height, 0, height, stride, cstride, 1, 0, 0, pick_stream(stream));
}

int vfgs_hip_add_grain_frame_part_dev(void* dY, void* dU, void* dV, unsigned width, unsigned frame_height,
                                      unsigned part_y, unsigned part_height, unsigned stride, unsigned cstride, void* stream)
{
	std::lock_guard<std::mutex> g(g_mu);
	S().gen++;
	if (part_y & 15) return fail(11, "part_y must be a multiple of 16");
	if (part_y > frame_height || part_height > frame_height - part_y) return fail(12, "part exceeds the frame");     // (no 32-bit wrap)
	return run_device(dY, dU, dV, dY, dU, dV, width, 0, frame_height, part_y, part_height, stride, cstride, 1, 0, 0, pick_stream(stream));
}

int vfgs_hip_add_grain_frames_dev(void* dY, void* dU, void* dV, unsigned width, unsigned height, unsigned stride,
                                  unsigned cstride, unsigned nframes, uint64_t y_frame_pitch_bytes,
                                  uint64_t c_frame_pitch_bytes, void* stream)
{
	std::lock_guard<std::mutex> g(g_mu);
	S().gen++;
	if ((y_frame_pitch_bytes | c_frame_pitch_bytes) & 15) return fail(13, "frame pitches must be multiples of 16 bytes");
	return run_device(dY, dU, dV, dY, dU, dV, width, 0, height, 0, height, stride, cstride, nframes,
	                  y_frame_pitch_bytes, c_frame_pitch_bytes, pick_stream(stream));
}

int vfgs_hip_add_grain_frames_part_dev(void* dY, void* dU, void* dV, unsigned width, unsigned frame_height,
                                       unsigned part_y, unsigned part_height, unsigned stride, unsigned cstride,
                                       unsigned nframes, uint64_t y_frame_pitch_bytes, uint64_t c_frame_pitch_bytes,
                                       void* stream)
{
	std::lock_guard<std::mutex> g(g_mu);
	S().gen++;
	if (part_y & 15) return fail(11, "part_y must be a multiple of 16");
	if (part_y > frame_height || part_height > frame_height - part_y) return fail(12, "part exceeds the frame");     // (no 32-bit wrap)
	if ((y_frame_pitch_bytes | c_frame_pitch_bytes) & 15) return fail(13, "frame pitches must be multiples of 16 bytes");
	return run_device(dY, dU, dV, dY, dU, dV, width, 0, frame_height, part_y, part_height, stride, cstride, nframes,
	                  y_frame_pitch_bytes, c_frame_pitch_bytes, pick_stream(stream));
}

// Frames anywhere in device memory (vfgs_hip.h): validated as a whole before anything moves, then launched in chunks of
// kListFrames frames whose plane pointers travel in the kernel arguments.
static int run_frame_list(const vfgs_hip_frame_ptrs* src, const vfgs_hip_frame_ptrs* dst, unsigned nframes, unsigned width, unsigned height,
                          unsigned stride, unsigned cstride, hipStream_t stream, DstGeom dg, bool whole = true, unsigned part_y = 0, unsigned part_h = 0)
{
	if (whole) { part_y = 0; part_h = height; }
	State& s = S();
	if (int e = ensure_init(-1)) return e;
	if (nframes == 0) return 0;
	if (!src || !dst) return fail(18, "frame list: null list");
	for (unsigned f = 0; f < nframes; f++)
	{
		if (!src[f].Y || !src[f].U || !src[f].V || !dst[f].Y || !dst[f].U || !dst[f].V) return fail(18, "frame list: frame %u has a null plane", f);
		if (int e = check_geometry(s, src[f].Y, src[f].U, src[f].V, width, stride, cstride)) return e;
		if ((((uintptr_t)dst[f].Y | (uintptr_t)dst[f].U | (uintptr_t)dst[f].V) & 15)) return fail(7, "plane pointers must be 16-byte aligned");
	}
	// The frames run concurrently: destination planes that overlap (the same plane listed twice, or two that share bytes) would be
	// a race where consecutive calls are not, and so would a source plane of one frame that shares bytes with the destination of
	// ANOTHER (that frame may already have been grained when it is read).  The bytes of a plane: its rows of the stripe, whole
	// 16-sample blocks each (check_geometry).  A frame's own source and destination are either the same plane (in place) or disjoint.
	{
		const uint64_t sz = s.bs ? 2 : 1, nblk = (width + 15) / 16;
		const uint64_t crows = part_h ? (uint64_t)(part_y + part_h - 1) / s.csuby - part_y / s.csuby + 1 : 0;
		const uint64_t dsz = dg.out8 ? 1 : sz, dstr = dg.out8 ? dg.stride : stride, dcstr = dg.out8 ? dg.cstride : cstride;
		auto extent = [&](uint64_t rows, uint64_t pitch, uint64_t rowb) { return rows ? (rows - 1) * pitch + rowb : 0; };
		const uint64_t ext_s[3] = {extent(part_h, stride * sz, nblk * 16 * sz), extent(crows, cstride * sz, nblk * 16 / s.csubx * sz), 0};
		const uint64_t ext_d[3] = {extent(part_h, dstr * dsz, nblk * 16 * dsz), extent(crows, dcstr * dsz, nblk * 16 / s.csubx * dsz), 0};
		struct Span { uintptr_t lo, hi; unsigned frame; };
		std::vector<Span> d;
		d.reserve(3 * (size_t)nframes);
		for (unsigned f = 0; f < nframes; f++)
		{
			const void* pl[3] = {dst[f].Y, dst[f].U, dst[f].V};
			for (int c = 0; c < 3; c++) d.push_back({(uintptr_t)pl[c], (uintptr_t)pl[c] + ext_d[c ? 1 : 0], f});
		}
		std::sort(d.begin(), d.end(), [](const Span& a, const Span& b) { return a.lo < b.lo; });
		for (size_t i = 1; i < d.size(); i++)
			if (d[i].lo < d[i - 1].hi || d[i].lo == d[i - 1].lo)
				return fail(18, d[i].lo == d[i - 1].lo ? "frame list: a destination plane is listed twice" : "frame list: destination planes of frames %u and %u overlap",
				            d[i - 1].frame, d[i].frame);
		for (unsigned f = 0; f < nframes && src != dst; f++)
		{
			const void* pl[3] = {src[f].Y, src[f].U, src[f].V};
			const void* own[3] = {dst[f].Y, dst[f].U, dst[f].V};
			for (int c = 0; c < 3; c++)
			{
				const uintptr_t lo = (uintptr_t)pl[c], hi = lo + ext_s[c ? 1 : 0];
				// destinations are disjoint and sorted: the first one that ends behind lo is the only candidate below hi ... and its successors
				auto it = std::upper_bound(d.begin(), d.end(), lo, [](uintptr_t v, const Span& x) { return v < x.hi; });
				for (; it != d.end() && it->lo < hi; ++it)
					if (!(it->frame == f && pl[c] == own[c] && it->lo == lo))
						return fail(18, "frame list: a source plane of frame %u shares bytes with a destination plane of frame %u", f, it->frame);
			}
		}
	}
	for (unsigned f0 = 0; f0 < nframes; f0 += vfgs::kListFrames)
	{
		const unsigned n = std::min<unsigned>(vfgs::kListFrames, nframes - f0);
		vfgs::FrameTable ft{};
		for (unsigned k = 0; k < n; k++)
		{
			const vfgs_hip_frame_ptrs &a = src[f0 + k], &b = dst[f0 + k];
			ft.src[0][k] = (const uint8_t*)a.Y; ft.src[1][k] = (const uint8_t*)a.U; ft.src[2][k] = (const uint8_t*)a.V;
			ft.dst[0][k] = (uint8_t*)b.Y; ft.dst[1][k] = (uint8_t*)b.U; ft.dst[2][k] = (uint8_t*)b.V;
		}
		if (int e = run_device(src[f0].Y, src[f0].U, src[f0].V, dst[f0].Y, dst[f0].U, dst[f0].V, width, 0, height, part_y, part_h, stride, cstride,
		                       n, 0, 0, stream, dg, &ft))
			return e;
	}
	return 0;
}

int vfgs_hip_add_grain_frame_list_dev(const vfgs_hip_frame_ptrs* frames, unsigned nframes, unsigned width, unsigned height,
                                      unsigned stride, unsigned cstride, void* stream)
{
	std::lock_guard<std::mutex> g(g_mu);
	S().gen++;
	return run_frame_list(frames, frames, nframes, width, height, stride, cstride, pick_stream(stream), DstGeom());
}

int vfgs_hip_add_grain_frame_list_part_dev(const vfgs_hip_frame_ptrs* frames, unsigned nframes, unsigned width, unsigned frame_height,
                                           unsigned part_y, unsigned part_height, unsigned stride, unsigned cstride, void* stream)
{
	std::lock_guard<std::mutex> g(g_mu);
	S().gen++;
	if (part_y & 15) return fail(11, "part_y must be a multiple of 16");
	if (part_y > frame_height || part_height > frame_height - part_y) return fail(12, "part exceeds the frame");     // (no 32-bit wrap)
	return run_frame_list(frames, frames, nframes, width, frame_height, stride, cstride, pick_stream(stream), DstGeom(), false, part_y, part_height);
}

int vfgs_hip_add_grain_frame_list_copy_dev(const vfgs_hip_frame_ptrs* src, const vfgs_hip_frame_ptrs* dst, unsigned nframes, unsigned width,
                                           unsigned height, unsigned stride, unsigned cstride, void* stream)
{
	std::lock_guard<std::mutex> g(g_mu);
	S().gen++;
	return run_frame_list(src, dst, nframes, width, height, stride, cstride, pick_stream(stream), DstGeom());
}

int vfgs_hip_add_grain_frame_list_copy8_dev(const vfgs_hip_frame_ptrs* src, const vfgs_hip_frame_ptrs* dst, unsigned nframes, unsigned width,
                                            unsigned height, unsigned stride, unsigned cstride, unsigned dst_stride, unsigned dst_cstride,
                                            void* stream)
{
	std::lock_guard<std::mutex> g(g_mu);
	S().gen++;
	DstGeom dg;
	dg.out8 = true;
	dg.stride = dst_stride; dg.cstride = dst_cstride;
	return run_frame_list(src, dst, nframes, width, height, stride, cstride, pick_stream(stream), dg);
}

int vfgs_hip_add_grain_frames_host(void* const* Y, void* const* U, void* const* V, unsigned nframes, unsigned width,
                                   unsigned height, unsigned stride, unsigned cstride)
{
	std::lock_guard<std::mutex> g(g_mu);
	S().gen++;
	S().la.valid = false;
	return run_host_frames_multi(Y, U, V, nframes, width, height, stride, cstride);
}

void* vfgs_hip_host_alloc(uint64_t bytes)
{
	std::lock_guard<std::mutex> g(g_mu);
	if (ensure_init(-1)) return nullptr;
	void* p = nullptr;
	if (hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); fail(2, "hipHostMalloc(%llu) failed", (unsigned long long)bytes); return nullptr; }
	return p;
}

void vfgs_hip_host_free(void* p)
{
	std::lock_guard<std::mutex> g(g_mu);
	if (p) (void)hipHostFree(p);
}

int vfgs_hip_add_grain_copy_dev(const void* sY, const void* sU, const void* sV, void* dY, void* dU, void* dV,
                                unsigned width, unsigned frame_height, unsigned part_y, unsigned part_height,
                                unsigned stride, unsigned cstride, unsigned nframes,
                                uint64_t y_frame_pitch_bytes, uint64_t c_frame_pitch_bytes, void* stream)
{
	std::lock_guard<std::mutex> g(g_mu);
	S().gen++;
	if (part_y & 15) return fail(11, "part_y must be a multiple of 16");
	if (part_y > frame_height || part_height > frame_height - part_y) return fail(12, "part exceeds the frame");     // (no 32-bit wrap)
	if ((y_frame_pitch_bytes | c_frame_pitch_bytes) & 15) return fail(13, "frame pitches must be multiples of 16 bytes");
	return run_device(sY, sU, sV, dY, dU, dV, width, 0, frame_height, part_y, part_height, stride, cstride, nframes,
	                  y_frame_pitch_bytes, c_frame_pitch_bytes, pick_stream(stream));
}

int vfgs_hip_add_grain_copy8_dev(const void* sY, const void* sU, const void* sV, void* dY, void* dU, void* dV,
                                 unsigned width, unsigned frame_height, unsigned part_y, unsigned part_height,
                                 unsigned stride, unsigned cstride, unsigned dst_stride, unsigned dst_cstride,
                                 unsigned nframes, uint64_t y_frame_pitch_bytes, uint64_t c_frame_pitch_bytes,
                                 uint64_t dst_y_frame_pitch_bytes, uint64_t dst_c_frame_pitch_bytes, void* stream)
{
	std::lock_guard<std::mutex> g(g_mu);
	S().gen++;
	if (part_y & 15) return fail(11, "part_y must be a multiple of 16");
	if (part_y > frame_height || part_height > frame_height - part_y) return fail(12, "part exceeds the frame");     // (no 32-bit wrap)
	if ((y_frame_pitch_bytes | c_frame_pitch_bytes) & 15) return fail(13, "frame pitches must be multiples of 16 bytes");
	DstGeom dg;
	dg.out8 = true;
	dg.stride = dst_stride; dg.cstride = dst_cstride;
	dg.ypitch = dst_y_frame_pitch_bytes; dg.cpitch = dst_c_frame_pitch_bytes;
	return run_device(sY, sU, sV, dY, dU, dV, width, 0, frame_height, part_y, part_height, stride, cstride, nframes,
	                  y_frame_pitch_bytes, c_frame_pitch_bytes, pick_stream(stream), dg);
}

void vfgs_hip_get_seed_state(uint32_t out[4])
{
	std::lock_guard<std::mutex> g(g_mu);
	State& s = S();
	out[0] = s.lfsr.window(s.rnd);
	out[1] = s.lfsr.window(s.rnd_up);
	out[2] = s.lfsr.window(s.line_rnd);
	out[3] = s.lfsr.window(s.line_rnd_up);
}

int vfgs_hip_generate_patterns(const vfgs_hip_pattern_job* jobs, int n)
{
	std::lock_guard<std::mutex> g(g_mu);
	S().gen++;
	return fw_generate(jobs, n);
}

int vfgs_hip_get_pattern(int chroma, int index, signed char out[64 * 64])
{
	std::lock_guard<std::mutex> g(g_mu);
	State& s = S();
	if (chroma < 0 || chroma > 1 || index < 0 || index >= vfgs::kSlots) return fail(20, "vfgs_hip_get_pattern: bank %d slot %d", chroma, index);
	if (s.dev_origin[chroma] >> index & 1)
	{
		if (int e = ensure_init(-1)) return e;
		if (int e = fw_flush(s, s.own_stream)) return e;
		if (int e = fw_bank_stream(s, s.own_stream)) return e;
		HIP_TRY(hipStreamSynchronize(s.own_stream));
		HIP_TRY(hipMemcpy(out, s.dev_bank + (size_t)(chroma * vfgs::kSlots + index) * 4096, 4096, hipMemcpyDeviceToHost));
	}
	else
		memcpy(out, s.bank[chroma][index], 4096);
	return 0;
}

int vfgs_hip_get_luts(int c, unsigned char scale[256], unsigned char pattern[256])
{
	std::lock_guard<std::mutex> g(g_mu);
	if (c < 0 || c > 2) return fail(21, "vfgs_hip_get_luts: component %d", c);
	if (scale) memcpy(scale, S().slut[c], 256);
	if (pattern) memcpy(pattern, S().plut[c], 256);
	return 0;
}

void vfgs_hip_get_params(int out[8])
{
	std::lock_guard<std::mutex> g(g_mu);
	const State& s = S();
	const int v[8] = {s.scale_shift, s.bs, s.ymin, s.ymax, s.cmin, s.cmax, s.csubx, s.csuby};
	memcpy(out, v, sizeof v);
}

void vfgs_hip_get_stream_stats(uint64_t out[4])
{
	std::lock_guard<std::mutex> g(g_mu);
	S().lfsr.stats(out);
}

int vfgs_hip_last_launch_info(vfgs_hip_launch_info* out)
{
	std::lock_guard<std::mutex> g(g_mu);
	if (!out || !g_last_launch_valid) return -1;
	*out = g_last_launch;
	return 0;
}

int vfgs_hip_lfsr_segments(unsigned int reg, uint64_t first_bit, uint64_t step_bits, unsigned nseg, unsigned seg_words, uint32_t* out)
{
	// (host only: a generator of its own, nothing of the library's state is touched)
	if (!out || !nseg || !seg_words) return fail(19, "vfgs_hip_lfsr_segments: nothing to fill");
	static std::mutex mu;
	std::lock_guard<std::mutex> g(mu);
	static StreamCache lfsr;
	static StripeStream gen;
	lfsr.reseed(reg);
	gen.generate(lfsr, first_bit, step_bits, nseg, seg_words, out);
	return 0;
}

void vfgs_hip_get_stripe_stream_stats(uint64_t out[4])
{
	std::lock_guard<std::mutex> g(g_mu);
	uint64_t st[3];
	S().stripes.stats(st);
	out[0] = st[0]; out[1] = st[1]; out[2] = st[2]; out[3] = S().stripe_stream_last ? 1 : 0;
}

int vfgs_hip_last_error(void) { return g_err; }
const char* vfgs_hip_last_error_string(void) { return g_errstr.c_str(); }

int vfgs_hip_timer_begin(void* stream)
{
	std::lock_guard<std::mutex> g(g_mu);
	if (int e = ensure_init(-1)) return e;
	HIP_TRY(hipEventRecord(S().ev0, (hipStream_t)stream));
	return 0;
}

int vfgs_hip_timer_end(void* stream, float* elapsed_ms)
{
	std::lock_guard<std::mutex> g(g_mu);
	if (int e = ensure_init(-1)) return e;
	HIP_TRY(hipEventRecord(S().ev1, (hipStream_t)stream));
	HIP_TRY(hipEventSynchronize(S().ev1));
	HIP_TRY(hipEventElapsedTime(elapsed_ms, S().ev0, S().ev1));
	return 0;
}

int vfgs_hip_dev_build(void)
{
#ifdef VFGS_DEV_BUILD
	return 1;      // built by a developer tool with tuning / ablation knobs: results may be wrong by design
#else
	return 0;
#endif
}

int vfgs_hip_device_info(int* cu_count, int* lds_bytes_per_cu, int* clock_khz, char* name, int name_len)
{
	std::lock_guard<std::mutex> g(g_mu);
	if (int e = ensure_init(-1)) return e;
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, S().device));
	if (cu_count) *cu_count = prop.multiProcessorCount;
	if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
	if (clock_khz) *clock_khz = prop.clockRate;
	if (name && name_len > 0) { strncpy(name, prop.name, name_len - 1); name[name_len - 1] = 0; }
	return 0;
}

}  // extern "C"
