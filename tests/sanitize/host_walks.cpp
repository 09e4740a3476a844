// TEST INFRASTRUCTURE.  Drives the host layer of libvfgs_hip (vfgs_host.cpp + vfgs_fw_host.cpp + vfgs_cfg_host.cpp, compiled
// with a sanitizer) over tests/sanitize/hip_stub.cpp: the walks of tests/test_gpu_parity.py (line calls in order, late edits,
// repeats, skips, setters in the middle of a frame, the ring of stripes, buffers of different pitches under a promised height),
// the host stripe and frame pipelines, the replica worker threads, the device-pointer entries with batches, parts, lists and
// regions, the refusals.  The stub's "kernel" copies rows unchanged, so every walk ALSO checks what it can without an oracle:
// the bytes a call hands back are the bytes that went in (a look-ahead that served a stale or foreign stripe shows as a
// difference), and nothing outside the rows the reference touches was written (vfgs_hw.c:288-312: whole 16-sample blocks of
// the lines handed over, nothing else).
//
// usage: host_walks [walk ...]     (no argument: all of them)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vfgs_hip.h"
#include "../../include/vfgs_hip_fw.h"

extern "C" void vfgs_stub_stats(uint64_t out[3]);
extern "C" void vfgs_stub_fail_pinned_allocs(int n);
extern "C" int hipMalloc(void** p, size_t n);
extern "C" int hipFree(void* p);
extern "C" int hipMemcpy(void* d, const void* s, size_t n, int kind);
extern "C" int hipDeviceSynchronize(void);
extern "C" int hipStreamCreateWithFlags(void** s, unsigned flags);
extern "C" int hipStreamDestroy(void* s);
extern "C" int hipStreamSynchronize(void* s);

static int g_fail = 0;
#define CHECK(c)                                                                     \
	do {                                                                             \
		if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); g_fail++; } \
	} while (0)
#define OK(call)                                                                                                    \
	do {                                                                                                            \
		const int rc_ = (call);                                                                                     \
		if (rc_) { fprintf(stderr, "FAILED %s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #call, rc_, vfgs_hip_last_error_string()); g_fail++; } \
	} while (0)

static uint32_t g_lcg = 1;
static uint32_t rnd() { g_lcg = g_lcg * 1664525u + 1013904223u; return g_lcg >> 8; }

// one picture in ordinary host memory, the reference's geometry (yuv.c:54-87): one allocation per plane here, so that a
// byte written past a plane's end is a heap overflow and not a neighbour's sample
struct Frame {
	int w, h, depth, sx, sy, stride, cstride, cw, ch, sz;
	std::vector<uint8_t> Y, U, V;
	Frame(int w_, int h_, int depth_, int sx_, int sy_, int stride_ = 0) : w(w_), h(h_), depth(depth_), sx(sx_), sy(sy_)
	{
		sz = depth > 8 ? 2 : 1;
		stride = stride_ ? stride_ : ((w % 64) ? (w + 64) & ~63 : w);
		cstride = stride / sx;
		cw = w / sx; ch = (h + sy - 1) / sy;
		Y.resize((size_t)stride * h * sz); U.resize((size_t)cstride * ch * sz); V.resize((size_t)cstride * ch * sz);
		fill();
	}
	void fill()
	{
		for (auto* p : {&Y, &U, &V})
			for (size_t i = 0; i < p->size(); i += sz)
			{
				const uint32_t v = rnd() & ((1u << depth) - 1);
				(*p)[i] = (uint8_t)v;
				if (sz == 2) (*p)[i + 1] = (uint8_t)(v >> 8);
			}
	}
	uint8_t* y(int line) { return Y.data() + (size_t)line * stride * sz; }
	uint8_t* u(int line) { return U.data() + (size_t)(line / sy) * cstride * sz; }
	uint8_t* v(int line) { return V.data() + (size_t)(line / sy) * cstride * sz; }
	bool same(const Frame& o) const { return Y == o.Y && U == o.U && V == o.V; }
};

static void program(int depth, int sx, int sy, bool one_luma, bool one_chroma, unsigned seed = 12345)
{
	vfgs_hip_reset_state();
	vfgs_set_depth(depth);
	vfgs_set_chroma_subsampling(sx, sy);
	signed char P[4096];
	for (int k = 0; k < 8; k++)
	{
		for (int i = 0; i < 4096; i++) P[i] = (signed char)((int)(rnd() % 255) - 127);
		vfgs_set_luma_pattern(k, P);
		vfgs_set_chroma_pattern(k, P);
	}
	unsigned char lut[256];
	for (int c = 0; c < 3; c++)
	{
		for (int i = 0; i < 256; i++) lut[i] = (unsigned char)(rnd() % 200);
		vfgs_set_scale_lut(c, lut);
		const bool one = c == 0 ? one_luma : one_chroma;
		for (int i = 0; i < 256; i++) lut[i] = (unsigned char)(one ? 0x10 : ((i >> 5) << 4));
		vfgs_set_pattern_lut(c, lut);
	}
	vfgs_set_scale_shift(5);
	vfgs_set_seed(seed);
}

static void line_loop(Frame& f, int width = 0)
{
	for (int y = 0; y < f.h; y++) vfgs_add_grain_line(f.y(y), f.u(y), f.v(y), y, width ? width : f.w);
}

// ---- walks ---------------------------------------------------------------------------------------------------------------

static void walk_lines_in_order()
{
	const int cases[][5] = {{192, 144, 10, 2, 2}, {200, 150, 10, 2, 2}, {192, 144, 8, 1, 1}, {264, 136, 8, 2, 1}, {1920, 1080, 10, 2, 2}, {720, 576, 8, 2, 2}};
	for (const auto& c : cases)
	{
		program(c[2], c[3], c[4], c[2] == 8, true);
		Frame f(c[0], c[1], c[2], c[3], c[4]);
		const int nwalks = c[0] > 1000 ? 3 : 5;
		vfgs_hip_launch_info li0{}, li1{};
		for (int n = 0; n < nwalks; n++)
		{
			f.fill();
			Frame before = f;
			line_loop(f);
			CHECK(f.same(before));      // (the stub's grain is zero: what comes back is what went in)
			if (n == 0) vfgs_hip_last_launch_info(&li0);
		}
		// the walks behind the first one were served from stripes computed ahead, not line by line: the code under test DID run
		vfgs_hip_last_launch_info(&li1);
		CHECK(li1.launches - li0.launches < (unsigned long long)(nwalks - 1) * c[1] / 4);
	}
}

static void walk_late_edits_and_irregular_calls()
{
	program(10, 2, 2, false, true);
	Frame f(320, 176, 10, 2, 2);
	for (int fi = 0; fi < 4; fi++)
	{
		f.fill();
		Frame want = f;
		for (int y = 0; y < f.h; y++)
		{
			vfgs_add_grain_line(f.y(y), f.u(y), f.v(y), y, f.w);
			CHECK(!memcmp(f.y(y), want.y(y), (size_t)f.w * 2));
			if (y % 23 == 5 && y + 3 < f.h)        // edit a line that was (probably) read ahead already
				for (Frame* g : {&f, &want})
				{
					for (int x = 0; x < g->w; x++) ((uint16_t*)g->y(y + 2))[x] = (uint16_t)((x * 7 + y) & 1023);
					for (int x = 0; x < 16; x++) ((uint16_t*)g->v(y + 2))[x] = 7;
				}
			if (fi == 1 && y == 40) vfgs_add_grain_line(f.y(y), f.u(y), f.v(y), y, f.w);       // repeat a line
			if (fi == 1 && y == 70) y += 5;                                                    // skip ahead
			if (fi == 2 && y == 50) vfgs_set_scale_shift(3);                                   // a setter in the middle of the frame
			if (fi == 2 && y == 90)                                                            // narrower calls from here on
			{
				for (int yy = y + 1; yy < y + 4; yy++) vfgs_add_grain_line(f.y(yy), f.u(yy), f.v(yy), yy, 200);
				y += 3;
			}
			if (fi == 3 && y == 100) { unsigned char lut[256]; memset(lut, 0x20, sizeof lut); vfgs_set_pattern_lut(0, lut); }
		}
		CHECK(f.same(want));
	}
}

static void walk_ring_of_stripes()
{
	program(10, 2, 2, true, true);
	Frame f(1920, 1080, 10, 2, 2);
	const int edits[][2] = {{3, 351}, {349, 352}, {350, 704}, {351, 353}, {600, 1079}, {703, 705}, {704, 1056}, {1000, 1057}};
	for (int fi = 0; fi < 4; fi++)
	{
		f.fill();
		Frame want = f;
		for (int y = 0; y < f.h; y++)
		{
			vfgs_add_grain_line(f.y(y), f.u(y), f.v(y), y, f.w);
			CHECK(!memcmp(f.y(y), want.y(y), (size_t)f.w * 2) && !memcmp(f.u(y), want.u(y), (size_t)f.cw * 2));
			if (fi >= 1)
				for (const auto& e : edits)
					if (e[0] == y)
						for (Frame* g : {&f, &want})
						{
							for (int x = 0; x < g->w; x++) ((uint16_t*)g->y(e[1]))[x] = (uint16_t)((x + e[1]) & 1023);
							for (int x = 5; x < 40; x++) ((uint16_t*)g->u(e[1]))[x] = 513;
						}
			if (fi == 2 && (y == 351 || y == 352 || y == 704)) vfgs_add_grain_line(f.y(y), f.u(y), f.v(y), y, f.w);
			if (fi == 2 && y == 700) y += 9;
			if (fi == 3 && y == 352) vfgs_set_legal_range(1);
		}
		CHECK(f.same(want));
	}
}

static void walk_declared_frames_and_pitches()
{
	// a promised frame: working ahead from the first line of the first walk; then buffers of other pitches and sizes behind it
	program(8, 2, 2, true, true);
	Frame a(640, 360, 8, 2, 2), b(640, 360, 8, 2, 2, 1024), c(512, 200, 8, 2, 2);
	OK(vfgs_hip_declare_frame(a.Y.data(), a.U.data(), a.V.data(), a.w, a.h, a.stride, a.cstride));
	for (int n = 0; n < 3; n++)
		for (Frame* f : {&a, &b, &c, &a})
		{
			Frame before = *f;
			line_loop(*f);
			CHECK(f->same(before));
		}
	OK(vfgs_hip_declare_frame(nullptr, nullptr, nullptr, 0, 0, 0, 0));
	// the same promise from the environment (read per walk): every walk that starts at line 0
	setenv("VFGS_HIP_FRAME_HEIGHT", "360", 1);
	for (Frame* f : {&b, &a, &b})
	{
		Frame before = *f;
		line_loop(*f);
		CHECK(f->same(before));
	}
	// ... and a buffer SHORTER than the promise would be the caller's lie: not walked here.  A walk that stops early:
	for (int y = 0; y < 100; y++) vfgs_add_grain_line(a.y(y), a.u(y), a.v(y), y, a.w);
	line_loop(b);
	unsetenv("VFGS_HIP_FRAME_HEIGHT");
	vfgs_hip_line_lookahead(0);
	line_loop(a);
	vfgs_hip_line_lookahead(1);
	line_loop(a);
}

static void walk_host_stripes_and_frames()
{
	for (int depth : {8, 10})
	{
		program(depth, 2, 2, depth == 8, true);
		Frame f(704, 400, depth, 2, 2);
		Frame before = f;
		vfgs_add_grain_stripe(f.y(0), f.u(0), f.v(0), 0, f.w, f.h, f.stride, f.cstride);
		vfgs_add_grain_stripe(f.y(32), f.u(32), f.v(32), 32, f.w, 100, f.stride, f.cstride);
		vfgs_add_grain_stripe(f.y(7), f.u(7), f.v(7), 7, f.w, 1, f.stride, f.cstride);
		vfgs_add_grain_stripe(f.y(0), f.u(0), f.v(0), 0, 200, f.h, f.stride, f.cstride);        // ragged width: 13 blocks
		CHECK(f.same(before));
		// pipelined frames: pageable, then pinned; more frames than ring slots
		std::vector<Frame> fr;
		for (int i = 0; i < 7; i++) fr.emplace_back(704, 400, depth, 2, 2);
		std::vector<Frame> want = fr;
		std::vector<void*> Y, U, V;
		for (auto& x : fr) { Y.push_back(x.Y.data()); U.push_back(x.U.data()); V.push_back(x.V.data()); }
		OK(vfgs_hip_add_grain_frames_host(Y.data(), U.data(), V.data(), 7, 704, 400, fr[0].stride, fr[0].cstride));
		for (int i = 0; i < 7; i++) CHECK(fr[i].same(want[i]));
		const size_t ny = fr[0].Y.size(), nc = fr[0].U.size();
		uint8_t* pin = (uint8_t*)vfgs_hip_host_alloc((ny + 2 * nc) * 5);
		CHECK(pin != nullptr);
		if (pin)
		{
			Y.clear(); U.clear(); V.clear();
			for (int i = 0; i < 5; i++)
			{
				uint8_t* b = pin + (size_t)i * (ny + 2 * nc);
				memcpy(b, fr[i].Y.data(), ny); memcpy(b + ny, fr[i].U.data(), nc); memcpy(b + ny + nc, fr[i].V.data(), nc);
				Y.push_back(b); U.push_back(b + ny); V.push_back(b + ny + nc);
			}
			OK(vfgs_hip_add_grain_frames_host(Y.data(), U.data(), V.data(), 5, 704, 400, fr[0].stride, fr[0].cstride));
			for (int i = 0; i < 5; i++) CHECK(!memcmp(Y[i], fr[i].Y.data(), ny) && !memcmp(V[i], fr[i].V.data(), nc));
			vfgs_hip_host_free(pin);
		}
		CHECK(vfgs_hip_add_grain_frames_host(Y.data(), U.data(), V.data(), 1, 100, 400, 128, 64) != 0);     // width <= 128: refused
	}
}

static void walk_several_devices()
{
	// replicas of the state on further devices (the stub offers two; a device may be listed twice): stripes and host frames are
	// split by block rows over worker threads
	program(10, 2, 2, false, true);
	const int two[2] = {0, 1}, three[3] = {0, 1, 0}, one[1] = {0};
	Frame f(1280, 720, 10, 2, 2);
	for (int round = 0; round < 2; round++)
	{
		OK(vfgs_hip_init_devices(round ? three : two, round ? 3 : 2));
		Frame before = f;
		vfgs_add_grain_stripe(f.y(0), f.u(0), f.v(0), 0, f.w, f.h, f.stride, f.cstride);
		vfgs_add_grain_stripe(f.y(16), f.u(16), f.v(16), 16, f.w, 40, f.stride, f.cstride);
		CHECK(f.same(before));
		std::vector<Frame> fr;
		for (int i = 0; i < 5; i++) fr.emplace_back(1280, 720, 10, 2, 2);
		std::vector<Frame> want = fr;
		std::vector<void*> Y, U, V;
		for (auto& x : fr) { Y.push_back(x.Y.data()); U.push_back(x.U.data()); V.push_back(x.V.data()); }
		OK(vfgs_hip_add_grain_frames_host(Y.data(), U.data(), V.data(), 5, 1280, 720, fr[0].stride, fr[0].cstride));
		for (int i = 0; i < 5; i++) CHECK(fr[i].same(want[i]));
		vfgs_set_seed(99 + round);           // the replicas follow a new seed, new LUTs and a line walk in between
		unsigned char lut[256];
		memset(lut, 77, sizeof lut);
		vfgs_set_scale_lut(0, lut);
		line_loop(f);
		vfgs_add_grain_stripe(f.y(0), f.u(0), f.v(0), 0, f.w, f.h, f.stride, f.cstride);
		// patterns generated "on the device" travel to the replicas device to device
		fgs_sei sei;
		memset(&sei, 0, sizeof sei);
		sei.model_id = 0; sei.log2_scale_factor = 5;
		sei.comp_model_present_flag[0] = 1; sei.num_intensity_intervals[0] = 1; sei.num_model_values[0] = 3;
		sei.intensity_interval_lower_bound[0][0] = 0; sei.intensity_interval_upper_bound[0][0] = 255;
		sei.comp_model_value[0][0][0] = 100; sei.comp_model_value[0][0][1] = 8; sei.comp_model_value[0][0][2] = 8;
		vfgs_init_sei(&sei);
		vfgs_add_grain_stripe(f.y(0), f.u(0), f.v(0), 0, f.w, f.h, f.stride, f.cstride);
	}
	OK(vfgs_hip_init_devices(one, 1));
	vfgs_add_grain_stripe(f.y(0), f.u(0), f.v(0), 0, f.w, f.h, f.stride, f.cstride);
}

struct DevPlanes {
	uint8_t *Y = nullptr, *U = nullptr, *V = nullptr;
	size_t ny, nc;
	DevPlanes(size_t ny_, size_t nc_) : ny(ny_), nc(nc_) { hipMalloc((void**)&Y, ny); hipMalloc((void**)&U, nc); hipMalloc((void**)&V, nc); }
	~DevPlanes() { hipFree(Y); hipFree(U); hipFree(V); }
	DevPlanes(const DevPlanes&) = delete;
};

static void walk_device_entries()
{
	void* st = nullptr;
	hipStreamCreateWithFlags(&st, 1);
	for (int depth : {10, 8})
		for (int fmt = 0; fmt < 3; fmt++)
		{
			const int sx = fmt == 0 ? 2 : 1, sy = fmt == 0 ? 2 : (fmt == 1 ? 1 : 2);
			for (int w : {1920, 520, 8208})
			{
				const int h = w > 4000 ? 48 : 270, sz = depth > 8 ? 2 : 1, stride = (w + 63) & ~63, cstride = stride / sx, ch = (h + sy - 1) / sy;
				const int nf = 5;
				const size_t ny = (size_t)stride * h * sz, nc = (size_t)cstride * ch * sz;
				program(depth, sx, sy, w != 520, true);
				DevPlanes d(ny * nf, nc * nf);        // frames at a constant pitch: exactly as large as the batch needs
				OK(vfgs_hip_add_grain_frame_dev(d.Y, d.U, d.V, w, h, stride, cstride, st));
				OK(vfgs_hip_add_grain_frames_dev(d.Y, d.U, d.V, w, h, stride, cstride, nf, ny, nc, st));
				OK(vfgs_hip_add_grain_stripe_dev(d.Y, d.U, d.V, 0, w, 33, stride, cstride, nullptr));
				if (h > 64)
				{
					const int py = 64, ph = h - 64 - 5;
					OK(vfgs_hip_add_grain_frame_part_dev(d.Y + (size_t)py * stride * sz, d.U + (size_t)(py / sy) * cstride * sz, d.V + (size_t)(py / sy) * cstride * sz,
					                                     w, h, py, ph, stride, cstride, st));
					// stripes of a batch in an allocation of their own: exactly the part's rows of every frame
					const int crows = (py + ph - 1) / sy - py / sy + 1;
					DevPlanes p((size_t)stride * ph * sz * nf, (size_t)cstride * crows * sz * nf);
					OK(vfgs_hip_add_grain_frames_part_dev(p.Y, p.U, p.V, w, h, py, ph, stride, cstride, nf, (size_t)stride * ph * sz, (size_t)cstride * crows * sz, st));
					OK(vfgs_hip_add_grain_copy_dev(d.Y + (size_t)py * stride * sz, d.U + (size_t)(py / sy) * cstride * sz, d.V + (size_t)(py / sy) * cstride * sz, p.Y, p.U,
					                               p.V, w, h, py, ph, stride, cstride, 1, 0, 0, st));
				}
				// out of place, and narrowed to 8 bit
				{
					DevPlanes o(ny * nf, nc * nf);
					OK(vfgs_hip_add_grain_copy_dev(d.Y, d.U, d.V, o.Y, o.U, o.V, w, h, 0, h, stride, cstride, nf, ny, nc, st));
					if (depth == 10)
					{
						DevPlanes o8((size_t)stride * h * nf, (size_t)cstride * ch * nf);
						OK(vfgs_hip_add_grain_copy8_dev(d.Y, d.U, d.V, o8.Y, o8.U, o8.V, w, h, 0, h, stride, cstride, stride, cstride, nf, ny, nc, (size_t)stride * h,
						                                (size_t)cstride * ch, st));
					}
				}
				// frames anywhere: every plane an allocation of exactly its size
				{
					std::vector<DevPlanes*> fr;
					std::vector<vfgs_hip_frame_ptrs> list;
					for (int i = 0; i < 35; i++) { fr.push_back(new DevPlanes(ny, nc)); list.push_back({fr.back()->Y, fr.back()->U, fr.back()->V}); }
					OK(vfgs_hip_add_grain_frame_list_dev(list.data(), 35, w, h, stride, cstride, st));
					std::vector<vfgs_hip_frame_ptrs> dst(list.rbegin(), list.rend());
					CHECK(vfgs_hip_add_grain_frame_list_copy_dev(list.data(), dst.data(), 35, w, h, stride, cstride, st) != 0);       // sources are other frames' destinations
					list.resize(17);
					dst.assign(list.begin(), list.end());
					for (int i = 0; i < 17; i++) dst[i] = {fr[18 + i]->Y, fr[18 + i]->U, fr[18 + i]->V};
					OK(vfgs_hip_add_grain_frame_list_copy_dev(list.data(), dst.data(), 17, w, h, stride, cstride, st));
					OK(vfgs_hip_overlap_begin(st));
					for (int i = 0; i < 6; i++) OK(vfgs_hip_add_grain_frame_dev(fr[i]->Y, fr[i]->U, fr[i]->V, w, h, stride, cstride, st));
					OK(vfgs_hip_add_grain_frame_list_dev(list.data() + 6, 11, w, h, stride, cstride, st));
					OK(vfgs_hip_overlap_end(st));
					hipStreamSynchronize(st);
					for (auto* p : fr) delete p;
				}
				hipStreamSynchronize(st);
			}
		}
	// a new seed per frame queued behind launches that still read the old stream window; 240 frames through the window refills
	program(10, 2, 2, false, true);
	{
		const int w = 1920, h = 1080, stride = 1920, cstride = 960;
		DevPlanes d((size_t)stride * h * 2 * 8, (size_t)cstride * (h / 2) * 2 * 8);
		for (int i = 0; i < 12; i++)
		{
			vfgs_set_seed(1000 + i);
			OK(vfgs_hip_add_grain_frame_dev(d.Y, d.U, d.V, w, h, stride, cstride, i & 1 ? st : nullptr));
		}
		for (int i = 0; i < 30; i++) OK(vfgs_hip_add_grain_frames_dev(d.Y, d.U, d.V, w, h, stride, cstride, 8, (size_t)stride * h * 2, (size_t)cstride * (h / 2) * 2, st));
		uint64_t ss[4];
		vfgs_hip_get_stream_stats(ss);
		CHECK(ss[0] + ss[2] >= 1);
	}
	hipDeviceSynchronize();
	hipStreamDestroy(st);
}

static void walk_stripe_batches()
{
	// what one rank of a stripe split runs per step (vfgs_hip_add_grain_frames_part_dev): a few block rows of every frame of a batch.
	// The library reads the LFSR windows of such a call from segments it reaches by jumps (StripeStream) -- the stub's kernel checks
	// that every window the launch addresses lies inside the image it was handed; the image of the next call is built ahead
	void* st = nullptr;
	hipStreamCreateWithFlags(&st, 1);
	program(10, 2, 2, false, true);
	const int w = 1920, h = 1080, stride = 1920, cstride = 960, nf = 16;
	uint64_t ss0[4], ss1[4];
	vfgs_hip_get_stripe_stream_stats(ss0);
	for (int rank : {0, 3, 7})
	{
		const int rows = 9, py = rank * rows * 16, ph = (rank == 7 ? h - py : rows * 16), crows = ph / 2;
		DevPlanes p((size_t)stride * ph * 2 * nf, (size_t)cstride * crows * 2 * nf);
		for (int call = 0; call < 14; call++)      // (an image holds four of these calls: the ninth finds its segments built ahead)
		{
			OK(vfgs_hip_add_grain_frames_part_dev(p.Y, p.U, p.V, w, h, py, ph, stride, cstride, call == 9 ? nf / 2 : nf, (size_t)stride * ph * 2, (size_t)cstride * crows * 2, st));
			if (call == 10) OK(vfgs_hip_add_grain_frame_dev(p.Y, p.U, p.V, w, 128, stride, cstride, st));     // something else in between: the chain starts over
			if (call == 12) vfgs_set_seed(77 + rank);                                                       // ... and a new seed
		}
		std::vector<vfgs_hip_frame_ptrs> list;
		for (int f = 0; f < nf; f++) list.push_back({p.Y + (size_t)f * stride * ph * 2, p.U + (size_t)f * cstride * crows * 2, p.V + (size_t)f * cstride * crows * 2});
		OK(vfgs_hip_add_grain_frame_list_part_dev(list.data(), nf, w, h, py, ph, stride, cstride, st));
		hipStreamSynchronize(st);
	}
	// the generator on its own, with segments shorter than the 32 jumped words (tiny pictures) and exactly sized buffers
	for (unsigned nw : {1u, 5u, 31u, 32u, 33u, 64u, 66u, 131u, 250u, 251u, 275u, 531u})
	{
		std::vector<uint32_t> seg((size_t)7 * nw);
		OK(vfgs_hip_lfsr_segments(12345u << 1, 1000, 4099, 7, nw, seg.data()));
	}
	vfgs_hip_get_stripe_stream_stats(ss1);
	CHECK(ss1[3] == 1 && ss1[0] > ss0[0] && ss1[2] > ss0[2]);      // built in stream at the start of a chain, then switched to images built ahead
	hipDeviceSynchronize();
	hipStreamDestroy(st);
}

static void walk_lookahead_without_pinned_memory()
{
	// The look-ahead needs pinned memory for its ring.  When the host has none to give, the drop-in call -- which returns void and would
	// otherwise have to abort -- computes the line alone: stripes run from a COPY of the seed registers, so nothing had moved
	// (vfgs_host.cpp line_call).  A fresh library state, so that the ring really has to be allocated; the failure is injected into the stub
	// behind the first walk (line by line: the buffer is not proven yet, the single lines' own bounce buffers get allocated).
	vfgs_hip_shutdown();
	program(10, 2, 2, false, true);
	Frame f(1280, 720, 10, 2, 2);
	Frame before = f;
	line_loop(f);
	vfgs_hip_launch_info li0{}, li1{}, li2{};
	vfgs_hip_last_launch_info(&li0);
	vfgs_stub_fail_pinned_allocs(1 << 20);
	for (int y = 0; y < 200; y++) vfgs_add_grain_line(f.y(y), f.u(y), f.v(y), y, f.w);
	vfgs_hip_last_launch_info(&li1);
	CHECK(li1.launches - li0.launches == 200);                 // every line a launch of its own: the look-ahead never got its ring
	vfgs_stub_fail_pinned_allocs(0);
	for (int y = 200; y < f.h; y++) vfgs_add_grain_line(f.y(y), f.u(y), f.v(y), y, f.w);
	vfgs_hip_last_launch_info(&li2);
	CHECK(li2.launches - li1.launches < 100);                  // ... and gets it as soon as the memory is there
	CHECK(f.same(before));
}

static void walk_refusals_and_restart()
{
	program(10, 2, 2, false, false);
	DevPlanes d(1 << 20, 1 << 19);
	CHECK(vfgs_hip_add_grain_frame_dev(d.Y + 8, d.U, d.V, 512, 64, 512, 256, nullptr) == 7);
	CHECK(vfgs_hip_add_grain_frame_dev(d.Y, d.U, d.V, 128, 64, 128, 64, nullptr) == 5);
	CHECK(vfgs_hip_add_grain_frame_dev(d.Y, d.U, d.V, 512, 64, 500, 256, nullptr) != 0);
	CHECK(vfgs_hip_add_grain_frame_part_dev(d.Y, d.U, d.V, 512, 64, 16, 0xFFFFFFFFu, 512, 256, nullptr) == 12);
	CHECK(vfgs_hip_add_grain_frame_part_dev(d.Y, d.U, d.V, 512, 64, 8, 16, 512, 256, nullptr) == 11);
	unsigned char lut[256];
	memset(lut, 0x90, sizeof lut);           // slot 9: undefined in the reference
	vfgs_set_pattern_lut(1, lut);
	CHECK(vfgs_hip_add_grain_frame_dev(d.Y, d.U, d.V, 512, 64, 512, 256, nullptr) == 4);
	uint32_t seeds[4];
	vfgs_hip_get_seed_state(seeds);
	vfgs_hip_shutdown();
	program(8, 2, 2, true, true);
	Frame f(320, 64, 8, 2, 2);
	line_loop(f);
	vfgs_hip_shutdown();
}

int main(int argc, char** argv)
{
	struct { const char* name; void (*fn)(); } walks[] = {
		{"lines_in_order", walk_lines_in_order},
		{"late_edits", walk_late_edits_and_irregular_calls},
		{"ring_of_stripes", walk_ring_of_stripes},
		{"declared_frames", walk_declared_frames_and_pitches},
		{"host_stripes_and_frames", walk_host_stripes_and_frames},
		{"several_devices", walk_several_devices},
		{"device_entries", walk_device_entries},
		{"stripe_batches", walk_stripe_batches},
		{"lookahead_without_pinned", walk_lookahead_without_pinned_memory},
		{"refusals_and_restart", walk_refusals_and_restart},
	};
	for (const auto& w : walks)
	{
		bool want = argc < 2;
		for (int i = 1; i < argc; i++) want = want || !strcmp(argv[i], w.name);
		if (!want) continue;
		const int before = g_fail;
		w.fn();
		uint64_t st[3];
		vfgs_stub_stats(st);
		printf("%-28s %s   (stub: %llu operations queued so far, %llu of them deferred copies / launches)\n", w.name, g_fail == before ? "ok" : "FAILED",
		       (unsigned long long)st[0], (unsigned long long)st[1]);
		fflush(stdout);
	}
	vfgs_hip_shutdown();
	return g_fail ? 1 : 0;
}
