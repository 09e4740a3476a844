// Developer utility: what ONE device-resident frame per call costs a C caller (the reference's call pattern, vfgs_main.c:771-790,
// with the frames already in HBM) -- host microseconds per call and device microseconds per frame, plain and inside an overlap
// region (include/vfgs_hip.h).  tools/bench_config.py measures the same through ctypes, which adds its own ~1.5 us per call.
//   hipcc -O2 -o tools/bin/host_call_bench tools/host_call_bench.cpp -Iinclude -Lversatilefilmgrain_amd -lvfgs_hip -Wl,-rpath,$PWD/versatilefilmgrain_amd
//   tools/bin/host_call_bench [width height [calls [null]]]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "vfgs_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define VK(x) do { int e_ = (x); if (e_) { fprintf(stderr, "%s: %d %s\n", #x, e_, vfgs_hip_last_error_string()); return 1; } } while (0)

int main(int argc, char** argv)
{
	const unsigned w = argc > 2 ? atoi(argv[1]) : 1920, h = argc > 2 ? atoi(argv[2]) : 1080;
	const int calls = argc > 3 ? atoi(argv[3]) : 2000;
	VK(vfgs_hip_init(0));
	// fgs_sei-like state: one luma and one chroma pattern, every intensity selects it, 10 bit 4:2:0
	std::vector<signed char> P(64 * 64);
	unsigned r = 12345;
	for (auto& v : P) { r = r * 1664525u + 1013904223u; v = (signed char)((int)(r >> 24) % 32 - 16); }
	unsigned char slut[256], plut[256];
	for (int i = 0; i < 256; i++) { slut[i] = (unsigned char)(20 + i / 4); plut[i] = 0; }
	vfgs_set_depth(10); vfgs_set_chroma_subsampling(2, 2); vfgs_set_legal_range(1); vfgs_set_scale_shift(5); vfgs_set_seed(0xdeadbeef);
	vfgs_set_luma_pattern(0, P.data()); vfgs_set_chroma_pattern(0, P.data());
	for (int c = 0; c < 3; c++) { vfgs_set_scale_lut(c, slut); vfgs_set_pattern_lut(c, plut); }

	const size_t ybytes = (size_t)w * h * 2, cbytes = (size_t)(w / 2) * (h / 2) * 2, fbytes = ybytes + 2 * cbytes;
	const int pool = (int)(1500000000ull / fbytes) + 1 > 256 ? 256 : (int)(1500000000ull / fbytes) + 1;   // > 1.5 GB in flight: no cache hits
	char* buf;
	CK(hipMalloc((void**)&buf, (size_t)pool * fbytes));
	CK(hipMemset(buf, 0x11, (size_t)pool * fbytes));
	hipStream_t st = nullptr;      // "null" as 4th argument: the caller's stream is the null stream
	if (!(argc > 4 && argv[4][0] == 'n')) CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	auto frame = [&](int i) { char* p = buf + (size_t)(i % pool) * fbytes; return vfgs_hip_add_grain_frame_dev(p, p + ybytes, p + ybytes + cbytes, w, h, w, w / 2, st); };
	for (int region = 0; region < 2; region++)
		for (int rep = 0; rep < 3; rep++)
		{
			for (int warm = 0; warm < 2; warm++)
			{
				if (region) VK(vfgs_hip_overlap_begin(st));
				for (int i = 0; i < 64; i++) VK(frame(i));
				if (region) VK(vfgs_hip_overlap_end(st));
			}
			CK(hipStreamSynchronize(st));
			CK(hipEventRecord(e0, st));
			auto t0 = std::chrono::steady_clock::now();
			if (region) VK(vfgs_hip_overlap_begin(st));
			for (int i = 0; i < calls; i++) VK(frame(i));
			if (region) VK(vfgs_hip_overlap_end(st));
			auto t1 = std::chrono::steady_clock::now();
			CK(hipEventRecord(e1, st));
			CK(hipStreamSynchronize(st));
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			const double host_us = std::chrono::duration<double, std::micro>(t1 - t0).count() / calls, dev_us = ms * 1e3 / calls;
			printf("{\"width\": %u, \"height\": %u, \"calls\": %d, \"overlap_region\": %s, \"host_us_per_call\": %.2f, \"us_per_frame\": %.3f, \"frac_of_8TBps\": %.4f}\n",
			       w, h, calls, region ? "true" : "false", host_us, dev_us, 2.0 * fbytes / dev_us / 1e3 / 8000);
			fflush(stdout);
		}
	vfgs_hip_shutdown();
	return 0;
}
