// Developer utility: what ONE device-resident frame per call costs a C caller (the reference's call pattern, vfgs_main.c:771-790,
// with the frames already in HBM) -- host microseconds per call and device microseconds per frame, plain and inside an overlap
// region (include/vfgs_hip.h).  tools/bench_config.py measures the same through ctypes, which adds its own ~1.5 us per call.
//   hipcc -O2 -o tools/bin/host_call_bench tools/host_call_bench.cpp -Iinclude -Lversatilefilmgrain_amd -lvfgs_hip -Wl,-rpath,$PWD/versatilefilmgrain_amd
//   tools/bin/host_call_bench [width height [calls [null]]]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "vfgs_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define VK(x) do { int e_ = (x); if (e_) { fprintf(stderr, "%s: %d %s\n", #x, e_, vfgs_hip_last_error_string()); return 1; } } while (0)

// `host_call_bench parts`: host microseconds per call at the call shapes one rank of bench.py --gpus N makes (its stripe of N x 8 frames
// per call, weak scaling; of 8 frames, strong), 4320p, from a C caller -- what tools/host_overhead.py measures through ctypes.  The
// stream bytes a call uploads: its stripe's LFSR windows reached by jumps (vfgs_hip_get_stripe_stream_stats), or, with
// VFGS_HIP_STRIPE_JUMP=0, whole frames' worth of the contiguous window.
static int parts()
{
	const unsigned W = 7680, H = 4320, nbr = (H + 15) / 16;
	VK(vfgs_hip_init(0));
	std::vector<signed char> P(64 * 64, 3);
	unsigned char slut[256], plut[256];
	for (int i = 0; i < 256; i++) { slut[i] = (unsigned char)(20 + i / 4); plut[i] = (unsigned char)((i >> 5) << 4); }
	vfgs_set_depth(10); vfgs_set_chroma_subsampling(2, 2); vfgs_set_scale_shift(5); vfgs_set_seed(12345);
	for (int k = 0; k < 8; k++) { vfgs_set_luma_pattern(k, P.data()); vfgs_set_chroma_pattern(k, P.data()); }
	for (int c = 0; c < 3; c++) { vfgs_set_scale_lut(c, slut); vfgs_set_pattern_lut(c, plut); }
	hipStream_t st;
	CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	for (int strong = 0; strong < 2; strong++)
		for (unsigned ranks : {1u, 2u, 4u, 8u})
		{
			const unsigned rows = (nbr + ranks - 1) / ranks, part_y = (ranks > 1 ? 1 : 0) * rows * 16, part_h = std::min(rows * 16, H - part_y);     // (rank 1's stripe: not the frame's first rows)
			const unsigned frames = strong ? 8 : 8 * ranks;
			char *Y, *U, *V;
			CK(hipMalloc((void**)&Y, (size_t)W * part_h * 2)); CK(hipMalloc((void**)&U, (size_t)(W / 2) * (part_h / 2) * 2)); CK(hipMalloc((void**)&V, (size_t)(W / 2) * (part_h / 2) * 2));
			auto once = [&] { return vfgs_hip_add_grain_frames_part_dev(Y, U, V, W, H, part_y, part_h, W, W / 2, frames, 0, 0, st); };     // (all frames alias one stripe: the content is irrelevant here)
			for (int i = 0; i < 4; i++) VK(once());
			CK(hipStreamSynchronize(st));
			std::vector<double> host, total;
			uint64_t s0[4], s1[4], w0[4], w1[4];
			vfgs_hip_get_stripe_stream_stats(s0); vfgs_hip_get_stream_stats(w0);
			const int calls = 3, reps = 12;      // short bursts: the host runs ahead of the GPU, the slot rings apply back-pressure on longer ones
			for (int rep = 0; rep < reps; rep++)
			{
				auto t0 = std::chrono::steady_clock::now();
				for (int i = 0; i < calls; i++) VK(once());
				auto t1 = std::chrono::steady_clock::now();
				CK(hipStreamSynchronize(st));
				auto t2 = std::chrono::steady_clock::now();
				host.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count() / calls);
				total.push_back(std::chrono::duration<double, std::micro>(t2 - t0).count() / calls);
			}
			vfgs_hip_get_stripe_stream_stats(s1); vfgs_hip_get_stream_stats(w1);
			std::sort(host.begin(), host.end()); std::sort(total.begin(), total.end());
			const unsigned nblk = (W + 15) / 16, seg_words = (32 + nblk + (part_h / 16) * nblk + 64 + 31) / 32 + 1;
			const double jump_bytes = s1[3] ? 4.0 * seg_words * frames : 0, window_bytes = double((w1[0] - w0[0]) + (w1[1] - w0[1])) * w1[3] * 4 / (calls * reps);
			printf("{\"scaling\": \"%s\", \"ranks\": %u, \"frames_per_call\": %u, \"lines\": %u, \"host_us_per_call\": %.1f, \"host_plus_gpu_us_per_call\": %.1f, "
			       "\"stripe_jump\": %s, \"stream_bytes_uploaded_per_call\": %.0f, \"images_built_ahead\": %llu, \"images_built_in_stream\": %llu}\n",
			       strong ? "strong" : "weak", ranks, frames, part_h, host[host.size() / 2], total[total.size() / 2], s1[3] ? "true" : "false",
			       s1[3] ? jump_bytes : window_bytes, (unsigned long long)(s1[1] - s0[1]), (unsigned long long)(s1[0] - s0[0]));
			fflush(stdout);
			CK(hipFree(Y)); CK(hipFree(U)); CK(hipFree(V));
		}
	vfgs_hip_shutdown();
	return 0;
}

int main(int argc, char** argv)
{
	if (argc > 1 && !strcmp(argv[1], "parts")) return parts();
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
