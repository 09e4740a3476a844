// Developer probe: where the first milliseconds of a process that uses the library go (the unchanged CLI's short runs):
// runtime initialisation, the first grain launch (which loads the library's code object onto the device), a launch of ANOTHER
// instantiation of the same code object, steady state.
//   hipcc -O2 -o tools/bin/startup_probe tools/dev/startup_probe.cpp -Iinclude -Lversatilefilmgrain_amd -lvfgs_hip -Wl,-rpath,$PWD/versatilefilmgrain_amd
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include "vfgs_hip.h"

static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define VK(x) do { int e_ = (x); if (e_) { fprintf(stderr, "%s: %d %s\n", #x, e_, vfgs_hip_last_error_string()); return 1; } } while (0)

int main()
{
	const double t0 = now();
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n == 0) { fprintf(stderr, "no device\n"); return 1; }
	const double t1 = now();                       // hipInit + device enumeration
	VK(vfgs_hip_init(0));
	const double t2 = now();                       // the library's state: streams, first allocations
	const unsigned w = 1920, h = 1080;
	void *Y, *U, *V;
	if (hipMalloc(&Y, w * h * 2) || hipMalloc(&U, w * h / 2) || hipMalloc(&V, w * h / 2)) return 1;
	hipMemset(Y, 0, w * h * 2); hipMemset(U, 0, w * h / 2); hipMemset(V, 0, w * h / 2);
	hipDeviceSynchronize();
	const double t3 = now();                       // allocations + the runtime's own first kernels (memset)
	vfgs_set_depth(10);
	vfgs_set_chroma_subsampling(2, 2);
	std::vector<signed char> pat(64 * 64, 3);
	unsigned char sl[256], pl[256];
	memset(sl, 40, 256); memset(pl, 0, 256);
	vfgs_set_luma_pattern(0, pat.data()); vfgs_set_chroma_pattern(0, pat.data());
	for (int c = 0; c < 3; c++) { vfgs_set_scale_lut(c, sl); vfgs_set_pattern_lut(c, pl); }
	vfgs_set_scale_shift(5);
	vfgs_set_seed(1);
	const double t4 = now();
	VK(vfgs_hip_add_grain_frame_dev(Y, U, V, w, h, w, w / 2, nullptr));
	hipDeviceSynchronize();
	const double t5 = now();                       // FIRST grain launch: code object onto the device + table image + LFSR window
	VK(vfgs_hip_add_grain_frame_dev(Y, U, V, w, h, w, w / 2, nullptr));
	hipDeviceSynchronize();
	const double t6 = now();                       // same kernel again
	vfgs_set_depth(8);                             // another instantiation of the same code object (8-bit): only a new table image
	VK(vfgs_hip_add_grain_frame_dev(Y, U, V, w, h, w, w / 2, nullptr));
	hipDeviceSynchronize();
	const double t7 = now();
	VK(vfgs_hip_add_grain_frame_dev(Y, U, V, w, h, w, w / 2, nullptr));
	hipDeviceSynchronize();
	const double t8 = now();
	printf("{\"runtime_init_ms\": %.2f, \"library_init_ms\": %.2f, \"alloc_memset_ms\": %.2f, \"setters_ms\": %.3f, \"first_grain_launch_ms\": %.2f, "
	       "\"second_launch_ms\": %.3f, \"first_launch_other_instantiation_ms\": %.3f, \"its_second_launch_ms\": %.3f, \"total_ms\": %.2f}\n",
	       t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6, t8 - t7, t8 - t0);
	return 0;
}
