// Developer probe (round 6): what the look-ahead ring's allocations cost a fresh process: hipHostMalloc / hipMalloc by size and count.
//   hipcc -O2 -o tools/bin/pinned_alloc_probe tools/dev/pinned_alloc_probe.cpp && tools/bin/pinned_alloc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
	const int mode = argc > 1 ? atoi(argv[1]) : 0;
	double t0 = now();
	hipFree(0);
	void* d0; hipMalloc(&d0, 1 << 20); hipDeviceSynchronize();
	printf("mode %d: runtime up after %.1f ms\n", mode, now() - t0);
	std::vector<void*> p;
	t0 = now();
	if (mode == 0) { void* a; hipHostMalloc(&a, 12u << 20, hipHostMallocDefault); p.push_back(a); }
	if (mode == 1) for (int i = 0; i < 18; i++) { void* a; hipHostMalloc(&a, (12u << 20) / 18, hipHostMallocDefault); p.push_back(a); }
	if (mode == 2) { void* a; hipHostMalloc(&a, 12u << 20, hipHostMallocNonCoherent); p.push_back(a); }
	if (mode == 3) { void* a; hipHostMalloc(&a, 12u << 20, hipHostMallocNumaUser); p.push_back(a); }
	if (mode == 4) for (int i = 0; i < 3; i++) { void* a; hipHostMalloc(&a, 4u << 20, hipHostMallocDefault); p.push_back(a); }
	if (mode == 5) { void* a = aligned_alloc(4096, 12u << 20); memset(a, 0, 12u << 20); double t1 = now(); hipHostRegister(a, 12u << 20, hipHostRegisterDefault); printf("   (memset %.1f ms)\n", t1 - t0); t0 = t1; }
	const double t_pin = now() - t0;
	t0 = now();
	void* d; hipMalloc(&d, 6u << 20);
	const double t_dev1 = now() - t0;
	t0 = now();
	for (int i = 0; i < 9; i++) { void* a; hipMalloc(&a, (6u << 20) / 9); }
	const double t_dev9 = now() - t0;
	t0 = now();
	for (void* a : p) memset(a, 1, mode == 1 ? (12u << 20) / 18 : mode == 4 ? 4u << 20 : 12u << 20);
	printf("mode %d: pinned %.2f ms, first touch of it %.2f ms; hipMalloc 6 MB %.2f ms, 9 x 0.67 MB %.2f ms\n", mode, t_pin, now() - t0, t_dev1, t_dev9);
	return 0;
}
