// Developer probe (round 6): what the FIRST USE of fresh HIP streams costs a process (the look-ahead of the line call creates three --
// upload / run / download -- at its first stripe, 32-54 ms into which the ring's allocations only put 3: tools/dev/pinned_alloc_probe.cpp).
//   hipcc -O2 --offload-arch=gfx950 -o tools/bin/stream_first_use_probe tools/dev/stream_first_use_probe.cpp && tools/bin/stream_first_use_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(int* p) { p[threadIdx.x + blockIdx.x * blockDim.x] = 1; }
int main()
{
	double t0 = now();
	void *d = nullptr, *h = nullptr;
	CK(hipMalloc(&d, 4 << 20)); CK(hipHostMalloc(&h, 4 << 20, hipHostMallocDefault));
	hipLaunchKernelGGL(touch, dim3(64), dim3(64), 0, 0, (int*)d); CK(hipDeviceSynchronize());
	CK(hipMemcpy(d, h, 4 << 20, hipMemcpyHostToDevice)); CK(hipMemcpy(h, d, 4 << 20, hipMemcpyDeviceToHost));
	printf("runtime up (null-stream kernel and copies done) after %.1f ms\n", now() - t0);
	hipStream_t s[3];
	hipEvent_t e[3];
	t0 = now(); for (int i = 0; i < 3; i++) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking)); const double tc = now() - t0;
	t0 = now(); for (int i = 0; i < 3; i++) CK(hipEventCreateWithFlags(&e[i], hipEventDisableTiming)); const double te = now() - t0;
	double t[6];
	for (int round = 0; round < 2; round++)
	{
		t0 = now(); CK(hipMemcpyAsync(d, h, 2 << 20, hipMemcpyHostToDevice, s[0])); CK(hipStreamSynchronize(s[0])); t[3 * round] = now() - t0;
		t0 = now(); hipLaunchKernelGGL(touch, dim3(64), dim3(64), 0, s[1], (int*)d); CK(hipStreamSynchronize(s[1])); t[3 * round + 1] = now() - t0;
		t0 = now(); CK(hipMemcpyAsync(h, d, 2 << 20, hipMemcpyDeviceToHost, s[2])); CK(hipStreamSynchronize(s[2])); t[3 * round + 2] = now() - t0;
	}
	printf("3 x hipStreamCreate %.2f ms, 3 x hipEventCreate %.2f ms; FIRST use: H2D 2 MB %.2f ms, kernel %.2f ms, D2H 2 MB %.2f ms; second use: %.2f / %.2f / %.2f ms\n",
	       tc, te, t[0], t[1], t[2], t[3], t[4], t[5]);
	return 0;
}
