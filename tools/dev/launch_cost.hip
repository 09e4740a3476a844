// Developer probe: what does the HIP runtime itself charge the calling thread per kernel launch (a 400-byte argument block, as the
// grain kernels take)?  tools/host_call_bench.cpp minus this is the library's own host work per call.
//   hipcc -O2 --offload-arch=gfx950 -o tools/bin/launch_cost tools/dev/launch_cost.hip && tools/bin/launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Args { unsigned char b[400]; };
__global__ void k(const Args a, int* out) { if (a.b[0] == 255 && out) out[0] = 1; }
int main()
{
	hipStream_t st[2];
	for (auto& s : st) if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return 1;
	Args a{};
	for (int two = 0; two < 2; two++)
		for (int rep = 0; rep < 3; rep++)
		{
			const int n = 20000;
			for (int i = 0; i < 2000; i++) hipLaunchKernelGGL(k, dim3(272), dim3(256), 0, st[two ? i & 1 : 0], a, nullptr);
			(void)hipDeviceSynchronize();
			auto t0 = std::chrono::steady_clock::now();
			for (int i = 0; i < n; i++) hipLaunchKernelGGL(k, dim3(272), dim3(256), 0, st[two ? i & 1 : 0], a, nullptr);
			auto t1 = std::chrono::steady_clock::now();
			(void)hipDeviceSynchronize();
			auto t2 = std::chrono::steady_clock::now();
			printf("%s: host %.2f us per launch, %.2f us per launch incl. drain\n", two ? "alternating two streams" : "one stream",
			       std::chrono::duration<double, std::micro>(t1 - t0).count() / n, std::chrono::duration<double, std::micro>(t2 - t0).count() / n);
		}
	return 0;
}
