// Experiment for the host-memory entry points: how fast can a frame that lives in ordinary
// (pageable) host memory make the round trip host -> HBM -> host, and what does pinning the
// caller's buffer in place (hipHostRegister) and splitting the frame over two streams buy?
//   hipcc -O2 --offload-arch=gfx950 tools/h2d_pipeline.hip -o tools/bin/h2d_pipeline
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(uint4* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { uint4 v = p[i]; v.x ^= 1; p[i] = v; } }

__global__ void copyk(const uint4* a, uint4* b, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { uint4 v = a[i]; v.x ^= 1; b[i] = v; } }

int main()
{
	const size_t bytes = 7680ull * 4320 * 3;       // one 4320p 10-bit 4:2:0 frame, all planes
	char* host = (char*)aligned_alloc(4096, bytes);
	memset(host, 1, bytes);
	char* dev; CK(hipMalloc(&dev, bytes));
	hipStream_t s[2]; CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
	auto kernel = [&](size_t off, size_t n, hipStream_t st) { hipLaunchKernelGGL(touch, dim3((n / 16 + 255) / 256), dim3(256), 0, st, (uint4*)(dev + off), n / 16); };
	auto seq = [&](const char* what) {
		double best = 1e9;
		for (int r = 0; r < 5; r++) {
			double t0 = now();
			CK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s[0]));
			kernel(0, bytes, s[0]);
			CK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s[0]));
			CK(hipStreamSynchronize(s[0]));
			best = std::min(best, now() - t0);
		}
		printf("%-44s %.3f ms  (%.1f GB/s each way if serial)\n", what, best * 1e3, 2 * bytes / best / 1e9);
	};
	auto piped = [&](const char* what, int parts) {
		double best = 1e9;
		const size_t chunk = ((bytes / parts) + 4095) & ~4095ull;
		for (int r = 0; r < 5; r++) {
			double t0 = now();
			int k = 0;
			for (size_t off = 0; off < bytes; off += chunk, k++) {
				const size_t n = std::min(chunk, bytes - off);
				hipStream_t st = s[k & 1];
				CK(hipMemcpyAsync(dev + off, host + off, n, hipMemcpyHostToDevice, st));
				kernel(off, n, st);
				CK(hipMemcpyAsync(host + off, dev + off, n, hipMemcpyDeviceToHost, st));
			}
			CK(hipStreamSynchronize(s[0])); CK(hipStreamSynchronize(s[1]));
			best = std::min(best, now() - t0);
		}
		printf("%-44s %.3f ms\n", what, best * 1e3);
	};
	seq("pageable, one stream");
	piped("pageable, 2 streams x 8 parts", 8);
	double t0 = now();
	CK(hipHostRegister(host, bytes, hipHostRegisterDefault));
	printf("%-44s %.3f ms\n", "hipHostRegister(149 MB)", (now() - t0) * 1e3);
	seq("registered, one stream");
	piped("registered, 2 streams x 2 parts", 2);
	piped("registered, 2 streams x 4 parts", 4);
	piped("registered, 2 streams x 8 parts", 8);
	piped("registered, 2 streams x 16 parts", 16);
	{
		// zero-copy: the kernel works in place on the caller's (registered) memory over PCIe
		char* hd = nullptr;
		CK(hipHostGetDevicePointer((void**)&hd, host, 0));
		for (int grid_div : {1, 4}) {
			double best = 1e9;
			for (int r = 0; r < 5; r++) {
				double t0 = now();
				const size_t n = bytes / 16;
				hipLaunchKernelGGL(touch, dim3((n + 255) / 256), dim3(256), 0, s[0], (uint4*)hd, n);
				CK(hipStreamSynchronize(s[0]));
				best = std::min(best, now() - t0);
			}
			printf("%-44s %.3f ms\n", "zero-copy in-place kernel on host memory", best * 1e3);
			(void)grid_div;
		}
		// H2D by DMA, kernel writes its result straight to host memory (out of place)
		double best = 1e9;
		const int parts = 8;
		const size_t chunk = ((bytes / parts) + 4095) & ~4095ull;
		for (int r = 0; r < 5; r++) {
			double t0 = now();
			int k = 0;
			for (size_t off = 0; off < bytes; off += chunk, k++) {
				const size_t nb = std::min(chunk, bytes - off);
				hipStream_t st = s[k & 1];
				CK(hipMemcpyAsync(dev + off, host + off, nb, hipMemcpyHostToDevice, st));
				hipLaunchKernelGGL(copyk, dim3((nb / 16 + 255) / 256), dim3(256), 0, st, (const uint4*)(dev + off), (uint4*)(hd + off), nb / 16);
			}
			CK(hipStreamSynchronize(s[0])); CK(hipStreamSynchronize(s[1]));
			best = std::min(best, now() - t0);
		}
		printf("%-44s %.3f ms\n", "DMA in + kernel stores to host, 2x8 parts", best * 1e3);
	}
	t0 = now();
	CK(hipHostUnregister(host));
	printf("%-44s %.3f ms\n", "hipHostUnregister", (now() - t0) * 1e3);
	// unaligned sub-range of a malloc'ed block
	t0 = now();
	hipError_t e = hipHostRegister(host + 100, bytes - 200, hipHostRegisterDefault);
	printf("register unaligned range: %s (%.3f ms)\n", hipGetErrorString(e), (now() - t0) * 1e3);
	if (e == hipSuccess) { seq("registered (unaligned), one stream"); CK(hipHostUnregister(host + 100)); }
	return 0;
}
