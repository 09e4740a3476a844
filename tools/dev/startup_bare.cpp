// Developer probe: the least a process pays for touching the GPU at all (no library): runtime init, one allocation, one runtime kernel (memset), sync.
#include <hip/hip_runtime.h>
#include <cstdio>
int main()
{
	void* p;
	if (hipMalloc(&p, 1 << 20) != hipSuccess) return 1;
	hipMemset(p, 0, 1 << 20);
	hipDeviceSynchronize();
	return 0;
}
