// Does hipStreamWaitValue32 hold a stream until a RUNNING kernel on another stream stores the value, and how soon after the
// store does the held stream's next kernel start?  (experiment for the receive loop: one chain launch instead of four)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_long(unsigned *flag, unsigned long long *t, int n_marks, unsigned long long gap_ticks)
{
	// one wave: every gap_ticks of the 100 MHz counter, store the next value
	unsigned long long t0 = wall_clock64();
	for (int k = 1; k <= n_marks; k++) {
		while (wall_clock64() - t0 < gap_ticks * k)
			__builtin_amdgcn_s_sleep(8);
		__threadfence_system();
		t[k] = wall_clock64();
		__hip_atomic_store(flag, (unsigned)k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}
__global__ void k_mark(unsigned long long *t, int slot) { if (threadIdx.x == 0) t[slot] = wall_clock64(); }

int main()
{
	int can = 0;
	CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
	printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
	unsigned *flag = nullptr;
	CK(hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory));
	CK(hipMemset(flag, 0, 8));
	unsigned long long *t = nullptr;
	CK(hipMalloc(&t, 64 * 8));
	CK(hipMemset(t, 0, 64 * 8));
	hipStream_t a, b;
	CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
	CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
	const int n = 4;
	hipLaunchKernelGGL(k_long, dim3(1), dim3(64), 0, a, flag, t, n, 20000ull);       // a mark every 200 us
	for (int k = 1; k <= n; k++) {
		CK(hipStreamWaitValue32(b, flag, (unsigned)k, hipStreamWaitValueGte, 0xffffffffu));
		hipLaunchKernelGGL(k_mark, dim3(1), dim3(64), 0, b, t, 16 + k);
	}
	CK(hipStreamSynchronize(a));
	CK(hipStreamSynchronize(b));
	unsigned long long h[64];
	CK(hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost));
	for (int k = 1; k <= n; k++)
		printf("value %d stored at %8.1f us, waiting stream's kernel ran %6.1f us later\n", k, (double)(h[k] - h[1]) / 100.0,
		       ((double)h[16 + k] - (double)h[k]) / 100.0);
	return 0;
}
