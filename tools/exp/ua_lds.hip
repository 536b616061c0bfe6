// experiment: does gfx950 take ds_read_b64 at a 4-byte aligned (not 8-byte aligned) LDS address?  hipcc --offload-arch=gfx950 ua_lds.hip -o ua_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(unsigned long long *out, int shift)
{
	__shared__ unsigned t[1024];
	for (int i = threadIdx.x; i < 1024; i += 64) t[i] = 0x1000u + i;
	__syncthreads();
	unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned *)t + 12u * threadIdx.x + 4u * shift;
	unsigned long long v;
	asm volatile("ds_read_b64 %0, %1 offset:48\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
	out[threadIdx.x] = v;
}
int main()
{
	unsigned long long *d, h[64];
	hipMalloc(&d, sizeof(h));
	int bad = 0;
	for (int shift = 0; shift < 2; shift++) {
		hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, shift);
		hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
		for (int i = 0; i < 64; i++) {
			const unsigned idx = 3u * i + shift + 12u;
			const unsigned long long want = (unsigned long long)(0x1000u + idx) | ((unsigned long long)(0x1000u + idx + 1) << 32);
			if (h[i] != want) { if (bad < 5) printf("shift %d lane %d: got %llx want %llx\n", shift, i, h[i], want); bad++; }
		}
	}
	printf("unaligned ds_read_b64: %s (%d mismatches)\n", bad ? "BROKEN" : "ok", bad);
	return bad != 0;
}
