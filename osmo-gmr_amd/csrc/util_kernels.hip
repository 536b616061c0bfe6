// util_kernels.hip -- diagnostics of the measurement (bench.py), not of the receive path.
//
// gmr1_hip_clock_probe_dev: the shader clock the GPU is holding RIGHT NOW, measured on the device -- one wave reads the
// shader-clock counter (s_memtime) and the constant-rate wall counter (s_memrealtime) around a busy wait of `micros`
// microseconds.  Launched on the stream of a timed region, right behind its last step, it says what clock those steps ran
// at: the pool's boxes differ by a few per cent in the clock they hold under the same load, and a bench line that
// carries the number explains such a gap in the record instead of in prose.
#include "capi_common.h"
#include "../../include/gmr1_hip.h"

namespace gmr1 {

__global__ __launch_bounds__(64) void k_clock_probe(unsigned long long wall_ticks, unsigned long long *out)
{
	const unsigned long long w0 = wall_clock64();
	const unsigned long long c0 = __builtin_readcyclecounter();
	unsigned long long w1 = w0;
	// (bounded: a wall counter that does not advance must not hold the wave)
	for (int i = 0; i < (1 << 22) && w1 - w0 < wall_ticks; i++) {
		__builtin_amdgcn_s_sleep(8);
		w1 = wall_clock64();
	}
	const unsigned long long c1 = __builtin_readcyclecounter();
	if (threadIdx.x == 0) {
		out[0] = c1 - c0;
		out[1] = w1 - w0;
	}
}

}  // namespace gmr1

extern "C" int gmr1_hip_clock_probe_dev(void *stream_, int micros, double *core_mhz, double *wall_mhz)
{
	using namespace gmr1;
	if (!core_mhz || micros < 1 || micros > 100000)
		return fail(-EINVAL, "gmr1_hip_clock_probe_dev: micros %d (1 ... 100000), core_mhz %p", micros, (void *)core_mhz);
	int dev = 0, wall_khz = 0;
	HIP_TRY(hipGetDevice(&dev));
	HIP_TRY(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, dev));
	if (wall_khz <= 0)
		return fail(-EIO, "gmr1_hip_clock_probe_dev: the device reports no wall-clock rate");
	static thread_local unsigned long long *d_out = nullptr;
	static thread_local int d_dev = -1;
	if (!d_out || d_dev != dev) {
		HIP_TRY(hipMalloc(&d_out, 2 * sizeof(unsigned long long)));
		d_dev = dev;
	}
	hipStream_t st = static_cast<hipStream_t>(stream_);
	const unsigned long long ticks = (unsigned long long)micros * (unsigned long long)wall_khz / 1000ull;
	hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, st, ticks, d_out);
	HIP_TRY(hipGetLastError());
	unsigned long long h[2] = {0, 0};
	HIP_TRY(hipMemcpyAsync(h, d_out, sizeof(h), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	if (!h[1])
		return fail(-EIO, "gmr1_hip_clock_probe_dev: the wall counter did not advance");
	*core_mhz = (double)h[0] / ((double)h[1] / ((double)wall_khz / 1000.0));
	if (wall_mhz)
		*wall_mhz = (double)wall_khz / 1000.0;
	return 0;
}
