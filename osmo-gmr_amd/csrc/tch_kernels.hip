// tch_kernels.hip -- the two pieces the TCH3 follow-up of gmr1_rx needs besides the normal-burst
// path (gfx950):
//
//   k_dkab : DKAB (dual keep-alive burst) demodulator, reference src/sdr/dkab.c:57-224
//            (gmr1_dkab_demod).  One burst per wavefront.  The reference normalises the window,
//            counter-rotates every sample by (freq_shift - pi/4)/sps, slides a 2 x 5-symbol energy
//            window, refines the peak by a parabola, validates it by the peak / valley energy ratio
//            and takes the eight differential soft bits arg(x[o] conj(x[o + sps])).  A rotation
//            changes neither an energy nor -- beyond the constant -(freq_shift - pi/4) -- a phase
//            difference, so here only the 16 samples of the soft bits are ever touched as complex
//            numbers; everything else runs on |x - mean|^2 / sigma^2 in LDS.
//   k_a5   : GMR-1 A5/1 keystream, reference src/l1/a5.c:56-282 (gmr1_a5 / gmr1_a5_1).  Bit-serial
//            and tiny (64 + 250 + 2 nbits clocks of four LFSRs): one (key, frame number) per LANE.
#include "gmr1_dev.h"

namespace gmr1 {

#define WSYNC()                                                   \
	do {                                                          \
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");    \
		__builtin_amdgcn_wave_barrier();                          \
	} while (0)

static constexpr float kPif = 3.14159265358979323846f;
static constexpr int kDkabSyms = 39 * 3;          // sdr/dkab.h:37
static constexpr float kDkabRatio = 10.0f;        // dkab.c:47

__device__ __forceinline__ float wsum(float v)
{
#pragma unroll
	for (int o = 32; o >= 1; o >>= 1)
		v += __shfl_xor(v, o, 64);
	return v;
}

__device__ __forceinline__ float arg_fast(float y, float x)
{
	// same octant-folded minimax polynomial as the demodulator's atan2 (rx_kernels.hip)
	const float ax = fabsf(x), ay = fabsf(y);
	const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
	const bool big = mn > 0.41421356237f * mx;
	const float num = big ? (mn - mx) : mn;
	const float den = big ? (mn + mx) : mx;
	const float t = num * __builtin_amdgcn_rcpf(den);
	const float z = t * t;
	float p = fmaf(z, 8.05374449538e-2f, -1.38776856032e-1f);
	p = fmaf(p, z, 1.99777106478e-1f);
	p = fmaf(p, z, -3.33329491539e-1f);
	float a = fmaf(p * z, t, t);
	a += big ? 0.785398163397448309f : 0.0f;
	a = (ay > ax) ? (1.57079632679489662f - a) : a;
	a = (x < 0.0f) ? (kPif - a) : a;
	a = (mx == 0.0f) ? 0.0f : a;
	return (y < 0.0f) ? -a : a;
}

__global__ __launch_bounds__(64) void k_dkab(DkabArgs a)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	float *e = reinterpret_cast<float *>(lds_raw);                 // [in_len] normalised sample energies
	float *pw = e + ((a.in_len + 3) & ~3);                         // [w] window energies
	const int lane = threadIdx.x;
	const int g = blockIdx.x;
	const int sps = a.sps, len = a.in_len;
	const float2 *__restrict__ in = a.iq + a.offset[g];
	const int p = a.p[g];
	const float fsh = a.freq_shift ? a.freq_shift[g] : 0.0f;

	const int w = len - kDkabSyms * sps + 1;                       // dkab.c:68-70 (checked on the host)

	// ---- osmo_cxvec_sig_normalize: mean, sigma ------------------------------------------------
	float sr = 0.f, si = 0.f;
	for (int i = lane; i < len; i += 64) {
		const float2 v = in[i];
		sr += v.x;
		si += v.y;
	}
	const float avr = wsum(sr) / (float)len, avi = wsum(si) / (float)len;
	float acc = 0.f;
	for (int i = lane; i < len; i += 64) {
		const float2 v = in[i];
		const float dx = v.x - avr, dy = v.y - avi;
		const float ee = fmaf(dx, dx, dy * dy);
		e[i] = ee;
		acc += ee;
	}
	float stddev = sqrtf(wsum(acc) / (float)len);
	if (stddev == 0.0f)
		stddev = 1.0f;
	const float inv2 = 1.0f / (stddev * stddev);
	WSYNC();
	for (int i = lane; i < len; i += 64)
		e[i] *= inv2;
	WSYNC();
	auto eat = [&](int i) -> float { return (i >= 0 && i < len) ? e[i] : 0.0f; };   // decision D7 of the oracle

	// ---- _gmr1_dkab_find_toa, dkab.c:57-151 ------------------------------------------------------
	const int ofs0 = sps * (2 + p), ofs1 = sps * (2 + p + 59), d = sps * 5;
	for (int j = lane; j < w; j += 64) {
		float s = 0.f;
		for (int i = 0; i < d; i++)
			s += eat(ofs0 + j + i) + eat(ofs1 + j + i);
		pw[j] = s;
	}
	WSYNC();
	// first maximum (strict '>' while scanning upwards)
	unsigned long long key = 0;
	for (int j = lane; j < w; j += 64) {
		const unsigned long long kk = ((unsigned long long)__builtin_bit_cast(uint32_t, pw[j]) << 32) | (uint32_t)(~j);
		key = kk > key ? kk : key;
	}
#pragma unroll
	for (int o = 32; o >= 1; o >>= 1) {
		const unsigned long long ot = __shfl_xor(key, o, 64);
		key = ot > key ? ot : key;
	}
	int mi = (int)(~(uint32_t)key);
	if (mi < 0 || mi >= w)
		mi = 0;
	float toa = (float)mi;
	if (mi > 0 && mi < w - 1)
		toa += 0.5f * (-pw[mi - 1] + pw[mi + 1]) / (-pw[mi - 1] + 2.0f * pw[mi] - pw[mi + 1]);
	toa += ((float)(sps - 1)) / 2.0f;
	const int toa_i = (int)roundf(toa);

	float pk = 0.f;
	for (int i = lane; i < d; i += 64)
		pk += eat(toa_i + ofs0 + i) + eat(toa_i + ofs1 + i);
	const float egy_peak = wsum(pk) / (float)(2 * d);
	const int l_valley = ofs1 - ofs0 - d;
	float vl = 0.f;
	for (int i = lane; i < l_valley; i += 64)
		vl += eat(toa_i + ofs0 + d + i);
	const float egy_valley = wsum(vl) / (float)l_valley;
	const int rv = ((egy_peak / egy_valley) > kDkabRatio) ? 0 : 1;

	// ---- _gmr1_dkab_soft_bits, dkab.c:161-181 ------------------------------------------------------
	if (lane < 8 && a.ebits) {
		int8_t out = 0;
		if (!rv) {
			const int o = toa_i + (lane >> 2 ? ofs1 : ofs0) + sps * (lane & 3);
			float2 x0 = make_float2(0.f, 0.f), x1 = make_float2(0.f, 0.f);
			if (o >= 0 && o < len) { x0 = in[o]; x0.x -= avr; x0.y -= avi; }
			if (o + sps >= 0 && o + sps < len) { x1 = in[o + sps]; x1.x -= avr; x1.y -= avi; }
			// x0 conj(x1), then the rotation the reference applied to every sample: -(fsh - pi/4) per symbol
			const float re = x0.x * x1.x + x0.y * x1.y;
			const float im = x0.y * x1.x - x0.x * x1.y;
			float pd = 0.0f;
			if (re != 0.0f || im != 0.0f) {
				pd = arg_fast(im, re) - (fsh - kPif / 4);
				pd -= 2.0f * kPif * rintf(pd / (2.0f * kPif));
			}
			out = (int8_t)roundf((0.5f - (fabsf(pd) / kPif)) * 254.0f);
		}
		a.ebits[(size_t)g * 8 + lane] = out;
	}
	if (lane == 0) {
		a.rv[g] = rv;
		if (a.toa) a.toa[g] = toa;
	}
}

hipError_t launch_dkab(const DkabArgs &a, hipStream_t stream)
{
	if (a.n <= 0)
		return hipSuccess;
	const int w = a.in_len - kDkabSyms * a.sps + 1;
	const size_t lds = (size_t)(((a.in_len + 3) & ~3) + ((w + 3) & ~3)) * 4;
	hipLaunchKernelGGL(k_dkab, dim3(a.n), dim3(64), lds, stream, a);
	return hipGetLastError();
}

// ---------------------------------------------------------------------------
// A5/1 (GMR-1 variant)
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t a5_clock(uint32_t r, uint32_t mask, uint32_t taps)
{
	return ((r << 1) & mask) | (uint32_t)(__popc(r & taps) & 1);
}

struct A5State { uint32_t r0, r1, r2, r3; };

__device__ __forceinline__ void a5_step(A5State &s)      // a5.c:163-185
{
	const uint32_t c0 = (s.r3 >> 15) & 1u, c1 = (s.r3 >> 6) & 1u, c2 = (s.r3 >> 1) & 1u;
	const uint32_t m = (c0 + c1 + c2) >= 2u ? 1u : 0u;
	if (c0 == m) s.r0 = a5_clock(s.r0, (1u << 19) - 1, 0x072000u);
	if (c1 == m) s.r1 = a5_clock(s.r1, (1u << 22) - 1, 0x311000u);
	if (c2 == m) s.r2 = a5_clock(s.r2, (1u << 23) - 1, 0x660000u);
	s.r3 = a5_clock(s.r3, (1u << 17) - 1, 0x013100u);
}

__device__ __forceinline__ uint32_t a5_maj(uint32_t r, int x, int y, int z)
{
	return (((r >> x) & 1u) + ((r >> y) & 1u) + ((r >> z) & 1u)) >= 2u ? 1u : 0u;
}

__device__ __forceinline__ uint32_t a5_out(const A5State &s)    // a5.c:191-216
{
	const uint32_t m0 = a5_maj(s.r0, 1, 6, 15) ^ ((s.r0 >> 11) & 1u);
	const uint32_t m1 = a5_maj(s.r1, 3, 8, 14) ^ ((s.r1 >> 1) & 1u);
	const uint32_t m2 = a5_maj(s.r2, 4, 15, 19) ^ (s.r2 & 1u);
	return m0 ^ m1 ^ m2;
}

__global__ __launch_bounds__(64) void k_a5(A5Args a)
{
	const int g = blockIdx.x * 64 + threadIdx.x;
	if (g >= a.n)
		return;
	uint8_t *dl = a.dl ? a.dl + (size_t)g * a.nbits : nullptr;
	uint8_t *ul = a.ul ? a.ul + (size_t)g * a.nbits : nullptr;
	if (a.alg == 0) {                    // a5.c:60-66
		for (int i = 0; i < a.nbits; i++) {
			if (dl) dl[i] = 0;
			if (ul) ul[i] = 0;
		}
		return;
	}
	if (a.alg != 1)                      // a5.c:72-75: A5/2..7 leave the buffers alone
		return;
	const uint8_t *key = a.keys + (size_t)g * 8;
	const uint32_t fn = a.fn[g];
	uint32_t lkey[8];
#pragma unroll
	for (int i = 0; i < 8; i++)
		lkey[i] = key[i ^ 1];
	lkey[6] ^= ((fn & 0x0000fu) << 4) & 0xffu;
	lkey[3] ^= ((fn & 0x00030u) << 2) & 0xffu;
	lkey[1] ^= ((fn & 0x007c0u) >> 3) & 0xffu;
	lkey[0] ^= ((fn & 0x0f800u) >> 11) & 0xffu;
	lkey[0] ^= ((fn & 0x70000u) >> 11) & 0xffu;
	A5State s = {0, 0, 0, 0};
#pragma unroll
	for (int by = 0; by < 8; by++) {
		for (int bi = 7; bi >= 0; bi--) {
			const uint32_t b = (lkey[by] >> bi) & 1u;
			s.r0 = a5_clock(s.r0, (1u << 19) - 1, 0x072000u) ^ b;
			s.r1 = a5_clock(s.r1, (1u << 22) - 1, 0x311000u) ^ b;
			s.r2 = a5_clock(s.r2, (1u << 23) - 1, 0x660000u) ^ b;
			s.r3 = a5_clock(s.r3, (1u << 17) - 1, 0x013100u) ^ b;
		}
	}
	s.r0 |= 1; s.r1 |= 1; s.r2 |= 1; s.r3 |= 1;
	for (int i = 0; i < 250; i++)
		a5_step(s);
	for (int i = 0; i < a.nbits; i++) {
		a5_step(s);
		if (dl) dl[i] = (uint8_t)a5_out(s);
	}
	if (!ul)
		return;
	for (int i = 0; i < a.nbits; i++) {
		a5_step(s);
		ul[i] = (uint8_t)a5_out(s);
	}
}

hipError_t launch_a5(const A5Args &a, hipStream_t stream)
{
	if (a.n <= 0)
		return hipSuccess;
	hipLaunchKernelGGL(k_a5, dim3((a.n + 63) / 64), dim3(64), 0, stream, a);
	return hipGetLastError();
}

}  // namespace gmr1
