// ambe_libm.h -- the two libm functions the reference's vocoder calls on arguments that cannot be tabulated, restated so
// that the GPU returns the bits the host's glibc returns.
//
// Third-party algorithm, absent from /root/reference: GNU libc 2.35 (the image's libm.so.6), `powf` =
// sysdeps/ieee754/flt-32/e_powf.c with e_powf_log2_data.c / e_exp2f_data.c, and `cosf` = s_cosf.c with s_sincosf.h /
// s_sincosf_data.c -- both taken by glibc from ARM Optimized Routines (Szabolcs Nagy, MIT licence); the algorithm and its
// constants are published there.  powf: x = 2^k z, log2 z from a 16-entry table (1/c, log2 c) and a degree-4 polynomial
// in r = z/c - 1, all in double; y log2 x split into k/32 + r, 2^(k/32) from a 32-entry table, a degree-3 polynomial in r;
// one rounding to float at the end.  cosf: reduction by pi/2 in double (a multiply for |x| < 120, 96 bits of 4/pi above),
// degree-8 / degree-7 polynomials in double, one rounding.
//
// The restatement is checked, not trusted: tests/test_codec_host.py runs these very functions on the host
// (gmr1_hip_codec_libm_check) against the libm the reference would call, on tens of millions of arguments, and requires
// identical bits; on the GPU the same double-precision operations give the same results (IEEE add / multiply, no
// contraction: -ffp-contract=off; glibc's FMA build of the same source was checked to return the same floats).
// Outside the range these cover (non-positive or subnormal x, results near overflow / underflow, |x| >= 2^28 for the
// cosine) the callers fall back to double-precision evaluation; the vocoder never gets there with finite parameters.
#pragma once

#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define AMBE_HD __host__ __device__ __forceinline__
#else
#define AMBE_HD inline
#endif

namespace gmr1 {
namespace ambe_libm {

AMBE_HD uint64_t bits_of(double d)
{
	uint64_t u;
	memcpy(&u, &d, 8);
	return u;
}

AMBE_HD double double_of(uint64_t u)
{
	double d;
	memcpy(&d, &u, 8);
	return d;
}

AMBE_HD uint32_t bits_of(float f)
{
	uint32_t u;
	memcpy(&u, &f, 4);
	return u;
}

// Tables in memory (device: a buffer filled by the host, capi_ambe.cpp): 2^(i/32) as a double bit pattern minus i << 47
// (e_exp2f_data.c: tab; computed with exp2l), and the 16 (1/c, log2 c) pairs below
struct LibmTab {
	uint64_t exp2_tab[32];
	double invc[16], logc[16];
};

// e_exp2f_data.c: poly; e_powf_log2_data.c: tab (invc, logc), poly
#define AMBE_EXP2_C0 0x1.c6af84b912394p-5
#define AMBE_EXP2_C1 0x1.ebfce50fac4f3p-3
#define AMBE_EXP2_C2 0x1.62e42ff0c52d6p-1

AMBE_HD void log2_entry(int i, double &invc, double &logc)
{
	switch (i) {
	case 0: invc = 0x1.661ec79f8f3bep+0; logc = -0x1.efec65b963019p-2; break;
	case 1: invc = 0x1.571ed4aaf883dp+0; logc = -0x1.b0b6832d4fca4p-2; break;
	case 2: invc = 0x1.49539f0f010bp+0; logc = -0x1.7418b0a1fb77bp-2; break;
	case 3: invc = 0x1.3c995b0b80385p+0; logc = -0x1.39de91a6dcf7bp-2; break;
	case 4: invc = 0x1.30d190c8864a5p+0; logc = -0x1.01d9bf3f2b631p-2; break;
	case 5: invc = 0x1.25e227b0b8eap+0; logc = -0x1.97c1d1b3b7afp-3; break;
	case 6: invc = 0x1.1bb4a4a1a343fp+0; logc = -0x1.2f9e393af3c9fp-3; break;
	case 7: invc = 0x1.12358f08ae5bap+0; logc = -0x1.960cbbf788d5cp-4; break;
	case 8: invc = 0x1.0953f419900a7p+0; logc = -0x1.a6f9db6475fcep-5; break;
	case 9: invc = 0x1p+0; logc = 0x0p+0; break;
	case 10: invc = 0x1.e608cfd9a47acp-1; logc = 0x1.338ca9f24f53dp-4; break;
	case 11: invc = 0x1.ca4b31f026aap-1; logc = 0x1.476a9543891bap-3; break;
	case 12: invc = 0x1.b2036576afce6p-1; logc = 0x1.e840b4ac4e4d2p-3; break;
	case 13: invc = 0x1.9c2d163a1aa2dp-1; logc = 0x1.40645f0c6651cp-2; break;
	case 14: invc = 0x1.886e6037841edp-1; logc = 0x1.88e9c2c1b9ff8p-2; break;
	default: invc = 0x1.767dcf5534862p-1; logc = 0x1.ce0a44eb17bccp-2; break;
	}
}

// log2 of a positive normal float, in double (e_powf.c: log2_inline)
AMBE_HD double log2_of(const LibmTab &T, uint32_t ix)
{
	const uint32_t tmp = ix - 0x3f330000u;
	const int i = (int)((tmp >> (23 - 4)) % 16u);
	const uint32_t top = tmp & 0xff800000u;
	const uint32_t iz = ix - top;
	const int k = (int32_t)top >> 23;
	float zf;
	memcpy(&zf, &iz, 4);
	const double invc = T.invc[i], logc = T.logc[i];
	const double z = (double)zf;
	const double r = z * invc - 1;
	const double y0 = logc + (double)k;
	const double r2 = r * r;
	double y = 0x1.27616c9496e0bp-2 * r + -0x1.71969a075c67ap-2;
	const double p = 0x1.ec70a6ca7baddp-2 * r + -0x1.7154748bef6c8p-1;
	const double r4 = r2 * r2;
	double q = 0x1.71547652ab82bp0 * r + y0;
	q = p * r2 + q;
	y = y * r4 + q;
	return y;
}

// 2^xd rounded to float, |xd| < 126 (e_powf.c: exp2_inline with sign_bias 0)
AMBE_HD float exp2_of(const LibmTab &T, double xd)
{
	const double shift = 0x1.8p+52 / 32;
	double kd = xd + shift;
	const uint64_t ki = bits_of(kd);
	kd -= shift;
	const double r = xd - kd;
	uint64_t t = T.exp2_tab[ki % 32];
	t += ki << (52 - 5);
	const double s = double_of(t);
	const double z = AMBE_EXP2_C0 * r + AMBE_EXP2_C1;
	const double r2 = r * r;
	double y = AMBE_EXP2_C2 * r + 1;
	y = z * r2 + y;
	y = y * s;
	return (float)y;
}

// powf(2.0f, y): log2(2) comes out of log2_of as exactly 1 (table entry 9: invc 1, logc 0), so this is exp2_of(y).
// *ok = false outside the range restated here.
AMBE_HD float pow2f(const LibmTab &T, float y, bool *ok)
{
	*ok = y > -125.0f && y < 125.0f;       // also false for NaN
	return *ok ? exp2_of(T, (double)y) : 0.0f;
}

// powf(x, y) for positive normal x and a result well inside the float range
AMBE_HD float powf_pos(const LibmTab &T, float x, float y, bool *ok)
{
	const uint32_t ix = bits_of(x);
	*ok = ix >= 0x00800000u && ix < 0x7f800000u;
	if (!*ok)
		return 0.0f;
	const double ylogx = (double)y * log2_of(T, ix);
	*ok = ylogx > -125.0 && ylogx < 125.0;
	return *ok ? exp2_of(T, ylogx) : 0.0f;
}

// ---- cosf (s_cosf.c, s_sincosf.h, s_sincosf_data.c) ----
// glibc picks its FMA build of this source on every x86-64 CPU that has FMA (sysdeps/x86_64/fpu/multiarch): the
// multiply-adds below are fused where that build fuses them.  (The plain build differs in about one argument in 10^7.)

AMBE_HD double fmadd(double a, double b, double c)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __fma_rn(a, b, c);
#else
	return __builtin_fma(a, b, c);
#endif
}

// polynomial for sin (n even) or cos (n odd) of a reduced argument; `neg`: the negated cosine coefficients (__sincosf_table[1])
AMBE_HD float sincos_poly(double x, double x2, bool neg, int n)
{
	const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
	if ((n & 1) == 0) {
		const double x3 = x * x2;
		const double t = fmadd(x2, s3, s2);
		const double x7 = x3 * x2;
		const double s = fmadd(x3, s1, x);
		return (float)fmadd(x7, t, s);
	}
	const double sg = neg ? -1.0 : 1.0;
	const double c0 = sg * 0x1p0, c1 = sg * -0x1.ffffffd0c621cp-2, c2 = sg * 0x1.55553e1068f19p-5,
	             c3 = sg * -0x1.6c087e89a359dp-10, c4 = sg * 0x1.99343027bf8c3p-16;
	const double x4 = x2 * x2;
	const double u = fmadd(x2, c4, c3);
	const double v = fmadd(x2, c1, c0);
	const double x6 = x4 * x2;
	const double c = fmadd(x4, c2, v);
	return (float)fmadd(x6, u, c);
}

AMBE_HD uint32_t inv_pio4_word(int i)
{
	// 4 / pi = 0x0.a2f9836e4e441529fc2757d1f534ddc0db6295993c439041..., one byte further per entry (__inv_pio4)
	const uint32_t w[24] = {0xa2, 0xa2f9, 0xa2f983, 0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529,
	                        0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0,
	                        0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};
	return w[i];
}

AMBE_HD float cosf_glibc(float y)
{
	const uint32_t xi0 = bits_of(y);
	const uint32_t top = (xi0 >> 20) & 0x7ffu;                 // abstop12
	double x = (double)y;
	const double sign4[4] = {1.0, -1.0, -1.0, 1.0};
	if (top < 0x3f4u) {                                        // |y| < pi/4
		if (top < 0x398u)                                      // |y| < 2^-12
			return 1.0f;
		return sincos_poly(x, x * x, false, 1);
	}
	if (top < 0x42fu) {                                        // |y| < 120: one multiply finds the quadrant
		const double r = x * 0x1.45F306DC9C883p+23;
		const int n = ((int32_t)r + 0x800000) >> 24;
		x = fmadd(-(double)n, 0x1.921FB54442D18p0, x);
		return sincos_poly(x * sign4[n & 3], x * x, (n & 2) != 0, n ^ 1);
	}
	if (top < 0x7f8u) {                                        // finite: 96 bits of 4 / pi
		const int sign = (int)(xi0 >> 31);
		const int at = (int)((xi0 >> 26) & 15u);
		const int shift = (int)((xi0 >> 23) & 7u);
		uint32_t xi = (xi0 & 0xffffffu) | 0x800000u;
		xi <<= shift;
		uint64_t res0 = (uint64_t)(uint32_t)(xi * inv_pio4_word(at));
		const uint64_t res1 = (uint64_t)xi * inv_pio4_word(at + 4);
		const uint64_t res2 = (uint64_t)xi * inv_pio4_word(at + 8);
		res0 = (res2 >> 32) | (res0 << 32);
		res0 += res1;
		const uint64_t n64 = (res0 + (1ull << 61)) >> 62;
		res0 -= n64 << 62;
		const int n = (int)n64;
		x = (double)(int64_t)res0 * 0x1.921FB54442D18p-62;
		return sincos_poly(x * sign4[(n + sign) & 3], x * x, ((n + sign) & 2) != 0, n ^ 1);
	}
	return y - y;                                              // inf, NaN -> NaN
}

}  // namespace ambe_libm
}  // namespace gmr1
