/*
 * oracle/orc_tch.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the two pieces the TCH3 follow-up
 * of gmr1_rx needs besides the normal-burst path: the DKAB (dual keep-alive burst) demodulator
 * (reference src/sdr/dkab.c) and the GMR-1 A5/1 keystream generator (reference src/l1/a5.c).
 * PARITY UNPINNED, see orc_3p.h: the reference holds no vectors for either.
 */
#include "orc_gmr1.h"

#include <errno.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PIf 3.14159265358979323846f
#define DKAB_SYMS (39 * 3)               /* sdr/dkab.h:37 */
#define DKAB_PWR_RATIO_THRESHOLD 10.0f   /* dkab.c:47 */

static float normsq(orc_cf v) { return crealf(v) * crealf(v) + cimagf(v) * cimagf(v); }

/* decision D7: the reference indexes burst->data without a bound (p > 51 or a late TOA runs past the
 * window); samples outside the burst read as 0 here and in the product */
static orc_cf at(const orc_cf *b, int len, int i) { return (i >= 0 && i < len) ? b[i] : 0.0f; }

/* dkab.c:57-151 */
static int dkab_find_toa(const orc_cf *burst, int len, int sps, int p, float *toa_p)
{
	int w, i, ofs[2], d, mi, rv;
	float mp, toa, egy_peak, egy_valley;
	int l_peak, l_valley, toa_i;
	float *pwr;

	w = len - (DKAB_SYMS * sps) + 1;
	if (w <= 0)
		return -EINVAL;
	pwr = (float *)calloc((size_t)w, sizeof(float));
	if (!pwr)
		return -ENOMEM;

	ofs[0] = sps * (2 + p);
	ofs[1] = sps * (2 + p + 59);
	d = sps * 5;

	pwr[0] = 0.0f;
	for (i = 0; i < d; i++)
		pwr[0] += normsq(at(burst, len, ofs[0] + i)) + normsq(at(burst, len, ofs[1] + i));
	mi = 0;
	mp = pwr[0];
	for (i = 0; i < w - 1; i++) {
		float np = pwr[i]
			- normsq(at(burst, len, ofs[0] + i)) - normsq(at(burst, len, ofs[1] + i))
			+ normsq(at(burst, len, ofs[0] + d + i)) + normsq(at(burst, len, ofs[1] + d + i));
		pwr[i + 1] = np;
		if (np > mp) {
			mi = i + 1;
			mp = np;
		}
	}
	toa = (float)mi;
	if ((mi > 0) && (mi < (w - 1)))
		toa += 0.5f * (-pwr[mi - 1] + pwr[mi + 1]) / (-pwr[mi - 1] + 2.0f * pwr[mi] - pwr[mi + 1]);
	toa += ((float)(sps - 1)) / 2.0f;
	*toa_p = toa;
	toa_i = (int)roundf(toa);

	egy_peak = 0.0f;
	l_peak = d * 2;
	for (i = 0; i < d; i++)
		egy_peak += normsq(at(burst, len, toa_i + ofs[0] + i)) + normsq(at(burst, len, toa_i + ofs[1] + i));
	egy_peak /= l_peak;
	egy_valley = 0.0f;
	l_valley = ofs[1] - ofs[0] - d;
	for (i = 0; i < l_valley; i++)
		egy_valley += normsq(at(burst, len, toa_i + ofs[0] + d + i));
	egy_valley /= l_valley;
	rv = ((egy_peak / egy_valley) > DKAB_PWR_RATIO_THRESHOLD) ? 0 : 1;
	free(pwr);
	return rv;
}

/* dkab.c:161-181 */
static void dkab_soft_bits(const orc_cf *burst, int len, int sps, int p, float toa, orc_sbit_t *ebits)
{
	int toa_i = (int)roundf(toa), ofs[2];
	ofs[0] = toa_i + sps * (2 + p);
	ofs[1] = toa_i + sps * (2 + p + 59);
	for (int i = 0; i < 8; i++) {
		int o = ofs[i >> 2] + sps * (i & 3);
		float pd = cargf(at(burst, len, o) * conjf(at(burst, len, o + sps)));
		ebits[i] = (orc_sbit_t)roundf((0.5f - (fabsf(pd) / PIf)) * 254.0f);
	}
}

/* dkab.c:196-224: 0 found, 1 not found, < 0 error.  toa is written whenever the search ran. */
int orc_dkab_demod(const orc_cf *in, int in_len, int sps, float freq_shift, int p,
                   orc_sbit_t *ebits, float *toa_p)
{
	orc_cf *burst = (orc_cf *)malloc(sizeof(orc_cf) * (size_t)(in_len > 0 ? in_len : 1));
	int rv;
	if (!burst)
		return -ENOMEM;
	orc_sig_normalize(in, in_len, 1, (freq_shift - (PIf / 4)) / sps, burst);
	rv = dkab_find_toa(burst, in_len, sps, p, toa_p);
	if (!rv)
		dkab_soft_bits(burst, in_len, sps, p, *toa_p, ebits);
	free(burst);
	return rv;
}

/* ---- A5/1, GMR-1 variant (a5.c:56-282) ---- */

static uint32_t parity32(uint32_t x)
{
	x ^= x >> 16; x ^= x >> 8; x ^= x >> 4; x &= 0xf;
	return (0x6996 >> x) & 1;
}
static uint32_t lfsr_clock(uint32_t r, uint32_t mask, uint32_t taps) { return ((r << 1) & mask) | parity32(r & taps); }

#define R1_MASK ((1u << 19) - 1)
#define R2_MASK ((1u << 22) - 1)
#define R3_MASK ((1u << 23) - 1)
#define R4_MASK ((1u << 17) - 1)
#define R1_TAPS 0x072000u
#define R2_TAPS 0x311000u
#define R3_TAPS 0x660000u
#define R4_TAPS 0x013100u

static void clock_force(uint32_t *r)
{
	r[0] = lfsr_clock(r[0], R1_MASK, R1_TAPS);
	r[1] = lfsr_clock(r[1], R2_MASK, R2_TAPS);
	r[2] = lfsr_clock(r[2], R3_MASK, R3_TAPS);
	r[3] = lfsr_clock(r[3], R4_MASK, R4_TAPS);
}

static void clock_rule(uint32_t *r)      /* a5.c:163-185 */
{
	int cb[3], m;
	cb[0] = !!(r[3] & (1u << 15));
	cb[1] = !!(r[3] & (1u << 6));
	cb[2] = !!(r[3] & (1u << 1));
	m = (cb[0] + cb[1] + cb[2]) >= 2;
	if (cb[0] == m) r[0] = lfsr_clock(r[0], R1_MASK, R1_TAPS);
	if (cb[1] == m) r[1] = lfsr_clock(r[1], R2_MASK, R2_TAPS);
	if (cb[2] == m) r[2] = lfsr_clock(r[2], R3_MASK, R3_TAPS);
	r[3] = lfsr_clock(r[3], R4_MASK, R4_TAPS);
}

static int maj3(uint32_t a, uint32_t b, uint32_t c) { return (!!a + !!b + !!c) >= 2; }

static orc_ubit_t output_bit(const uint32_t *r)   /* a5.c:191-216 */
{
	int m0 = maj3(r[0] & (1u << 1), r[0] & (1u << 6), r[0] & (1u << 15));
	int m1 = maj3(r[1] & (1u << 3), r[1] & (1u << 8), r[1] & (1u << 14));
	int m2 = maj3(r[2] & (1u << 4), r[2] & (1u << 15), r[2] & (1u << 19));
	m0 ^= !!(r[0] & (1u << 11));
	m1 ^= !!(r[1] & (1u << 1));
	m2 ^= !!(r[2] & (1u << 0));
	return (orc_ubit_t)(m0 ^ m1 ^ m2);
}

void orc_a5_1(const uint8_t *key, uint32_t fn, int nbits, orc_ubit_t *dl, orc_ubit_t *ul)
{
	uint32_t r[4] = {0, 0, 0, 0};
	uint8_t lkey[8];
	int i;
	for (i = 0; i < 8; i++)
		lkey[i] = key[i ^ 1];
	lkey[6] ^= (uint8_t)((fn & 0x0000f) << 4);
	lkey[3] ^= (uint8_t)((fn & 0x00030) << 2);
	lkey[1] ^= (uint8_t)((fn & 0x007c0) >> 3);
	lkey[0] ^= (uint8_t)((fn & 0x0f800) >> 11);
	lkey[0] ^= (uint8_t)((fn & 0x70000) >> 11);
	for (i = 0; i < 64; i++) {
		uint32_t b = (lkey[i >> 3] >> (7 - (i & 7))) & 1;
		clock_force(r);
		r[0] ^= b; r[1] ^= b; r[2] ^= b; r[3] ^= b;
	}
	r[0] |= 1; r[1] |= 1; r[2] |= 1; r[3] |= 1;
	for (i = 0; i < 250; i++)
		clock_rule(r);
	for (i = 0; i < nbits; i++) {
		clock_rule(r);
		if (dl)
			dl[i] = output_bit(r);
	}
	if (!ul)
		return;
	for (i = 0; i < nbits; i++) {
		clock_rule(r);
		ul[i] = output_bit(r);
	}
}

/* a5.c:56-78: n = 0 -> zeros, n = 1 -> A5/1, anything else leaves the buffers alone */
void orc_a5(int n, const uint8_t *key, uint32_t fn, int nbits, orc_ubit_t *dl, orc_ubit_t *ul)
{
	if (n == 0) {
		if (dl) memset(dl, 0, (size_t)nbits);
		if (ul) memset(ul, 0, (size_t)nbits);
	} else if (n == 1) {
		orc_a5_1(key, fn, nbits, dl, ul);
	}
}
