/*
 * oracle/orc_l1.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the
 * reference's layer-1 channel coding chains (reference src/l1/).
 * PARITY UNPINNED, see orc_3p.h.
 */
#include "orc_gmr1.h"

#include <string.h>

/* ---- scrambler: reference src/l1/scramb.c:39-93 -------------------------
 * 15-bit LFSR in a uint16, seed 0x4d4b, feedback = bit14 ^ bit0, the
 * feedback bit is also the scrambling bit. */
static int lfsr_step(uint16_t *r)
{
	int b = ((*r >> 14) ^ *r) & 1;
	*r = (uint16_t)((*r << 1) | b);
	return b;
}

void orc_scramble_sbit(orc_sbit_t *out, const orc_sbit_t *in, int len)
{
	uint16_t r = 0x4d4b;
	for (int i = 0; i < len; i++) {
		orc_sbit_t v = in[i];
		out[i] = lfsr_step(&r) ? (orc_sbit_t)-v : v;
	}
}

void orc_scramble_ubit(orc_ubit_t *out, const orc_ubit_t *in, int len)
{
	uint16_t r = 0x4d4b;
	for (int i = 0; i < len; i++)
		out[i] = in[i] ^ (orc_ubit_t)lfsr_step(&r);
}

/* ---- intra-burst interleaver: reference src/l1/interleave.c:48-87 ------- */
void orc_interleave_intra(void *out, const void *in, int N)
{
	const uint8_t *s = in;
	uint8_t *d = out;
	for (int kc = 0; kc < 8 * N; kc++)
		d[N * ((5 * kc) & 7) + (kc >> 3)] = s[kc];
}

void orc_deinterleave_intra(void *out, const void *in, int N)
{
	const uint8_t *s = in;
	uint8_t *d = out;
	for (int kc = 0; kc < 8 * N; kc++)
		d[kc] = s[N * ((5 * kc) & 7) + (kc >> 3)];
}

/* ---- codes: polynomials from the comments of reference src/l1/conv.c
 * (:123-128 k5_12, :174-181 k5_14, :518-523 tch3) -- bit i = D^i --------- */
static const struct orc_crc_code crc16 = { 16, 0x1021, 0, 0 };  /* src/l1/crc.c:58-63 */

static struct orc_conv_code code_bcch, code_facch3, code_tch3;
static int codes_ready;

static void codes_init(void)
{
	static const unsigned k5_12[2] = { 0x19, 0x17 };             /* 1+D3+D4 ; 1+D+D2+D4 */
	static const unsigned k5_14[4] = { 0x19, 0x17, 0x15, 0x1f }; /* + 1+D2+D4 ; 1+D+D2+D3+D4 */
	static const unsigned k7_12[2] = { 0x6d, 0x4f };             /* 1+D2+D3+D5+D6 ; 1+D+D2+D3+D6 */
	if (codes_ready)
		return;
	orc_conv_make(&code_bcch,   2, 5, 208, ORC_TERM_FLUSH, k5_12);       /* bcch.c:44-50, ccch.c:44-50 */
	orc_conv_make(&code_facch3, 4, 5,  92, ORC_TERM_FLUSH, k5_14);       /* facch3.c:44-50 */
	orc_conv_make(&code_tch3,   2, 7,  48, ORC_TERM_TAIL_BITING, k7_12); /* tch3.c:42-49 */
	/* P(1;2) mask {1,1,1,0}, 0 = punctured (punct.c:239-248), expanded as
	 * gmr1_puncturer_generate does (punct.c:48-133): every index = 3 mod 4 */
	code_tch3.n_punct = 24;
	for (int i = 0; i < 24; i++)
		code_tch3.punct[i] = 4 * i + 3;
	code_tch3.punct[24] = -1;
	codes_ready = 1;
}

/* for the tests: the specialised code a chain decodes with (0 = BCCH / CCCH, 1 = FACCH3, 2 = TCH3 speech) */
const struct orc_conv_code *orc_l1_code(int which)
{
	codes_init();
	return which == 0 ? &code_bcch : which == 1 ? &code_facch3 : which == 2 ? &code_tch3 : NULL;
}

/* ---- BCCH: reference src/l1/bcch.c:60-103 ------------------------------- */
void orc_bcch_encode(orc_ubit_t *bits_e, const uint8_t *l2)
{
	orc_ubit_t u[208], c[424], ep[424];
	codes_init();
	orc_pbit2ubit_lsb(u, l2, 192);
	orc_crc_set_bits(&crc16, u, 192, u + 192);
	orc_conv_encode(&code_bcch, u, c);
	orc_interleave_intra(ep, c, 53);
	orc_scramble_ubit(bits_e, ep, 424);
}

int orc_bcch_decode(uint8_t *l2, const orc_sbit_t *bits_e, int *conv_rv)
{
	orc_sbit_t ep[424], c[424];
	orc_ubit_t u[208];
	int rv;
	codes_init();
	orc_scramble_sbit(ep, bits_e, 424);
	orc_deinterleave_intra(c, ep, 53);
	rv = orc_conv_decode(&code_bcch, c, u);
	if (conv_rv) *conv_rv = rv;
	rv = orc_crc_check_bits(&crc16, u, 192, u + 192);
	orc_ubit2pbit_lsb(l2, u, 192);
	return rv;
}

/* ---- CCCH: reference src/l1/ccch.c:60-107 ------------------------------- */
void orc_ccch_encode(orc_ubit_t *bits_e, const uint8_t *l2)
{
	orc_ubit_t u[208], c[424], ep[432];
	codes_init();
	for (int i = 0; i < 4; i++)
		ep[i] = ep[431 - i] = 0;
	orc_pbit2ubit_lsb(u, l2, 192);
	orc_crc_set_bits(&crc16, u, 192, u + 192);
	orc_conv_encode(&code_bcch, u, c);
	orc_interleave_intra(ep + 4, c, 53);
	orc_scramble_ubit(bits_e, ep, 432);
}

int orc_ccch_decode(uint8_t *l2, const orc_sbit_t *bits_e, int *conv_rv)
{
	orc_sbit_t ep[432], c[424];
	orc_ubit_t u[208];
	int rv;
	codes_init();
	orc_scramble_sbit(ep, bits_e, 432);
	orc_deinterleave_intra(c, ep + 4, 53);
	rv = orc_conv_decode(&code_bcch, c, u);
	if (conv_rv) *conv_rv = rv;
	rv = orc_crc_check_bits(&crc16, u, 192, u + 192);
	orc_ubit2pbit_lsb(l2, u, 192);
	return rv;
}

/* ---- FACCH3: reference src/l1/facch3.c:65-170 --------------------------- */
void orc_facch3_encode(orc_ubit_t *bits_e, const uint8_t *l2,
                       const orc_ubit_t *bits_s, const orc_ubit_t *ciph)
{
	orc_ubit_t u[92], c[384], cp[384], ep[384], xmy[384];
	codes_init();
	orc_pbit2ubit_lsb(u, l2, 76);
	orc_crc_set_bits(&crc16, u, 76, u + 76);
	orc_conv_encode(&code_facch3, u, c);
	for (int i = 0; i < 384; i++)
		cp[(i & 3) * 96 + (i >> 2)] = c[i];
	for (int b = 0; b < 4; b++) {
		orc_interleave_intra(ep + 96 * b, cp + 96 * b, 12);
		orc_scramble_ubit(xmy + 96 * b, ep + 96 * b, 96);
		if (ciph)
			for (int j = 0; j < 96; j++)
				xmy[96 * b + j] ^= ciph[96 * b + j];
		memcpy(bits_e + 104 * b,      xmy + 96 * b,      22);
		memcpy(bits_e + 104 * b + 22, bits_s + 8 * b,     8);
		memcpy(bits_e + 104 * b + 30, xmy + 96 * b + 22, 74);
	}
}

int orc_facch3_decode(uint8_t *l2, orc_ubit_t *bits_s, const orc_sbit_t *bits_e,
                      const orc_ubit_t *ciph, int *conv_rv)
{
	orc_sbit_t xmy[384], ep[384], cp[384], c[384];
	orc_ubit_t u[92];
	int rv;
	codes_init();
	for (int b = 0; b < 4; b++) {
		const orc_sbit_t *e = bits_e + 104 * b;
		for (int j = 0; j < 8; j++)
			bits_s[8 * b + j] = e[22 + j] < 0;
		memcpy(xmy + 96 * b,      e,      22);
		memcpy(xmy + 96 * b + 22, e + 30, 74);
		if (ciph)
			for (int j = 0; j < 96; j++)
				if (ciph[96 * b + j])
					xmy[96 * b + j] = (orc_sbit_t)-xmy[96 * b + j];
		orc_scramble_sbit(ep + 96 * b, xmy + 96 * b, 96);
		orc_deinterleave_intra(cp + 96 * b, ep + 96 * b, 12);
	}
	for (int i = 0; i < 384; i++)
		c[i] = cp[(i & 3) * 96 + (i >> 2)];
	rv = orc_conv_decode(&code_facch3, c, u);
	if (conv_rv) *conv_rv = rv;
	rv = orc_crc_check_bits(&crc16, u, 76, u + 76);
	l2[9] = 0;
	orc_ubit2pbit_lsb(l2, u, 76);
	return rv;
}

/* ---- TCH3 speech: reference src/l1/tch3.c:60-183 ------------------------ */
static int tch3_perm(int kc)
{
	int ii = kc % 24, ij = kc / 24;
	return (ii < 8) ? (ij + 5 * ii) : (ij + 4 * ii + 8);
}

void orc_tch3_encode(orc_ubit_t *bits_e, const uint8_t *frame0, const uint8_t *frame1,
                     const orc_ubit_t *bits_s, const orc_ubit_t *ciph, int m)
{
	orc_ubit_t epp[208], xmy[208];
	codes_init();
	for (int i = 0; i < 2; i++) {
		orc_ubit_t d[80], c[104], ep[104];
		orc_pbit2ubit_msb(d, i ? frame1 : frame0, 80);
		orc_conv_encode(&code_tch3, d, c);   /* 48 bits -> 72 punctured coded bits */
		memcpy(c + 72, d + 48, 32);
		for (int kc = 0; kc < 104; kc++)
			ep[tch3_perm(kc)] = c[kc];
		for (int j = 0; j < 104; j++)
			epp[m ? (104 * i + j) : ((j << 1) + i)] = ep[j];
	}
	orc_scramble_ubit(xmy, epp, 208);
	if (ciph)
		for (int i = 0; i < 208; i++)
			xmy[i] ^= ciph[i];
	memcpy(bits_e,      xmy,       52);
	memcpy(bits_e + 52, bits_s,     4);
	memcpy(bits_e + 56, xmy + 52, 156);
}

void orc_tch3_decode(uint8_t *frame0, uint8_t *frame1, orc_ubit_t *bits_s,
                     const orc_sbit_t *bits_e, const orc_ubit_t *ciph, int m,
                     int *conv0_rv, int *conv1_rv)
{
	orc_sbit_t xmy[208], epp[208];
	codes_init();
	for (int i = 0; i < 4; i++)
		bits_s[i] = bits_e[52 + i] < 0;
	memcpy(xmy,      bits_e,       52);
	memcpy(xmy + 52, bits_e + 56, 156);
	if (ciph)
		for (int i = 0; i < 208; i++)
			if (ciph[i])
				xmy[i] = (orc_sbit_t)-xmy[i];
	orc_scramble_sbit(epp, xmy, 208);
	for (int i = 0; i < 2; i++) {
		orc_sbit_t ep[104], c[104];
		orc_ubit_t d[80];
		int rv;
		for (int j = 0; j < 104; j++)
			ep[j] = epp[m ? (104 * i + j) : ((j << 1) + i)];
		for (int kc = 0; kc < 104; kc++)
			c[kc] = ep[tch3_perm(kc)];
		rv = orc_conv_decode(&code_tch3, c, d);
		if (i ? (conv1_rv != 0) : (conv0_rv != 0))
			*(i ? conv1_rv : conv0_rv) = rv;
		for (int j = 48; j < 80; j++)
			d[j] = c[j + 24] < 0;
		orc_ubit2pbit_msb(i ? frame1 : frame0, d, 80);
	}
}
