/*
 * oracle/orc_xch.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the two uplink / DC12 layer-1
 * codecs that gmr1_rx itself never calls: xCH over DC12 (reference
 * src/l1/xch_dc12.c: K=9 rate 1/3, tail-biting, punctured P(12;13)) and RACH (reference src/l1/rach.c:
 * K=5 rate 1/4, CRC8 + CRC12, class-1 bits sent twice).  PARITY UNPINNED, see orc_3p.h.
 */
#include "orc_gmr1.h"

#include <string.h>

static const struct orc_crc_code crc8  = {  8, 0x9b,   0, 0 };  /* src/l1/crc.c:36-44  */
static const struct orc_crc_code crc12 = { 12, 0x80f,  0, 0 };  /* src/l1/crc.c:46-54  */
static const struct orc_crc_code crc16 = { 16, 0x1021, 0, 0 };  /* src/l1/crc.c:58-63  */

static struct orc_conv_code code_xch, code_rach;
static int ready;

static void codes_init(void)
{
	/* conv.c:345-351: g0 = 1+D2+D3+D5+D6+D7+D8, g1 = 1+D+D3+D4+D7+D8, g2 = 1+D+D2+D5+D8 (bit i = D^i) */
	static const unsigned k9_13[3] = { 0x1ed, 0x19b, 0x127 };
	static const unsigned k5_14[4] = { 0x19, 0x17, 0x15, 0x1f };     /* conv.c:174-181 */
	/* gmr1_punct_k9_13_P1213, punct.c:1105-1125: 0 = punctured, 13 trellis steps of 3 bits */
	static const uint8_t p1213[39] = {
		1,1,0, 1,0,1, 0,1,1, 1,1,0, 1,0,1, 0,1,1, 1,1,0, 1,0,1, 0,1,1, 1,1,0, 1,0,1, 0,1,1, 1,1,1,
	};
	int io = 0;
	if (ready)
		return;
	/* xch_dc12.c:45-54: len 208, tail biting; gmr1_puncturer_generate(code, NULL, P1213, NULL, 0)
	 * (punct.c:48-133) repeats the mask over the 624 unpunctured bits: 16 x 12 = 192 positions */
	orc_conv_make(&code_xch, 3, 9, 208, ORC_TERM_TAIL_BITING, k9_13);
	for (int ii = 0; ii < 624; ii++)
		if (p1213[ii % 39] == 0)
			code_xch.punct[io++] = ii;
	code_xch.punct[io] = -1;
	code_xch.n_punct = io;
	/* rach.c:44-66: len 159, flush; only b[0..539] punctured: bits 2 and 3 of the first 135 steps */
	orc_conv_make(&code_rach, 4, 5, 159, ORC_TERM_FLUSH, k5_14);
	for (int i = 0; i < 135; i++) {
		code_rach.punct[2 * i] = 4 * i + 2;
		code_rach.punct[2 * i + 1] = 4 * i + 3;
	}
	code_rach.punct[270] = -1;
	code_rach.n_punct = 270;
	ready = 1;
}

/* for the tests: 0 = xCH over DC12, 1 = RACH */
const struct orc_conv_code *orc_xch_code(int which)
{
	codes_init();
	return which == 0 ? &code_xch : which == 1 ? &code_rach : NULL;
}

/* osmo_pbit2ubit_ext / osmo_ubit2pbit_ext in lsb mode with bit offsets on the packed side */
static void p2u_lsb(orc_ubit_t *out, const uint8_t *in, int in_ofs, int n)
{
	for (int i = 0; i < n; i++) {
		int k = in_ofs + i;
		out[i] = (in[k >> 3] >> (k & 7)) & 1;
	}
}

static void u2p_lsb(uint8_t *out, int out_ofs, const orc_ubit_t *in, int n)
{
	for (int i = 0; i < n; i++) {
		int k = out_ofs + i;
		if (in[i])
			out[k >> 3] |= (uint8_t)(1 << (k & 7));
		else
			out[k >> 3] &= (uint8_t)~(1 << (k & 7));
	}
}

/* ---- xCH over DC12, xch_dc12.c:64-108 ---- */
void orc_xch_dc12_encode(orc_ubit_t *bits_e, const uint8_t *l2)
{
	orc_ubit_t u[208], c[432], ep[432];
	codes_init();
	orc_pbit2ubit_lsb(u, l2, 192);
	orc_crc_set_bits(&crc16, u, 192, u + 192);
	orc_conv_encode(&code_xch, u, c);
	orc_interleave_intra(ep, c, 54);
	orc_scramble_ubit(bits_e, ep, 432);
}

int orc_xch_dc12_decode(uint8_t *l2, const orc_sbit_t *bits_e, int *conv_rv)
{
	orc_sbit_t ep[432], c[432];
	orc_ubit_t u[208];
	int rv;
	codes_init();
	orc_scramble_sbit(ep, bits_e, 432);
	orc_deinterleave_intra(c, ep, 54);
	rv = orc_conv_decode(&code_xch, c, u);
	if (conv_rv)
		*conv_rv = rv;
	rv = orc_crc_check_bits(&crc16, u, 192, u + 192);
	orc_ubit2pbit_lsb(l2, u, 192);
	return rv;
}

/* ---- RACH, rach.c:78-200 ---- */
void orc_rach_encode(orc_ubit_t *bits_e, const uint8_t *rach, uint8_t sb_mask)
{
	orc_ubit_t u[159], *u1 = u + 135, *u2 = u;
	orc_ubit_t c[382], e1p[112], e2p[270], ep[494], x[494];
	codes_init();
	p2u_lsb(u1, rach, 0, 16);
	p2u_lsb(u2, rach, 16, 123);
	orc_crc_set_bits(&crc8, u1, 16, u1 + 16);
	orc_crc_set_bits(&crc12, u2, 123, u2 + 123);
	for (int i = 0; i < 8; i++)
		u1[16 + i] ^= (sb_mask >> (7 - i)) & 1;
	orc_conv_encode(&code_rach, u, c);
	orc_interleave_intra(e1p, c + 270, 14);
	orc_interleave_intra(e2p, c, 33);
	memcpy(e2p + 264, c + 264, 6);
	memcpy(ep, e1p, 112);
	memcpy(ep + 112, e2p, 270);
	memcpy(ep + 382, e1p, 112);
	orc_scramble_ubit(x, ep, 494);
	memcpy(bits_e, x + 112, 136);
	memcpy(bits_e + 136, x, 112);
	memcpy(bits_e + 248, x + 382, 112);
	memcpy(bits_e + 360, x + 248, 134);
}

int orc_rach_decode(uint8_t *rach, const orc_sbit_t *bits_e, uint8_t sb_mask, int *conv_rv, int *crc_rv)
{
	orc_sbit_t x[494], ep[494], e1p[112], e2p[270], c[382];
	orc_ubit_t u[159], *u1 = u + 135, *u2 = u;
	int rv, crc[2];
	codes_init();
	memcpy(x, bits_e + 136, 112);
	memcpy(x + 112, bits_e, 136);
	memcpy(x + 248, bits_e + 360, 134);
	memcpy(x + 382, bits_e + 248, 112);
	orc_scramble_sbit(ep, x, 494);
	memcpy(e2p, ep + 112, 270);
	for (int i = 0; i < 112; i++)                    /* the two copies of the class-1 part, rach.c:161-162 */
		e1p[i] = (orc_sbit_t)(((int)ep[i] + (int)ep[i + 382]) >> 1);
	orc_deinterleave_intra(c + 270, e1p, 14);
	orc_deinterleave_intra(c, e2p, 33);
	memcpy(c + 264, e2p + 264, 6);
	rv = orc_conv_decode(&code_rach, c, u);
	if (conv_rv)
		*conv_rv = rv;
	crc[0] = orc_crc_check_bits(&crc8, u1, 16, u1 + 16);
	crc[1] = orc_crc_check_bits(&crc12, u2, 123, u2 + 123);
	if (crc[0]) {                                    /* rach.c:180-184: retried with the SB mask removed */
		for (int i = 0; i < 8; i++)
			u1[16 + i] ^= (sb_mask >> (7 - i)) & 1;
		crc[0] = orc_crc_check_bits(&crc8, u1, 16, u1 + 16);
	}
	if (crc_rv) {
		crc_rv[0] = crc[0];
		crc_rv[1] = crc[1];
	}
	rach[17] = 0x00;
	u2p_lsb(rach, 0, u1, 16);
	u2p_lsb(rach, 16, u2, 123);
	return crc[0] || crc[1];
}
