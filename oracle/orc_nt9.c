/*
 * oracle/orc_nt9.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the NT9 burst codecs: FACCH9
 * (reference src/l1/facch9.c), TCH9 in its three modes (reference src/l1/tch9.c), the puncturing
 * array generator (reference src/l1/punct.c:48-133) and the inter-burst interleaver (reference
 * src/l1/interleave.c:95-190).  PARITY UNPINNED, see orc_3p.h.
 */
#include "orc_gmr1.h"

#include <string.h>

static const struct orc_crc_code crc16 = { 16, 0x1021, 0, 0 };  /* src/l1/crc.c:58-63 */

/* ---- puncturing schemes: mask entry 0 = punctured (punct.c:137-175, 389-428, 448-481) ---- */
struct punct { int r, L, N; uint8_t mask[15]; };
static const struct punct k5_12_P23  = { 2, 3, 2, { 0,1, 1,0, 1,1 } };
static const struct punct k5_12_P25  = { 2, 5, 2, { 1,0, 1,1, 1,0, 1,1, 1,1 } };
static const struct punct k5_12_Ps25 = { 2, 5, 2, { 1,1, 1,1, 1,0, 1,1, 1,0 } };
static const struct punct k5_13_P25  = { 2, 5, 3, { 1,1,1, 1,1,1, 1,0,1, 1,1,1, 1,0,1 } };
static const struct punct k5_13_P15  = { 1, 5, 3, { 1,0,1, 1,1,1, 1,1,1, 1,1,1, 1,1,1 } };
static const struct punct k5_13_Ps15 = { 1, 5, 3, { 1,1,1, 1,1,1, 1,1,1, 1,1,1, 1,0,1 } };
static const struct punct k5_15_P23  = { 2, 3, 5, { 1,1,1,1,1, 1,1,0,1,1, 1,1,1,1,0 } };
static const struct punct k5_15_P53  = { 5, 3, 5, { 1,1,1,0,1, 1,0,0,1,1, 1,1,1,0,0 } };
static const struct punct k5_15_Ps53 = { 5, 3, 5, { 1,1,1,0,0, 1,0,0,1,1, 1,1,1,0,1 } };

/* gmr1_puncturer_generate, punct.c:48-133 */
static void puncturer_generate(struct orc_conv_code *code, const struct punct *pre,
                               const struct punct *main_, const struct punct *post, int repeat)
{
	int N = code->N, cl, d, ii = 0, io = 0, ip, i;
	int *p = code->punct;
	cl = (code->len + code->K - 1) * N;          /* osmo_conv_get_output_length(code, 0), unpunctured */
	if (pre) {
		d = pre->L * N;
		for (ip = 0; ii < cl && ip < d; ii++, ip++)
			if (pre->mask[ip] == 0)
				p[io++] = ii;
	}
	if (post)
		cl -= post->L * N;
	for (i = 0; i < repeat; i++) {
		d = main_->L * N;
		for (ip = 0; ii < cl && ip < d; ii++, ip++)
			if (main_->mask[ip] == 0)
				p[io++] = ii;
	}
	if (post) {
		d = post->L * N;
		ii = cl;
		for (ip = 0; ii > 0 && ip < d; ii++, ip++)
			if (post->mask[ip] == 0)
				p[io++] = ii;
	}
	p[io] = -1;
	code->n_punct = io;
}

static struct orc_conv_code code_facch9, code_tch9[3];
static int ready;

static void codes_init(void)
{
	static const unsigned k5_12[2] = { 0x19, 0x17 };
	static const unsigned k5_13[3] = { 0x15, 0x1b, 0x1f };             /* conv.c:148-154 */
	static const unsigned k5_15[5] = { 0x15, 0x1b, 0x1f, 0x1d, 0x17 }; /* conv.c:201-209 */
	if (ready)
		return;
	orc_conv_make(&code_facch9, 2, 5, 316, ORC_TERM_FLUSH, k5_12);     /* facch9.c:42-48 */
	/* tch9.c:56-79 */
	orc_conv_make(&code_tch9[ORC_TCH9_2k4], 5, 5, 144, ORC_TERM_FLUSH, k5_15);
	puncturer_generate(&code_tch9[ORC_TCH9_2k4], &k5_15_P53, &k5_15_P23, &k5_15_Ps53, 41);
	orc_conv_make(&code_tch9[ORC_TCH9_4k8], 3, 5, 240, ORC_TERM_FLUSH, k5_13);
	puncturer_generate(&code_tch9[ORC_TCH9_4k8], &k5_13_P15, &k5_13_P25, &k5_13_Ps15, 41);
	orc_conv_make(&code_tch9[ORC_TCH9_9k6], 2, 5, 480, ORC_TERM_FLUSH, k5_12);
	puncturer_generate(&code_tch9[ORC_TCH9_9k6], &k5_12_P25, &k5_12_P23, &k5_12_Ps25, 158);
	ready = 1;
}

/* for the tests: 0 = FACCH9, 1 + mode = TCH9 */
const struct orc_conv_code *orc_nt9_code(int which)
{
	codes_init();
	return which == 0 ? &code_facch9 : (which >= 1 && which <= 3) ? &code_tch9[which - 1] : NULL;
}

int orc_tch9_punct(int mode, int *idx)      /* for the tests: the punctured positions of a mode */
{
	codes_init();
	memcpy(idx, code_tch9[mode].punct, sizeof(int) * (size_t)(code_tch9[mode].n_punct + 1));
	return code_tch9[mode].n_punct;
}

/* ---- inter-burst interleaver, interleave.c:95-190 (bits are bytes: ubit or sbit alike) ---- */
void orc_interleaver_init(struct orc_interleaver *il, int N, int K)
{
	memset(il, 0, sizeof(*il));
	il->N = N;
	il->K = K;
}

void orc_interleave_inter(struct orc_interleaver *il, void *bits_epp, const void *bits_ep)
{
	uint8_t tmp[ORC_IL_MAXK];
	int i, jk;
	i = il->n % il->N;
	memcpy(&il->bits_cpp[i * il->K], bits_ep, (size_t)il->K);
	for (jk = 0; jk < il->K; jk++) {
		i = ((il->n % il->N) - (jk % il->N) + il->N) % il->N;
		tmp[jk] = il->bits_cpp[(i * il->K) + jk];
	}
	memcpy(bits_epp, tmp, (size_t)il->K);
	il->n++;
}

void orc_deinterleave_inter(struct orc_interleaver *il, void *bits_ep, const void *bits_epp)
{
	const uint8_t *s = bits_epp;
	int i, jk;
	for (jk = 0; jk < il->K; jk++) {
		i = ((il->n % il->N) - (jk % il->N) + il->N) % il->N;
		il->bits_cpp[(i * il->K) + jk] = s[jk];
	}
	i = (il->n + 1) % il->N;
	memcpy(bits_ep, &il->bits_cpp[i * il->K], (size_t)il->K);
	il->n++;
}

/* ---- FACCH9, facch9.c:60-144 ---- */
void orc_facch9_encode(orc_ubit_t *bits_e, const uint8_t *l2, const orc_ubit_t *bits_sacch,
                       const orc_ubit_t *bits_status, const orc_ubit_t *ciph)
{
	orc_ubit_t u[316], c[640], x[648], my[658];
	codes_init();
	orc_pbit2ubit_lsb(u, l2, 300);
	orc_crc_set_bits(&crc16, u, 300, u + 300);
	orc_conv_encode(&code_facch9, u, c);
	memset(x, 0, 4);
	memset(x + 644, 0, 4);
	orc_interleave_intra(x + 4, c, 80);
	orc_scramble_ubit(x, x, 648);
	memcpy(my, x, 52);
	memcpy(my + 52, bits_sacch, 10);
	memcpy(my + 62, x + 52, 596);
	if (ciph)
		for (int i = 0; i < 658; i++)
			my[i] ^= ciph[i];
	memcpy(bits_e, my, 52);
	memcpy(bits_e + 52, bits_status, 4);
	memcpy(bits_e + 56, my + 52, 606);
}

int orc_facch9_decode(uint8_t *l2, orc_sbit_t *bits_sacch, orc_sbit_t *bits_status,
                      const orc_sbit_t *bits_e, const orc_ubit_t *ciph, int *conv_rv)
{
	orc_sbit_t my[658], x[648], c[640];
	orc_ubit_t u[316];
	int rv;
	codes_init();
	memcpy(my, bits_e, 52);
	memcpy(bits_status, bits_e + 52, 4);
	memcpy(my + 52, bits_e + 56, 606);
	if (ciph)
		for (int i = 0; i < 658; i++)
			if (ciph[i])
				my[i] = (orc_sbit_t)(-my[i]);
	memcpy(x, my, 52);
	memcpy(bits_sacch, my + 52, 10);
	memcpy(x + 52, my + 62, 596);
	orc_scramble_sbit(x, x, 648);
	orc_deinterleave_intra(c, x + 4, 80);
	rv = orc_conv_decode(&code_facch9, c, u);
	if (conv_rv)
		*conv_rv = rv;
	rv = orc_crc_check_bits(&crc16, u, 300, u + 300);
	l2[37] = 0;
	orc_ubit2pbit_lsb(l2, u, 300);
	return rv;
}

/* ---- TCH9, tch9.c:81-175 ---- */
void orc_tch9_encode(orc_ubit_t *bits_e, const uint8_t *l2, int mode, const orc_ubit_t *bits_sacch,
                     const orc_ubit_t *bits_status, const orc_ubit_t *ciph, struct orc_interleaver *il)
{
	const struct orc_conv_code *cc;
	orc_ubit_t u[480], c[648], x[648], my[658];
	codes_init();
	cc = &code_tch9[mode];
	orc_pbit2ubit_lsb(u, l2, cc->len);
	orc_conv_encode(cc, u, c);
	orc_interleave_intra(x, c, 81);
	orc_interleave_inter(il, x, x);
	orc_scramble_ubit(x, x, 648);
	memcpy(my, x, 52);
	memcpy(my + 52, bits_sacch, 10);
	memcpy(my + 62, x + 52, 596);
	if (ciph)
		for (int i = 0; i < 658; i++)
			my[i] ^= ciph[i];
	memcpy(bits_e, my, 52);
	memcpy(bits_e + 52, bits_status, 4);
	memcpy(bits_e + 56, my + 52, 606);
}

void orc_tch9_decode(uint8_t *l2, orc_sbit_t *bits_sacch, orc_sbit_t *bits_status, const orc_sbit_t *bits_e,
                     int mode, const orc_ubit_t *ciph, struct orc_interleaver *il, int *conv_rv)
{
	const struct orc_conv_code *cc;
	orc_sbit_t my[658], x[648], c[648];
	orc_ubit_t u[480];
	int rv;
	codes_init();
	cc = &code_tch9[mode];
	memcpy(my, bits_e, 52);
	memcpy(bits_status, bits_e + 52, 4);
	memcpy(my + 52, bits_e + 56, 606);
	if (ciph)
		for (int i = 0; i < 658; i++)
			if (ciph[i])
				my[i] = (orc_sbit_t)(-my[i]);
	memcpy(x, my, 52);
	memcpy(bits_sacch, my + 52, 10);
	memcpy(x + 52, my + 62, 596);
	orc_scramble_sbit(x, x, 648);
	orc_deinterleave_inter(il, x, x);
	orc_deinterleave_intra(c, x, 81);
	rv = orc_conv_decode(cc, c, u);
	if (conv_rv)
		*conv_rv = rv;
	orc_ubit2pbit_lsb(l2, u, cc->len);
}
