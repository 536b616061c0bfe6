/*
 * oracle/orc_sdr.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the
 * reference's FCCH acquisition and pi/4-CxPSK burst demodulation
 * (reference src/sdr/fcch.c, src/sdr/pi4cxpsk.c, burst data src/sdr/nb.c).
 * PARITY UNPINNED, see orc_3p.h.
 */
#include "orc_gmr1.h"

#include <errno.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PIf 3.14159265358979323846f
#define SYM_RATE 23400   /* reference include/osmocom/gmr1/sdr/defs.h:33 */

static float normsq(orc_cf c) { return crealf(c) * crealf(c) + cimagf(c) * cimagf(c); }

/* ------------------------------------------------------------------------ */
/* Burst formats (ETSI TS 101 376-5-2 section 7.4; reference src/sdr/nb.c)    */
/* Compact text form: "pos:symbols" sync chunks, "pos+len" data chunks.      */
/* ------------------------------------------------------------------------ */

struct burst_src {
	const char *name;
	int rot_div;      /* rotation = pi / rot_div */
	int nbits, len, ebits;
	const char *sync[ORC_MAX_SYNC];
	const char *data;
};

static const struct burst_src burst_src[ORC_BURST__COUNT] = {
	/* nb.c:36-62 */
	{ "bcch", 4, 2, 234, 424,
	  { "28:02200020222 119:220 197:220" },
	  "2+26 39+80 122+75 200+31" },
	/* nb.c:67-89 */
	{ "dc2", 4, 2, 78, 132,
	  { "28:0123030" },
	  "2+26 35+40" },
	/* nb.c:94-120 */
	{ "dc6", 4, 2, 234, 432,
	  { "28:0002202 119:030 197:311" },
	  "2+26 35+84 122+75 200+31" },
	/* nb.c:125-151 */
	{ "dc12", 2, 1, 468, 432,
	  { "10:0010001111 228:00100011101 447:0010001111" },
	  "2+8 20+208 239+208 457+8" },
	/* nb.c:156-178 */
	{ "nt3_speech", 4, 2, 117, 212,
	  { "28:033123" },
	  "2+26 34+80" },
	/* nb.c:183-210 */
	{ "nt3_facch", 4, 1, 117, 104,
	  { "28:10101010", "28:11001001" },
	  "2+26 36+78" },
	/* nb.c:215-248 */
	{ "nt6", 4, 2, 234, 434,
	  { "28:022323 119:010 197:230", "28:000220 119:130 197:213" },
	  "2+26 34+85 122+75 200+31" },
	/* nb.c:253-289 */
	{ "nt9", 4, 2, 351, 662,
	  { "28:022323 119:122 197:010 275:230", "28:000220 119:020 197:130 275:213" },
	  "2+26 34+85 122+75 200+75 278+70" },
	/* nb.c:294-325 */
	{ "rach", 4, 2, 351, 494,
	  { "78:02200020222220220 127:22222222222222222222222222222222 "
	    "191:22222222222222222222222222222222 255:02200020222220220 347:0" },
	  "2+76 95+32 159+32 223+32 272+75" },
	/* nb.c:330-377 */
	{ "sdcch", 4, 1, 234, 208,
	  { "28:0101010 115:1010101 197:0101011", "28:0011001 115:1001100 197:1100111",
	    "28:0000111 115:1000011 197:1100001", "28:0110100 115:1011010 197:0101101" },
	  "2+26 35+80 122+75 204+27" },
};

static struct orc_burst bursts[ORC_BURST__COUNT];
static int bursts_ready;

static int parse_chunks(const char *s, struct orc_chunk *out, int is_sync)
{
	int n = 0;
	while (*s) {
		char *end;
		while (*s == ' ') s++;
		if (!*s) break;
		out[n].pos = (int)strtol(s, &end, 10);
		s = end + 1;   /* skip ':' or '+' */
		if (is_sync) {
			int l = 0;
			while (*s >= '0' && *s <= '9')
				out[n].syms[l++] = (uint8_t)(*s++ - '0');
			out[n].len = l;
		} else {
			out[n].len = (int)strtol(s, &end, 10);
			s = end;
		}
		n++;
	}
	return n;
}

const struct orc_burst *orc_burst_get(int id)
{
	if (!bursts_ready) {
		for (int i = 0; i < ORC_BURST__COUNT; i++) {
			const struct burst_src *s = &burst_src[i];
			struct orc_burst *b = &bursts[i];
			memset(b, 0, sizeof(*b));
			b->name = s->name;
			b->rotation = PIf / (float)s->rot_div;
			b->nbits = s->nbits;
			b->guard_pre = 2;
			b->guard_post = 3;
			b->len = s->len;
			b->ebits = s->ebits;
			for (int k = 0; k < ORC_MAX_SYNC && s->sync[k]; k++) {
				b->n_sync_chunks[k] = parse_chunks(s->sync[k], b->sync[k], 1);
				b->n_sync = k + 1;
			}
			b->n_data = parse_chunks(s->data, b->data, 0);
		}
		bursts_ready = 1;
	}
	if (id < 0 || id >= ORC_BURST__COUNT)
		return NULL;
	return &bursts[id];
}

/* symbol idx -> modulating value (pi4cxpsk.c:47-115): CQPSK idx*pi/2; CBPSK 0 / pi */
static orc_cf sym_val(const struct orc_burst *bt, int s)
{
	static const orc_cf q[4] = { 1, I, -1, -I };
	static const orc_cf b[2] = { 1, -1 };
	return bt->nbits == 2 ? q[s & 3] : b[s & 1];
}

/* symbol idx -> data bit j (pi4cxpsk.c:71-107): CQPSK 0:00 1:01 2:11 3:10 */
static int sym_bit(const struct orc_burst *bt, int s, int j)
{
	static const uint8_t q[4][2] = { {0,0}, {0,1}, {1,1}, {1,0} };
	return bt->nbits == 2 ? q[s & 3][j] : (s & 1);
}

/* data bits (MSB first) -> symbol idx (the '.bits' tables) */
static int bits_sym(const struct orc_burst *bt, int v)
{
	static const uint8_t q[4] = { 0, 1, 3, 2 };
	return bt->nbits == 2 ? q[v & 3] : (v & 1);
}

/* ------------------------------------------------------------------------ */
/* pi/4-CxPSK                                                                */
/* ------------------------------------------------------------------------ */

/* pi4cxpsk.c:184-268 -- note the accumulator is NOT cleared between sync
 * sequences, preserved here. */
static int sync_find(const struct orc_burst *bt, const orc_cf *burst, int len, int sps,
                     float *toa, float *pwr)
{
	int w = len - bt->len * sps + 1;
	float p_toa = 0.0f, p_pwr = 0.0f, p_idx = -1;
	orc_cf *corr, *tmp;

	if (w < 1)
		return -EINVAL;
	corr = calloc((size_t)w, sizeof(orc_cf));
	tmp = calloc((size_t)w, sizeof(orc_cf));

	for (int i = 0; i < bt->n_sync; i++) {
		int tl = 0;
		float s_toa, s_pwr;
		orc_cf s_peak;
		for (int c = 0; c < bt->n_sync_chunks[i]; c++) {
			const struct orc_chunk *cs = &bt->sync[i][c];
			orc_cf ref[ORC_MAX_SYNC_SYMS];
			for (int j = 0; j < cs->len; j++)
				ref[j] = sym_val(bt, cs->syms[j]);
			orc_correlate(ref, cs->len, burst + cs->pos * sps, cs->len * sps + w - 1, sps, tmp);
			for (int j = 0; j < w; j++)
				corr[j] += cabsf(tmp[j]);
			tl += cs->len;
		}
		s_toa = orc_peak_energy_find(corr, w, 3, ORC_PEAK_EARLY_LATE, &s_peak);
		s_peak /= (float)tl;
		s_pwr = normsq(s_peak);
		if (s_pwr > p_pwr) {
			p_pwr = s_pwr;
			p_toa = s_toa;
			p_idx = (float)i;
		}
	}
	free(tmp);
	free(corr);
	if (toa) *toa = p_toa;
	if (pwr) *pwr = p_pwr;
	return (int)p_idx;
}

/* pi4cxpsk.c:280-348 */
static void align(const struct orc_burst *bt, orc_cf *burst, int len, int sps, float toa)
{
	if (sps >= 4) {
		/* The reference indexes burst->data[i*sps+d] unchecked
		 * (pi4cxpsk.c:294-295); d can reach -1/-2 when the peak sits at
		 * lag 0, which only touches guard symbol 0.  Out-of-range reads
		 * are defined as 0 here (they never reach an output). */
		int d = (int)roundf(toa);
		for (int i = 0; i < bt->len; i++) {
			int j = i * sps + d;
			burst[i] = (j < 0 || j >= len) ? 0.0f : burst[j];
		}
	} else {
		int ofs_int = (int)roundf(toa);
		float ofs_frac = toa - (float)ofs_int;
		orc_cf *src = burst, *conv = NULL;
		if (fabs(ofs_frac) > 0.1f) {
			float pulse[21];
			for (int i = 0; i < 21; i++)
				pulse[i] = orc_sinc(PIf * ((float)(i - 10) + ofs_frac));
			conv = malloc(sizeof(orc_cf) * (size_t)len);
			orc_convolve_nodelay_real(pulse, 21, burst, len, conv);
			src = conv;
		}
		for (int i = 0; i < bt->len; i++) {
			int j = i * sps + ofs_int;
			burst[i] = (j < 0 || j >= len) ? 0.0f : src[j];
		}
		free(conv);
	}
}

/* pi4cxpsk.c:360-406 */
static float freq_err(const struct orc_burst *bt, const orc_cf *burst, int sync_id)
{
	int n = bt->n_sync_chunks[sync_id];
	orc_cf corr[ORC_MAX_CHUNKS];
	float pos[ORC_MAX_CHUNKS], f = 0.0f;

	if (n <= 1)
		return 0.0f;
	for (int i = 0; i < n; i++) {
		const struct orc_chunk *cs = &bt->sync[sync_id][i];
		corr[i] = 0.0f;
		pos[i] = (float)cs->pos + (float)cs->len / 2.0f;
		for (int j = 0; j < cs->len; j++)
			corr[i] += conjf(sym_val(bt, cs->syms[j])) * burst[cs->pos + j];
	}
	for (int i = 1; i < n; i++)
		f += cargf(corr[i] * conjf(corr[i - 1])) / (pos[i] - pos[i - 1]);
	f /= (float)(n - 1);
	return f;
}

/* pi4cxpsk.c:415-433 */
static orc_cf phase(const struct orc_burst *bt, const orc_cf *burst, int sync_id)
{
	orc_cf corr = 0.0f;
	for (int c = 0; c < bt->n_sync_chunks[sync_id]; c++) {
		const struct orc_chunk *cs = &bt->sync[sync_id][c];
		for (int i = 0; i < cs->len; i++)
			corr += conjf(sym_val(bt, cs->syms[i])) * burst[cs->pos + i];
	}
	return corr / cabsf(corr);
}

/* pi4cxpsk.c:468-503 */
static void soft_bits(const struct orc_burst *bt, const float *ssyms, orc_sbit_t *ebits)
{
	int mask = (1 << bt->nbits) - 1, k = 0;
	for (int c = 0; c < bt->n_data; c++) {
		const struct orc_chunk *dc = &bt->data[c];
		for (int i = dc->pos; i < dc->pos + dc->len; i++) {
			float sv = ssyms[i], svr = roundf(sv);
			int sp = (int)svr & mask;
			int ss = (svr > sv ? (sp - 1) : (sp + 1)) & mask;
			int d = (int)roundf((2.0f * (float)fabs(svr - sv)) * 64.0f);
			for (int j = 0; j < bt->nbits; j++) {
				int vp = sym_bit(bt, sp, j), vs = sym_bit(bt, ss, j);
				orc_sbit_t v = (orc_sbit_t)(127 - ((vp ^ vs) ? d : (d >> 1)));
				ebits[k++] = vp ? (orc_sbit_t)-v : v;
			}
		}
	}
}

/* pi4cxpsk.c:520-602 */
int orc_pi4cxpsk_demod(const struct orc_burst *bt, const orc_cf *in, int in_len,
                       int sps, float freq_shift, orc_sbit_t *ebits,
                       int *sync_id_p, float *toa_p, float *freq_err_p, float *ssyms_out)
{
	orc_cf *burst = malloc(sizeof(orc_cf) * (size_t)in_len);
	float *ssyms = malloc(sizeof(float) * (size_t)bt->len);
	float toa, ffe, d;
	orc_cf ph;
	int sync_id, rv = 0;

	orc_sig_normalize(in, in_len, 1, (freq_shift - bt->rotation) / (float)sps, burst);

	sync_id = sync_find(bt, burst, in_len, sps, &toa, NULL);
	if (sync_id < 0) { rv = sync_id; goto out; }
	if (sync_id_p) *sync_id_p = sync_id;
	if (toa_p) *toa_p = toa;

	align(bt, burst, in_len, sps, toa);

	ffe = freq_err(bt, burst, sync_id);
	if (freq_err_p) *freq_err_p = ffe;
	if (ffe != 0.0f)
		orc_rotate(burst, bt->len, -ffe);

	ph = phase(bt, burst, sync_id);
	orc_scale(burst, bt->len, conjf(ph));

	/* pi4cxpsk.c:442-460 */
	d = (2.0f * PIf) / (float)(1 << bt->nbits);
	for (int i = 0; i < bt->len; i++)
		ssyms[i] = cargf(burst[i]) / d;
	if (ssyms_out)
		memcpy(ssyms_out, ssyms, sizeof(float) * (size_t)bt->len);

	soft_bits(bt, ssyms, ebits);
out:
	free(ssyms);
	free(burst);
	return rv;
}

/* pi4cxpsk.c:617-682 */
int orc_pi4cxpsk_detect(const struct orc_burst *const *bts, int n_bts, float e_toa,
                        const orc_cf *in, int in_len, int sps, float freq_shift,
                        int *bt_id_p, int *sync_id_p, float *toa_p)
{
	orc_cf *burst = malloc(sizeof(orc_cf) * (size_t)in_len);
	int p_id = -1, p_sid = -1, rv = 0;
	float p_toa = 0.0f, p_pwr = 0.0f;

	orc_sig_normalize(in, in_len, 1, (freq_shift - bts[0]->rotation) / (float)sps, burst);
	for (int id = 0; id < n_bts; id++) {
		float toa, pwr;
		int sid = sync_find(bts[id], burst, in_len, sps, &toa, &pwr);
		if (sid < 0) { rv = sid; goto out; }
		if (e_toa >= 0.0f)
			pwr /= (float)fabs(e_toa - toa);
		if (pwr > p_pwr) {
			p_id = id; p_sid = sid; p_pwr = pwr; p_toa = toa;
		}
	}
	if (bt_id_p) *bt_id_p = p_id;
	if (sync_id_p) *sync_id_p = p_sid;
	if (toa_p) *toa_p = p_toa;
out:
	free(burst);
	return rv;
}

/* pi4cxpsk.c:693-729 */
int orc_pi4cxpsk_mod_order(const orc_cf *in, int in_len, int sps, float freq_shift)
{
	orc_cf *burst = malloc(sizeof(orc_cf) * (size_t)in_len);
	orc_cf sb = 0.0f, sq = 0.0f;
	orc_sig_normalize(in, in_len, 1, (freq_shift - (PIf / 4)) / (float)sps, burst);
	for (int i = 0; i < in_len; i++) {
		orc_cf v = burst[i];
		v = (v * v) / normsq(v);
		sb += v;
		sq += v * v;
	}
	free(burst);
	return normsq(sb) < (normsq(sq) / 2.0f) ? 4 : 2;
}

/* pi4cxpsk.c:741-799 (1 sample per symbol) */
int orc_pi4cxpsk_mod(const struct orc_burst *bt, const orc_ubit_t *ebits, int sync_id, orc_cf *out)
{
	int k = 0;
	for (int i = 0; i < bt->len; i++)
		out[i] = 0.0f;
	for (int c = 0; c < bt->n_sync_chunks[sync_id]; c++) {
		const struct orc_chunk *cs = &bt->sync[sync_id][c];
		for (int i = 0; i < cs->len; i++)
			out[cs->pos + i] = sym_val(bt, cs->syms[i]);
	}
	for (int c = 0; c < bt->n_data; c++) {
		const struct orc_chunk *dc = &bt->data[c];
		for (int i = 0; i < dc->len; i++) {
			int v = 0;
			for (int j = 0; j < bt->nbits; j++)
				v = (v << 1) | ebits[k++];
			out[dc->pos + i] = sym_val(bt, bits_sym(bt, v));
		}
	}
	orc_rotate(out, bt->len, bt->rotation);
	return 0;
}

/* ------------------------------------------------------------------------ */
/* FCCH                                                                      */
/* ------------------------------------------------------------------------ */

const struct orc_fcch_burst orc_fcch_burst        = { 0.32f, 117 };  /* fcch.c:50-53 */
const struct orc_fcch_burst orc_fcch3_lband_burst = { 0.32f, 468 };  /* fcch.c:59-62 */
const struct orc_fcch_burst orc_fcch3_sband_burst = { 0.16f, 468 };  /* fcch.c:67-70 */

/* fcch.c:92-121: sign=+1 "up" (up_down=0), -1 "down" ; sps fixed to 1 by all callers */
static void gen_chirp(const struct orc_fcch_burst *bt, int sign, orc_cf *out)
{
	float sq2d2 = sqrtf(2.0f) / 2.0f;
	float phase_base = bt->freq * 2.0f * PIf / (float)bt->len;
	float halfpos = (float)bt->len / 2.0f;
	if (sign < 0)
		phase_base *= -1.0f;
	for (int i = 0; i < bt->len; i++) {
		float pos = ((float)i / 1.0f) - halfpos;
		float ph = phase_base * (pos * pos);
		out[i] = sq2d2 * (cosf(ph) + I * sinf(ph));
	}
}

/* fcch.c:167-193 (real-only) */
static void gen_dual_chirp(const struct orc_fcch_burst *bt, orc_cf *out)
{
	float sq2 = sqrtf(2.0f);
	float phase_base = bt->freq * 2.0f * PIf / (float)bt->len;
	float halfpos = (float)bt->len / 2.0f;
	for (int i = 0; i < bt->len; i++) {
		float pos = ((float)i / 1.0f) - halfpos;
		out[i] = sq2 * cosf(phase_base * (pos * pos));
	}
}

/* fcch.c:211-250 */
int orc_fcch_rough(const struct orc_fcch_burst *bt, const orc_cf *in, int in_len,
                   int sps, float freq_shift, int *toa)
{
	orc_cf *ref = malloc(sizeof(orc_cf) * (size_t)bt->len);
	orc_cf *win = malloc(sizeof(orc_cf) * (size_t)(in_len / sps + 1));
	orc_cf *corr;
	int l, cl;
	float pos;

	gen_dual_chirp(bt, ref);
	l = orc_sig_normalize(in, in_len, sps, freq_shift, win);
	corr = malloc(sizeof(orc_cf) * (size_t)(l > bt->len ? l : bt->len));
	cl = orc_correlate(ref, bt->len, win, l, 1, corr);
	pos = orc_peak_energy_find(corr, cl, 5, ORC_PEAK_WEIGH_WIN, NULL);
	*toa = (int)round(pos * sps);
	free(corr); free(win); free(ref);
	return 0;
}

/* fcch.c:264-326 */
static void peak_record(const struct orc_fcch_burst *bt, int *toa, float *pwr, int *n,
                        int N, int Lp, int sps, int peak_toa, float peak_pwr)
{
	int i, j, has_dupe = 0;
	for (i = 0; i < *n; i++) {
		int th = (bt->len * sps) >> 1;
		int d = (toa[i] % Lp) - (peak_toa % Lp);
		if (abs(d) > th)
			continue;
		if (pwr[i] > peak_pwr) {
			if (!has_dupe)
				has_dupe = 1;
			continue;
		}
		for (j = i; j < (*n) - 1; j++) {
			toa[j] = toa[j + 1];
			pwr[j] = pwr[j + 1];
		}
		*n = *n - 1;
		has_dupe = -1;
	}
	if (has_dupe > 0)
		return;
	for (i = 0; i < *n; i++)
		if (peak_pwr > pwr[i])
			break;
	if (i == N)
		return;
	for (j = N - 1; j > i; j--) {
		toa[j] = toa[j - 1];
		pwr[j] = pwr[j - 1];
	}
	toa[i] = peak_toa;
	pwr[i] = peak_pwr;
	if (*n != N)
		*n = *n + 1;
}

/* fcch.c:341-496 */
int orc_fcch_rough_multi(const struct orc_fcch_burst *bt, const orc_cf *in, int in_len,
                         int sps, float freq_shift, int *peaks_toa, int N)
{
	orc_cf *ref, *win, *corr;
	float *cp, pwr_max, pwrs[2], peaks[2], avg, stddev, th;
	float *peaks_pwr;
	int Lw, Lp, nLp, l, cl, pwr_max_idx, a, peaks_cnt, rv;

	if (in_len < ((650 * SYM_RATE * sps) / 1000))
		return -EINVAL;

	ref = malloc(sizeof(orc_cf) * (size_t)bt->len);
	win = malloc(sizeof(orc_cf) * (size_t)(in_len / sps + 1));
	peaks_pwr = calloc((size_t)N, sizeof(float));
	gen_dual_chirp(bt, ref);
	l = orc_sig_normalize(in, in_len, sps, freq_shift, win);
	corr = malloc(sizeof(orc_cf) * (size_t)l);
	cl = orc_correlate(ref, bt->len, win, l, 1, corr);
	/* The reference reads corr_pwr[i+Lp] for i < Lw (fcch.c:438) which runs up to 8 entries past
	 * the array when the measured period exceeds 7488 on a minimum-length (650 ms) window; those
	 * reads are defined as 0 here. */
	cp = calloc((size_t)cl + 64, sizeof(float));

	Lw = (320 * SYM_RATE) / 1000 + bt->len;
	Lp = (320 * SYM_RATE) / 1000;

	pwr_max_idx = 0;
	pwr_max = 0.0f;
	for (int i = 0; i < cl; i++) {
		float e = normsq(corr[i]);
		cp[i] = e;
		if (e > pwr_max && i < Lw) { pwr_max = e; pwr_max_idx = i; }
	}

	pwrs[0] = pwrs[1] = peaks[0] = peaks[1] = 0.0f;
	for (int i = -10; i <= 10; i++) {
		int j = pwr_max_idx + i;
		if (j > 0 && j < cl) { pwrs[0] += cp[j]; peaks[0] += cp[j] * (float)j; }
		j += Lp;
		if (j > 0 && j < cl) { pwrs[1] += cp[j]; peaks[1] += cp[j] * (float)j; }
	}
	peaks[0] /= pwrs[0];
	peaks[1] /= pwrs[1];
	nLp = (int)round(peaks[1] - peaks[0]);
	if (abs(nLp - Lp) > 10) { rv = -EINVAL; goto out; }
	Lp = nLp;

	avg = 0.0f;
	for (int i = 0; i < Lw; i++) {
		float v = sqrtf(cp[i] * cp[i + Lp]);
		cp[i] = v;
		avg += v;
	}
	avg /= (float)Lw;
	stddev = 0.0f;
	for (int i = 0; i < Lw; i++) {
		float v = cp[i] - avg;
		stddev += v * v;
	}
	stddev = sqrtf(stddev / (float)Lw);
	th = avg + 3.0f * stddev;

	peaks_cnt = 0;
	a = 0;
	for (int i = 1; i < Lw - 1; i++) {
		if (cp[i] > th) {
			float p_pwr, p_fpos;
			int p_pos;
			if (a) continue;
			a = 1;
			p_pwr = cp[i - 1] + cp[i] + cp[i + 1];
			p_fpos = (-cp[i - 1] + cp[i + 1]) / p_pwr;
			p_pos = (int)round(((float)i + p_fpos) * (float)sps);
			peak_record(bt, peaks_toa, peaks_pwr, &peaks_cnt, N, Lp, sps, p_pos, p_pwr);
		} else {
			a = 0;
		}
	}
	rv = peaks_cnt;
out:
	free(cp); free(corr); free(peaks_pwr); free(win); free(ref);
	return rv;
}

/* fcch.c:512-628 */
int orc_fcch_fine(const struct orc_fcch_burst *bt, const orc_cf *in, int in_len,
                  int sps, float freq_shift, int *toa, float *freq_error)
{
	int len = bt->len, mid;
	orc_cf *up, *down, *burst, *mu, *md;
	float bin_hz, peak_up, peak_down, freq_err_hz, chirp_rate, toa_ms, toa_samples;

	if (in_len / sps != len)
		return -EINVAL;
	up = malloc(sizeof(orc_cf) * (size_t)len);
	down = malloc(sizeof(orc_cf) * (size_t)len);
	burst = malloc(sizeof(orc_cf) * (size_t)len);
	mu = malloc(sizeof(orc_cf) * (size_t)len);
	md = malloc(sizeof(orc_cf) * (size_t)len);
	gen_chirp(bt, +1, up);
	gen_chirp(bt, -1, down);
	orc_sig_normalize(in, in_len, sps, freq_shift, burst);
	for (int i = 0; i < len; i++) {
		mu[i] = burst[i] * up[i];
		md[i] = burst[i] * down[i];
	}
	mid = (int)(float)(len >> 1);
	for (int i = 0; i < len; i++) {
		/* float phase, double cexp, result narrowed to float (fcch.c:575-580) */
		float phf = 2.0f * PIf * (float)mid / (float)len * (float)i;
		orc_cf fs = (orc_cf)(cos((double)phf) + I * sin((double)phf));
		mu[i] *= fs;
		md[i] *= fs;
	}
	orc_dft_forward(mu, len);
	orc_dft_forward(md, len);
	peak_up = orc_peak_energy_find(mu, len, 5, ORC_PEAK_WEIGH_WIN, NULL);
	peak_down = orc_peak_energy_find(md, len, 5, ORC_PEAK_WEIGH_WIN, NULL);
	bin_hz = (float)SYM_RATE / (float)len;
	peak_up = (peak_up - (float)mid) * bin_hz;
	peak_down = (peak_down - (float)mid) * bin_hz;
	freq_err_hz = (peak_up + peak_down) / 2.0f;
	*freq_error = (2.0f * PIf * freq_err_hz) / SYM_RATE;
	chirp_rate = (2.0f * bt->freq * SYM_RATE * SYM_RATE) / (float)(bt->len * 1000);
	toa_ms = ((peak_up - peak_down) / 2.0f) / chirp_rate;
	toa_samples = (toa_ms * SYM_RATE * (float)sps) / 1000.0f;
	*toa = (int)round(toa_samples);
	free(md); free(mu); free(burst); free(down); free(up);
	return 0;
}

/* fcch.c:643-708 */
int orc_fcch_snr(const struct orc_fcch_burst *bt, const orc_cf *in, int in_len,
                 int sps, float freq_shift, float *snr)
{
	int len = bt->len, peaks[6];
	orc_cf *ref, *burst;

	if (in_len / sps != len)
		return -EINVAL;
	ref = malloc(sizeof(orc_cf) * (size_t)len);
	burst = malloc(sizeof(orc_cf) * (size_t)len);
	gen_dual_chirp(bt, ref);
	orc_sig_normalize(in, in_len, sps, freq_shift, burst);
	for (int i = 0; i < len; i++)
		burst[i] *= crealf(ref[i]);
	orc_dft_forward(burst, len);
	orc_peaks_scan(burst, len, peaks, 6);
	*snr = (normsq(burst[peaks[0]]) + normsq(burst[peaks[1]])) /
	       (normsq(burst[peaks[4]]) + normsq(burst[peaks[5]]));
	free(burst); free(ref);
	return 0;
}

/* ------------------------------------------------------------------------ */
/* Batch driver (tests + bench cpu_baseline only)                            */
/* ------------------------------------------------------------------------ */

void orc_demod_decode_batch(int n, const orc_cf *iq, const uint64_t *offset,
                            const uint8_t *kind, int sps, const float *freq_shift,
                            uint8_t *l2, int32_t *crc, int32_t *conv,
                            float *toa, float *freq_err_out,
                            orc_sbit_t *ebits_out, float *ssyms_out, int32_t *rv_out)
{
	const struct orc_burst *bcch = orc_burst_get(ORC_BURST_BCCH);
	const struct orc_burst *dc6 = orc_burst_get(ORC_BURST_DC6);

	for (int i = 0; i < n; i++) {
		const struct orc_burst *bt = kind[i] ? dc6 : bcch;
		int win = kind[i] ? 10 * sps : 20 * sps;   /* gmr1_rx.c:759,809 */
		int in_len = bt->len * sps + win;
		orc_sbit_t eb[432];
		float ss[234];
		float t = 0.0f, fe = 0.0f;
		int cv = 0, rv, c;

		memset(eb, 0, sizeof(eb));
		memset(l2 + 24 * i, 0, 24);
		rv = orc_pi4cxpsk_demod(bt, iq + offset[i], in_len, sps,
		                        freq_shift ? freq_shift[i] : 0.0f,
		                        eb, NULL, &t, &fe, ss);
		if (rv_out) rv_out[i] = rv;
		if (rv) {
			crc[i] = -1; conv[i] = 0; toa[i] = 0; freq_err_out[i] = 0;
			if (ebits_out) memset(ebits_out + 432 * i, 0, 432);
			if (ssyms_out) memset(ssyms_out + 234 * i, 0, 234 * sizeof(float));
			continue;
		}
		c = kind[i] ? orc_ccch_decode(l2 + 24 * i, eb, &cv)
		            : orc_bcch_decode(l2 + 24 * i, eb, &cv);
		crc[i] = c; conv[i] = cv; toa[i] = t; freq_err_out[i] = fe;
		if (ebits_out) memcpy(ebits_out + 432 * i, eb, 432);
		if (ssyms_out) memcpy(ssyms_out + 234 * i, ss, 234 * sizeof(float));
	}
}
