/* oracle/orc_ambe.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's GMR-1 AMBE speech decoder (10-byte frame -> 160 samples of 8 kHz PCM), the
 * checker of the GPU vocoder kernel.  Pinned: `make -C oracle ref` compiles the reference's own src/codec into
 * oracle/_ref/libgmr1_codec_ref.so and tests/test_oracle_ambe.py requires this file to give the same samples,
 * bit for bit, on random, structured, tone, silence and mixed streams; tests/golden/ambe_vectors.npz holds outputs
 * of that reference build for the GPU box, where /root/reference does not exist.
 *
 * Every sum below adds its terms in the reference's order and every product keeps its grouping: float arithmetic
 * is not associative and the cosine is a 1024-entry table indexed by a truncated product, so one changed rounding
 * moves a table index.  Build with -ffp-contract=off (oracle/Makefile).
 *
 *   frame layout, parameter decode    src/codec/frame.c
 *   cosine table, DCT / DFT            src/codec/math.c
 *   enhancement, unvoiced / voiced     src/codec/synth.c
 *   tone frames                        src/codec/tone.c
 *   frame dispatch, state              src/codec/ambe.c, src/codec/private.h:89-111
 */
#include <errno.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "orc_ambe.h"
#include "orc_ambe_tables.h"

#define PI_F 3.141592653589793f   /* private.h:117 */

static float bits_f(uint32_t w)
{
	float f;
	memcpy(&f, &w, 4);
	return f;
}

/* ---- cosine by table (math.c:40-69) ---- */

static float g_cos[1024];
static float g_win[121];            /* synthesis window: 0.025 steps up over 40 samples, flat, down (synth.c:36-54) */
static int g_ready;

static void tables_once(void)
{
	if (g_ready)
		return;
	for (int i = 0; i < 1024; i++)
		g_cos[i] = cosf((PI_F * i) / 512.0f);
	/* the reference lists the window as literals 0.000f, 0.025f ...: a correctly rounded quotient of the exact
	 * integers is the float nearest to that decimal, i.e. the literal's value */
	for (int i = 0; i < 121; i++) {
		int k = i < 40 ? i : i > 80 ? 120 - i : 40;
		g_win[i] = (float)(25 * k) / 1000.0f;
	}
	g_ready = 1;
}

static inline float tcos(float a)
{
	const float scale = 512.0f / PI_F;
	return g_cos[(int)(a * scale) & 1023];
}

static inline float tsin(float a)
{
	const float scale = 512.0f / PI_F;
	return g_cos[((int)(a * scale) + 768) & 1023];
}

/* inverse DCT, M coefficients -> N points (math.c:99-114) */
static void inv_dct(float *out, const float *coef, int N, int M)
{
	for (int n = 0; n < N; n++) {
		float acc = coef[0];
		for (int k = 1; k < M; k++)
			acc += 2.0f * coef[k] * tcos((PI_F / N) * k * (n + .5f));
		out[n] = acc;
	}
}

/* ---- frame fields (frame.c:40-75) ---- */

static unsigned field(const uint8_t *fr, int pos, int len, int up)
{
	/* `len` bits starting at bit `pos` (MSB first), moved up by `up`; the reference narrows to 8 bits before masking */
	uint8_t v;
	int in_byte = pos & 7;
	if (in_byte + len > 8)
		v = (uint8_t)(((fr[pos >> 3] << 8) | fr[(pos >> 3) + 1]) >> (16 - in_byte - len));
	else
		v = (uint8_t)(fr[pos >> 3] >> (8 - in_byte - len));
	return (uint8_t)((v & ((1 << len) - 1)) << up);
}

struct raw {
	unsigned pitch, pitch_rule, gain, vuv, prba12, prba34, prba57, hoc[4], mag_rule, perr14, perr58;
};

static void unpack(struct raw *r, const uint8_t *fr)
{
	r->pitch = field(fr, 0, 7, 0);
	r->pitch_rule = field(fr, 48, 2, 0);
	r->gain = (uint8_t)(field(fr, 7, 6, 2) | field(fr, 50, 2, 0));
	r->vuv = field(fr, 13, 6, 0);
	r->prba12 = (uint8_t)(field(fr, 19, 6, 1) | field(fr, 52, 1, 0));
	r->prba34 = (uint8_t)(field(fr, 25, 3, 3) | field(fr, 53, 3, 0));
	r->prba57 = (uint8_t)(field(fr, 28, 3, 4) | field(fr, 56, 4, 0));
	r->hoc[0] = (uint8_t)(field(fr, 31, 3, 4) | field(fr, 60, 4, 0));
	r->hoc[1] = (uint8_t)(field(fr, 34, 3, 3) | field(fr, 64, 3, 0));
	r->hoc[2] = (uint8_t)(field(fr, 37, 2, 4) | field(fr, 67, 4, 0));
	r->hoc[3] = (uint8_t)(field(fr, 39, 2, 3) | field(fr, 71, 3, 0));
	r->mag_rule = field(fr, 46, 2, 0);
	r->perr14 = (uint8_t)(field(fr, 41, 3, 3) | field(fr, 74, 3, 0));
	r->perr58 = (uint8_t)(field(fr, 44, 2, 3) | field(fr, 77, 3, 0));
}

void orc_ambe_unpack(const uint8_t *frame, unsigned out[14])
{
	struct raw r;
	unpack(&r, frame);
	unsigned v[14] = {r.pitch, r.pitch_rule, r.gain, r.vuv, r.prba12, r.prba34, r.prba57, r.hoc[0], r.hoc[1],
	                  r.hoc[2], r.hoc[3], r.mag_rule, r.perr14, r.perr58};
	memcpy(out, v, sizeof(v));
}

/* ---- parameter decode (frame.c:77-338) ---- */

/* log2 of the first subframe's fundamental from the two frames' pitch values (frame.c:79-118) */
float orc_ambe_f0log_sf0(float before, float now, int rule)
{
	if (now != before) {
		switch (rule) {
		case 0: return now;
		case 1: return (0.65f * now) + (0.35f * before);
		case 2: return (now + before) / 2.0f;
		default: return before;
		}
	}
	const float step = 4.2672e-2f;
	switch (rule) {
	case 0:
	case 1: return now;
	case 2: return now + step;
	default: return now - step;
	}
}

float orc_ambe_f0log_sf1(int pitch)
{
	return -4.312f - 2.1336e-2f * pitch;   /* frame.c:300 */
}

int orc_ambe_harmonics(float f0)
{
	int L = (int)floorf(0.4751f / f0);     /* frame.c:125-130 */
	return L < 9 ? 9 : L > 56 ? 56 : L;
}

static void set_harmonics(struct orc_ambe_sub *s)
{
	s->L = orc_ambe_harmonics(s->f0);
	for (int b = 0; b < 4; b++)
		s->Lb[b] = orc_ambe_hpg[(s->L - 9) * 4 + b];
}

/* magnitudes of a subframe with Ls harmonics seen on a grid of Ld harmonics, mean removed (frame.c:140-171) */
static void regrid(float *dst, int Ld, const float *src, int Ls)
{
	float mean = 0.0f;
	const float step = (float)Ls / (float)Ld;
	float at = step;
	for (int i = 0; i < Ld; i++) {
		int k = (int)floorf(at);
		if (k == 0)
			dst[i] = src[0];
		else if (k >= Ls)
			dst[i] = src[Ls - 1];
		else {
			float frac = at - k;
			dst[i] = src[k - 1] * (1.0f - frac) + src[k] * frac;
		}
		mean += dst[i];
		at += step;
	}
	mean /= Ld;
	for (int i = 0; i < Ld; i++)
		dst[i] -= mean;
}

/* second subframe: prediction from the previous frame + PRBA / HOC blocks (frame.c:175-242) */
static void mags_sf1(struct orc_ambe_sub *s, const struct orc_ambe_sub *before, const struct raw *r)
{
	regrid(s->Mlog, s->L, before->Mlog, before->L);
	for (int i = 0; i < s->L; i++)
		s->Mlog[i] *= 0.65f;

	float g[8], R[8];
	g[0] = 0.0f;
	g[1] = bits_f(orc_ambe_prba12[r->prba12 * 2 + 0]);
	g[2] = bits_f(orc_ambe_prba12[r->prba12 * 2 + 1]);
	g[3] = bits_f(orc_ambe_prba34[r->prba34 * 2 + 0]);
	g[4] = bits_f(orc_ambe_prba34[r->prba34 * 2 + 1]);
	g[5] = bits_f(orc_ambe_prba57[r->prba57 * 3 + 0]);
	g[6] = bits_f(orc_ambe_prba57[r->prba57 * 3 + 1]);
	g[7] = bits_f(orc_ambe_prba57[r->prba57 * 3 + 2]);
	inv_dct(R, g, 8, 8);

	const uint32_t *hoc[4] = {&orc_ambe_hoc0[r->hoc[0] * 4], &orc_ambe_hoc1[r->hoc[1] * 4],
	                          &orc_ambe_hoc2[r->hoc[2] * 4], &orc_ambe_hoc3[r->hoc[3] * 4]};
	const float half_rsqrt2 = (1.0f / (2.0f * (float)M_SQRT2));
	float weighted = 0.0f;
	int at = 0;
	for (int b = 0; b < 4; b++) {
		float C[6], c[17];
		C[0] = (R[2 * b] + R[2 * b + 1]) * 0.5f;
		C[1] = (R[2 * b] - R[2 * b + 1]) * half_rsqrt2;
		for (int k = 0; k < 4; k++)
			C[2 + k] = bits_f(hoc[b][k]);
		inv_dct(c, C, s->Lb[b], 6);
		for (int j = 0; j < s->Lb[b]; j++)
			s->Mlog[at++] += c[j];
		weighted += C[0] * s->Lb[b];
	}
	const float shift = s->gain - (0.5f * log2f(s->L)) - (weighted / s->L);
	for (int i = 0; i < s->L; i++)
		s->Mlog[i] += shift;
}

/* first subframe: between the previous frame and this frame's second subframe + correction (frame.c:246-286) */
static void mags_sf0(struct orc_ambe_sub *s, const struct orc_ambe_sub *before, const struct orc_ambe_sub *after,
                     const struct raw *r)
{
	float from_before[56], from_after[56], e[9], fix[56];
	regrid(from_before, s->L, before->Mlog, before->L);
	regrid(from_after, s->L, after->Mlog, after->L);
	const float a = bits_f(orc_ambe_sf0_interp[r->mag_rule]);
	e[0] = 0.0f;
	for (int k = 0; k < 4; k++) {
		e[1 + k] = bits_f(orc_ambe_sf0_perr14[r->perr14 * 4 + k]);
		e[5 + k] = bits_f(orc_ambe_sf0_perr58[r->perr58 * 4 + k]);
	}
	inv_dct(fix, e, s->L, 9);
	const float level = s->gain - (0.5f * log2f(s->L));
	for (int i = 0; i < s->L; i++)
		s->Mlog[i] = level + fix[i] + (a * from_before[i]) + ((1.0f - a) * from_after[i]);
}

static void decode_params(struct orc_ambe_sub sf[2], const struct orc_ambe_sub *before, const struct raw *r)
{
	sf[1].f0log = orc_ambe_f0log_sf1((int)r->pitch);
	sf[1].f0 = powf(2.0f, sf[1].f0log);
	sf[0].f0log = orc_ambe_f0log_sf0(before->f0log, sf[1].f0log, (int)r->pitch_rule);
	sf[0].f0 = powf(2.0f, sf[0].f0log);
	set_harmonics(&sf[0]);
	set_harmonics(&sf[1]);

	const unsigned pat = orc_ambe_vuv[r->vuv];   /* low byte: first subframe, MSB = lowest band (frame.c:313-318) */
	for (int b = 0; b < 8; b++) {
		sf[0].band_v[b] = (pat >> (7 - b)) & 1;
		sf[1].band_v[b] = (pat >> (15 - b)) & 1;
	}

	for (int k = 0; k < 2; k++) {
		sf[k].gain = (0.5f * before->gain) + bits_f(orc_ambe_gain[r->gain * 2 + k]);
		if (sf[k].gain > 13.0f)
			sf[k].gain = 13.0f;
	}
	mags_sf1(&sf[1], before, r);
	mags_sf0(&sf[0], before, &sf[1], r);
}

/* per-harmonic voicing and linear magnitudes (frame.c:342-359) */
static void expand(struct orc_ambe_sub *s)
{
	s->w0 = s->f0 * (2.0f * PI_F);
	const float unv = 0.2046f / sqrtf(s->w0);
	for (int l = 0; l < s->L; l++) {
		int band = (int)(l * 16.0f * s->f0);
		if (band > 7)
			band = 7;               /* same first-frame case: the reference reads past v_uv[8] (D10) */
		s->V[l] = s->band_v[band];
		s->M[l] = powf(2.0, s->Mlog[l]) / 6.0f;
		if (!s->V[l])
			s->M[l] *= unv;
	}
}

/* ---- synthesis (synth.c) ---- */

/* spectral enhancement of the magnitudes (synth.c:314-379) */
static void enhance(struct orc_ambe_dec *d, struct orc_ambe_sub *s)
{
	float r0 = 0.0f, r1 = 0.0f;
	for (int l = 0; l < s->L; l++) {
		float p = s->M[l] * s->M[l];
		r0 += p;
		r1 += p * tcos(s->w0 * (l + 1));
	}
	const float k1 = 0.96f * PI_F / (s->w0 * r0 * (r0 * r0 - r1 * r1));
	const float k2 = r0 * r0 + r1 * r1;
	const float k3 = 2.0f * r0 * r1;
	float after = 0.0f;
	for (int l = 0; l < s->L; l++) {
		float w;
		if ((l + 1) * 8 <= s->L)
			w = 1.0f;
		else {
			w = sqrtf(s->M[l]) * powf(k1 * (k2 - k3 * tcos(s->w0 * (l + 1))), 0.25f);
			if (w > 1.2f)
				w = 1.2f;
			else if (w < 0.5f)
				w = 0.5f;
		}
		s->M[l] *= w;
		after += s->M[l] * s->M[l];
	}
	const float norm = sqrtf(r0 / after);
	for (int l = 0; l < s->L; l++)
		s->M[l] *= norm;
	d->SE = 0.95f * d->SE + 0.05f * r0;
	if (d->SE < 1e4f)
		d->SE = 1e4f;
}

/* noise: 121 numbers of the spec's generator, continuing 80 further on each subframe (synth.c:98-110, 127-128).
 * The carried value is kept in an int16_t but handed over as a uint16_t: numbers >= 32768 survive the round trip. */
static void noise(uint16_t *u, int16_t *carry)
{
	uint32_t x = (uint16_t)*carry;
	for (int i = 0; i < 121; i++) {
		x = (x * 171 + 11213) % 53125;
		u[i] = (uint16_t)x;
	}
	*carry = (int16_t)u[79];
}

/* unvoiced part: shaped noise, 128-point DFT pair, overlap-add with the previous subframe (synth.c:114-214) */
static void unvoiced(struct orc_ambe_dec *d, float *out, const struct orc_ambe_sub *s)
{
	uint16_t u[121];
	float t[121], re[65], im[65];
	noise(u, &d->u_last);
	for (int i = 0; i < 121; i++)
		t[i] = (float)u[i] * g_win[i];

	for (int k = 0; k <= 64; k++) {                /* math.c:118-138 */
		float a = 0.0f, b = 0.0f;
		for (int n = 0; n < 121; n++) {
			float ang = (-2.0f * PI_F / 128) * k * n;
			a += t[n] * tcos(ang);
			b += t[n] * tsin(ang);
		}
		re[k] = a;
		im[k] = b;
	}

	/* Band edges past the last bin: the reference walks off its 65-entry arrays there (synth.c:141-171).  It happens
	 * only when a stream's FIRST speech frame asks for pitch interpolation (rules 1-3 mix in the initial f0log of 0,
	 * giving f0 = 0.14 ... 1.0, frame.c:303-305); defined here as "no such bins" (decision D10). */
	int hi = ceilf(128.0f / (2 * PI_F) * (.5f) * s->w0);
	if (hi > 65)
		hi = 65;
	for (int k = 0; k < hi; k++)
		re[k] = im[k] = 0.0f;
	for (int l = 0; l < s->L; l++) {
		int lo = hi;
		hi = ceilf(128.0f / (2 * PI_F) * (l + 1.5f) * s->w0);
		if (hi > 65)
			hi = 65;
		float e = 0.0f;
		for (int k = lo; k < hi; k++)
			e += re[k] * re[k] + im[k] * im[k];
		const float scale = 76.89f * s->M[l] / sqrtf(e / (hi - lo));
		for (int k = lo; k < hi; k++) {
			if (s->V[l])
				re[k] = im[k] = 0.0f;
			else {
				re[k] *= scale;
				im[k] *= scale;
			}
		}
	}
	for (int k = hi; k <= 64; k++)
		re[k] = im[k] = 0.0f;

	for (int n = 0; n < 121; n++) {               /* math.c:142-163 */
		float acc = 0.0f;
		for (int k = 0; k <= 64; k++) {
			float ang = (-2.0f * PI_F / 128) * k * n;
			float twice = (k == 0 || k == 64) ? 1.0f : 2.0f;
			acc += twice * (re[k] * tcos(ang) + im[k] * tsin(ang));
		}
		t[n] = acc / 128;
	}

	for (int i = 0; i < 21; i++)
		out[i] = d->uw_last[i + 60];
	for (int i = 21; i < 60; i++)
		out[i] = (g_win[i + 60] * d->uw_last[i + 60] + g_win[i - 20] * t[i - 20]) /
		         (g_win[i + 60] * g_win[i + 60] + g_win[i - 20] * g_win[i - 20]);
	for (int i = 60; i < 80; i++)
		out[i] = t[i - 20];
	memcpy(d->uw_last, t, sizeof(t));
}

/* voiced part: one oscillator per harmonic, phase-continuous where it can be (synth.c:218-302) */
static void voiced(struct orc_ambe_dec *d, float *out, const struct orc_ambe_sub *s, const struct orc_ambe_sub *before)
{
	memset(out, 0, 80 * sizeof(float));
	const int Lmax = before->L > s->L ? before->L : s->L;
	int n_unv = 0;
	for (int l = 0; l < Lmax; l++)
		n_unv += s->V[l] ? 0 : 1;
	d->psi1 = remainderf(d->psi1 + (s->w0 + before->w0) * 40.0f, 2 * PI_F);

	for (int l = 0; l < Lmax; l++) {
		const int v_now = l >= s->L ? 0 : s->V[l];
		const int v_was = l >= before->L ? 0 : before->V[l];
		const float m_now = l >= s->L ? 0.0f : s->M[l];
		const float m_was = l >= before->L ? 0.0f : before->M[l];
		const float w_now = (l + 1) * s->w0;
		const float w_was = (l + 1) * before->w0;
		const float ph_was = d->phi[l];
		float ph_now = d->psi1 * (l + 1);
		if (l >= (s->L / 4))
			ph_now += ((float)n_unv / (float)s->L) * bits_f(orc_ambe_rho[l]);
		d->phi[l] = ph_now;

		const int smooth = v_now && v_was && (l < 7) && (fabsf(w_now - w_was) < (.1f * w_now));
		if (smooth) {
			const float dm = (m_now - m_was) / 80.0f;
			const float dp = ph_now - ph_was - (w_now + w_was) * 40.0f;
			const float dw = (dp - 2 * PI_F * floorf((dp + PI_F) / (2 * PI_F))) / 80.0f;
			const float ta = w_was + dw;
			const float tb = (w_now - w_was) / 160.0f;
			for (int i = 0; i < 80; i++)
				out[i] += (m_was + i * dm) * tcos(ph_was + (ta + tb * i) * i);
		}
		if (!smooth && v_now)
			for (int i = 21; i < 80; i++)
				out[i] += g_win[i - 20] * m_now * tcos(ph_now + w_now * (i - 80));
		if (!smooth && v_was)
			for (int i = 0; i < 60; i++)
				out[i] += g_win[i + 60] * m_was * tcos(ph_was + w_was * i);
	}
	for (int l = Lmax; l < 56; l++)
		d->phi[l] = (d->psi1 * (l + 1)) + (((float)n_unv / (float)s->L) * bits_f(orc_ambe_rho[l]));
}

static void subframe_audio(struct orc_ambe_dec *d, int16_t *pcm, const struct orc_ambe_sub *s,
                           const struct orc_ambe_sub *before)
{
	float nu[80], vo[80];
	unvoiced(d, nu, s);
	voiced(d, vo, s, before);
	for (int i = 0; i < 80; i++)
		pcm[i] = (int16_t)((nu[i] + 2.0f * vo[i]) * 4.0f);   /* synth.c:381-395 */
}

/* ---- tone frames (tone.c) ---- */

static void tone_freqs(int code, int *f1, int *f2)
{
	static const int dtmf_col[4] = {1209, 1336, 1477, 1633}, dtmf_row[4] = {697, 770, 852, 941};
	static const int knox_col[4] = {1052, 1162, 1297, 1430}, knox_row[4] = {606, 672, 743, 820};
	static const int prog[4][2] = {{440, 350}, {480, 440}, {630, 480}, {490, 350}};
	const int k = code & 0xf;
	if (code >= 0xa0) {
		*f1 = prog[k][0];
		*f2 = prog[k][1];
	} else if (code >= 0x90) {
		*f1 = knox_col[k >> 2];
		*f2 = knox_row[k & 3];
	} else {
		*f1 = dtmf_col[k >> 2];
		*f2 = dtmf_row[k & 3];
	}
}

static void add_tone(int16_t *pcm, int n, int ampl, int hz, float *phase)
{
	float ph = *phase;
	const float step = (2.0f * PI_F * hz) / 8000;      /* tone.c:93-110 */
	for (int i = 0; i < n; i++) {
		pcm[i] += (int16_t)(ampl * cosf(ph));
		ph += step;
	}
	*phase = ph;
}

int orc_ambe_tone_ampl(int log_ampl)
{
	return (int)(32767.0f * exp2f(((float)log_ampl - 255.0f) / 17.0f));   /* tone.c:146 */
}

static int tone(struct orc_ambe_dec *d, int16_t *pcm, int N, const uint8_t *fr)
{
	const int sel = fr[0] & 3, log_ampl = fr[1];
	int code = 0;
	for (int bit = 0; bit < 8; bit++) {           /* majority over the first eight bytes, per bit (tone.c:127-133) */
		int ones = 0;
		for (int j = 0; j < 8; j++)
			ones += (fr[j] >> (7 - bit)) & 1;
		code = (code << 1) | (ones >= 4);
	}
	memset(pcm, 0, sizeof(int16_t) * N);
	const int start = (sel & 2) ? 0 : N >> 1;
	const int stop = (sel & 1) ? (N - 1) : ((N >> 1) - 1);
	if (start >= stop)
		return 0;
	const int ampl = orc_ambe_tone_ampl(log_ampl);
	const int n = stop - start + 1;
	if (code == 0xff)
		return 0;
	if (code >= 0x80 && code <= 0xa3) {
		int f1, f2;
		tone_freqs(code, &f1, &f2);
		add_tone(pcm + start, n, ampl >> 1, f1, &d->tone_ph1);
		add_tone(pcm + start, n, ampl >> 1, f2, &d->tone_ph2);
		return 0;
	}
	if (code < 0x7f) {
		add_tone(pcm + start, n, ampl, (code * 125) >> 2, &d->tone_ph1);
		return 0;
	}
	return -EINVAL;
}

/* ---- decoder object (ambe.c) ---- */

/* Decision D9.  The reference keeps a frame's two subframes in a local array it never clears (ambe.c:81-83), fills
 * the per-harmonic voicing only below L (frame.c:353-358) and then counts unvoiced harmonics up to max(L, L_prev)
 * (synth.c:229-233): when the previous subframe had more harmonics it reads memory it did not write.  In the
 * reference's own program (gmr1_ambe_decode: one decoder, one call site) that memory is the same stack slot on every
 * call, so what it reads is what the same subframe of an earlier frame left there.  That is the behaviour defined
 * here - each decoder carries the two voicing arrays from frame to frame, zero before the first - and
 * tests/test_oracle_ambe.py checks it, bit for bit, against the output of that program built from the reference's
 * sources.  `cleared` = 1 gives the other reading (entries above L are 0), which is what the reference computes when
 * it is entered on a zeroed stack (oracle/ref_codec_shim.c); also checked. */
void orc_ambe_set_cleared(struct orc_ambe_dec *d, int on)
{
	d->cleared = on;
}

void orc_ambe_init(struct orc_ambe_dec *d)
{
	tables_once();
	memset(d, 0, sizeof(*d));
	d->u_last = 3147;                           /* synth.c:306-311 */
	d->prev.w0 = 0.09378f;                      /* ambe.c:43-45 */
	d->prev.f0 = d->prev.w0 / (2 * PI_F);
	d->prev.L = 30;
}

int orc_ambe_decode_frame(struct orc_ambe_dec *d, int16_t *pcm, int N, const uint8_t *frame, int bad)
{
	(void)bad;
	tables_once();
	switch (frame[0] & 0xfc) {                  /* ambe.c:59-73 */
	case 0xfc:
		return tone(d, pcm, N, frame);
	case 0xf8:
		memset(pcm, 0, 160 * sizeof(int16_t));
		return 0;
	}
	struct raw r;
	struct orc_ambe_sub sf[2];
	memset(sf, 0, sizeof(sf));
	if (!d->cleared) {
		memcpy(sf[0].V, d->V_slot[0], sizeof(sf[0].V));
		memcpy(sf[1].V, d->V_slot[1], sizeof(sf[1].V));
	}
	unpack(&r, frame);
	decode_params(sf, &d->prev, &r);
	expand(&sf[0]);
	expand(&sf[1]);
	enhance(d, &sf[0]);
	subframe_audio(d, pcm, &sf[0], &d->prev);
	enhance(d, &sf[1]);
	subframe_audio(d, pcm + 80, &sf[1], &sf[0]);
	d->prev = sf[1];
	memcpy(d->V_slot[0], sf[0].V, sizeof(sf[0].V));
	memcpy(d->V_slot[1], sf[1].V, sizeof(sf[1].V));
	return 0;
}

int orc_ambe_decode_dtx(struct orc_ambe_dec *d, int16_t *pcm, int N)
{
	(void)d;
	memset(pcm, 0, sizeof(int16_t) * N);       /* ambe.c:134-141 */
	return 0;
}

/* a whole stream: n frames of 10 bytes -> n x 160 samples; returns the number of frames that failed */
int orc_ambe_decode_stream(struct orc_ambe_dec *d, const uint8_t *frames, int n, int16_t *pcm, int *rv)
{
	int bad = 0;
	for (int i = 0; i < n; i++) {
		int r = orc_ambe_decode_frame(d, pcm + 160 * (size_t)i, 160, frames + 10 * (size_t)i, 0);
		if (rv)
			rv[i] = r;
		bad += r != 0;
	}
	return bad;
}

size_t orc_ambe_state_size(void)
{
	return sizeof(struct orc_ambe_dec);
}

float orc_ambe_cos_entry(int i)
{
	tables_once();
	return g_cos[i & 1023];
}

float orc_ambe_pow2(float x)
{
	return powf(2.0f, x);                       /* as frame.c:301, 305 call it */
}

float orc_ambe_log2_int(int L)
{
	return log2f(L);                            /* frame.c:238 */
}

/* libm as the reference calls it, over arrays: the product's restatement of glibc's powf is checked against these */
void orc_ambe_powf_array(int n, const float *x, float y, int x_is_base, float *out)
{
	for (int i = 0; i < n; i++)
		out[i] = x_is_base ? powf(x[i], y) : powf(y, x[i]);
}

void orc_ambe_cosf_array(int n, const float *x, float *out)
{
	for (int i = 0; i < n; i++)
		out[i] = cosf(x[i]);                    /* tone.c:104 */
}
