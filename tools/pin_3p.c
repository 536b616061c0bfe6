/*
 * tools/pin_3p.c -- the pinning kit: turns "parity unpinned"  into "pinned" in one run on a
 * machine that HAS the reference's third-party libraries.
 *
 * The arithmetic of the reference's hot path lives in libosmocore (osmo_conv_decode: src/l1/bcch.c:94, ccch.c:98,
 * facch3.c:160, tch3.c:174, facch9.c:134, tch9.c:170, rach.c:167, xch_dc12.c:97) and libosmo-dsp
 * (osmo_cxvec_sig_normalize / _correlate / _peak_energy_find / _peaks_scan / _rotate / _convolve, osmo_sinc:
 * src/sdr/pi4cxpsk.c:229-240, 317-325, 539, 575; src/sdr/fcch.c:230-238, 596-597, 696).  Neither library exists in the
 * image this repository was built in, so oracle/orc_3p.c and oracle/orc_3p_acc.c restate them from their published
 * algorithms, with decisions D1 / D1b (which Viterbi decoder, tie-breaking, metric), D2 (normalisation), D3 (peak
 * search) D4 (tail-biting) left open.  This program calls the REAL functions on inputs chosen to land on exactly those
 * decisions -- metric ties, erasures, junk, saturated soft bits, near-tie correlation peaks -- and writes what they
 * return, inputs included, as JSON.  tests/test_oracle_3p.py::test_third_party_pins loads that file when it is present
 * and requires the oracle to reproduce it: every convolutional code either as D1 or as D1b (and says which, i.e. which
 * decoder your libosmocore runs -- the value to hand to gmr1_hip_set_conv_decoder), the DSP functions within float
 * rounding, the peak position exactly.
 *
 *   cc -std=c99 -O2 tools/pin_3p.c $(pkg-config --cflags --libs libosmocore libosmodsp) -lm -o pin_3p
 *   ./pin_3p > tests/golden/third_party_pins.json
 *   python -m pytest tests/test_oracle_3p.py -k third_party_pins
 *
 * Inputs are generated here from a fixed integer generator (no libc rand, no floating-point in the generator), so
 * every machine produces the same vectors; they are written into the JSON, so the checking side needs no copy of it.
 * The program includes nothing of this repository and nothing of the reference: only the two libraries' public
 * headers.  In the build container it is compiled against declaration-only headers and linked against the ORACLE as
 * a self-test of the kit (tests/test_pin_kit.py); that run pins nothing and says so in its output ("library").
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <osmocom/core/bits.h>
#include <osmocom/core/conv.h>
#include <osmocom/dsp/cxvec.h>
#include <osmocom/dsp/cxvec_math.h>

#ifndef PIN_3P_LIBRARY
#define PIN_3P_LIBRARY "libosmocore + libosmo-dsp"
#endif

/* ---- deterministic inputs ------------------------------------------------------------------------------------ */
static uint32_t g_rng = 0x4d4b2017u;
static uint32_t rnd(void)
{
	g_rng ^= g_rng << 13;
	g_rng ^= g_rng >> 17;
	g_rng ^= g_rng << 5;
	return g_rng;
}
static int rnd_range(int lo, int hi) { return lo + (int)(rnd() % (uint32_t)(hi - lo + 1)); }
/* a multiple of 1/256 in [-4, 4): exact in binary floating point on every machine */
static float rnd_f(void) { return (float)rnd_range(-1024, 1023) / 256.0f; }

/* ---- convolutional codes ----------------------------------------------------------------------------------------
 * The codes of the reference's chains, written as libosmocore wants them: trellis tables expanded from the generator
 * polynomials in the comments of src/l1/conv.c (bit i of a polynomial = D^i, reg = (state << 1) | bit, MSB of the output
 * word = first generator), length / termination / puncturing as the chain constructors set them. */
struct code_def {
	const char *name, *site;
	int N, K, len;
	enum osmo_conv_term term;
	unsigned poly[5];
	int punct_kind;           /* 0 none, 1 idx % 4 == 3 (tch3.c:42-49), 2 TCH9 9k6 (tch9.c:74-79), 3 RACH (rach.c:53-66) */
};

static const struct code_def k_codes[] = {
	{ "bcch_k5_12",     "src/l1/bcch.c:44-50,94",     2, 5, 208, CONV_TERM_FLUSH,       { 0x19, 0x17 }, 0 },
	{ "facch3_k5_14",   "src/l1/facch3.c:44-50,160",  4, 5,  92, CONV_TERM_FLUSH,       { 0x19, 0x17, 0x15, 0x1f }, 0 },
	{ "tch3_k7",        "src/l1/tch3.c:42-49,174",    2, 7,  48, CONV_TERM_TAIL_BITING, { 0x6d, 0x4f }, 1 },
	{ "facch9_k5_12",   "src/l1/facch9.c:42-48,134",  2, 5, 316, CONV_TERM_FLUSH,       { 0x19, 0x17 }, 0 },
	{ "tch9_9k6_k5_12", "src/l1/tch9.c:74-79,170",    2, 5, 480, CONV_TERM_FLUSH,       { 0x19, 0x17 }, 2 },
	{ "k5_13_len240",   "src/l1/conv.c:148-170",      3, 5, 240, CONV_TERM_FLUSH,       { 0x15, 0x1b, 0x1f }, 0 },
	{ "rach_k5_14",     "src/l1/rach.c:44-66,167",    4, 5, 159, CONV_TERM_FLUSH,       { 0x19, 0x17, 0x15, 0x1f }, 3 },
	/* the next two never reach osmo_conv_decode_acc (N = 5, K = 9): they pin the GENERIC decoder on any libosmocore */
	{ "k5_15_len144",   "src/l1/conv.c:201-228",      5, 5, 144, CONV_TERM_FLUSH,       { 0x15, 0x1b, 0x1f, 0x1d, 0x17 }, 0 },
	{ "k9_13_len208",   "src/l1/conv.c:345-415",      3, 9, 208, CONV_TERM_TAIL_BITING, { 0x1ed, 0x19b, 0x127 }, 0 },
};

static unsigned parity(unsigned v)
{
	v ^= v >> 16; v ^= v >> 8; v ^= v >> 4; v ^= v >> 2; v ^= v >> 1;
	return v & 1u;
}

static uint8_t g_next_output[256][2], g_next_state[256][2];
static int g_punct[1024];

static int build_code(const struct code_def *d, struct osmo_conv_code *c)
{
	const int ns = 1 << (d->K - 1);
	int n_punct = 0;
	memset(c, 0, sizeof(*c));
	for (int s = 0; s < ns; s++)
		for (int b = 0; b < 2; b++) {
			const unsigned reg = ((unsigned)s << 1) | (unsigned)b;
			unsigned o = 0;
			for (int i = 0; i < d->N; i++)
				o = (o << 1) | parity(reg & d->poly[i]);
			g_next_output[s][b] = (uint8_t)o;
			g_next_state[s][b] = (uint8_t)(reg & (unsigned)(ns - 1));
		}
	const int coded = (d->len + (d->term == CONV_TERM_FLUSH ? d->K - 1 : 0)) * d->N;
	if (d->punct_kind == 1) {
		for (int i = 3; i < coded; i += 4)
			g_punct[n_punct++] = i;
	} else if (d->punct_kind == 2) {
		/* P(2;5) once, P(2;3) 158 times, P*(2;5) at the end: {1, 5}, {10 + 6k, 13 + 6k}, {963, 967} */
		g_punct[n_punct++] = 1; g_punct[n_punct++] = 5;
		for (int k = 0; k < 158; k++) { g_punct[n_punct++] = 10 + 6 * k; g_punct[n_punct++] = 13 + 6 * k; }
		g_punct[n_punct++] = 963; g_punct[n_punct++] = 967;
	} else if (d->punct_kind == 3) {
		for (int k = 0; k < 135; k++) { g_punct[n_punct++] = 4 * k + 2; g_punct[n_punct++] = 4 * k + 3; }
	}
	g_punct[n_punct] = -1;
	c->N = d->N;
	c->K = d->K;
	c->len = d->len;
	c->term = d->term;
	c->next_output = (const uint8_t (*)[2])g_next_output;
	c->next_state = (const uint8_t (*)[2])g_next_state;
	c->puncture = n_punct ? g_punct : NULL;
	return n_punct;
}

/* a code word of the code for random data, as hard bits (so that "clean + noise" inputs sit around a real path) */
static void encode(const struct osmo_conv_code *c, const ubit_t *u, ubit_t *coded /* unpunctured */)
{
	const int ns = 1 << (c->K - 1);
	unsigned s = 0;
	const int steps = c->len + (c->term == CONV_TERM_FLUSH ? c->K - 1 : 0);
	if (c->term == CONV_TERM_TAIL_BITING)
		for (int i = 0; i < c->K - 1; i++)
			s = ((s << 1) | u[c->len - c->K + 1 + i]) & (unsigned)(ns - 1);
	for (int i = 0; i < steps; i++) {
		const unsigned b = i < c->len ? u[i] : 0;
		const unsigned o = c->next_output[s][b];
		for (int j = 0; j < c->N; j++)
			coded[i * c->N + j] = (ubit_t)((o >> (c->N - 1 - j)) & 1u);
		s = c->next_state[s][b];
	}
}

static const char *const k_kinds[] = { "noisy", "erased", "junk", "coarse_ties", "tiny", "saturated", "all_erased" };
#define N_KINDS 7
#define N_PER_KIND 4

static void make_input(const struct osmo_conv_code *c, int n_in, int kind, sbit_t *in)
{
	static ubit_t u[512], coded[2600];
	const int steps = c->len + (c->term == CONV_TERM_FLUSH ? c->K - 1 : 0);
	for (int i = 0; i < c->len; i++)
		u[i] = (ubit_t)(rnd() & 1u);
	encode(c, u, coded);
	/* the transmitted (punctured) stream */
	int o = 0, p = 0;
	for (int idx = 0; idx < steps * c->N; idx++) {
		if (c->puncture && c->puncture[p] == idx) { p++; continue; }
		const int sign = coded[idx] ? -1 : 1;
		int v;
		switch (kind) {
		case 0: v = sign * rnd_range(20, 127); if (rnd() % 8 == 0) v = -v; break;             /* 12 % of the bits wrong */
		case 1: v = rnd() % 5 == 0 ? 0 : sign * rnd_range(40, 127); if (rnd() % 16 == 0) v = -v; break;
		case 2: v = rnd_range(-128, 127); break;                                            /* no code word at all */
		case 3: v = 50 * rnd_range(-2, 2); break;                                           /* five levels: metric ties */
		case 4: v = rnd_range(-2, 2); break;                                                /* the generic decoder's >> 9 sees one level */
		case 5: v = rnd() % 4 == 0 ? (sign > 0 ? -128 : 127) : (sign > 0 ? 127 : -128); break;
		default: v = 0; break;
		}
		in[o++] = (sbit_t)v;
	}
	if (o != n_in) { fprintf(stderr, "pin_3p: internal length error (%d != %d)\n", o, n_in); exit(2); }
}

static void conv_section(void)
{
	printf(" \"conv\": [\n");
	for (size_t ci = 0; ci < sizeof(k_codes) / sizeof(k_codes[0]); ci++) {
		const struct code_def *d = &k_codes[ci];
		struct osmo_conv_code code;
		const int n_punct = build_code(d, &code);
		const int n_in = osmo_conv_get_output_length(&code, 0);
		printf("  {\"name\": \"%s\", \"site\": \"%s\", \"N\": %d, \"K\": %d, \"len\": %d, \"term\": %d, \"n_in\": %d,\n",
		       d->name, d->site, d->N, d->K, d->len, (int)d->term, n_in);
		printf("   \"polys\": [");
		for (int i = 0; i < d->N; i++)
			printf("%s%u", i ? ", " : "", d->poly[i]);
		printf("],\n   \"punct\": [");
		for (int i = 0; i < n_punct; i++)
			printf("%s%d", i ? "," : "", g_punct[i]);
		printf("],\n   \"vectors\": [\n");
		for (int kind = 0; kind < N_KINDS; kind++)
			for (int r = 0; r < N_PER_KIND; r++) {
				static sbit_t in[2600];
				static ubit_t out[512];
				make_input(&code, n_in, kind, in);
				memset(out, 0xff, sizeof(out));
				const int rv = osmo_conv_decode(&code, in, out);
				printf("    {\"kind\": \"%s\", \"rv\": %d, \"in\": \"", k_kinds[kind], rv);
				for (int i = 0; i < n_in; i++)
					printf("%02x", (unsigned)(uint8_t)in[i]);
				printf("\", \"out\": \"");
				for (int i = 0; i < d->len; i++)
					putchar(out[i] == 0 ? '0' : (out[i] == 1 ? '1' : '?'));
				printf("\"}%s\n", (kind == N_KINDS - 1 && r == N_PER_KIND - 1) ? "" : ",");
			}
		printf("   ]}%s\n", ci + 1 < sizeof(k_codes) / sizeof(k_codes[0]) ? "," : "");
	}
	printf(" ],\n");
}

/* ---- libosmo-dsp ------------------------------------------------------------------------------------------------ */
static void put_cvec(const char *key, const float complex *v, int n, const char *tail)
{
	printf("\"%s\": [", key);
	for (int i = 0; i < n; i++)
		printf("%s[%.9g, %.9g]", i ? ", " : "", (double)crealf(v[i]), (double)cimagf(v[i]));
	printf("]%s", tail);
}

static struct osmo_cxvec *rnd_vec(int n)
{
	struct osmo_cxvec *v = osmo_cxvec_alloc(n);
	if (!v) { fprintf(stderr, "pin_3p: osmo_cxvec_alloc failed\n"); exit(2); }
	v->len = n;
	for (int i = 0; i < n; i++)
		v->data[i] = rnd_f() + I * rnd_f();
	return v;
}

/* a correlation-like vector: a sinc-shaped bump at a fractional position over a small floor */
static struct osmo_cxvec *bump_vec(int n, int pos256 /* peak position in 1/256 sample */, int floor_level)
{
	struct osmo_cxvec *v = osmo_cxvec_alloc(n);
	if (!v) { fprintf(stderr, "pin_3p: osmo_cxvec_alloc failed\n"); exit(2); }
	v->len = n;
	for (int i = 0; i < n; i++) {
		const int d = 256 * i - pos256;             /* distance in 1/256 sample */
		/* triangle of half-width 2 samples, height 1024/256 = 4: exact arithmetic, a clean single maximum */
		int h = 1024 - (d < 0 ? -d : d) * 2;
		if (h < 0) h = 0;
		v->data[i] = (float)(h + rnd_range(0, floor_level)) / 256.0f + I * ((float)rnd_range(-floor_level, floor_level) / 256.0f);
	}
	return v;
}

static void dsp_section(void)
{
	printf(" \"dsp\": {\n");

	/* osmo_cxvec_sig_normalize: pi4cxpsk.c:539 (decim 1, shift), fcch.c:230 (decim sps, shift) -- decision D2 */
	printf("  \"sig_normalize\": [\n");
	{
		static const int decim[4] = { 1, 1, 4, 4 };
		static const float shift[4] = { 0.0f, -0.1963495f, 0.0f, 0.0123f };
		for (int k = 0; k < 4; k++) {
			struct osmo_cxvec *in = rnd_vec(96);
			struct osmo_cxvec *out = osmo_cxvec_sig_normalize(in, decim[k], shift[k], NULL);
			printf("   {\"decim\": %d, \"freq_shift\": %.9g, ", decim[k], (double)shift[k]);
			put_cvec("in", in->data, in->len, ", ");
			put_cvec("out", out->data, out->len, k < 3 ? "},\n" : "}\n");
			osmo_cxvec_free(in);
			osmo_cxvec_free(out);
		}
	}
	printf("  ],\n");

	/* osmo_cxvec_correlate: pi4cxpsk.c:229 (step sps), fcch.c:233 (step 1) */
	printf("  \"correlate\": [\n");
	for (int k = 0; k < 2; k++) {
		const int step = k ? 4 : 1;
		struct osmo_cxvec *f = rnd_vec(11), *g = rnd_vec(120);
		struct osmo_cxvec *out = osmo_cxvec_correlate(f, g, step, NULL);
		printf("   {\"step\": %d, ", step);
		put_cvec("f", f->data, f->len, ", ");
		put_cvec("g", g->data, g->len, ", ");
		put_cvec("out", out->data, out->len, k < 1 ? "},\n" : "}\n");
		osmo_cxvec_free(f); osmo_cxvec_free(g); osmo_cxvec_free(out);
	}
	printf("  ],\n");

	/* osmo_cxvec_peak_energy_find: pi4cxpsk.c:240 (win 3, PEAK_EARLY_LATE, peak value), fcch.c:238, 596 (win 5,
	 * PEAK_WEIGH_WIN) -- decision D3.  Peaks on a grid of 1/256 sample incl. exact half-sample positions (ties between
	 * the early and the late point), near either end of the vector (the interpolation's clipping, D3b), flat tops. */
	printf("  \"peak_energy_find\": [\n");
	{
		static const int pos256[12] = { 20 * 256, 20 * 256 + 128, 20 * 256 + 37, 20 * 256 - 91, 3 * 256 + 64, 37 * 256 + 200,
		                                256 + 128, 39 * 256, 12 * 256 + 1, 12 * 256 + 255, 25 * 256 + 127, 25 * 256 + 129 };
		for (int k = 0; k < 24; k++) {
			const int early_late = k < 12;
			struct osmo_cxvec *cv = bump_vec(41, pos256[k % 12], k % 3 == 0 ? 0 : 40);
			float complex pv = 0;
			const float pos = early_late ? osmo_cxvec_peak_energy_find(cv, 3, PEAK_EARLY_LATE, &pv)
			                             : osmo_cxvec_peak_energy_find(cv, 5, PEAK_WEIGH_WIN, NULL);
			printf("   {\"win\": %d, \"alg\": \"%s\", ", early_late ? 3 : 5, early_late ? "early_late" : "weigh_win");
			put_cvec("cv", cv->data, cv->len, ", ");
			printf("\"pos\": %.9g, \"peak\": [%.9g, %.9g]}%s\n", (double)pos, (double)crealf(pv), (double)cimagf(pv), k < 23 ? "," : "");
			osmo_cxvec_free(cv);
		}
	}
	printf("  ],\n");

	/* osmo_cxvec_peaks_scan: fcch.c:696 */
	printf("  \"peaks_scan\": [\n");
	for (int k = 0; k < 2; k++) {
		struct osmo_cxvec *cv = rnd_vec(117);
		int idx[6];
		if (k) { cv->data[7] = cv->data[90]; cv->data[31] = cv->data[90]; }      /* equal energies */
		osmo_cxvec_peaks_scan(cv, idx, 6);
		printf("   {");
		put_cvec("cv", cv->data, cv->len, ", ");
		printf("\"idx\": [%d, %d, %d, %d, %d, %d]}%s\n", idx[0], idx[1], idx[2], idx[3], idx[4], idx[5], k < 1 ? "," : "");
		osmo_cxvec_free(cv);
	}
	printf("  ],\n");

	/* osmo_cxvec_rotate (pi4cxpsk.c:575, 793), osmo_cxvec_interpolate_point, osmo_sinc (pi4cxpsk.c:317) */
	printf("  \"rotate\": [\n");
	{
		struct osmo_cxvec *in = rnd_vec(64), *out = osmo_cxvec_alloc(64);
		out->len = 64;
		osmo_cxvec_rotate(in, 0.7853982f, out);
		printf("   {\"rps\": %.9g, ", (double)0.7853982f);
		put_cvec("in", in->data, 64, ", ");
		put_cvec("out", out->data, 64, "}\n");
		osmo_cxvec_free(in); osmo_cxvec_free(out);
	}
	printf("  ],\n  \"interpolate_point\": [\n");
	{
		struct osmo_cxvec *cv = rnd_vec(41);
		static const float at[6] = { 20.0f, 20.5f, 3.25f, 38.75f, 0.5f, 39.99f };
		printf("   {");
		put_cvec("cv", cv->data, 41, ", ");
		printf("\"at\": [");
		for (int i = 0; i < 6; i++) printf("%s%.9g", i ? ", " : "", (double)at[i]);
		printf("], \"val\": [");
		for (int i = 0; i < 6; i++) {
			const float complex v = osmo_cxvec_interpolate_point(cv, at[i]);
			printf("%s[%.9g, %.9g]", i ? ", " : "", (double)crealf(v), (double)cimagf(v));
		}
		printf("]}\n");
		osmo_cxvec_free(cv);
	}
	printf("  ],\n  \"sinc\": {\"x\": [");
	{
		static const float xs[8] = { 0.0f, 1e-8f, 1e-4f, 0.5f, 1.5707964f, 3.1415927f, -2.0f, 31.0f };
		for (int i = 0; i < 8; i++) printf("%s%.9g", i ? ", " : "", (double)xs[i]);
		printf("], \"y\": [");
		for (int i = 0; i < 8; i++) printf("%s%.9g", i ? ", " : "", (double)osmo_sinc(xs[i]));
		printf("]},\n");
	}

	/* osmo_cxvec_convolve(CONV_NO_DELAY): pi4cxpsk.c:325, real 21-tap pulse against a burst (the sps < 4 branch) */
	printf("  \"convolve_no_delay\": [\n");
	{
		struct osmo_cxvec *f = osmo_cxvec_alloc(21), *g = rnd_vec(60);
		f->len = 21;
		f->flags |= CXVEC_FLG_REAL_ONLY;
		for (int i = 0; i < 21; i++)
			f->data[i] = osmo_sinc(3.1415927f * ((float)(i - 10) - 0.3125f));
		struct osmo_cxvec *out = osmo_cxvec_convolve(f, g, CONV_NO_DELAY, NULL);
		printf("   {");
		put_cvec("f", f->data, 21, ", ");
		put_cvec("g", g->data, 60, ", ");
		put_cvec("out", out->data, out->len, "}\n");
		osmo_cxvec_free(f); osmo_cxvec_free(g); osmo_cxvec_free(out);
	}
	printf("  ]\n },\n");
}

int main(void)
{
	printf("{\n \"format\": 1,\n \"library\": \"%s\",\n", PIN_3P_LIBRARY);
	conv_section();
	dsp_section();
	printf(" \"end\": true\n}\n");
	return 0;
}
