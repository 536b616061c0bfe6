/*
 * tests/c/pin_3p_oracle_shim.c -- TEST INFRASTRUCTURE ONLY: self-test of the pinning kit (tools/pin_3p.c).
 *
 * The kit is meant to be linked against the real libosmocore / libosmo-dsp on an integrator's machine.  Neither exists
 * in this image, so tests/test_pin_kit.py links it against THIS file instead, which answers the handful of library
 * calls the kit makes with the oracle's restatements (oracle/orc_3p.c, orc_3p_acc.c).  That run checks the kit's own
 * plumbing -- code tables, puncturing lists, input generator, JSON -- and that tests/pin_check.py recognises each decoder;
 * it pins nothing (its output says "library": "oracle self-test").  PIN_SHIM_CONV_MODE=1 makes the stand-in behave like a
 * libosmocore with the accelerated decoder.
 */
#include <complex.h>
#include <stdlib.h>
#include <string.h>

#include <osmocom/core/bits.h>
#include <osmocom/core/conv.h>
#include <osmocom/dsp/cxvec.h>
#include <osmocom/dsp/cxvec_math.h>

#include "orc_3p.h"

static void to_orc(const struct osmo_conv_code *c, struct orc_conv_code *o)
{
	const int ns = 1 << (c->K - 1);
	memset(o, 0, sizeof(*o));
	o->N = c->N; o->K = c->K; o->len = c->len;
	o->term = c->term == CONV_TERM_FLUSH ? ORC_TERM_FLUSH : (c->term == CONV_TERM_TAIL_BITING ? ORC_TERM_TAIL_BITING : ORC_TERM_TRUNCATION);
	for (int s = 0; s < ns; s++)
		for (int b = 0; b < 2; b++) {
			o->next_output[s][b] = c->next_output[s][b];
			o->next_state[s][b] = c->next_state[s][b];
		}
	if (c->puncture)
		for (o->n_punct = 0; c->puncture[o->n_punct] >= 0; o->n_punct++)
			o->punct[o->n_punct] = c->puncture[o->n_punct];
	o->punct[o->n_punct] = -1;
}

int osmo_conv_get_output_length(const struct osmo_conv_code *code, int len)
{
	struct orc_conv_code o;
	(void)len;
	to_orc(code, &o);
	return orc_conv_output_length(&o);
}

int osmo_conv_decode(const struct osmo_conv_code *code, const sbit_t *input, ubit_t *output)
{
	static struct orc_conv_code o;
	const char *m = getenv("PIN_SHIM_CONV_MODE");
	to_orc(code, &o);
	orc_conv_set_mode(m && m[0] == '1');
	return orc_conv_decode(&o, input, output);
}

struct osmo_cxvec *osmo_cxvec_alloc(int max_len)
{
	struct osmo_cxvec *v = calloc(1, sizeof(*v) + sizeof(float complex) * (size_t)max_len);
	v->max_len = max_len;
	v->data = v->_data;
	return v;
}
void osmo_cxvec_free(struct osmo_cxvec *cv) { free(cv); }

struct osmo_cxvec *osmo_cxvec_sig_normalize(const struct osmo_cxvec *sig, int decim, float freq_shift, struct osmo_cxvec *out)
{
	if (!out) out = osmo_cxvec_alloc(sig->len / decim + 1);
	out->len = orc_sig_normalize(sig->data, sig->len, decim, freq_shift, out->data);
	return out;
}
struct osmo_cxvec *osmo_cxvec_correlate(const struct osmo_cxvec *f, const struct osmo_cxvec *g, int step, struct osmo_cxvec *out)
{
	if (!out) out = osmo_cxvec_alloc(g->len);
	out->len = orc_correlate(f->data, f->len, g->data, g->len, step, out->data);
	return out;
}
float osmo_cxvec_peak_energy_find(const struct osmo_cxvec *cv, int win, enum osmo_cxvec_peak_alg alg, float complex *pv)
{
	return orc_peak_energy_find(cv->data, cv->len, win, alg == PEAK_EARLY_LATE ? ORC_PEAK_EARLY_LATE : ORC_PEAK_WEIGH_WIN, pv);
}
void osmo_cxvec_peaks_scan(const struct osmo_cxvec *cv, int *idx, int N) { orc_peaks_scan(cv->data, cv->len, idx, N); }
struct osmo_cxvec *osmo_cxvec_rotate(const struct osmo_cxvec *in, float rps, struct osmo_cxvec *out)
{
	memcpy(out->data, in->data, sizeof(float complex) * (size_t)in->len);
	out->len = in->len;
	orc_rotate(out->data, out->len, rps);
	return out;
}
float complex osmo_cxvec_interpolate_point(const struct osmo_cxvec *cv, float pos) { return orc_interpolate_point(cv->data, cv->len, pos); }
float osmo_sinc(float x) { return orc_sinc(x); }
struct osmo_cxvec *osmo_cxvec_convolve(const struct osmo_cxvec *f, const struct osmo_cxvec *g, enum osmo_cxvec_conv_type type, struct osmo_cxvec *out)
{
	float taps[64];
	(void)type;
	for (int i = 0; i < f->len; i++) taps[i] = crealf(f->data[i]);
	if (!out) out = osmo_cxvec_alloc(g->len);
	out->len = g->len;
	orc_convolve_nodelay_real(taps, f->len, g->data, g->len, out->data);
	return out;
}
