/*
 * oracle/orc_3p_acc.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Decision D1b: the OTHER Viterbi decoder a libosmo-gmr build may run.  libosmocore since 2017 sends codes with
 * K in {5, 7} and N in {2, 3, 4} -- BCCH, CCCH, FACCH3, FACCH9, TCH3 speech, TCH9 4k8 / 9k6, RACH; not the K = 9 xCH
 * code, not the rate-1/5 TCH9 2k4 code -- to osmo_conv_decode_acc (src/conv_acc.c, conv_acc_generic.c /
 * conv_acc_sse*.h) and only the rest to the generic decoder restated as D1 in orc_3p.c.  That library is absent
 * from this image and from /root/reference; this is its published algorithm as recollected [3P-recollection], and
 * the tests that use it (tests/test_oracle_d1b.py, tests/test_gpu_d1b.py) do not pin it either: they BOUND what the
 * choice between D1 and D1b can change in decoded frames.
 *
 * What differs from D1:
 *  - branch metric: the correlation  sum_j in[j] * (coded bit j ? -1 : +1)  is MAXIMISED (16-bit accumulated sums,
 *    renormalised by subtracting the minimum every INT16_MAX / (N * 127) - K steps, which cannot change a decision
 *    -- evaluated here in 32 bits without the renormalisation) instead of minimising sum ((in -+ 127)^2 >> 9); the two
 *    rank paths identically in exact arithmetic, the >> 9 of D1 quantises;
 *  - punctured positions count 0 (as in D1); a soft bit of 0 counts 0 (as in D1);
 *  - flushed codes: every start state is allowed, state 0 leads by 127 * N * K (D1: only state 0), every step runs the
 *    full butterfly (D1: input 0 only in the K - 1 flush steps), the traceback starts in state 0;
 *  - tail-biting: two passes over the data with the sums carried over, the traceback starts in the state with the
 *    largest sum, the first one in the decoder's own state numbering (newest bit on top: bit-reversed with respect
 *    to struct osmo_conv_code) on ties;
 *  - ties between the two paths into a state: `sum0 >= sum1` keeps the predecessor whose OLDEST bit is 0 -- the same
 *    lower-numbered predecessor D1 keeps;
 *  - the return value is 0, not a path metric (so `conv_rv` carries no information under D1b).
 */
#include "orc_3p.h"

#include <stdlib.h>
#include <string.h>

static int g_conv_mode;                  /* 0: D1 for every code, 1: D1b where libosmocore would use it */

void orc_conv_set_mode(int mode) { g_conv_mode = mode; }
int  orc_conv_get_mode(void) { return g_conv_mode; }

int orc_conv_acc_applies(const struct orc_conv_code *c)
{
	return g_conv_mode == 1 && (c->K == 5 || c->K == 7) && c->N >= 2 && c->N <= 4;
}

static unsigned bitrev(unsigned v, int bits)
{
	unsigned r = 0;
	for (int i = 0; i < bits; i++)
		r |= ((v >> i) & 1u) << (bits - 1 - i);
	return r;
}

int orc_conv_decode_acc(const struct orc_conv_code *c, const orc_sbit_t *in, orc_ubit_t *out)
{
	const int ns = 1 << (c->K - 1);
	const int steps = c->len + (c->term == ORC_TERM_FLUSH ? c->K - 1 : 0);
	const int passes = c->term == ORC_TERM_TAIL_BITING ? 2 : 1;
	int32_t sum[256], nsum[256];
	int8_t *sym = calloc((size_t)steps * (size_t)c->N, 1);
	uint8_t *hist = calloc((size_t)steps * (size_t)ns, 1);
	int p = 0, o = 0, state;

	/* depuncture into a zero-filled stream (conv_acc.c does the same before decoding) */
	for (int idx = 0; idx < steps * c->N; idx++) {
		if (c->n_punct && c->punct[p] == idx) { p++; continue; }
		sym[idx] = in[o++];
	}
	memset(sum, 0, sizeof(sum));
	if (c->term == ORC_TERM_FLUSH)
		sum[0] = 127 * c->N * c->K;

	for (int pass = 0; pass < passes; pass++)
		for (int i = 0; i < steps; i++) {
			for (int t = 0; t < ns; t++) {
				/* predecessors of t: the state without and with the oldest bit */
				const int b = t & 1;
				const int lo = t >> 1, hi = lo | (ns >> 1);
				int32_t m[2];
				for (int k = 0; k < 2; k++) {
					const int s = k ? hi : lo;
					const unsigned w = c->next_output[s][b];
					int32_t v = 0;
					for (int j = 0; j < c->N; j++)
						v += ((w >> (c->N - 1 - j)) & 1) ? -(int32_t)sym[i * c->N + j] : (int32_t)sym[i * c->N + j];
					m[k] = sum[s] + v;
				}
				if (m[0] >= m[1]) { nsum[t] = m[0]; hist[i * ns + t] = (uint8_t)lo; }
				else              { nsum[t] = m[1]; hist[i * ns + t] = (uint8_t)hi; }
			}
			memcpy(sum, nsum, sizeof(int32_t) * (size_t)ns);
		}

	if (c->term == ORC_TERM_FLUSH) {
		state = 0;
	} else {
		int32_t best = -1;
		state = -1;
		for (int r = 0; r < ns; r++) {            /* the decoder's own numbering */
			const int s = (int)bitrev((unsigned)r, c->K - 1);
			if (sum[s] > best) { best = sum[s]; state = s; }
		}
		if (state < 0) { free(sym); free(hist); return -1; }
	}
	for (int i = steps - 1; i >= 0; i--) {
		if (i < c->len)
			out[i] = (orc_ubit_t)(state & 1);     /* the input bit that led into `state` */
		state = hist[i * ns + state];
	}
	free(sym);
	free(hist);
	return 0;
}
