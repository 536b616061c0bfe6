// tests/c/host_san.cpp -- host-side parts of the library under AddressSanitizer + UBSan (tests/test_sanitize.py):
// the burst tables and their flattening (host_tables.cpp), the code / puncturing descriptions (l1_tables.cpp) and
// gmr1_puncturer_generate (l1_punct.cpp) with every chain's arguments, the FCCH tables.  No GPU call anywhere: these
// translation units are plain host code, compiled here with g++ and the HIP headers only for their types.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <osmocom/gmr1/l1/conv.h>
#include <osmocom/gmr1/l1/punct.h>

#include "host_tables.h"

using namespace gmr1;

static int fails;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)

static struct osmo_conv_code copy_code(const struct osmo_conv_code &c, int len)
{
	struct osmo_conv_code o = c;
	o.len = len;
	o.puncture = nullptr;
	return o;
}

static int count(const int *p)
{
	int n = 0;
	while (p && p[n] >= 0)
		n++;
	return n;
}

int main()
{
	// ---- burst tables: every built-in format flattens and converts; the flat form has the geometry of nb.c
	tables_init();
	int ebits_total = 0;
	for (int i = 0; i < GMR1_HIP_N_BURSTS; i++) {
		gmr1_hip_burst_flat f;
		DevBurst d;
		CHECK(flatten(kBuiltin[i], &f, kBuiltinName[i]) == 0);
		CHECK(to_dev(f, &d) == 0);
		CHECK(d.len == kBuiltin[i]->len && d.ebits == kBuiltin[i]->ebits);
		int data = 0;
		for (int c = 0; c < d.n_data; c++)
			data += d.dlen[c];
		CHECK(data * d.nbits == d.ebits);
		ebits_total += d.ebits;
	}
	CHECK(ebits_total == 424 + 132 + 432 + 432 + 212 + 104 + 434 + 662 + 494 + 208);
	FcchTables *ft = new FcchTables;
	fcch_tables_init(ft);
	CHECK(ft->len[0] == 117 && ft->len[1] == 468 && ft->len[2] == 468);
	delete ft;

	// ---- gmr1_puncturer_generate with the arguments of the chains' constructors (tch3.c:42-49, tch9.c:55-79,
	// xch_dc12.c:45-54): list lengths, order, termination
	{
		struct osmo_conv_code c = copy_code(gmr1_conv_tch3, 48);
		CHECK(gmr1_puncturer_generate(&c, nullptr, &gmr1_punct_k5_12_P12, nullptr, 0) == 0);
		CHECK(count(c.puncture) == 24 && c.puncture[0] == 3 && c.puncture[23] == 95);
		free((void *)c.puncture);
	}
	{
		struct osmo_conv_code c = copy_code(gmr1_conv_k5_12, 480);
		CHECK(gmr1_puncturer_generate(&c, &gmr1_punct_k5_12_P25, &gmr1_punct_k5_12_P23, &gmr1_punct_k5_12_Ps25, 158) == 0);
		CHECK(count(c.puncture) == 320 && c.puncture[0] == 1 && c.puncture[1] == 5 && c.puncture[319] == 967);
		for (int i = 1; i < 320; i++)
			CHECK(c.puncture[i] > c.puncture[i - 1]);
		free((void *)c.puncture);
	}
	{
		struct osmo_conv_code c = copy_code(gmr1_conv_k5_13, 240);
		CHECK(gmr1_puncturer_generate(&c, &gmr1_punct_k5_13_P15, &gmr1_punct_k5_13_P25, &gmr1_punct_k5_13_Ps15, 41) == 0);
		CHECK(count(c.puncture) == 244 * 3 - 648);
		free((void *)c.puncture);
	}
	{
		struct osmo_conv_code c = copy_code(gmr1_conv_k5_15, 144);
		CHECK(gmr1_puncturer_generate(&c, &gmr1_punct_k5_15_P53, &gmr1_punct_k5_15_P23, &gmr1_punct_k5_15_Ps53, 41) == 0);
		CHECK(count(c.puncture) == 148 * 5 - 648);
		free((void *)c.puncture);
	}
	{
		struct osmo_conv_code c = copy_code(gmr1_conv_k9_13, 208);
		c.term = CONV_TERM_TAIL_BITING;
		CHECK(gmr1_puncturer_generate(&c, nullptr, &gmr1_punct_k9_13_P1213, nullptr, 0) == 0);
		CHECK(count(c.puncture) == 208 * 3 - 432);
		free((void *)c.puncture);
	}
	// argument checks: mismatched rate, missing main scheme
	{
		struct osmo_conv_code c = copy_code(gmr1_conv_k5_12, 100);
		CHECK(gmr1_puncturer_generate(&c, nullptr, &gmr1_punct_k5_13_P25, nullptr, 0) < 0);
		CHECK(gmr1_puncturer_generate(&c, nullptr, nullptr, nullptr, 0) < 0);
		CHECK(gmr1_puncturer_generate(nullptr, nullptr, &gmr1_punct_k5_12_P23, nullptr, 0) < 0);
	}
	// every exported trellis is a shift register: next_state follows ((s << 1) | b) & mask (conv.c:35-40)
	const struct osmo_conv_code *codes[] = {&gmr1_conv_k5_12, &gmr1_conv_k5_13, &gmr1_conv_k5_14, &gmr1_conv_k5_15, &gmr1_conv_k6_14,
	                                        &gmr1_conv_k9_12, &gmr1_conv_k9_13, &gmr1_conv_k9_14, &gmr1_conv_tch3};
	for (const struct osmo_conv_code *c : codes) {
		const int ns = 1 << (c->K - 1);
		for (int s = 0; s < ns; s++)
			for (int b = 0; b < 2; b++) {
				CHECK(c->next_state[s][b] == (((s << 1) | b) & (ns - 1)));
				CHECK(c->next_output[s][b] < (1 << c->N));
			}
	}
	printf("host_san: %d failures\n", fails);
	return fails ? 1 : 0;
}
