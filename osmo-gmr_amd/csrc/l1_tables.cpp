// l1_tables.cpp -- the code description objects of libgmr1-l1's API (reference include/osmocom/gmr1/l1/conv.h:36-44,
// l1/punct.h:54-106, l1/crc.h:36-38), exported for consumers that drive libosmocore's own encoder / decoder with them.
// Nothing on the GPU path reads these: the kernels build their trellises from the same polynomials at compile time and
// fold the punctured positions into their gather maps (capi_nt9.cpp, capi_xch.cpp, l1_kernels.hip).
//
// Stored compactly and expanded at compile time: a trellis is its generator polynomials (bit i = D^i; the output word
// has g0 in its MSB), a puncturing scheme is the list of its punctured mask positions.  tests/test_ref_tables.py
// compares every expanded table with the one printed in the reference's conv.c / punct.c / crc.c.
#include <stdint.h>

#include <osmocom/gmr1/l1/conv.h>
#include <osmocom/gmr1/l1/crc.h>
// (not l1/punct.h: its objects end in a flexible array member, which C++ cannot initialise; they are defined below
// with a fixed-size twin of the same layout)

namespace {

template <int K>
struct Trellis {
	uint8_t out[1 << (K - 1)][2];
	uint8_t nxt[1 << (K - 1)][2];
};

constexpr unsigned parity(unsigned v)
{
	unsigned p = 0;
	for (; v; v >>= 1)
		p ^= v & 1u;
	return p;
}

template <int K, int N>
constexpr Trellis<K> make_trellis(const unsigned (&polys)[N])
{
	Trellis<K> t{};
	for (unsigned s = 0; s < (1u << (K - 1)); s++)
		for (unsigned b = 0; b < 2; b++) {
			const unsigned reg = (s << 1) | b;            // bit i = D^i
			unsigned w = 0;
			for (int j = 0; j < N; j++)
				w = (w << 1) | parity(reg & polys[j]);
			t.out[s][b] = (uint8_t)w;
			t.nxt[s][b] = (uint8_t)(reg & ((1u << (K - 1)) - 1u));
		}
	return t;
}

// conv.c:123-128, 148-154, 174-181, 201-209, 229-236, 260-265, 345-351, 431-438 (k9_14: g3 as its TABLE has it), 518-523
constexpr unsigned p_k5_12[] = {0x19, 0x17};
constexpr unsigned p_k5_13[] = {0x15, 0x1b, 0x1f};
constexpr unsigned p_k5_14[] = {0x19, 0x17, 0x15, 0x1f};
constexpr unsigned p_k5_15[] = {0x15, 0x1b, 0x1f, 0x1d, 0x17};
constexpr unsigned p_k6_14[] = {0x25, 0x2d, 0x3b, 0x3f};
constexpr unsigned p_k9_12[] = {0x11d, 0x1af};
constexpr unsigned p_k9_13[] = {0x1ed, 0x19b, 0x127};
constexpr unsigned p_k9_14[] = {0x1b9, 0x1a5, 0x13b, 0x15f};
constexpr unsigned p_tch3[]  = {0x6d, 0x4f};

constexpr Trellis<5> t_k5_12 = make_trellis<5>(p_k5_12);
constexpr Trellis<5> t_k5_13 = make_trellis<5>(p_k5_13);
constexpr Trellis<5> t_k5_14 = make_trellis<5>(p_k5_14);
constexpr Trellis<5> t_k5_15 = make_trellis<5>(p_k5_15);
constexpr Trellis<6> t_k6_14 = make_trellis<6>(p_k6_14);
constexpr Trellis<9> t_k9_12 = make_trellis<9>(p_k9_12);
constexpr Trellis<9> t_k9_13 = make_trellis<9>(p_k9_13);
constexpr Trellis<9> t_k9_14 = make_trellis<9>(p_k9_14);
constexpr Trellis<7> t_tch3  = make_trellis<7>(p_tch3);

// same layout as struct gmr1_puncturer { int r, L, N; const uint8_t mask[]; } with the mask given its size
template <int M>
struct PunctStore {
	int r, L, N;
	uint8_t mask[M];
};

template <int M, int NZ>
constexpr PunctStore<M> make_punct(int r, int L, int N, int two, const int (&zeros)[NZ], int nz)
{
	PunctStore<M> p{r, L, N, {}};
	for (int i = 0; i < M; i++)
		p.mask[i] = 1;
	for (int i = 0; i < nz; i++)
		if (zeros[i] >= 0)        // (the lists start with a -1 so that an empty one is still an array)
			p.mask[zeros[i]] = 0;
	if (two >= 0)
		p.mask[two] = 2;      // scheme E repeats a bit: the reference writes a 2 there (punct.c:313-324)
	return p;
}

}  // namespace

extern "C" {

#define CONV(name, n, k, term, t)                                                                        \
	extern const struct osmo_conv_code gmr1_conv_##name;                                                 \
	const struct osmo_conv_code gmr1_conv_##name = {n, k, 0, term, t.out, t.nxt, nullptr, nullptr, nullptr};

CONV(k5_12, 2, 5, CONV_TERM_FLUSH, t_k5_12)
CONV(k5_13, 3, 5, CONV_TERM_FLUSH, t_k5_13)
CONV(k5_14, 4, 5, CONV_TERM_FLUSH, t_k5_14)
CONV(k5_15, 5, 5, CONV_TERM_FLUSH, t_k5_15)
CONV(k6_14, 4, 6, CONV_TERM_FLUSH, t_k6_14)
CONV(k9_12, 2, 9, CONV_TERM_FLUSH, t_k9_12)
CONV(k9_13, 3, 9, CONV_TERM_FLUSH, t_k9_13)
CONV(k9_14, 4, 9, CONV_TERM_FLUSH, t_k9_14)
CONV(tch3, 2, 7, CONV_TERM_TAIL_BITING, t_tch3)

// crc.c:40-63
extern const struct osmo_crc8gen_code gmr1_crc8;
extern const struct osmo_crc16gen_code gmr1_crc12, gmr1_crc16;
const struct osmo_crc8gen_code gmr1_crc8 = {8, 0x9b, 0x00, 0x00};
const struct osmo_crc16gen_code gmr1_crc12 = {12, 0x80f, 0x0000, 0x0000};
const struct osmo_crc16gen_code gmr1_crc16 = {16, 0x1021, 0x0000, 0x0000};

// scheme, mask steps L, code outputs N, r, position of the one "2" (-1: none), punctured mask positions
#define PUNCT(name, L, N, r, two, ...)                                                                   \
	namespace { constexpr int z_##name[] = {-1, __VA_ARGS__}; }                                          \
	extern const PunctStore<(L) * (N)> gmr1_punct_##name;                                                \
	const PunctStore<(L) * (N)> gmr1_punct_##name =                                                      \
		make_punct<(L) * (N)>(r, L, N, two, z_##name, (int)(sizeof(z_##name) / sizeof(int)));

PUNCT(k5_12_P23, 3, 2, 2, -1, 0, 3)
PUNCT(k5_12_P25, 5, 2, 2, -1, 1, 5)
PUNCT(k5_12_Ps25, 5, 2, 2, -1, 5, 9)
PUNCT(k5_12_P311, 11, 2, 3, -1, 1, 5, 11)
PUNCT(k5_12_P412, 12, 2, 4, -1, 1, 5, 9, 13)
PUNCT(k5_12_Ps412, 12, 2, 4, -1, 11, 15, 19, 23)
PUNCT(k5_12_P12, 2, 2, 1, -1, 3)
PUNCT(k5_12_Ps12, 2, 2, 1, -1, 1)
PUNCT(k5_12_A, 4, 2, 0, -1)
PUNCT(k5_12_B, 4, 2, 1, -1, 1)
PUNCT(k5_12_C, 4, 2, 2, -1, 1, 5)
PUNCT(k5_12_D, 4, 2, 3, -1, 0, 3, 4)
PUNCT(k5_12_E, 4, 2, 1, 1)
PUNCT(k5_12_P38, 8, 2, 3, -1, 0, 4, 13)
PUNCT(k5_12_P26, 6, 2, 2, -1, 1, 7)
PUNCT(k5_12_P37, 7, 2, 3, -1, 1, 5, 9)
PUNCT(k5_13_P16, 6, 3, 1, -1, 2)
PUNCT(k5_13_P25, 5, 3, 2, -1, 7, 13)
PUNCT(k5_13_P15, 5, 3, 1, -1, 1)
PUNCT(k5_13_Ps15, 5, 3, 1, -1, 13)
PUNCT(k5_13_P78, 8, 3, 7, -1, 0, 1, 5, 9, 17, 19, 22)
PUNCT(k5_15_P23, 3, 5, 2, -1, 7, 14)
PUNCT(k5_15_P53, 3, 5, 5, -1, 3, 6, 7, 13, 14)
PUNCT(k5_15_Ps53, 3, 5, 5, -1, 3, 4, 6, 7, 13)
PUNCT(k7_12_P23, 3, 2, 2, -1, 3, 4)
PUNCT(k7_12_P410, 10, 2, 4, -1, 1, 5, 9, 17)
PUNCT(k7_12_P512, 12, 2, 5, -1, 3, 7, 15, 19, 23)
PUNCT(k7_12_P116, 16, 2, 1, -1, 1)
PUNCT(k7_12_P148, 48, 2, 1, -1, 1)
PUNCT(k7_12_P184, 84, 2, 1, -1, 1)
PUNCT(k7_12_P1152, 152, 2, 1, -1, 1)
PUNCT(k7_12_P45, 5, 2, 4, -1, 0, 5, 6, 9)
PUNCT(k7_12_P245, 5, 2, 4, -1, 1, 2, 5, 6)
PUNCT(k9_12_P13, 3, 2, 1, -1, 1)
PUNCT(k9_12_P47, 7, 2, 4, -1, 0, 5, 9, 13)
PUNCT(k9_12_P34, 4, 2, 3, -1, 3, 4, 7)
PUNCT(k9_12_P17, 7, 2, 1, -1, 1)
PUNCT(k9_12_P19, 9, 2, 1, -1, 0)
PUNCT(k9_12_P26, 6, 2, 2, -1, 1, 7)
PUNCT(k9_12_P110, 10, 2, 1, -1, 0)
PUNCT(k9_12_P14, 4, 2, 1, -1, 1)
PUNCT(k9_12_P45, 5, 2, 4, -1, 0, 4, 7, 9)
PUNCT(k9_12_P234, 4, 2, 3, -1, 1, 2, 5)
PUNCT(k6_14_P45, 5, 4, 4, -1, 1, 5, 11, 19)
PUNCT(k9_14_P148, 8, 4, 14, -1, 1, 2, 5, 7, 9, 11, 13, 14, 18, 21, 22, 25, 26, 30)
PUNCT(k9_14_P65, 5, 4, 6, -1, 0, 5, 12, 13, 15, 17)
PUNCT(k9_13_P12, 2, 3, 1, -1, 3)
PUNCT(k9_13_P1213, 13, 3, 12, -1, 2, 4, 6, 11, 13, 15, 20, 22, 24, 29, 31, 33)
PUNCT(k9_13_P44, 4, 3, 4, -1, 2, 3, 7, 11)
PUNCT(k9_13_P33, 3, 3, 3, -1, 0, 4, 8)
PUNCT(k9_13_P65, 5, 3, 6, -1, 1, 3, 7, 8, 9, 14)
}  // extern "C"
