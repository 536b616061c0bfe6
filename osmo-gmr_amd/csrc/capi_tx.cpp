// capi_tx.cpp -- C ABI of the transmit direction: the layer-1 channel ENCODERS (reference
// include/osmocom/gmr1/l1/bcch.h:37, ccch.h:37, facch3.h:37-38, tch3.h:37-39, facch9.h:37-39, tch9.h:47-49,
// rach.h:37, xch_dc12.h:37) and the modulator gmr1_pi4cxpsk_mod (include/osmocom/gmr1/sdr/pi4cxpsk.h:115-117).
//
// The host's part is to write each chain down ONCE as an EncPlan -- the position map from payload bits to burst
// bits (CRC tables, trellis-step windows, puncturing, interleavers, scrambler, multiplexing; the structure follows
// the reference's encoders line by line, in index space instead of on data).  Every bit of every burst is then
// computed on the GPU by k_encode (tx_kernels.hip); there is no CPU path.

#include "capi_common.h"

#include <cmath>
#include <mutex>
#include <vector>

#include "../../include/gmr1_hip.h"
#include "../../include/osmocom/gmr1/l1/bcch.h"
#include "../../include/osmocom/gmr1/l1/ccch.h"
#include "../../include/osmocom/gmr1/l1/facch3.h"
#include "../../include/osmocom/gmr1/l1/facch9.h"
#include "../../include/osmocom/gmr1/l1/interleave.h"
#include "../../include/osmocom/gmr1/l1/rach.h"
#include "../../include/osmocom/gmr1/l1/scramb.h"
#include "../../include/osmocom/gmr1/l1/tch3.h"
#include "../../include/osmocom/gmr1/l1/tch9.h"
#include "../../include/osmocom/gmr1/l1/xch_dc12.h"
#include "../../include/osmocom/gmr1/sdr/pi4cxpsk.h"

using namespace gmr1;

namespace {

// ---- a burst bit in index space -------------------------------------------------------------------------
struct Sym {
	int kind = 1;      // 0 coded (window t, poly slot), 1 constant 0, 2 multiplexed-in bit t
	int t = 0, poly = 0;
	int scr = 0;       // scrambler bit xor-ed in
	int ci = 0;        // keystream index + 1
	int d = 0;         // burst delay (inter-burst interleaver)
};
typedef std::vector<Sym> Bits;

Sym aux_bit(int idx)
{
	Sym s;
	s.kind = 2;
	s.t = idx;
	return s;
}

uint16_t pay_lsb(int k) { return (uint16_t)k; }                                       // osmo_pbit2ubit_ext, lsb mode
uint16_t pay_msb(int k) { return (uint16_t)((k >> 3) * 8 + 7 - (k & 7)); }            // osmo_pbit2ubit
uint16_t crc_a(int bit) { return (uint16_t)(0x4000 | bit); }
uint16_t crc_b(int bit) { return (uint16_t)(0x8000 | bit); }

struct Code {
	int N, K, len;
	bool tail_biting;
	unsigned polys[5];                 // bit i = D^i (the comments of reference src/l1/conv.c)
	std::vector<uint8_t> punct;        // per unpunctured coded bit: 1 = not sent (empty: nothing punctured)
};

struct Builder {
	EncPlan P;
	int n_poly = 0;

	Builder() { std::memset(&P, 0, sizeof(P)); P.depth = 1; }

	int poly_slot(uint32_t mask)
	{
		for (int i = 0; i < n_poly; i++)
			if (P.poly[i] == mask)
				return i;
		if (n_poly == 8)
			abort();
		P.poly[n_poly] = mask;
		return n_poly++;
	}

	// CRC of `n` payload-sourced bits (osmo_crcXXgen_set_bits, init 0 / no final xor: crc.c:36-63): the register is the
	// xor of one table entry per set bit
	void crc_table(uint16_t *tab, int bits, uint32_t poly, const std::vector<uint16_t> &src)
	{
		const int n = (int)src.size();
		const uint32_t top = 1u << (bits - 1), mask = (top << 1) - 1;
		for (int k = 0; k < n; k++) {
			uint32_t crc = top;                                   // the unit vector's only set bit enters here
			crc = ((crc << 1) ^ poly) & mask;
			for (int i = k + 1; i < n; i++)
				crc = ((crc & top) ? ((crc << 1) ^ poly) : (crc << 1)) & mask;
			tab[src[k]] ^= (uint16_t)crc;
		}
	}

	// extended information word of one code: [K-1 preset | info | K-1 flush zeros]; returns the offset of step 0's window
	int add_word(int K, bool tail_biting, const std::vector<uint16_t> &info)
	{
		const int base = P.n_ext, len = (int)info.size();
		int o = base;
		for (int i = 0; i < K - 1; i++)
			P.ext_src[o++] = tail_biting ? info[len - (K - 1) + i] : (uint16_t)0xffff;
		for (int i = 0; i < len; i++)
			P.ext_src[o++] = info[i];
		if (!tail_biting)
			for (int i = 0; i < K - 1; i++)
				P.ext_src[o++] = 0xffff;
		if (o > kEncMaxExt)
			abort();
		P.n_ext = o;
		return base;
	}

	// osmo_conv_encode over the word at `base`: step t reads the register (state << 1 | bit), bit i = D^i = info[t - i];
	// in the extended word that is window bit K-1-i.  Output bit j of a step = poly j, MSB of next_output first.
	Bits conv(const Code &c, int base)
	{
		Bits out;
		const int steps = c.len + (c.tail_biting ? 0 : c.K - 1);
		int slots[5];
		for (int j = 0; j < c.N; j++) {
			uint32_t m = 0;
			for (int i = 0; i < c.K; i++)
				if ((c.polys[j] >> i) & 1)
					m |= 1u << (c.K - 1 - i);
			slots[j] = poly_slot(m);
		}
		for (int t = 0, idx = 0; t < steps; t++)
			for (int j = 0; j < c.N; j++, idx++) {
				if (!c.punct.empty() && c.punct[idx])
					continue;
				Sym s;
				s.kind = 0;
				s.t = base + t;
				s.poly = slots[j];
				out.push_back(s);
			}
		return out;
	}

	// bits that are sent as they are (class-2 speech bits): a window of one bit
	Bits plain(int base, int n)
	{
		Bits out;
		const int slot = poly_slot(1u);
		for (int i = 0; i < n; i++) {
			Sym s;
			s.kind = 0;
			s.t = base + i;
			s.poly = slot;
			out.push_back(s);
		}
		return out;
	}

	void finish(const Bits &e)
	{
		if ((int)e.size() > kEncMaxOut)
			abort();
		P.n_out = (int)e.size();
		for (int i = 0; i < P.n_out; i++) {
			const Sym &s = e[i];
			P.out[i] = (uint32_t)s.t | ((uint32_t)s.poly << 10) | ((uint32_t)s.kind << 13) | ((uint32_t)s.scr << 15) |
			           ((uint32_t)s.ci << 16) | ((uint32_t)s.d << 26);
		}
	}
};

// gmr1_interleave_intra (interleave.c:48-61)
Bits interleave_intra(const Bits &in, int off, int N)
{
	Bits out(8 * N);
	for (int kc = 0; kc < 8 * N; kc++)
		out[N * ((5 * kc) & 7) + (kc >> 3)] = in[off + kc];
	return out;
}

// gmr1_scramble_ubit (scramb.c:39-52, 83-93): 15-bit LFSR 0x4d4b, restarted for every call
void scramble(Bits &b, int off, int n)
{
	uint16_t r = 0x4d4b;
	for (int i = 0; i < n; i++) {
		const int bit = ((r >> 14) ^ r) & 1;
		r = (uint16_t)((r << 1) | bit);
		b[off + i].scr ^= bit;
	}
}

void cipher(Bits &b, int off, int n, int c0)
{
	for (int i = 0; i < n; i++)
		b[off + i].ci = c0 + i + 1;
}

void append(Bits &dst, const Bits &src, int off, int n) { dst.insert(dst.end(), src.begin() + off, src.begin() + off + n); }

std::vector<uint16_t> lsb_bits(int first, int n)
{
	std::vector<uint16_t> v(n);
	for (int i = 0; i < n; i++)
		v[i] = pay_lsb(first + i);
	return v;
}

const unsigned k5_12[2] = {0x19, 0x17};                          // conv.c:123-128
const unsigned k5_13[3] = {0x15, 0x1b, 0x1f};                    // conv.c:148-154
const unsigned k5_14[4] = {0x19, 0x17, 0x15, 0x1f};              // conv.c:174-181
const unsigned k5_15[5] = {0x15, 0x1b, 0x1f, 0x1d, 0x17};        // conv.c:201-209
const unsigned k9_13[3] = {0x1ed, 0x19b, 0x127};                 // conv.c:345-351
const unsigned k7_tch3[2] = {0x6d, 0x4f};                        // conv.c:518-523

Code make_code(int N, int K, int len, bool tb, const unsigned *polys)
{
	Code c;
	c.N = N; c.K = K; c.len = len; c.tail_biting = tb;
	for (int j = 0; j < N; j++)
		c.polys[j] = polys[j];
	return c;
}

// 192 information bits + CRC16 (crc.c:58-63, poly 0x1021): BCCH, CCCH, xCH
std::vector<uint16_t> info_crc16(Builder &b, int n_bits)
{
	std::vector<uint16_t> info = lsb_bits(0, n_bits);
	b.crc_table(b.P.crc_tab, 16, 0x1021, info);
	for (int i = 0; i < 16; i++)
		info.push_back(crc_a(15 - i));
	return info;
}

// ---- the chains --------------------------------------------------------------------------------------------
enum PlanId { kPlBcch = 0, kPlCcch, kPlFacch3, kPlTch3M0, kPlTch3M1, kPlFacch9, kPlTch9_2k4, kPlTch9_4k8, kPlTch9_9k6,
              kPlRach, kPlXch, kPlCount };

void plan_bcch_ccch(Builder &b, bool ccch)          // bcch.c:60-81, ccch.c:60-83
{
	b.P.n_in0 = 24;
	const Code c = make_code(2, 5, 208, false, k5_12);
	const int base = b.add_word(5, false, info_crc16(b, 192));
	const Bits coded = b.conv(c, base);                            // 424
	const Bits il = interleave_intra(coded, 0, 53);
	Bits e;
	if (ccch) e.resize(4);                                         // 4 + 4 padding zeros, scrambled with the rest
	append(e, il, 0, 424);
	if (ccch) e.resize(432);
	scramble(e, 0, (int)e.size());
	b.finish(e);
}

void plan_facch3(Builder &b)                        // facch3.c:65-116
{
	b.P.n_in0 = 10;
	b.P.n_aux0 = 32;
	b.P.n_ciph = 384;
	const Code c = make_code(4, 5, 92, false, k5_14);
	const int base = b.add_word(5, false, info_crc16(b, 76));
	const Bits coded = b.conv(c, base);                            // 384
	Bits cp(384);
	for (int i = 0; i < 384; i++)
		cp[(i & 3) * 96 + (i >> 2)] = coded[i];
	Bits e;
	for (int bu = 0; bu < 4; bu++) {
		Bits x = interleave_intra(cp, 96 * bu, 12);
		scramble(x, 0, 96);
		cipher(x, 0, 96, 96 * bu);
		append(e, x, 0, 22);
		for (int j = 0; j < 8; j++)
			e.push_back(aux_bit(8 * bu + j));
		append(e, x, 22, 74);
	}
	b.finish(e);
}

int tch3_perm(int kc)                               // tch3.c:60-76 (position of coded bit kc in the frame's 104 bits)
{
	const int ii = kc % 24, ij = kc / 24;
	return ii < 8 ? ij + 5 * ii : ij + 4 * ii + 8;
}

void plan_tch3(Builder &b, int m)                   // tch3.c:78-118 with the conv_encode arguments the right way round
{
	b.P.n_in0 = 20;                                                // frame0 | frame1
	b.P.n_aux0 = 4;
	b.P.n_ciph = 208;
	Code c = make_code(2, 7, 48, true, k7_tch3);
	c.punct.assign(96, 0);
	for (int i = 0; i < 24; i++)
		c.punct[4 * i + 3] = 1;                                    // P(1;2), punct.c:239-248
	Bits epp(208);
	for (int f = 0; f < 2; f++) {
		std::vector<uint16_t> cls1(48), cls2(32);
		for (int k = 0; k < 48; k++) cls1[k] = pay_msb(80 * f + k);
		for (int k = 0; k < 32; k++) cls2[k] = pay_msb(80 * f + 48 + k);
		const int b1 = b.add_word(7, true, cls1);
		const int b2 = b.add_word(1, false, cls2);
		Bits cb = b.conv(c, b1);                                   // 72
		const Bits raw = b.plain(b2, 32);
		append(cb, raw, 0, 32);                                    // 104
		Bits ep(104);
		for (int kc = 0; kc < 104; kc++)
			ep[tch3_perm(kc)] = cb[kc];
		for (int j = 0; j < 104; j++)
			epp[m ? 104 * f + j : 2 * j + f] = ep[j];
	}
	scramble(epp, 0, 208);
	cipher(epp, 0, 208, 0);
	Bits e;
	append(e, epp, 0, 52);
	for (int j = 0; j < 4; j++)
		e.push_back(aux_bit(j));
	append(e, epp, 52, 156);
	b.finish(e);
}

struct Punct { int L; uint8_t mask[15]; };          // mask 0 = punctured (punct.c)
const Punct P12_23 = {3, {0, 1, 1, 0, 1, 1}}, P12_25 = {5, {1, 0, 1, 1, 1, 0, 1, 1, 1, 1}}, P12_s25 = {5, {1, 1, 1, 1, 1, 0, 1, 1, 1, 0}};
const Punct P13_25 = {5, {1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 1}}, P13_15 = {5, {1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}},
            P13_s15 = {5, {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1}};
const Punct P15_23 = {3, {1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 0}}, P15_53 = {3, {1, 1, 1, 0, 1, 1, 0, 0, 1, 1, 1, 1, 1, 0, 0}},
            P15_s53 = {3, {1, 1, 1, 0, 0, 1, 0, 0, 1, 1, 1, 1, 1, 0, 1}};

// gmr1_puncturer_generate (punct.c:48-133) as a flag per unpunctured coded bit
std::vector<uint8_t> puncture(int cl, int N, const Punct *pre, const Punct *mn, const Punct *post, int repeat)
{
	std::vector<uint8_t> p(cl, 0);
	int ii = 0, lim = cl;
	if (pre)
		for (int ip = 0; ii < lim && ip < pre->L * N; ii++, ip++)
			if (!pre->mask[ip]) p[ii] = 1;
	if (post)
		lim -= post->L * N;
	for (int i = 0; i < repeat; i++)
		for (int ip = 0; ii < lim && ip < mn->L * N; ii++, ip++)
			if (!mn->mask[ip]) p[ii] = 1;
	if (post) {
		ii = lim;
		for (int ip = 0; ii > 0 && ip < post->L * N && ii < cl; ii++, ip++)
			if (!post->mask[ip]) p[ii] = 1;
	}
	return p;
}

// the NT9 burst around 648 coded positions: scramble, SACCH, cipher, status (facch9.c:84-103, tch9.c:114-136)
void nt9_finish(Builder &b, Bits &x)
{
	b.P.n_aux0 = 10;
	b.P.n_aux1 = 4;
	b.P.n_ciph = 658;
	scramble(x, 0, 648);
	Bits my;
	append(my, x, 0, 52);
	for (int j = 0; j < 10; j++)
		my.push_back(aux_bit(j));
	append(my, x, 52, 596);
	cipher(my, 0, 658, 0);
	Bits e;
	append(e, my, 0, 52);
	for (int j = 0; j < 4; j++)
		e.push_back(aux_bit(10 + j));
	append(e, my, 52, 606);
	b.finish(e);
}

void plan_facch9(Builder &b)                        // facch9.c:60-104
{
	b.P.n_in0 = 38;
	const Code c = make_code(2, 5, 316, false, k5_12);
	const int base = b.add_word(5, false, info_crc16(b, 300));
	const Bits coded = b.conv(c, base);                            // 640
	const Bits il = interleave_intra(coded, 0, 80);
	Bits x(4);
	append(x, il, 0, 640);
	x.resize(648);
	nt9_finish(b, x);
}

void plan_tch9(Builder &b, int mode)                // tch9.c:56-79, 81-137
{
	static const int len[3] = {144, 240, 480}, N[3] = {5, 3, 2};
	Code c = make_code(N[mode], 5, len[mode], false, mode == 0 ? k5_15 : mode == 1 ? k5_13 : k5_12);
	const int cl = (len[mode] + 4) * N[mode];
	if (mode == 0) c.punct = puncture(cl, 5, &P15_53, &P15_23, &P15_s53, 41);
	if (mode == 1) c.punct = puncture(cl, 3, &P13_15, &P13_25, &P13_s15, 41);
	if (mode == 2) c.punct = puncture(cl, 2, &P12_25, &P12_23, &P12_s25, 158);
	b.P.n_in0 = len[mode] / 8;
	b.P.depth = 3;
	const int base = b.add_word(5, false, lsb_bits(0, len[mode]));
	const Bits coded = b.conv(c, base);
	if (coded.size() != 648)
		abort();
	Bits x = interleave_intra(coded, 0, 81);
	for (int jk = 0; jk < 648; jk++)
		x[jk].d = jk % 3;                                          // gmr1_interleave_inter, N = 3 (interleave.c:128-158)
	nt9_finish(b, x);
}

void plan_rach(Builder &b)                          // rach.c:44-66, 78-136
{
	b.P.n_in0 = 18;
	b.P.n_in1 = 1;                                                 // the SB mask rides as payload byte 18
	const std::vector<uint16_t> u1 = lsb_bits(0, 16), u2 = lsb_bits(16, 123);
	b.crc_table(b.P.crc_tab, 8, 0x9b, u1);                         // crc.c:36-44
	for (int i = 0; i < 8; i++)
		b.P.crc_tab[144 + i] ^= (uint16_t)(1u << i);               // crc bit ^= sb_mask bit, rach.c:104-105
	b.crc_table(b.P.crc_tab2, 12, 0x80f, u2);                      // crc.c:46-54
	std::vector<uint16_t> info = u2;
	for (int i = 0; i < 12; i++) info.push_back(crc_b(11 - i));
	info.insert(info.end(), u1.begin(), u1.end());
	for (int i = 0; i < 8; i++) info.push_back(crc_a(7 - i));      // 159
	Code c = make_code(4, 5, 159, false, k5_14);
	c.punct.assign(163 * 4, 0);
	for (int i = 0; i < 135; i++)
		c.punct[4 * i + 2] = c.punct[4 * i + 3] = 1;
	const int base = b.add_word(5, false, info);
	const Bits coded = b.conv(c, base);                            // 382
	const Bits e1p = interleave_intra(coded, 270, 14);             // 112
	Bits e2p = interleave_intra(coded, 0, 33);                     // 264
	append(e2p, coded, 264, 6);
	Bits x;
	append(x, e1p, 0, 112);
	append(x, e2p, 0, 270);
	append(x, e1p, 0, 112);
	scramble(x, 0, 494);
	Bits e;
	append(e, x, 112, 136);
	append(e, x, 0, 112);
	append(e, x, 382, 112);
	append(e, x, 248, 134);
	b.finish(e);
}

void plan_xch(Builder &b)                           // xch_dc12.c:45-54, 64-84
{
	static const uint8_t p1213[39] = {                             // gmr1_punct_k9_13_P1213, punct.c:1105-1125
		1, 1, 0, 1, 0, 1, 0, 1, 1, 1, 1, 0, 1, 0, 1, 0, 1, 1, 1, 1, 0, 1, 0, 1, 0, 1, 1, 1, 1, 0, 1, 0, 1, 0, 1, 1, 1, 1, 1,
	};
	b.P.n_in0 = 24;
	Code c = make_code(3, 9, 208, true, k9_13);
	c.punct.assign(624, 0);
	for (int ii = 0; ii < 624; ii++)
		c.punct[ii] = p1213[ii % 39] == 0;
	const int base = b.add_word(9, true, info_crc16(b, 192));
	const Bits coded = b.conv(c, base);                            // 432
	Bits e = interleave_intra(coded, 0, 54);
	scramble(e, 0, 432);
	b.finish(e);
}

void build_plan(int id, EncPlan *out)
{
	Builder b;
	switch (id) {
	case kPlBcch: plan_bcch_ccch(b, false); break;
	case kPlCcch: plan_bcch_ccch(b, true); break;
	case kPlFacch3: plan_facch3(b); break;
	case kPlTch3M0: plan_tch3(b, 0); break;
	case kPlTch3M1: plan_tch3(b, 1); break;
	case kPlFacch9: plan_facch9(b); break;
	case kPlTch9_2k4: plan_tch9(b, 0); break;
	case kPlTch9_4k8: plan_tch9(b, 1); break;
	case kPlTch9_9k6: plan_tch9(b, 2); break;
	case kPlRach: plan_rach(b); break;
	default: plan_xch(b); break;
	}
	*out = b.P;
}

std::mutex g_mu;
struct DevPlan { int device, id; EncPlan host; EncPlan *d; };
std::vector<DevPlan> g_plans;

int get_plan(int id, const EncPlan **host, const EncPlan **dev_p)
{
	int dev = 0;
	HIP_TRY(hipGetDevice(&dev));
	std::lock_guard<std::mutex> lk(g_mu);
	for (const DevPlan &p : g_plans)
		if (p.device == dev && p.id == id) {
			*host = &p.host;
			*dev_p = p.d;
			return 0;
		}
	g_plans.reserve(64);                       // the pointers handed out stay valid
	DevPlan np;
	np.device = dev;
	np.id = id;
	build_plan(id, &np.host);
	np.d = nullptr;
	HIP_TRY(hipMalloc(&np.d, sizeof(EncPlan)));
	HIP_TRY(hipMemcpy(np.d, &np.host, sizeof(EncPlan), hipMemcpyHostToDevice));
	g_plans.push_back(np);
	*host = &g_plans.back().host;
	*dev_p = g_plans.back().d;
	return 0;
}

int check_args(const EncPlan &p, int n, int seq_len, const uint8_t *in0, const uint8_t *in1, const uint8_t *aux0,
               const uint8_t *aux1, const uint8_t *ebits, const char *what)
{
	if (!in0 || !ebits || (p.n_in1 && !in1) || (p.n_aux0 && !aux0) || (p.n_aux1 && !aux1))
		return fail(-EINVAL, "%s: a required array is NULL", what);
	if (p.depth > 1 && (seq_len < 1 || n % seq_len))
		return fail(-EINVAL, "%s: %d bursts are not a whole number of runs of %d", what, n, seq_len);
	return 0;
}

// every pointer is device memory
int encode_dev(hipStream_t st, int id, int n, int seq_len, const uint8_t *in0, const uint8_t *in1,
               const uint8_t *aux0, const uint8_t *aux1, const uint8_t *ciph, uint8_t *ebits, const char *what)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n < 0)
		return fail(-EINVAL, "%s: n < 0", what);
	if (n == 0)
		return 0;
	const EncPlan *hp, *dp;
	r = get_plan(id, &hp, &dp);
	if (r) return r;
	r = check_args(*hp, n, seq_len, in0, in1, aux0, aux1, ebits, what);
	if (r) return r;
	EncArgs a;
	std::memset(&a, 0, sizeof(a));
	a.n = n; a.seq_len = hp->depth > 1 ? seq_len : 1; a.plan = dp;
	a.in0 = in0; a.in1 = in1; a.aux0 = aux0; a.aux1 = aux1; a.ciph = ciph; a.ebits = ebits;
	HIP_TRY(launch_encode(a, st));
	return 0;
}

// host pointers: H2D -> kernel -> D2H, blocking
int encode_host(int id, int n, int seq_len, const uint8_t *in0, const uint8_t *in1, const uint8_t *aux0,
                const uint8_t *aux1, const uint8_t *ciph, uint8_t *ebits, const char *what)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n < 0)
		return fail(-EINVAL, "%s: n < 0", what);
	if (n == 0)
		return 0;
	const EncPlan *hp, *dp;
	r = get_plan(id, &hp, &dp);
	if (r) return r;
	r = check_args(*hp, n, seq_len, in0, in1, aux0, aux1, ebits, what);
	if (r) return r;
	struct Up { const uint8_t *h; size_t per; DBuf d; };
	Up up[5] = {{in0, (size_t)hp->n_in0, {}}, {in1, (size_t)hp->n_in1, {}}, {aux0, (size_t)hp->n_aux0, {}},
	            {aux1, (size_t)hp->n_aux1, {}}, {ciph, (size_t)hp->n_ciph, {}}};
	for (Up &x : up) {
		if (!x.h || !x.per)
			continue;
		HIP_TRY(x.d.alloc(x.per * (size_t)n));
		HIP_TRY(hipMemcpy(x.d.p, x.h, x.per * (size_t)n, hipMemcpyHostToDevice));
	}
	DBuf d_e;
	HIP_TRY(d_e.alloc((size_t)hp->n_out * (size_t)n));
	r = encode_dev(nullptr, id, n, seq_len, up[0].d.as<uint8_t>(), up[1].d.as<uint8_t>(), up[2].d.as<uint8_t>(),
	               up[3].d.as<uint8_t>(), up[4].d.as<uint8_t>(), d_e.as<uint8_t>(), what);
	if (r) return r;
	HIP_TRY(hipStreamSynchronize(nullptr));
	HIP_TRY(hipMemcpy(ebits, d_e.p, (size_t)hp->n_out * (size_t)n, hipMemcpyDeviceToHost));
	return 0;
}

int tch9_plan(int mode) { return mode == GMR1_TCH9_2k4 ? kPlTch9_2k4 : mode == GMR1_TCH9_4k8 ? kPlTch9_4k8 : mode == GMR1_TCH9_9k6 ? kPlTch9_9k6 : -1; }

// ---- modulator -----------------------------------------------------------------------------------------------
// per-symbol plan of one (burst format, sync sequence): -1 guard, 0..3 training symbol, 4 + k data symbol from ebits[k..]
int mod_plan(const gmr1_hip_burst_flat &f, int sync_id, std::vector<int16_t> *out)
{
	if (sync_id < 0 || sync_id >= f.n_sync)
		return -EINVAL;
	if (f.len < 1 || f.len > 4096 || (f.nbits != 1 && f.nbits != 2))
		return -EINVAL;
	out->assign(f.len, -1);
	for (int c = 0; c < f.n_sync_chunks[sync_id]; c++) {
		const gmr1_hip_chunk &cs = f.sync[sync_id][c];
		for (int i = 0; i < cs.len; i++) {
			if (cs.pos + i < 0 || cs.pos + i >= f.len)
				return -EINVAL;
			(*out)[cs.pos + i] = (int16_t)(cs.syms[i] & 3);
		}
	}
	int k = 0;
	for (int c = 0; c < f.n_data; c++)
		for (int i = 0; i < f.data[c].len; i++, k += f.nbits) {
			const int p = f.data[c].pos + i;
			if (p < 0 || p >= f.len)
				return -EINVAL;
			(*out)[p] = (int16_t)(4 + k);
		}
	return 0;
}

int mod_dev(hipStream_t st, const gmr1_hip_burst_flat &f, int sync_id, int n, const uint8_t *ebits, float *out)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n < 0 || (n > 0 && (!ebits || !out)))
		return fail(-EINVAL, "mod: ebits / out are required");
	std::vector<int16_t> plan;
	r = mod_plan(f, sync_id, &plan);
	if (r) return fail(r, "mod: unsupported burst description or sync_id %d", sync_id);
	if (n == 0)
		return 0;
	// osmo_cxvec_rotate: sample i times e^{j (rotation * i)}, phase and phasor in single precision
	std::vector<float2> rot((size_t)f.len);
	for (int i = 0; i < f.len; i++) {
		const float ph = f.rotation * (float)i;
		rot[i] = make_float2(cosf(ph), sinf(ph));
	}
	DBuf d_p, d_r;
	HIP_TRY(d_p.alloc(plan.size() * 2));
	HIP_TRY(d_r.alloc(rot.size() * 8));
	HIP_TRY(hipMemcpyAsync(d_p.p, plan.data(), plan.size() * 2, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(d_r.p, rot.data(), rot.size() * 8, hipMemcpyHostToDevice, st));
	ModArgs a;
	a.n = n; a.len = f.len; a.nbits = f.nbits; a.n_ebits = f.ebits; a.rotation = f.rotation;
	a.plan = d_p.as<int16_t>(); a.rot = d_r.as<float2>(); a.ebits = ebits; a.out = reinterpret_cast<float2 *>(out);
	HIP_TRY(launch_mod(a, st));
	HIP_TRY(hipStreamSynchronize(st));         // the plan buffer is released on return
	return 0;
}

int mod_host(const gmr1_hip_burst_flat &f, int sync_id, int n, const uint8_t *ebits, float *out)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n < 0 || (n > 0 && (!ebits || !out)))
		return fail(-EINVAL, "mod: ebits / out are required");
	if (n == 0)
		return 0;
	DBuf d_e, d_o;
	HIP_TRY(d_e.alloc((size_t)n * f.ebits));
	HIP_TRY(d_o.alloc((size_t)n * f.len * 8));
	HIP_TRY(hipMemcpy(d_e.p, ebits, (size_t)n * f.ebits, hipMemcpyHostToDevice));
	r = mod_dev(nullptr, f, sync_id, n, d_e.as<uint8_t>(), d_o.as<float>());
	if (r) return r;
	HIP_TRY(hipMemcpy(out, d_o.p, (size_t)n * f.len * 8, hipMemcpyDeviceToHost));
	return 0;
}

int builtin_flat(int burst_id, gmr1_hip_burst_flat *f)
{
	if (burst_id < 0 || burst_id >= GMR1_HIP_N_BURSTS)
		return fail(-EINVAL, "mod: burst id %d unknown", burst_id);
	tables_init();
	return flatten(kBuiltin[burst_id], f, kBuiltinName[burst_id]);
}

// ---- stand-alone primitives -------------------------------------------------------------------------------------
// one blocking H2D -> k_bitmap -> D2H; perm / mask are position tables the host writes down (no data passes through them)
int bitmap_host(int n_in, int n_out, int soft, const void *in, const std::vector<int32_t> *perm,
                const std::vector<uint8_t> *mask, void *out, const char *what)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n_out < 0 || n_in < 0 || (n_out > 0 && (!in || !out)))
		return fail(-EINVAL, "%s: bad argument", what);
	if (n_out == 0)
		return 0;
	DBuf d_in, d_out, d_perm, d_mask;
	HIP_TRY(d_in.alloc((size_t)n_in));
	HIP_TRY(d_out.alloc((size_t)n_out));
	HIP_TRY(hipMemcpy(d_in.p, in, (size_t)n_in, hipMemcpyHostToDevice));
	BitMapArgs a;
	std::memset(&a, 0, sizeof(a));
	a.n = n_out; a.soft = soft; a.in = d_in.as<uint8_t>(); a.out = d_out.as<uint8_t>();
	if (perm) {
		HIP_TRY(d_perm.alloc((size_t)n_out * 4));
		HIP_TRY(hipMemcpy(d_perm.p, perm->data(), (size_t)n_out * 4, hipMemcpyHostToDevice));
		a.perm = d_perm.as<int32_t>();
	}
	if (mask) {
		HIP_TRY(d_mask.alloc((size_t)n_out));
		HIP_TRY(hipMemcpy(d_mask.p, mask->data(), (size_t)n_out, hipMemcpyHostToDevice));
		a.mask = d_mask.as<uint8_t>();
	}
	HIP_TRY(launch_bitmap(a, nullptr));
	HIP_TRY(hipStreamSynchronize(nullptr));
	HIP_TRY(hipMemcpy(out, d_out.p, (size_t)n_out, hipMemcpyDeviceToHost));
	return 0;
}

std::vector<uint8_t> scramble_mask(int len)         // scramb.c:39-52: 15-bit LFSR, seed 0x4d4b
{
	std::vector<uint8_t> m((size_t)(len > 0 ? len : 0));
	uint16_t r = 0x4d4b;
	for (int i = 0; i < len; i++) {
		const int b = ((r >> 14) ^ r) & 1;
		r = (uint16_t)((r << 1) | b);
		m[i] = (uint8_t)b;
	}
	return m;
}

}  // namespace

extern "C" {

// ---- batch forms (include/gmr1_hip.h) ----
int gmr1_hip_bcch_encode_batch_dev(void *stream, int n, const uint8_t *l2, uint8_t *ebits)
{
	return encode_dev((hipStream_t)stream, kPlBcch, n, 1, l2, nullptr, nullptr, nullptr, nullptr, ebits, "bcch_encode");
}
int gmr1_hip_bcch_encode_batch(int n, const uint8_t *l2, uint8_t *ebits)
{
	return encode_host(kPlBcch, n, 1, l2, nullptr, nullptr, nullptr, nullptr, ebits, "bcch_encode");
}
int gmr1_hip_ccch_encode_batch_dev(void *stream, int n, const uint8_t *l2, uint8_t *ebits)
{
	return encode_dev((hipStream_t)stream, kPlCcch, n, 1, l2, nullptr, nullptr, nullptr, nullptr, ebits, "ccch_encode");
}
int gmr1_hip_ccch_encode_batch(int n, const uint8_t *l2, uint8_t *ebits)
{
	return encode_host(kPlCcch, n, 1, l2, nullptr, nullptr, nullptr, nullptr, ebits, "ccch_encode");
}
int gmr1_hip_xch_dc12_encode_batch_dev(void *stream, int n, const uint8_t *l2, uint8_t *ebits)
{
	return encode_dev((hipStream_t)stream, kPlXch, n, 1, l2, nullptr, nullptr, nullptr, nullptr, ebits, "xch_dc12_encode");
}
int gmr1_hip_xch_dc12_encode_batch(int n, const uint8_t *l2, uint8_t *ebits)
{
	return encode_host(kPlXch, n, 1, l2, nullptr, nullptr, nullptr, nullptr, ebits, "xch_dc12_encode");
}
int gmr1_hip_facch3_encode_batch_dev(void *stream, int n, const uint8_t *l2, const uint8_t *bits_s, const uint8_t *ciph,
                                     uint8_t *ebits)
{
	return encode_dev((hipStream_t)stream, kPlFacch3, n, 1, l2, nullptr, bits_s, nullptr, ciph, ebits, "facch3_encode");
}
int gmr1_hip_facch3_encode_batch(int n, const uint8_t *l2, const uint8_t *bits_s, const uint8_t *ciph, uint8_t *ebits)
{
	return encode_host(kPlFacch3, n, 1, l2, nullptr, bits_s, nullptr, ciph, ebits, "facch3_encode");
}
int gmr1_hip_tch3_encode_batch_dev(void *stream, int n, int m, const uint8_t *frames, const uint8_t *bits_s,
                                   const uint8_t *ciph, uint8_t *ebits)
{
	return encode_dev((hipStream_t)stream, m ? kPlTch3M1 : kPlTch3M0, n, 1, frames, nullptr, bits_s, nullptr, ciph, ebits,
	                  "tch3_encode");
}
int gmr1_hip_tch3_encode_batch(int n, int m, const uint8_t *frames, const uint8_t *bits_s, const uint8_t *ciph,
                               uint8_t *ebits)
{
	return encode_host(m ? kPlTch3M1 : kPlTch3M0, n, 1, frames, nullptr, bits_s, nullptr, ciph, ebits, "tch3_encode");
}
int gmr1_hip_facch9_encode_batch_dev(void *stream, int n, const uint8_t *l2, const uint8_t *sacch, const uint8_t *status,
                                     const uint8_t *ciph, uint8_t *ebits)
{
	return encode_dev((hipStream_t)stream, kPlFacch9, n, 1, l2, nullptr, sacch, status, ciph, ebits, "facch9_encode");
}
int gmr1_hip_facch9_encode_batch(int n, const uint8_t *l2, const uint8_t *sacch, const uint8_t *status,
                                 const uint8_t *ciph, uint8_t *ebits)
{
	return encode_host(kPlFacch9, n, 1, l2, nullptr, sacch, status, ciph, ebits, "facch9_encode");
}
int gmr1_hip_tch9_encode_batch_dev(void *stream, int mode, int n, int seq_len, const uint8_t *l2, const uint8_t *sacch,
                                   const uint8_t *status, const uint8_t *ciph, uint8_t *ebits)
{
	if (tch9_plan(mode) < 0)
		return fail(-EINVAL, "tch9_encode: mode %d unknown", mode);
	return encode_dev((hipStream_t)stream, tch9_plan(mode), n, seq_len, l2, nullptr, sacch, status, ciph, ebits, "tch9_encode");
}
int gmr1_hip_tch9_encode_batch(int mode, int n, int seq_len, const uint8_t *l2, const uint8_t *sacch,
                               const uint8_t *status, const uint8_t *ciph, uint8_t *ebits)
{
	if (tch9_plan(mode) < 0)
		return fail(-EINVAL, "tch9_encode: mode %d unknown", mode);
	return encode_host(tch9_plan(mode), n, seq_len, l2, nullptr, sacch, status, ciph, ebits, "tch9_encode");
}
int gmr1_hip_rach_encode_batch_dev(void *stream, int n, const uint8_t *rach, const uint8_t *sb_mask, uint8_t *ebits)
{
	return encode_dev((hipStream_t)stream, kPlRach, n, 1, rach, sb_mask, nullptr, nullptr, nullptr, ebits, "rach_encode");
}
int gmr1_hip_rach_encode_batch(int n, const uint8_t *rach, const uint8_t *sb_mask, uint8_t *ebits)
{
	return encode_host(kPlRach, n, 1, rach, sb_mask, nullptr, nullptr, nullptr, ebits, "rach_encode");
}

// the position map of one chain as the kernel reads it (struct EncPlan, csrc/gmr1_dev.h); host-only, works without a GPU
int gmr1_hip_encoder_plan(int chain, void *buf, int buf_len)
{
	if (chain < 0 || chain >= kPlCount)
		return fail(-EINVAL, "gmr1_hip_encoder_plan: chain %d unknown", chain);
	if (buf) {
		if (buf_len < (int)sizeof(EncPlan))
			return fail(-EINVAL, "gmr1_hip_encoder_plan: the plan has %d bytes", (int)sizeof(EncPlan));
		build_plan(chain, static_cast<EncPlan *>(buf));
	}
	return (int)sizeof(EncPlan);
}

int gmr1_hip_mod_batch_dev(void *stream, int burst_id, int sync_id, int n, const uint8_t *ebits, float *out)
{
	gmr1_hip_burst_flat f;
	int r = builtin_flat(burst_id, &f);
	if (r) return r;
	return mod_dev((hipStream_t)stream, f, sync_id, n, ebits, out);
}
int gmr1_hip_mod_batch(int burst_id, int sync_id, int n, const uint8_t *ebits, float *out)
{
	gmr1_hip_burst_flat f;
	int r = builtin_flat(burst_id, &f);
	if (r) return r;
	return mod_host(f, sync_id, n, ebits, out);
}

// ---- the reference's own single calls; the void ones report a device failure through gmr1_hip_last_error() ----
void gmr1_bcch_encode(ubit_t *bits_e, const uint8_t *l2)
{
	(void)encode_host(kPlBcch, 1, 1, l2, nullptr, nullptr, nullptr, nullptr, bits_e, "gmr1_bcch_encode");
}
void gmr1_ccch_encode(ubit_t *bits_e, const uint8_t *l2)
{
	(void)encode_host(kPlCcch, 1, 1, l2, nullptr, nullptr, nullptr, nullptr, bits_e, "gmr1_ccch_encode");
}
int gmr1_xch_dc12_encode(ubit_t *bits_e, const uint8_t *l2)
{
	return encode_host(kPlXch, 1, 1, l2, nullptr, nullptr, nullptr, nullptr, bits_e, "gmr1_xch_dc12_encode");
}
void gmr1_facch3_encode(ubit_t *bits_e, const uint8_t *l2, const ubit_t *bits_s, const ubit_t *ciph)
{
	(void)encode_host(kPlFacch3, 1, 1, l2, nullptr, bits_s, nullptr, ciph, bits_e, "gmr1_facch3_encode");
}
void gmr1_tch3_encode(ubit_t *bits_e, const uint8_t *frame0, const uint8_t *frame1, const ubit_t *bits_s,
                      const ubit_t *ciph, int m)
{
	if (!frame0 || !frame1) {
		(void)fail(-EINVAL, "gmr1_tch3_encode: NULL frame");
		return;
	}
	uint8_t frames[20];
	std::memcpy(frames, frame0, 10);
	std::memcpy(frames + 10, frame1, 10);
	(void)encode_host(m ? kPlTch3M1 : kPlTch3M0, 1, 1, frames, nullptr, bits_s, nullptr, ciph, bits_e, "gmr1_tch3_encode");
}
void gmr1_facch9_encode(ubit_t *bits_e, const uint8_t *l2, const ubit_t *bits_sacch, const ubit_t *bits_status,
                        const ubit_t *ciph)
{
	(void)encode_host(kPlFacch9, 1, 1, l2, nullptr, bits_sacch, bits_status, ciph, bits_e, "gmr1_facch9_encode");
}
// tch9.h:47-49: stateful, one burst per call.  What the depth-3 interleaver remembers is kept as the payloads of the
// previous two bursts (first 60 bytes of the two private slots of struct gmr1_interleaver, see capi_nt9.cpp); each call
// encodes the run [n-2, n-1, n] on the GPU and returns the newest burst (slots start zeroed = empty interleaver).
void gmr1_tch9_encode(ubit_t *bits_e, const uint8_t *l2, enum gmr1_tch9_mode mode, const ubit_t *bits_sacch,
                      const ubit_t *bits_status, const ubit_t *ciph, struct gmr1_interleaver *il)
{
	static const int kBytes[3] = {18, 30, 60};
	static const size_t kSlot = 662 + 658;
	if (!bits_e || !l2 || !bits_sacch || !bits_status || !il || !il->bits_cpp || il->N != 3 || il->K != 648 ||
	    tch9_plan((int)mode) < 0) {
		(void)fail(-EINVAL, "gmr1_tch9_encode: bad argument");
		return;
	}
	const int nb = kBytes[(int)mode];
	uint8_t *older = il->bits_cpp + (size_t)(il->n & 1) * kSlot, *newer = il->bits_cpp + (size_t)((il->n + 1) & 1) * kSlot;
	uint8_t pay[3 * 60], sa[3 * 10] = {0}, stt[3 * 4] = {0}, out[3 * 662];
	std::vector<uint8_t> cs;
	std::memcpy(pay, older, (size_t)nb);
	std::memcpy(pay + nb, newer, (size_t)nb);
	std::memcpy(pay + 2 * nb, l2, (size_t)nb);
	std::memcpy(sa + 20, bits_sacch, 10);
	std::memcpy(stt + 8, bits_status, 4);
	if (ciph) {
		cs.assign(3 * 658, 0);
		std::memcpy(cs.data() + 2 * 658, ciph, 658);
	}
	if (encode_host(tch9_plan((int)mode), 3, 3, pay, nullptr, sa, stt, ciph ? cs.data() : nullptr, out, "gmr1_tch9_encode"))
		return;
	std::memcpy(bits_e, out + 2 * 662, 662);
	std::memcpy(older, l2, (size_t)nb);                        // this burst replaces burst n-2
	il->n++;
}

void gmr1_rach_encode(ubit_t *bits_e, const uint8_t *rach, uint8_t sb_mask)
{
	(void)encode_host(kPlRach, 1, 1, rach, &sb_mask, nullptr, nullptr, nullptr, bits_e, "gmr1_rach_encode");
}

int gmr1_pi4cxpsk_mod(struct gmr1_pi4cxpsk_burst *burst_type, ubit_t *ebits, int sync_id, struct osmo_cxvec *burst_out)
{
	if (!burst_type || !ebits || !burst_out || !burst_out->data)
		return fail(-EINVAL, "gmr1_pi4cxpsk_mod: NULL argument");
	if (burst_out->max_len < burst_type->len)
		return -ENOMEM;                                            // pi4cxpsk.c:752-756
	tables_init();
	gmr1_hip_burst_flat f;
	int r = flatten(burst_type, &f, "mod");
	if (r) return fail(r, "gmr1_pi4cxpsk_mod: unsupported burst description");
	burst_out->len = burst_type->len;
	return mod_host(f, sync_id, 1, ebits, reinterpret_cast<float *>(burst_out->data));
}

// ---- the stand-alone layer-1 primitives (scramb.h:36-37, interleave.h:36-56); each is one blocking call to the GPU ----
void gmr1_scramble_sbit(sbit_t *out, const sbit_t *in, int len)
{
	const std::vector<uint8_t> m = scramble_mask(len);
	(void)bitmap_host(len, len, 1, in, nullptr, &m, out, "gmr1_scramble_sbit");
}

void gmr1_scramble_ubit(ubit_t *out, const ubit_t *in, int len)
{
	const std::vector<uint8_t> m = scramble_mask(len);
	(void)bitmap_host(len, len, 0, in, nullptr, &m, out, "gmr1_scramble_ubit");
}

void gmr1_interleave_intra(void *out, const void *in, int N)
{
	if (N < 0) { (void)fail(-EINVAL, "gmr1_interleave_intra: N < 0"); return; }
	std::vector<int32_t> perm((size_t)8 * N);
	for (int kc = 0; kc < 8 * N; kc++)
		perm[N * ((5 * kc) & 7) + (kc >> 3)] = kc;                 // interleave.c:48-61
	(void)bitmap_host(8 * N, 8 * N, 0, in, &perm, nullptr, out, "gmr1_interleave_intra");
}

void gmr1_deinterleave_intra(void *out, const void *in, int N)
{
	if (N < 0) { (void)fail(-EINVAL, "gmr1_deinterleave_intra: N < 0"); return; }
	std::vector<int32_t> perm((size_t)8 * N);
	for (int kc = 0; kc < 8 * N; kc++)
		perm[kc] = N * ((5 * kc) & 7) + (kc >> 3);                 // interleave.c:73-87
	(void)bitmap_host(8 * N, 8 * N, 0, in, &perm, nullptr, out, "gmr1_deinterleave_intra");
}

// interleave.c:128-158.  The state is the reference's: N rows of K bits in il->bits_cpp (an object used with these two
// calls must not also be handed to gmr1_tch9_encode / gmr1_tch9_decode, which keep their own history there).
void gmr1_interleave_inter(struct gmr1_interleaver *il, void *bits_epp, void *bits_ep)
{
	if (!il || !il->bits_cpp || il->N != 3 || il->K != 648 || !bits_epp || !bits_ep) {
		(void)fail(-EINVAL, "gmr1_interleave_inter: bad argument");
		return;
	}
	const int N = il->N, K = il->K, cur = il->n % N;
	// staging = [state with this burst in row cur]: the row copy is the upload itself
	std::vector<uint8_t> stage(il->bits_cpp, il->bits_cpp + (size_t)N * K);
	std::memcpy(stage.data() + (size_t)cur * K, bits_ep, (size_t)K);
	std::vector<int32_t> perm((size_t)K);
	for (int jk = 0; jk < K; jk++)
		perm[jk] = ((cur - (jk % N) + N) % N) * K + jk;
	if (bitmap_host(N * K, K, 0, stage.data(), &perm, nullptr, bits_epp, "gmr1_interleave_inter"))
		return;
	std::memcpy(il->bits_cpp + (size_t)cur * K, stage.data() + (size_t)cur * K, (size_t)K);
	il->n++;
}

// interleave.c:163-190
void gmr1_deinterleave_inter(struct gmr1_interleaver *il, void *bits_ep, void *bits_epp)
{
	if (!il || !il->bits_cpp || il->N != 3 || il->K != 648 || !bits_ep || !bits_epp) {
		(void)fail(-EINVAL, "gmr1_deinterleave_inter: bad argument");
		return;
	}
	const int N = il->N, K = il->K, cur = il->n % N;
	// source = [state | received burst]; the new state takes bit jk of row (cur - jk mod N) from the burst
	std::vector<uint8_t> src((size_t)(N + 1) * K);
	std::memcpy(src.data(), il->bits_cpp, (size_t)N * K);
	std::memcpy(src.data() + (size_t)N * K, bits_epp, (size_t)K);
	std::vector<int32_t> perm((size_t)N * K);
	for (int i = 0; i < N; i++)
		for (int jk = 0; jk < K; jk++)
			perm[(size_t)i * K + jk] = (i == (cur - (jk % N) + N) % N) ? N * K + jk : i * K + jk;
	std::vector<uint8_t> state((size_t)N * K);
	if (bitmap_host((N + 1) * K, N * K, 0, src.data(), &perm, nullptr, state.data(), "gmr1_deinterleave_inter"))
		return;
	std::memcpy(il->bits_cpp, state.data(), (size_t)N * K);
	std::memcpy(bits_ep, state.data() + (size_t)((il->n + 1) % N) * K, (size_t)K);
	il->n++;
}

}  // extern "C"
