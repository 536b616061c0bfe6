// host_tables.cpp -- GMR-1 burst formats and modulations as exported C data.
//
// The objects have the reference's names and layouts (include/osmocom/gmr1/sdr/
// nb.h:37-46, pi4cxpsk.h:60-62) because callers pass their addresses
// (gmr1_rx.c:759,809,...).  The numbers are ETSI TS 101 376-5-2 section 7.4 data
// (sync words, field positions); they are assembled at load time from the
// compact descriptions below.
#include <cmath>
#include <cstring>
#include <initializer_list>
#include <vector>

#include "host_tables.h"

namespace {

using Syms = std::initializer_list<int>;
struct SyncDef { int pos; Syms syms; };
struct DataDef { int pos, len; };

gmr1_pi4cxpsk_symbol g_bpsk_syms[2], g_qpsk_syms[4], g_qpsk_bits[4];

// storage pools for the chunk lists the burst structs point to
gmr1_pi4cxpsk_sync g_sync_pool[128];
gmr1_pi4cxpsk_data g_data_pool[64];
int g_sync_used, g_data_used;

gmr1_pi4cxpsk_symbol mk_sym(int idx, int b0, int b1, int quarter_turns)
{
	gmr1_pi4cxpsk_symbol s;
	std::memset(&s, 0, sizeof(s));
	static const float re[4] = {1.f, 0.f, -1.f, 0.f}, im[4] = {0.f, 1.f, 0.f, -1.f};
	s.idx = (short)idx;
	s.data[0] = (ubit_t)b0;
	s.data[1] = (ubit_t)b1;
	s.mod_phase = (float)quarter_turns * (float)M_PI / 2.0f;
	s.mod_val.re = re[quarter_turns & 3];
	s.mod_val.im = im[quarter_turns & 3];
	return s;
}

void build(gmr1_pi4cxpsk_burst &b, gmr1_pi4cxpsk_modulation *mod, int len, int ebits,
           std::initializer_list<std::initializer_list<SyncDef>> seqs,
           std::initializer_list<DataDef> data)
{
	std::memset(&b, 0, sizeof(b));
	b.mod = mod;
	b.guard_pre = 2;
	b.guard_post = 3;
	b.len = len;
	b.ebits = ebits;
	int si = 0;
	for (const auto &seq : seqs) {
		b.sync[si++] = &g_sync_pool[g_sync_used];
		for (const auto &c : seq) {
			gmr1_pi4cxpsk_sync &o = g_sync_pool[g_sync_used++];
			std::memset(&o, 0, sizeof(o));
			o.pos = c.pos;
			o.len = (int)c.syms.size();
			int k = 0;
			for (int v : c.syms)
				o.syms[k++] = (uint8_t)v;
		}
		gmr1_pi4cxpsk_sync &t = g_sync_pool[g_sync_used++];
		std::memset(&t, 0, sizeof(t));
		t.pos = -1;
	}
	b.data = &g_data_pool[g_data_used];
	for (const auto &d : data) {
		g_data_pool[g_data_used].pos = d.pos;
		g_data_pool[g_data_used].len = d.len;
		g_data_used++;
	}
	g_data_pool[g_data_used].pos = -1;
	g_data_pool[g_data_used].len = 0;
	g_data_used++;
}

const Syms kOnes32 = {2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2,
                      2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2};
const Syms kRach17 = {0, 2, 2, 0, 0, 0, 2, 0, 2, 2, 2, 2, 2, 0, 2, 2, 0};

}  // namespace

extern "C" {
struct gmr1_pi4cxpsk_modulation gmr1_pi2cbpsk, gmr1_pi4cbpsk, gmr1_pi4cqpsk;
struct gmr1_pi4cxpsk_burst gmr1_bcch_burst, gmr1_dc2_burst, gmr1_dc6_burst, gmr1_dc12_burst,
	gmr1_nt3_speech_burst, gmr1_nt3_facch_burst, gmr1_nt6_burst, gmr1_nt9_burst,
	gmr1_rach_burst, gmr1_sdcch_burst;
}

namespace gmr1 {

gmr1_pi4cxpsk_burst *const kBuiltin[GMR1_HIP_N_BURSTS] = {
	&gmr1_bcch_burst, &gmr1_dc2_burst, &gmr1_dc6_burst, &gmr1_dc12_burst,
	&gmr1_nt3_speech_burst, &gmr1_nt3_facch_burst, &gmr1_nt6_burst, &gmr1_nt9_burst,
	&gmr1_rach_burst, &gmr1_sdcch_burst,
};
const char *const kBuiltinName[GMR1_HIP_N_BURSTS] = {
	"bcch", "dc2", "dc6", "dc12", "nt3_speech", "nt3_facch", "nt6", "nt9", "rach", "sdcch",
};

static bool g_ready = false;

void tables_init()
{
	if (g_ready)
		return;
	g_ready = true;

	// symbol alphabets (pi4cxpsk.c:47-115): CBPSK 0 -> +1, 1 -> -1;
	// CQPSK symbol n has phase n*pi/2 and Gray bits 00,01,11,10
	g_bpsk_syms[0] = mk_sym(0, 0, 0, 0);
	g_bpsk_syms[1] = mk_sym(1, 1, 0, 2);
	g_qpsk_syms[0] = mk_sym(0, 0, 0, 0);
	g_qpsk_syms[1] = mk_sym(1, 0, 1, 1);
	g_qpsk_syms[2] = mk_sym(2, 1, 1, 2);
	g_qpsk_syms[3] = mk_sym(3, 1, 0, 3);
	g_qpsk_bits[0] = g_qpsk_syms[0];   // 00
	g_qpsk_bits[1] = g_qpsk_syms[1];   // 01
	g_qpsk_bits[2] = g_qpsk_syms[3];   // 10
	g_qpsk_bits[3] = g_qpsk_syms[2];   // 11

	gmr1_pi2cbpsk = {(float)M_PI / 2.0f, 1, g_bpsk_syms, g_bpsk_syms};
	gmr1_pi4cbpsk = {(float)M_PI / 4.0f, 1, g_bpsk_syms, g_bpsk_syms};
	gmr1_pi4cqpsk = {(float)M_PI / 4.0f, 2, g_qpsk_syms, g_qpsk_bits};

	auto *Q = &gmr1_pi4cqpsk;
	auto *B4 = &gmr1_pi4cbpsk;
	auto *B2 = &gmr1_pi2cbpsk;

	build(gmr1_bcch_burst, Q, 234, 424,
	      {{{28, {0, 2, 2, 0, 0, 0, 2, 0, 2, 2, 2}}, {119, {2, 2, 0}}, {197, {2, 2, 0}}}},
	      {{2, 26}, {39, 80}, {122, 75}, {200, 31}});
	build(gmr1_dc2_burst, Q, 78, 132,
	      {{{28, {0, 1, 2, 3, 0, 3, 0}}}},
	      {{2, 26}, {35, 40}});
	build(gmr1_dc6_burst, Q, 234, 432,
	      {{{28, {0, 0, 0, 2, 2, 0, 2}}, {119, {0, 3, 0}}, {197, {3, 1, 1}}}},
	      {{2, 26}, {35, 84}, {122, 75}, {200, 31}});
	build(gmr1_dc12_burst, B2, 468, 432,
	      {{{10, {0, 0, 1, 0, 0, 0, 1, 1, 1, 1}},
	        {228, {0, 0, 1, 0, 0, 0, 1, 1, 1, 0, 1}},
	        {447, {0, 0, 1, 0, 0, 0, 1, 1, 1, 1}}}},
	      {{2, 8}, {20, 208}, {239, 208}, {457, 8}});
	build(gmr1_nt3_speech_burst, Q, 117, 212,
	      {{{28, {0, 3, 3, 1, 2, 3}}}},
	      {{2, 26}, {34, 80}});
	build(gmr1_nt3_facch_burst, B4, 117, 104,
	      {{{28, {1, 0, 1, 0, 1, 0, 1, 0}}}, {{28, {1, 1, 0, 0, 1, 0, 0, 1}}}},
	      {{2, 26}, {36, 78}});
	build(gmr1_nt6_burst, Q, 234, 434,
	      {{{28, {0, 2, 2, 3, 2, 3}}, {119, {0, 1, 0}}, {197, {2, 3, 0}}},
	       {{28, {0, 0, 0, 2, 2, 0}}, {119, {1, 3, 0}}, {197, {2, 1, 3}}}},
	      {{2, 26}, {34, 85}, {122, 75}, {200, 31}});
	build(gmr1_nt9_burst, Q, 351, 662,
	      {{{28, {0, 2, 2, 3, 2, 3}}, {119, {1, 2, 2}}, {197, {0, 1, 0}}, {275, {2, 3, 0}}},
	       {{28, {0, 0, 0, 2, 2, 0}}, {119, {0, 2, 0}}, {197, {1, 3, 0}}, {275, {2, 1, 3}}}},
	      {{2, 26}, {34, 85}, {122, 75}, {200, 75}, {278, 70}});
	build(gmr1_rach_burst, Q, 351, 494,
	      {{{78, kRach17}, {127, kOnes32}, {191, kOnes32}, {255, kRach17}, {347, {0}}}},
	      {{2, 76}, {95, 32}, {159, 32}, {223, 32}, {272, 75}});
	build(gmr1_sdcch_burst, B4, 234, 208,
	      {{{28, {0, 1, 0, 1, 0, 1, 0}}, {115, {1, 0, 1, 0, 1, 0, 1}}, {197, {0, 1, 0, 1, 0, 1, 1}}},
	       {{28, {0, 0, 1, 1, 0, 0, 1}}, {115, {1, 0, 0, 1, 1, 0, 0}}, {197, {1, 1, 0, 0, 1, 1, 1}}},
	       {{28, {0, 0, 0, 0, 1, 1, 1}}, {115, {1, 0, 0, 0, 0, 1, 1}}, {197, {1, 1, 0, 0, 0, 0, 1}}},
	       {{28, {0, 1, 1, 0, 1, 0, 0}}, {115, {1, 0, 1, 1, 0, 1, 0}}, {197, {0, 1, 0, 1, 1, 0, 1}}}},
	      {{2, 26}, {35, 80}, {122, 75}, {204, 27}});
}

namespace {
struct AutoInit { AutoInit() { tables_init(); } } g_auto_init;
}

int flatten(const gmr1_pi4cxpsk_burst *b, gmr1_hip_burst_flat *out, const char *name)
{
	tables_init();
	if (!b || !b->mod || !b->data)
		return -22;
	std::memset(out, 0, sizeof(*out));
	if (name)
		std::strncpy(out->name, name, sizeof(out->name) - 1);
	out->rotation = b->mod->rotation;
	out->nbits = b->mod->nbits;
	out->guard_pre = b->guard_pre;
	out->guard_post = b->guard_post;
	out->len = b->len;
	out->ebits = b->ebits;
	if (out->nbits < 1 || out->nbits > 2)
		return -22;
	for (int i = 0; i < GMR1_HIP_MAX_SYNC && b->sync[i]; i++) {
		int n = 0;
		for (const gmr1_pi4cxpsk_sync *c = b->sync[i]; c->pos >= 0; c++) {
			if (n >= GMR1_HIP_MAX_CHUNKS || c->len < 0 || c->len > GMR1_HIP_MAX_SYNC_SYMS)
				return -22;
			out->sync[i][n].pos = c->pos;
			out->sync[i][n].len = c->len;
			std::memcpy(out->sync[i][n].syms, c->syms, (size_t)c->len);
			n++;
		}
		out->n_sync_chunks[i] = n;
		out->n_sync = i + 1;
	}
	int n = 0;
	for (const gmr1_pi4cxpsk_data *d = b->data; d->pos >= 0; d++) {
		if (n >= GMR1_HIP_MAX_CHUNKS)
			return -22;
		out->data[n].pos = d->pos;
		out->data[n].len = d->len;
		n++;
	}
	out->n_data = n;
	return 0;
}

int to_dev(const gmr1_hip_burst_flat &f, DevBurst *d)
{
	std::memset(d, 0, sizeof(*d));
	d->rotation = f.rotation;
	d->nbits = f.nbits;
	d->len = f.len;
	d->ebits = f.ebits;
	d->n_sync = f.n_sync;
	for (int i = 0; i < f.n_sync; i++) {
		d->n_chunks[i] = f.n_sync_chunks[i];
		int tl = 0;
		for (int c = 0; c < f.n_sync_chunks[i]; c++) {
			d->sync[i][c].pos = (int16_t)f.sync[i][c].pos;
			d->sync[i][c].len = (int16_t)f.sync[i][c].len;
			std::memcpy(d->sync[i][c].syms, f.sync[i][c].syms, kMaxSyncSyms);
			tl += f.sync[i][c].len;
			if (f.sync[i][c].pos + f.sync[i][c].len > f.len)
				return -22;
		}
		if (tl > kMaxCoef)
			return -22;
		d->sync_tl[i] = tl;
	}
	if (f.len > kMaxLen)
		return -22;
	for (int i = 0; i < kMaxLen; i++)
		d->ord_of_sym[i] = -1;
	d->n_data = f.n_data;
	int cum = 0;
	for (int c = 0; c < f.n_data; c++) {
		d->dpos[c] = (int16_t)f.data[c].pos;
		d->dlen[c] = (int16_t)f.data[c].len;
		d->dcum[c] = (int16_t)cum;
		if (f.data[c].pos >= 0 && f.data[c].pos + f.data[c].len <= f.len)
			for (int i = 0; i < f.data[c].len; i++)
				d->ord_of_sym[f.data[c].pos + i] = (int16_t)(cum + i);
		cum += f.data[c].len;
		if (f.data[c].pos + f.data[c].len > f.len)
			return -22;
	}
	if (cum * f.nbits != f.ebits)
		return -22;
	return 0;
}

}  // namespace gmr1

// ---------------------------------------------------------------------------
// FCCH
// ---------------------------------------------------------------------------
extern "C" {
const struct gmr1_fcch_burst gmr1_fcch_burst        = {0.32f, 3 * 39};     // fcch.c:50-53
const struct gmr1_fcch_burst gmr1_fcch3_lband_burst = {0.32f, 12 * 39};    // fcch.c:59-62
const struct gmr1_fcch_burst gmr1_fcch3_sband_burst = {0.16f, 12 * 39};    // fcch.c:67-70
}

namespace gmr1 {

const struct gmr1_fcch_burst *const kFcchBuiltin[kFcchTabs] = {
	&gmr1_fcch_burst, &gmr1_fcch3_lband_burst, &gmr1_fcch3_sband_burst,
};

void fcch_tables_init(FcchTables *t)
{
	std::memset(t, 0, sizeof(*t));
	const float pif = 3.14159265358979323846f;
	for (int k = 0; k < kFcchTabs; k++) {
		const struct gmr1_fcch_burst *b = kFcchBuiltin[k];
		const int len = b->len;
		t->freq[k] = b->freq;
		t->len[k] = len;
		// single-precision formulas exactly as the reference evaluates them at sps = 1
		const float sq2 = sqrtf(2.0f), sq2d2 = sqrtf(2.0f) / 2.0f;
		const float phase_base = b->freq * 2.0f * pif / (float)len;
		const float halfpos = (float)len / 2.0f;
		const int mid = len >> 1;
		for (int i = 0; i < len; i++) {
			const float pos = ((float)i / 1.0f) - halfpos;
			const float ph = phase_base * (pos * pos);
			t->dual[k][i] = sq2 * cosf(ph);
			t->up[k][i] = make_float2(sq2d2 * cosf(ph), sq2d2 * sinf(ph));
			const float phf = 2.0f * pif * (float)mid / (float)len * (float)i;
			t->shift[k][i] = make_float2((float)cos((double)phf), (float)sin((double)phf));
			const double tw = -2.0 * M_PI * (double)i / (double)len;
			t->twid[k][i] = make_float2((float)cos(tw), (float)sin(tw));
		}
	}
}

}  // namespace gmr1
