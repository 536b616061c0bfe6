// capi_nt9.cpp -- C ABI of the NT9 burst decoders: FACCH9 (reference include/osmocom/gmr1/l1/facch9.h:39-41)
// and TCH9 (include/osmocom/gmr1/l1/tch9.h:40-53).  The host builds, once per kind, the map that tells the
// kernel where the soft bit of every coded bit of every trellis step sits in the burst (puncturing,
// intra- and inter-burst de-interleaving, descrambling, demultiplexing folded together); everything per
// burst runs on the GPU.

#include "capi_common.h"

#include <mutex>
#include <vector>

#include "../../include/gmr1_hip.h"
#include "../../include/osmocom/gmr1/l1/facch9.h"
#include "../../include/osmocom/gmr1/l1/interleave.h"
#include "../../include/osmocom/gmr1/l1/tch9.h"

using namespace gmr1;

namespace {

struct Punct { int r, L, N; uint8_t mask[15]; };       // mask 0 = punctured (punct.c)
const Punct k5_12_P23 = {2, 3, 2, {0, 1, 1, 0, 1, 1}};
const Punct k5_12_P25 = {2, 5, 2, {1, 0, 1, 1, 1, 0, 1, 1, 1, 1}};
const Punct k5_12_Ps25 = {2, 5, 2, {1, 1, 1, 1, 1, 0, 1, 1, 1, 0}};
const Punct k5_13_P25 = {2, 5, 3, {1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 1}};
const Punct k5_13_P15 = {1, 5, 3, {1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}};
const Punct k5_13_Ps15 = {1, 5, 3, {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1}};
const Punct k5_15_P23 = {2, 3, 5, {1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 0}};
const Punct k5_15_P53 = {5, 3, 5, {1, 1, 1, 0, 1, 1, 0, 0, 1, 1, 1, 1, 1, 0, 0}};
const Punct k5_15_Ps53 = {5, 3, 5, {1, 1, 1, 0, 0, 1, 0, 0, 1, 1, 1, 1, 1, 0, 1}};

// gmr1_puncturer_generate (punct.c:48-133): punctured[ii] for the cl unpunctured coded bits
std::vector<uint8_t> punctured_bits(int cl, int N, const Punct *pre, const Punct *mn, const Punct *post, int repeat)
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

struct Nt9Kind { int N, len, l2_bytes, intra, x_off, inter; };
const Nt9Kind kKinds[4] = {
	{5, 144, 18, 81, 0, 1},     // TCH9 2k4: k5_15, tch9.c:56-62
	{3, 240, 30, 81, 0, 1},     // TCH9 4k8: k5_13, tch9.c:64-70
	{2, 480, 60, 81, 0, 1},     // TCH9 9k6: k5_12, tch9.c:72-78
	{2, 316, 38, 80, 4, 0},     // FACCH9:   k5_12 len 316, facch9.c:42-48, 4 + 4 pad bits
};

std::vector<uint32_t> build_map(int kind)
{
	const Nt9Kind &kd = kKinds[kind];
	const int S = kd.len + 4, cl = S * kd.N;
	std::vector<uint8_t> punct(cl, 0);
	if (kind == 0) punct = punctured_bits(cl, 5, &k5_15_P53, &k5_15_P23, &k5_15_Ps53, 41);
	if (kind == 1) punct = punctured_bits(cl, 3, &k5_13_P15, &k5_13_P25, &k5_13_Ps15, 41);
	if (kind == 2) punct = punctured_bits(cl, 2, &k5_12_P25, &k5_12_P23, &k5_12_Ps25, 158);
	// scrambler over the 648 positions of bits_epp_x (scramb.c:39-52)
	bool scr[648];
	{
		uint16_t r = 0x4d4b;
		for (int i = 0; i < 648; i++) {
			const uint32_t b = ((r >> 14) ^ r) & 1u;
			r = (uint16_t)((r << 1) | b);
			scr[i] = b != 0;
		}
	}
	std::vector<uint32_t> map(cl);
	int q = 0;                                   // index into bits_c (the sent coded bits)
	for (int ii = 0; ii < cl; ii++) {
		if (punct[ii]) {
			map[ii] = 0x80000000u;
			continue;
		}
		// bits_c[q] = ep[intra * ((5 q) & 7) + (q >> 3)]   (gmr1_deinterleave_intra, interleave.c:73-87)
		const int xi = kd.intra * ((5 * q) & 7) + (q >> 3) + kd.x_off;
		const int back = kd.inter ? 2 - (xi % 3) : 0;      // gmr1_deinterleave_inter, N = 3 (interleave.c:163-186)
		const int mi = xi < 52 ? xi : xi + 10;             // bits_my: 52 | sacch 10 | 596
		const int ei = mi < 52 ? mi : mi + 4;              // bits_e:  52 | status 4 | 606
		map[ii] = (uint32_t)ei | (scr[xi] ? 0x400u : 0u) | ((uint32_t)mi << 11) | ((uint32_t)back << 21);
		q++;
	}
	return map;
}

std::mutex g_mu;
struct DevMap { int device; int kind; uint32_t *d; };
std::vector<DevMap> g_maps;

int get_map(int kind, const uint32_t **out)
{
	int dev = 0;
	HIP_TRY(hipGetDevice(&dev));
	std::lock_guard<std::mutex> lk(g_mu);
	for (const DevMap &m : g_maps)
		if (m.device == dev && m.kind == kind) {
			*out = m.d;
			return 0;
		}
	const std::vector<uint32_t> map = build_map(kind);
	uint32_t *d = nullptr;
	HIP_TRY(hipMalloc(&d, map.size() * 4));
	HIP_TRY(hipMemcpy(d, map.data(), map.size() * 4, hipMemcpyHostToDevice));
	g_maps.push_back({dev, kind, d});
	*out = d;
	return 0;
}

int nt9_dev(hipStream_t st, int kind, int n, int seq_len, const int8_t *ebits, const uint8_t *ciph,
            uint8_t *l2, int8_t *sacch, int8_t *status, int32_t *crc, int32_t *conv)
{
	if (n < 0 || !ebits || !l2)
		return fail(-EINVAL, "nt9: ebits / l2 are required");
	if (kind < 0 || kind > 3)
		return fail(-EINVAL, "nt9: mode %d unknown", kind);
	if (seq_len < 1 || (n % seq_len) != 0)
		return fail(-EINVAL, "nt9: %d bursts are not a whole number of sequences of %d", n, seq_len);
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	const uint32_t *map;
	r = get_map(kind, &map);
	if (r) return r;
	Nt9Args a;
	std::memset(&a, 0, sizeof(a));
	a.n = n; a.seq_len = seq_len; a.kind = kind; a.conv_acc = conv_acc(); a.N = kKinds[kind].N; a.len = kKinds[kind].len;
	a.map = map; a.ebits = ebits; a.ciph = ciph; a.l2 = l2; a.l2_bytes = kKinds[kind].l2_bytes;
	a.sacch = sacch; a.status = status; a.crc = crc; a.conv = conv;
	HIP_TRY(launch_nt9(a, st));
	return 0;
}

}  // namespace

namespace gmr1 {
// TCH9 bursts of several interleaver runs of unequal length, run after run, in one launch: seq_pos[i] (device)
// = position of burst i in its run (the receive loop's TCH9 follow-up, capi_rx.cpp)
int tch9_runs_dev_impl(hipStream_t st, int mode, int n, const int32_t *seq_pos, const int8_t *ebits, const uint8_t *ciph,
                       uint8_t *l2, int32_t *conv)
{
	if (n < 0 || mode < 0 || mode > 2 || !seq_pos || !ebits || !l2)
		return fail(-EINVAL, "tch9 runs: bad arguments");
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	const uint32_t *map;
	r = get_map(mode, &map);
	if (r) return r;
	Nt9Args a;
	std::memset(&a, 0, sizeof(a));
	a.n = n; a.seq_len = 1; a.seq_pos = seq_pos; a.kind = mode; a.conv_acc = conv_acc(); a.N = kKinds[mode].N; a.len = kKinds[mode].len;
	a.map = map; a.ebits = ebits; a.ciph = ciph; a.l2 = l2; a.l2_bytes = kKinds[mode].l2_bytes; a.conv = conv;
	HIP_TRY(launch_nt9(a, st));
	return 0;
}
}  // namespace gmr1

namespace {

int nt9_host(int kind, int n, int seq_len, const int8_t *ebits, const uint8_t *ciph,
             uint8_t *l2, int8_t *sacch, int8_t *status, int32_t *crc, int32_t *conv)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n <= 0) return 0;
	if (!ebits || !l2)
		return fail(-EINVAL, "nt9: ebits / l2 are required");
	if (kind < 0 || kind > 3)
		return fail(-EINVAL, "nt9: mode %d unknown", kind);
	const int nb = kKinds[kind].l2_bytes;
	DBuf d_e, d_c, d_l2, d_sa, d_st, d_crc, d_cv;
	HIP_TRY(d_e.alloc((size_t)n * 662));
	HIP_TRY(d_l2.alloc((size_t)n * nb));
	HIP_TRY(d_sa.alloc((size_t)n * 10));
	HIP_TRY(d_st.alloc((size_t)n * 4));
	HIP_TRY(d_crc.alloc((size_t)n * 4));
	HIP_TRY(d_cv.alloc((size_t)n * 4));
	HIP_TRY(hipMemcpy(d_e.p, ebits, (size_t)n * 662, hipMemcpyHostToDevice));
	if (ciph) {
		HIP_TRY(d_c.alloc((size_t)n * 658));
		HIP_TRY(hipMemcpy(d_c.p, ciph, (size_t)n * 658, hipMemcpyHostToDevice));
	}
	r = nt9_dev(nullptr, kind, n, seq_len, d_e.as<int8_t>(), ciph ? d_c.as<uint8_t>() : nullptr, d_l2.as<uint8_t>(),
	            d_sa.as<int8_t>(), d_st.as<int8_t>(), d_crc.as<int32_t>(), d_cv.as<int32_t>());
	if (r) return r;
	HIP_TRY(hipStreamSynchronize(nullptr));
	HIP_TRY(hipMemcpy(l2, d_l2.p, (size_t)n * nb, hipMemcpyDeviceToHost));
	if (sacch) HIP_TRY(hipMemcpy(sacch, d_sa.p, (size_t)n * 10, hipMemcpyDeviceToHost));
	if (status) HIP_TRY(hipMemcpy(status, d_st.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	if (crc) HIP_TRY(hipMemcpy(crc, d_crc.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	if (conv) HIP_TRY(hipMemcpy(conv, d_cv.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	return 0;
}

}  // namespace

extern "C" {

int gmr1_hip_facch9_decode_batch_dev(void *stream, int n, const int8_t *ebits, const uint8_t *ciph,
                                     uint8_t *l2, int8_t *sacch, int8_t *status, int32_t *crc, int32_t *conv)
{
	if (!crc)
		return fail(-EINVAL, "facch9: crc is required");
	return nt9_dev((hipStream_t)stream, 3, n, 1, ebits, ciph, l2, sacch, status, crc, conv);
}

int gmr1_hip_facch9_decode_batch(int n, const int8_t *ebits, const uint8_t *ciph,
                                 uint8_t *l2, int8_t *sacch, int8_t *status, int32_t *crc, int32_t *conv)
{
	if (n > 0 && !crc)
		return fail(-EINVAL, "facch9: crc is required");
	return nt9_host(3, n, 1, ebits, ciph, l2, sacch, status, crc, conv);
}

int gmr1_hip_tch9_decode_batch_dev(void *stream, int n_chan, int seq_len, int mode, const int8_t *ebits,
                                   const uint8_t *ciph, uint8_t *l2, int8_t *sacch, int8_t *status, int32_t *conv)
{
	if (mode < 0 || mode > 2)
		return fail(-EINVAL, "tch9: mode %d unknown (0 2k4, 1 4k8, 2 9k6)", mode);
	if (n_chan < 0 || seq_len < 1)
		return fail(-EINVAL, "tch9: n_chan / seq_len");
	return nt9_dev((hipStream_t)stream, mode, n_chan * seq_len, seq_len, ebits, ciph, l2, sacch, status, nullptr, conv);
}

int gmr1_hip_tch9_decode_batch(int n_chan, int seq_len, int mode, const int8_t *ebits, const uint8_t *ciph,
                               uint8_t *l2, int8_t *sacch, int8_t *status, int32_t *conv)
{
	if (mode < 0 || mode > 2)
		return fail(-EINVAL, "tch9: mode %d unknown (0 2k4, 1 4k8, 2 9k6)", mode);
	if (n_chan < 0 || seq_len < 1)
		return fail(-EINVAL, "tch9: n_chan / seq_len");
	return nt9_host(mode, n_chan * seq_len, seq_len, ebits, ciph, l2, sacch, status, nullptr, conv);
}

// reference-compatible single call (facch9.h:39-41)
int gmr1_facch9_decode(uint8_t *l2, sbit_t *bits_sacch, sbit_t *bits_status,
                       const sbit_t *bits_e, const ubit_t *ciph, int *conv_rv)
{
	if (!l2 || !bits_sacch || !bits_status || !bits_e)
		return fail(-EINVAL, "gmr1_facch9_decode: NULL argument");
	int32_t crc = -1, conv = 0;
	int r = nt9_host(3, 1, 1, reinterpret_cast<const int8_t *>(bits_e), reinterpret_cast<const uint8_t *>(ciph), l2,
	                 reinterpret_cast<int8_t *>(bits_sacch), reinterpret_cast<int8_t *>(bits_status), &crc, &conv);
	if (r) return r;
	if (conv_rv) *conv_rv = conv;
	return crc;
}

// ---- the reference's stateful single-burst TCH9 call (tch9.h:47-50, interleave.h:40-56) -----------------
// The history the depth-3 de-interleaver needs is kept as the previous two bursts AS RECEIVED (662 soft
// bits + 658 key stream bits each); every call decodes the three-burst sequence on the GPU, where the
// de-interleaving is part of the gather (nt9_kernels.hip), and returns the newest burst's outputs.
static constexpr int kIlSlot = 662 + 658;

int gmr1_interleaver_init(struct gmr1_interleaver *il, int N, int K)
{
	if (!il)
		return fail(-EINVAL, "gmr1_interleaver_init: NULL");
	std::memset(il, 0, sizeof(*il));
	if (N != 3 || K != 648)
		return fail(-EINVAL, "gmr1_interleaver_init: only the (3, 648) geometry of TCH9 is provided");
	uint8_t *b = static_cast<uint8_t *>(std::calloc(2, kIlSlot));
	if (!b)
		return fail(-ENOMEM, "gmr1_interleaver_init: out of memory");
	il->N = N;
	il->K = K;
	il->bits_cpp = b;
	return 0;
}

void gmr1_interleaver_fini(struct gmr1_interleaver *il)
{
	if (!il)
		return;
	std::free(il->bits_cpp);
	std::memset(il, 0, sizeof(*il));
}

void gmr1_tch9_decode(uint8_t *l2, sbit_t *bits_sacch, sbit_t *bits_status, const sbit_t *bits_e,
                      enum gmr1_tch9_mode mode, const ubit_t *ciph, struct gmr1_interleaver *il, int *conv_rv)
{
	static const int kBytes[3] = {18, 30, 60};
	if (!l2 || !bits_e || !il || !il->bits_cpp || il->N != 3 || il->K != 648 || (int)mode < 0 || (int)mode > 2) {
		(void)fail(-EINVAL, "gmr1_tch9_decode: bad argument");
		return;
	}
	const int nb = kBytes[(int)mode];
	// sequence = [burst n-2, burst n-1, this burst]; slot (n & 1) holds burst n-2
	uint8_t *older = il->bits_cpp + (size_t)(il->n & 1) * kIlSlot, *newer = il->bits_cpp + (size_t)((il->n + 1) & 1) * kIlSlot;
	int8_t seq_e[3 * 662];
	uint8_t seq_c[3 * 658];
	std::memcpy(seq_e, older, 662);
	std::memcpy(seq_c, older + 662, 658);
	std::memcpy(seq_e + 662, newer, 662);
	std::memcpy(seq_c + 658, newer + 662, 658);
	std::memcpy(seq_e + 2 * 662, bits_e, 662);
	if (ciph)
		std::memcpy(seq_c + 2 * 658, ciph, 658);
	else
		std::memset(seq_c + 2 * 658, 0, 658);
	uint8_t out[3 * 60] = {0};
	int8_t sa[3 * 10] = {0}, stt[3 * 4] = {0};
	int32_t conv[3] = {0, 0, 0};
	(void)nt9_host((int)mode, 3, 3, seq_e, seq_c, out, sa, stt, nullptr, conv);
	std::memcpy(l2, out + 2 * nb, (size_t)nb);
	if (bits_sacch) std::memcpy(bits_sacch, sa + 20, 10);
	if (bits_status) std::memcpy(bits_status, stt + 8, 4);
	if (conv_rv) *conv_rv = conv[2];
	// this burst replaces burst n-2
	std::memcpy(older, seq_e + 2 * 662, 662);
	std::memcpy(older + 662, seq_c + 2 * 658, 658);
	il->n++;
}

}  // extern "C"
