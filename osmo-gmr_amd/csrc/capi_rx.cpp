// capi_rx.cpp -- the receive control loop of the reference's gmr1_rx application
// (reference src/gmr1_rx.c:605-895, main() :897-975) over MANY BCCH carriers at once.
//
// The reference walks one carrier frame by frame on the CPU: one FCCH acquisition, then per
// 40 ms frame one BCCH or CCCH burst through demod + decode, feeding time / frequency / TDMA
// position back into the next frame.  That feedback only crosses a BCCH frame (rx_bcch is the
// only writer of align / freq_err / fn / sa_*), so between two BCCH frames of one chain every
// burst is independent, and different chains / carriers are independent throughout.  The frame
// loop therefore runs in ROUNDS -- a chain's CCCH bursts up to and including its next BCCH burst,
// then the BCCH feedback -- and it runs them ON THE GPU: k_rx_chain (rx_kernels.hip) walks every
// chain's feedback path from the first frame to the end of the capture and lists its CCCH bursts, one
// k_rx4 batch takes those, k_rx_merge writes the records (the integer control logic is in rx_loop.h);
// the host only collects them.  FCCH acquisition is three batched sweeps
// (rough / rough_multi / fine + snr) over all carriers.  The arithmetic of every step runs on the
// GPU; the host keeps only the per-chain integers the reference keeps in struct chan_desc.
// There is no CPU fallback.
//
// The traffic channels never feed back into that loop, so their follow-ups run after it as batched
// passes of their own (RxRun::tch3_pass, RxRun::tch9_pass).  GSMTAP transport and per-burst stderr
// logging are out of scope; what GSMTAP would have carried comes back as records.

#include "capi_common.h"
#include "fcch_acq.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <vector>

#include "../../include/gmr1_hip.h"

using namespace gmr1;

namespace {

constexpr int kStartDiscard = 8000;   // gmr1_rx.c:52
constexpr int kSymRate = 23400;
constexpr int kFcchLen = 117;         // gmr1_fcch_burst.len, fcch.c:50-54
constexpr int kMaxPeaks = 16;         // gmr1_rx.c:650

float to_hz(float f_rps) { return (kSymRate * f_rps) / (2.0f * 3.14159265358979323846f); }

struct FrameCtx { int align; float freq_err; int fn; };       // what rx_tch3 sees in a frame
struct AssEvt { int frame; int tn, p; float ref_energy; };     // an IMMEDIATE ASSIGNMENT taken from the CCCH

struct RxChain {
	int a;                // carrier index
	int chain;            // chain index within the carrier
	uint64_t base;        // first sample of the carrier in iq
	int len;              // samples of the carrier
	int align;
	float freq_err;
	int fn, delay, stn;
	float bcch_energy;
	bool done;
	std::vector<gmr1_hip_rx_record> rec;
	int n_rec = 0;                       // records of the chain when they went straight to the caller (RxRun::direct)
	std::vector<int> rec_frame;          // frame (index into log) each record belongs to
	std::vector<FrameCtx> log;           // one entry per loop iteration of process_bcch (only with a traffic carrier)
	std::vector<AssEvt> events;
	std::vector<AssEvt> events9;         // ASSIGNMENT COMMAND 1 taken from a FACCH3 (frame, tn)
	std::vector<gmr1_hip_rx_big_record> big;
};

void emit(RxChain &c, uint16_t arfcn, int type, int fn, int tn, const uint8_t *l2, int conv, int frame, int len = 24)
{
	gmr1_hip_rx_record r;
	std::memset(&r, 0, sizeof(r));
	r.arfcn = arfcn;
	r.chain = (uint8_t)c.chain;
	r.type = (uint8_t)type;
	r.fn = (uint32_t)fn;
	r.tn = (uint8_t)tn;
	r.crc = 0;
	r.len = (uint8_t)len;
	r.conv = conv;
	std::memcpy(r.l2, l2, (size_t)len);
	c.rec.push_back(r);
	c.rec_frame.push_back(frame);
}

// TCH3 state of a chain (struct tch3_state, gmr1_rx.c:59-78)
struct Tch3State {
	int active = 0;
	int tn = 0, p = 0, ciph = 0;
	float energy_dkab = 0.f, energy_burst = 0.f;
	int weak_cnt = 0;
	int8_t ebits[104 * 4] = {};
	uint32_t bi_fn[4] = {0, 0, 0, 0};
	int sync_id = 0, burst_cnt = 0;
};

struct TchItem {          // one frame of a chain in which rx_tch3 maps a burst
	int chain_idx, frame;
	int tn, p, e_toa;
};

struct TchJob {           // a decode the walk asks for: a speech burst or a FACCH3 flush
	int chain_idx, frame;
	int is_flush;
	int fn;               // cd->fn when it happens
	int tn;
	int item;             // speech: index of the TchItem whose soft bits are decoded
	int8_t ebits[104 * 4];    // flush: the four stored bursts
	uint32_t bi_fn[4];
};

size_t up16(size_t x) { return (x + 15) & ~(size_t)15; }
size_t up128(size_t x) { return (x + 127) & ~(size_t)127; }

// grow-only pinned host buffer of the calling thread: the burst log of the receive loop comes back through it
// (a fresh hipHostMalloc of some megabytes per call would cost more than the loop itself)
int host_log(size_t bytes, unsigned char **out)
{
	struct Buf {                 // never freed: the runtime may be gone by the time thread-locals are destroyed
		void *p = nullptr;
		size_t n = 0;
	};
	static thread_local Buf b;
	if (b.n < bytes) {
		if (b.p) (void)hipHostFree(b.p);
		b.p = nullptr;
		b.n = 0;
		HIP_TRY(hipHostMalloc(&b.p, bytes + bytes / 4, hipHostMallocDefault));
		b.n = bytes + bytes / 4;
	}
	*out = static_cast<unsigned char *>(b.p);
	return 0;
}


// Device scratch of a traffic pass: one carve-up of the grow-only device workspace (none of the kernels the
// passes launch uses it) instead of dozens of hipMalloc / hipFree pairs per call.
struct Arena {
	unsigned char *base = nullptr;
	size_t cap = 0, off = 0;
	int init(size_t bytes)
	{
		DevState *ds;
		int r = dev_state(&ds);
		if (r) return r;
		void *ws;
		r = dev_workspace(ds, bytes + 256, &ws);
		if (r) return r;
		base = reinterpret_cast<unsigned char *>(((uintptr_t)ws + 127) & ~(uintptr_t)127);
		cap = bytes;
		off = 0;
		return 0;
	}
	void *take(size_t n)
	{
		n = up128(n ? n : 1);
		if (off + n > cap)
			return nullptr;
		void *p = base + off;
		off += n;
		return p;
	}
};
thread_local Arena *g_arena = nullptr;
struct ABuf {                      // DBuf's interface on the current arena
	void *p = nullptr;
	hipError_t alloc(size_t n)
	{
		p = g_arena ? g_arena->take(n) : nullptr;
		return p ? hipSuccess : hipErrorOutOfMemory;
	}
	template <typename T> T *as() { return static_cast<T *>(p); }
};

// One call of gmr1_hip_rx_run*: what the phases share.  The phases run in the order the reference's main()
// runs them (gmr1_rx.c:897-975); each is one member function below.
struct RxRun {
	hipStream_t st;
	int sps;
	const float *iq, *tch, *csd;
	const uint64_t *offset, *length;
	const uint16_t *arfcn;
	const uint8_t *kc;
	int A;                                   // carriers
	int flen;                                // samples of an FCCH burst
	int r = 0;
	std::vector<int32_t> stat, nch;          // per carrier: status, chains followed
	std::vector<int> align, base_align;
	std::vector<float> ferr;
	std::vector<RxChain> chains;
	double t_loop_gpu_us = 0;                // launch to log-on-host
	double t_chain_us = 0;                   // ... of which: launches until the loop's kernels are through (counters on the host)
	// hand-back of a plain BCCH / CCCH run (no traffic follow-up): the records are closed up on the device in the order
	// they are returned in (k_rx_pack) and copied ONCE, as many as there are -- straight into the caller's buffer when
	// that is device memory or pinned host memory, else through the library's pinned block
	gmr1_hip_rx_record *out = nullptr;
	int max_records = 0;
	bool direct = false;
	int direct_total = 0;

	int acquire();        // fcch_single_init + fcch_multi_process
	int frame_loop();     // process_bcch: BCCH / CCCH, in rounds
	int tch3_pass();      // rx_tch3 and its helpers
	int tch9_pass();      // rx_tch9
};


// grow-only device scratch of the acquisition (its sweeps use the library's shared workspace themselves)
static int acq_scratch(size_t bytes, unsigned char **out)
{
	// (per thread AND per device: a thread that moves on to another GPU must not hand that GPU's kernels this one's memory)
	struct Buf { void *p = nullptr; size_t n = 0; int dev = -1; };
	static thread_local Buf b;
	int dev = 0;
	HIP_TRY(hipGetDevice(&dev));
	if (b.n < bytes || b.dev != dev) {
		if (b.p) {
			int back = dev;
			if (b.dev >= 0 && b.dev != dev && hipSetDevice(b.dev) == hipSuccess) {
				(void)hipFree(b.p);
				(void)hipSetDevice(back);
			} else {
				(void)hipFree(b.p);
			}
		}
		b.p = nullptr;
		b.n = 0;
		b.dev = dev;
		HIP_TRY(hipMalloc(&b.p, bytes + bytes / 4));
		b.n = bytes + bytes / 4;
	}
	*out = static_cast<unsigned char *>(b.p);
	return 0;
}

// wall time of the phases of this thread's last gmr1_hip_rx_run* call, microseconds (gmr1_hip_rx_run_last_timing)
static thread_local double t_last_timing[5] = {0, 0, 0, 0, 0};

int RxRun::acquire()
{
	// The five sweeps of fcch_single_init / fcch_multi_process (gmr1_rx.c:605-744) follow each other on the stream
	// without the host: k_acq_glue (fcch_kernels.hip) does the additions and bound checks between them on the device
	// and lays out each next sweep's windows; candidate stages run over all kMaxPeaks slots of every carrier (a slot
	// without a candidate gets a harmless window).  One copy brings every raw sweep result back, and the decisions are
	// then taken here exactly as before, from those numbers.
	static_assert(kMaxPeaks == kAcqPeaks, "candidate slots");
	// (profiling build, GMR1_HIP_RX_TIMING: host-side stamps of this call's stages on stderr)
	static const bool timing = profile_env("GMR1_HIP_RX_TIMING") != nullptr;
	std::chrono::steady_clock::time_point tp[6];
	int n_tp = 0;
	auto stamp = [&] { if (timing && n_tp < 6) tp[n_tp++] = std::chrono::steady_clock::now(); };
	stamp();
	const int wl1 = (330 * kSymRate * sps) / 1000, wl3 = (650 * kSymRate * sps) / 1000;
	std::vector<int> idx;
	for (int i = 0; i < A; i++) {
		if ((uint64_t)align[i] + wl1 > length[i]) { stat[i] = -1; continue; }
		idx.push_back(i);
	}
	const int n = (int)idx.size();
	if (!n)
		return 0;
	const size_t S = (size_t)n * kMaxPeaks;
	// one block, device and pinned host mirror: [per carrier ... | per slot ...]
	size_t o = 0;
	auto take = [&](size_t bytes) { const size_t at = o; o += up128(bytes); return at; };
	// (what the host sends first -- the carriers' parameters, the first sweep's windows and a zeroed peak list -- in front,
	// so that ONE copy starts the chain; what comes back -- toa1 .. snr -- contiguous behind it)
	const size_t o_base = take(n * 8), o_len = take(n * 8), o_stat = take(n * 4), o_align = take(n * 4), o_ba = take(n * 4),
	             o_ferr = take(n * 4), o_can3 = take(n * 4), o_off = take(S * 8), o_peaks = take(S * 4), o_toa1 = take(n * 4),
	             o_rv1 = take(n * 4), o_ftoa = take(n * 4), o_fe = take(n * 4), o_count = take(n * 4), o_ctoa = take(S * 4),
	             o_cfe = take(S * 4), o_snr = take(S * 4), o_live = take(S * 4), o_fs = take(S * 4);
	const size_t total = o;
	unsigned char *d, *h;
	if ((r = acq_scratch(total, &d))) return r;
	if ((r = host_log(total, &h))) return r;
	auto H = [&](size_t at) { return h + at; };
	auto D = [&](size_t at) { return d + at; };
	for (int k = 0; k < n; k++) {
		const int i = idx[k];
		reinterpret_cast<uint64_t *>(H(o_base))[k] = offset[i];
		reinterpret_cast<uint64_t *>(H(o_len))[k] = length[i];
		reinterpret_cast<int32_t *>(H(o_stat))[k] = 0;
		reinterpret_cast<int32_t *>(H(o_align))[k] = align[i];
		reinterpret_cast<int32_t *>(H(o_ba))[k] = 0;
		reinterpret_cast<float *>(H(o_ferr))[k] = 0.f;
		reinterpret_cast<int32_t *>(H(o_can3))[k] = length[i] >= (uint64_t)wl3 ? 1 : 0;
		reinterpret_cast<uint64_t *>(H(o_off))[k] = offset[i] + (uint64_t)align[i];
	}
	// rough_multi leaves slots past `count` unwritten: the copy back must not carry stale numbers
	std::memset(H(o_off) + (size_t)n * 8, 0, (o_toa1 - o_off) - (size_t)n * 8);
	// inputs: everything up to can3, the first sweep's windows, the zeroed peak list -- one copy
	HIP_TRY(hipMemcpyAsync(d, h, o_toa1, hipMemcpyHostToDevice, st));
	stamp();

	AcqArgs g;
	std::memset(&g, 0, sizeof(g));
	g.n = n; g.sps = sps; g.flen = flen; g.wl3 = wl3;
	g.base = reinterpret_cast<const uint64_t *>(D(o_base));
	g.len = reinterpret_cast<const uint64_t *>(D(o_len));
	g.stat = reinterpret_cast<int32_t *>(D(o_stat));
	g.align = reinterpret_cast<int32_t *>(D(o_align));
	g.base_align = reinterpret_cast<int32_t *>(D(o_ba));
	g.ferr = reinterpret_cast<float *>(D(o_ferr));
	g.can3 = reinterpret_cast<const int32_t *>(D(o_can3));
	g.toa1 = reinterpret_cast<const int32_t *>(D(o_toa1));
	g.rv1 = reinterpret_cast<const int32_t *>(D(o_rv1));
	g.ftoa = reinterpret_cast<const int32_t *>(D(o_ftoa));
	g.fe = reinterpret_cast<const float *>(D(o_fe));
	g.peaks = reinterpret_cast<const int32_t *>(D(o_peaks));
	g.count = reinterpret_cast<const int32_t *>(D(o_count));
	g.ctoa = reinterpret_cast<const int32_t *>(D(o_ctoa));
	g.cfe = reinterpret_cast<const float *>(D(o_cfe));
	g.off = reinterpret_cast<uint64_t *>(D(o_off));
	g.fs = reinterpret_cast<float *>(D(o_fs));
	g.live = reinterpret_cast<int32_t *>(D(o_live));
	uint64_t *d_off = g.off;
	float *d_fs = g.fs;

	// What the reference does between two sweeps (k_acq_glue's steps) is done by the producing sweep's last thread
	// (AcqTail, fcch_acq.h) -- four launches fewer in a chain of small dependent ones; the profiling build keeps the
	// other form for the comparison (GMR1_HIP_ACQ_UNFUSED).
	static const bool unfused = profile_env("GMR1_HIP_ACQ_UNFUSED") != nullptr;
	auto tail = [&](int step, bool skip_dead) {
		AcqTail t;
		std::memset(&t, 0, sizeof(t));
		if (!unfused) {
			t.step = step;
			t.skip_dead = skip_dead ? g.live : nullptr;
			t.g = g;
		}
		return t;
	};
	// fcch_single_init (gmr1_rx.c:605-639): rough over 330 ms, then fine
	if ((r = fcch_rough_tail(st, 0, n, sps, wl1, iq, d_off, nullptr, reinterpret_cast<int32_t *>(D(o_toa1)),
	                         reinterpret_cast<int32_t *>(D(o_rv1)), tail(1, false)))) return r;
	if (unfused) HIP_TRY(launch_acq_glue(1, g, st));
	if ((r = fcch_fine_tail(st, 0, 0, n, sps, iq, d_off, nullptr, reinterpret_cast<int32_t *>(D(o_ftoa)),
	                        reinterpret_cast<float *>(D(o_fe)), nullptr, tail(2, false)))) return r;
	if (unfused) HIP_TRY(launch_acq_glue(2, g, st));
	// fcch_multi_process (gmr1_rx.c:643-744); a carrier shorter than 650 ms can only fail here, its dummy window would
	// not fit either: the sweep runs over the others
	std::vector<int> k3;
	for (int k = 0; k < n; k++)
		if (length[idx[k]] >= (uint64_t)wl3)
			k3.push_back(k);
	const bool all3 = (int)k3.size() == n;
	if (!k3.empty()) {
		if (all3) {
			if ((r = fcch_rough_multi_tail(st, 0, n, sps, wl3, iq, d_off, d_fs, reinterpret_cast<int32_t *>(D(o_peaks)),
			                               kMaxPeaks, reinterpret_cast<int32_t *>(D(o_count)), tail(3, false)))) return r;
		} else {
			// mixed lengths: the long-enough carriers one by one at their own slots (rare; captures come in equal lengths)
			for (int k : k3)
				if ((r = gmr1_hip_fcch_rough_multi_batch_dev(st, 0, 1, sps, wl3, iq, d_off + k, d_fs + k,
				                                             reinterpret_cast<int32_t *>(D(o_peaks)) + (size_t)k * kMaxPeaks, kMaxPeaks,
				                                             reinterpret_cast<int32_t *>(D(o_count)) + k))) return r;
		}
	}
	// (mixed lengths, or no carrier long enough: the step that lays out the candidate slots runs as its own launch)
	if (unfused || !all3) HIP_TRY(launch_acq_glue(3, g, st));
	if ((r = fcch_fine_tail(st, 0, 0, (int)S, sps, iq, d_off, d_fs, reinterpret_cast<int32_t *>(D(o_ctoa)),
	                        reinterpret_cast<float *>(D(o_cfe)), nullptr, tail(4, true)))) return r;
	if (unfused) HIP_TRY(launch_acq_glue(4, g, st));
	if ((r = fcch_fine_tail(st, 0, 1, (int)S, sps, iq, d_off, d_fs, nullptr, nullptr, reinterpret_cast<float *>(D(o_snr)),
	                        tail(0, true)))) return r;
	// results back: the peak list (in front of toa1) and toa1 .. snr
	stamp();
	HIP_TRY(hipMemcpyAsync(H(o_peaks), D(o_peaks), o_live - o_peaks, hipMemcpyDeviceToHost, st));
	stamp();
	HIP_TRY(hipStreamSynchronize(st));
	stamp();
	if (timing) {
		auto us = [](auto a, auto b) { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() / 1e3; };
		fprintf(stderr, "acquire: prepare + first copy %.1f us, launches %.1f us, copy back enqueued %.1f us, waited %.1f us\n",
		        us(tp[0], tp[1]), us(tp[1], tp[2]), us(tp[2], tp[3]), us(tp[3], tp[4]));
	}

	// ---- the decisions, from the raw sweep results, in the reference's order ------------------------------------
	const int32_t *toa1 = reinterpret_cast<const int32_t *>(H(o_toa1)), *rv1 = reinterpret_cast<const int32_t *>(H(o_rv1)),
	              *ftoa = reinterpret_cast<const int32_t *>(H(o_ftoa)), *count = reinterpret_cast<const int32_t *>(H(o_count)),
	              *peaks = reinterpret_cast<const int32_t *>(H(o_peaks)), *ctoa = reinterpret_cast<const int32_t *>(H(o_ctoa));
	const float *fe = reinterpret_cast<const float *>(H(o_fe)), *cfe = reinterpret_cast<const float *>(H(o_cfe)),
	            *snr = reinterpret_cast<const float *>(H(o_snr));
	for (int k = 0; k < n; k++) {
		const int i = idx[k];
		if (rv1[k]) { stat[i] = rv1[k]; continue; }
		align[i] += toa1[k];
		if ((uint64_t)align[i] + flen > length[i]) { stat[i] = -1; continue; }
		align[i] += ftoa[k];
		ferr[i] = fe[k];
		base_align[i] = std::max(0, align[i] - flen);
		if ((uint64_t)base_align[i] + wl3 > length[i]) { stat[i] = -1; continue; }
		if (count[k] < 0) { stat[i] = count[k]; continue; }
		// candidates; a carrier with any candidate out of its samples is dropped as a whole (the oracle's early
		// return, see orc_rx.c), before or after the refinement
		bool ok = true;
		for (int q = 0; q < count[k]; q++) {
			const int64_t p = (int64_t)base_align[i] + peaks[(size_t)k * kMaxPeaks + q];
			if (p < 0 || p + flen > (int64_t)length[i]) ok = false;
		}
		if (!ok) { stat[i] = -1; continue; }
		for (int q = 0; q < count[k]; q++) {
			const size_t sl = (size_t)k * kMaxPeaks + q;
			const int64_t p = (int64_t)base_align[i] + peaks[sl] + ctoa[sl];
			if (p < 0 || p + flen > (int64_t)length[i]) stat[i] = -1;
		}
		if (stat[i]) continue;
		// survivor selection, candidate order, first one is the reference (gmr1_rx.c:704-733)
		float ref_snr = 0.f, ref_fe = 0.f;
		for (int q = 0; q < count[k]; q++) {
			const size_t sl = (size_t)k * kMaxPeaks + q;
			if (q == 0) {
				ref_snr = snr[sl];
				ref_fe = cfe[sl];
			} else {
				if (snr[sl] < 2.0f) continue;
				if (snr[sl] < ref_snr / 6.0f) continue;
				if (to_hz(std::fabs(ref_fe - cfe[sl])) > 500.0f) continue;
			}
			RxChain c;
			c.a = i;
			c.chain = nch[i]++;
			c.base = offset[i];
			c.len = (int)length[i];
			c.align = base_align[i] + peaks[sl] + ctoa[sl];
			c.freq_err = ferr[i];
			c.fn = 0; c.delay = 0; c.stn = 0;
			c.bcch_energy = std::nanf("inf");
			c.done = false;
			chains.push_back(std::move(c));
		}
	}
	if (timing)
		fprintf(stderr, "acquire: decisions %.1f us\n",
		        (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tp[4]).count() / 1e3);
	return 0;
}

int RxRun::frame_loop()
{
	// ---- process_bcch (gmr1_rx.c:852-895) for every chain ------------------------------------------
	// Three launches (launch_rx_loop): k_rx_chain walks each chain through all of its frames on the GPU (rounds of
	// CCCH bursts up to the next BCCH burst, whose result feeds back before the next round; rx_loop.h), k_rx4 takes the
	// CCCH bursts it listed, k_rx_merge writes what the reference hands to GSMTAP -- the records, in frame order -- plus,
	// when a traffic pass follows, the per-frame context rx_tch3 sees.  The host only collects them.
	const int nc = (int)chains.size();
	if (!nc)
		return 0;
	const auto t_start = std::chrono::steady_clock::now();
	const int frame_len = sps * 24 * 39;
	int max_frames = 0;
	for (const RxChain &c : chains)
		max_frames = std::max(max_frames, c.len / frame_len + 2);
	// every round but the last covers at least seven frames (seven CCCH bursts, or fewer and the BCCH burst
	// that closes its eight-frame cycle); burst_map only refuses windows at the very ends of the capture
	const int max_rounds = max_frames / 7 + 8;
	const int rec_stride = max_rounds * kLoopPerRound;
	const bool want_ctx = tch != nullptr;
	std::vector<RxLoopState> st0((size_t)nc);
	for (int ci = 0; ci < nc; ci++) {
		const RxChain &c = chains[ci];
		st0[ci] = {c.base, c.len, c.align, c.freq_err, c.fn, c.delay, c.stn, c.done ? 1 : 0, c.bcch_energy,
		           (uint16_t)(arfcn ? arfcn[c.a] : (uint16_t)c.a), (uint16_t)c.chain};
	}
	// one block of device memory and its mirror in pinned host memory:
	// [records | counters (n_rounds, n_rec, n_frames) | states | frame index + gate level per record | frame log]
	const size_t rec_bytes = up128((size_t)nc * rec_stride * sizeof(gmr1_hip_rx_record));
	const size_t cnt_bytes = up128(((size_t)nc * 3 + 1) * 4);      // + the packed total
	const bool pack = !tch && !csd;
	const size_t st_bytes = up128((size_t)nc * sizeof(RxLoopState));
	const size_t rf_bytes = want_ctx ? up128((size_t)nc * rec_stride * 4) : 0;
	const size_t fl_bytes = want_ctx ? up128((size_t)nc * max_frames * sizeof(RxLoopFrame)) : 0;
	const size_t total = rec_bytes + cnt_bytes + st_bytes + 2 * rf_bytes + fl_bytes;
	// ... and, device only, what passes between the loop's three launches (RxLoopArgs): the round logs, the CCCH lists
	// (a frame holds at most one burst of the list; each time slice of the walk starts its part at a multiple of four)
	const int c_stride = ((max_frames + 3) & ~3) + 4 * kLoopSlices;
	const size_t nslot = (size_t)nc * c_stride;
	const size_t rl_bytes = up128((size_t)nc * max_rounds * sizeof(RxLoopRound));
	const size_t s8 = up128(nslot * 8), s4 = up128(nslot * 4), s1 = up128(nslot), s12 = up128(nslot * sizeof(RxLoopCcch)),
	             s24 = up128(nslot * 24);
	const size_t scratch = rl_bytes + up128((size_t)nc * 4) * (2 + kLoopSlices + 1) + s8 + s4 + s1 + s12 + s24 + 4 * s4;
	DevState *ds;
	r = dev_state(&ds);
	if (r) return r;
	void *ws;
	r = dev_workspace(ds, total + scratch + (pack ? rec_bytes : 0) + 128, &ws);
	if (r) return r;
	unsigned char *d = reinterpret_cast<unsigned char *>(((uintptr_t)ws + 127) & ~(uintptr_t)127);
	unsigned char *h;
	r = host_log(total, &h);
	if (r) return r;
	const size_t o_cnt = rec_bytes, o_st = o_cnt + cnt_bytes, o_rf = o_st + st_bytes, o_me = o_rf + rf_bytes,
	             o_fl = o_me + rf_bytes;
	RxLoopArgs la;
	std::memset(&la, 0, sizeof(la));
	la.state = reinterpret_cast<RxLoopState *>(d + o_st);
	la.rec = reinterpret_cast<gmr1_hip_rx_record *>(d);
	la.rec_stride = rec_stride;
	la.max_rounds = max_rounds;
	la.n_rounds = reinterpret_cast<int32_t *>(d + o_cnt);
	la.n_rec = la.n_rounds + nc;
	la.n_frames = la.n_rec + nc;
	if (want_ctx) {
		la.rec_frame = reinterpret_cast<int32_t *>(d + o_rf);
		la.rec_minen = reinterpret_cast<float *>(d + o_me);
		la.flog = reinterpret_cast<RxLoopFrame *>(d + o_fl);
		la.flog_stride = max_frames;
	}
	{
		unsigned char *q = d + total;
		auto take = [&](size_t bytes) { unsigned char *p = q; q += bytes; return p; };
		la.rounds = reinterpret_cast<RxLoopRound *>(take(rl_bytes));
		la.n_ccch = reinterpret_cast<int32_t *>(take(up128((size_t)nc * 4)));
		la.fin = reinterpret_cast<int32_t *>(take(up128((size_t)nc * 4)));
		la.slice_end = reinterpret_cast<int32_t *>(take(up128((size_t)nc * 4) * (kLoopSlices + 1)));
		la.c_stride = c_stride;
		la.c_off = reinterpret_cast<uint64_t *>(take(s8));
		la.c_fs = reinterpret_cast<float *>(take(s4));
		la.c_kind = reinterpret_cast<uint8_t *>(take(s1));
		la.c_meta = reinterpret_cast<RxLoopCcch *>(take(s12));
		la.c_l2 = reinterpret_cast<uint8_t *>(take(s24));
		la.c_crc = reinterpret_cast<int32_t *>(take(s4));
		la.c_conv = reinterpret_cast<int32_t *>(take(s4));
		la.c_rv = reinterpret_cast<int32_t *>(take(s4));
		la.c_en = reinterpret_cast<float *>(take(s4));
		if (pack) {
			la.packed = reinterpret_cast<gmr1_hip_rx_record *>(take(rec_bytes));
			la.n_packed = la.n_frames + nc;
		}
	}
	static const bool timing = profile_env("GMR1_HIP_RX_TIMING") != nullptr;
	const auto t_a = std::chrono::steady_clock::now();
	HIP_TRY(hipMemcpyAsync(la.state, st0.data(), (size_t)nc * sizeof(RxLoopState), hipMemcpyHostToDevice, st));
	const auto t_b = std::chrono::steady_clock::now();
	r = rx_loop_dev_impl(st, nc, sps, iq, la);
	if (r) return r;
	if (timing) {
		auto us = [](auto a, auto b) { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() / 1e3; };
		fprintf(stderr, "frame loop: prepare %.1f us, states copy enqueued %.1f us, launches %.1f us\n", us(t_start, t_a), us(t_a, t_b),
		        us(t_b, std::chrono::steady_clock::now()));
	}
	if (pack) {
		// counters and states first (a few KB), then exactly the records there are
		HIP_TRY(hipMemcpyAsync(h + o_cnt, d + o_cnt, cnt_bytes + st_bytes, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		t_chain_us = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_start).count() / 1e3;
		const int n_total = reinterpret_cast<const int32_t *>(h + o_cnt)[3 * nc];
		if (n_total < 0 || (size_t)n_total > (size_t)nc * rec_stride)
			return fail(-EIO, "rx loop: packed record count %d out of range", n_total);
		const int fit = std::max(0, std::min(n_total, max_records));
		if (fit) {
			// device memory and pinned / registered host memory take the copy directly; pageable memory goes through the pinned block
			hipPointerAttribute_t at;
			const bool known = hipPointerGetAttributes(&at, out) == hipSuccess &&
			                   (at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeHost || at.type == hipMemoryTypeManaged);
			if (!known)
				(void)hipGetLastError();
			const size_t nb = (size_t)fit * sizeof(gmr1_hip_rx_record);
			if (known) {
				HIP_TRY(hipMemcpyAsync(out, la.packed, nb, hipMemcpyDefault, st));
				HIP_TRY(hipStreamSynchronize(st));
			} else {
				HIP_TRY(hipMemcpyAsync(h, la.packed, nb, hipMemcpyDeviceToHost, st));
				HIP_TRY(hipStreamSynchronize(st));
				std::memcpy(out, h, nb);
			}
		}
		direct = true;
		direct_total = n_total;
	} else {
		HIP_TRY(hipMemcpyAsync(h, d, total, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
	}
	t_loop_gpu_us = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_start).count() / 1e3;

	const int32_t *h_nr = reinterpret_cast<const int32_t *>(h + o_cnt), *h_nrec = h_nr + nc, *h_nfr = h_nrec + nc;
	const RxLoopState *h_st = reinterpret_cast<const RxLoopState *>(h + o_st);
	for (int ci = 0; ci < nc; ci++) {
		RxChain &c = chains[ci];
		if (h_nr[ci] >= max_rounds || h_nrec[ci] > rec_stride || (want_ctx && h_nfr[ci] > max_frames)) {
			// this chain outgrew its buffers: its carrier's status says so, its records are dropped, the others go on
			fail(-EIO, "rx loop: chain %d of carrier %d outgrew its buffers (%d rounds, %d records, %d frames)", c.chain,
			     c.a, h_nr[ci], h_nrec[ci], h_nfr[ci]);
			stat[c.a] = -EIO;
			c.done = true;
			continue;
		}
		const gmr1_hip_rx_record *rp = reinterpret_cast<const gmr1_hip_rx_record *>(h) + (size_t)ci * rec_stride;
		if (!pack)
			c.rec.assign(rp, rp + h_nrec[ci]);
		c.n_rec = h_nrec[ci];
		if (want_ctx) {
			const int32_t *fp = reinterpret_cast<const int32_t *>(h + o_rf) + (size_t)ci * rec_stride;
			const float *mp = reinterpret_cast<const float *>(h + o_me) + (size_t)ci * rec_stride;
			const RxLoopFrame *lp = reinterpret_cast<const RxLoopFrame *>(h + o_fl) + (size_t)ci * max_frames;
			c.rec_frame.assign(fp, fp + h_nrec[ci]);
			c.log.resize((size_t)h_nfr[ci]);
			for (int f = 0; f < h_nfr[ci]; f++)
				c.log[f] = {lp[f].align, lp[f].freq_err, lp[f].fn};
			// IMM.ASS on the CCCH starts the TCH3 follow-up in that very frame (gmr1_rx.c:235-246, 836-841)
			for (int k = 0; k < h_nrec[ci]; k++) {
				const uint8_t *l2 = rp[k].l2;
				if (rp[k].type == 2 && l2[1] == 0x06 && l2[2] == 0x3f)
					c.events.push_back({fp[k], ((l2[8] & 0x03) << 3) | (l2[9] >> 5), (l2[8] & 0xfc) >> 2, mp[k]});
			}
		} else if (!pack) {
			c.rec_frame.assign((size_t)h_nrec[ci], 0);
		}
		const RxLoopState &s = h_st[ci];
		c.align = s.align; c.freq_err = s.freq_err; c.fn = s.fn; c.delay = s.delay; c.stn = s.stn;
		c.bcch_energy = s.bcch_energy;
		c.done = s.done != 0;
	}
	return 0;
}

int RxRun::tch3_pass()
{
	// ---- TCH3 follow-up (rx_tch3, gmr1_rx.c:355-600) ----------------------------------------------
	// Nothing the traffic channel does feeds back into the BCCH / CCCH loop, so it runs afterwards,
	// for all chains at once, in four steps:
	//   A. for every frame from a chain's first assignment on, speculatively: burst energy, FACCH3
	//      and speech demodulation, burst type detection, DKAB search         (4 batched launches)
	//   B. host: the per-frame state machine (energy thresholds, DKAB / weak count, FACCH3 burst
	//      grouping by sync sequence) over those results, producing the list of decodes it calls for
	//   C. A5/1 keystreams and the TCH3 / FACCH3 decodes, each with and without deciphering
	//   D. host: the ciphering state (a FACCH3 that only decodes ciphered switches it on) picks
	//      the variant, records are emitted in frame order.
	if (tch) {
		std::vector<TchItem> titems;
		const int twin = sps + (sps / 2);            // gmr1_rx.c:551
		const int t_in_len = 117 * sps + twin;
		const int t_etoa = twin >> 1;
		for (size_t ci = 0; ci < chains.size(); ci++) {
			RxChain &c = chains[ci];
			if (c.events.empty())
				continue;
			size_t ev = 0;
			for (int f = c.events[0].frame; f < (int)c.log.size(); f++) {
				while (ev + 1 < c.events.size() && c.events[ev + 1].frame <= f)
					ev++;
				const int tn = c.events[ev].tn;
				const int64_t begin = (int64_t)c.log[f].align + sps * tn * 39 - t_etoa;
				if (begin < 0 || begin + t_in_len > c.len)
					continue;                         // burst_map fails: rx_tch3 returns before touching anything
				titems.push_back({(int)ci, f, tn, c.events[ev].p, t_etoa});
			}
		}
		const int nt = (int)titems.size();
		if (nt) {
			// step A needs 376 B per frame, the decodes of step C at most 866 B per frame
			Arena arena;
			if ((r = arena.init((size_t)nt * 1300 + 64 * 1024))) return r;
			g_arena = &arena;
			struct Reset { ~Reset() { g_arena = nullptr; } } reset;
			std::vector<uint64_t> t_off(nt);
			std::vector<float> t_fs(nt), t_et(nt);
			std::vector<int32_t> t_p(nt);
			for (int k = 0; k < nt; k++) {
				const RxChain &c = chains[titems[k].chain_idx];
				const FrameCtx &x = c.log[titems[k].frame];
				t_off[k] = c.base + (uint64_t)((int64_t)x.align + sps * titems[k].tn * 39 - t_etoa);
				t_fs[k] = -x.freq_err;
				t_et[k] = (float)t_etoa;
				t_p[k] = titems[k].p;
			}
			ABuf d_off2, d_fs2, d_et, d_pp, d_feb, d_fsid, d_frv, d_en, d_seb, d_srv, d_bt, d_dsid, d_dtoa, d_drv, d_krv, d_ftoa;
			HIP_TRY(d_off2.alloc((size_t)nt * 8)); HIP_TRY(d_fs2.alloc((size_t)nt * 4)); HIP_TRY(d_et.alloc((size_t)nt * 4));
			HIP_TRY(d_pp.alloc((size_t)nt * 4)); HIP_TRY(d_feb.alloc((size_t)nt * 104)); HIP_TRY(d_fsid.alloc((size_t)nt * 4));
			HIP_TRY(d_frv.alloc((size_t)nt * 4)); HIP_TRY(d_en.alloc((size_t)nt * 4)); HIP_TRY(d_seb.alloc((size_t)nt * 212));
			HIP_TRY(d_srv.alloc((size_t)nt * 4)); HIP_TRY(d_bt.alloc((size_t)nt * 4)); HIP_TRY(d_dsid.alloc((size_t)nt * 4));
			HIP_TRY(d_dtoa.alloc((size_t)nt * 4)); HIP_TRY(d_drv.alloc((size_t)nt * 4)); HIP_TRY(d_krv.alloc((size_t)nt * 4));
			HIP_TRY(d_ftoa.alloc((size_t)nt * 4));
			HIP_TRY(hipMemcpyAsync(d_off2.p, t_off.data(), (size_t)nt * 8, hipMemcpyHostToDevice, st));
			HIP_TRY(hipMemcpyAsync(d_fs2.p, t_fs.data(), (size_t)nt * 4, hipMemcpyHostToDevice, st));
			HIP_TRY(hipMemcpyAsync(d_et.p, t_et.data(), (size_t)nt * 4, hipMemcpyHostToDevice, st));
			HIP_TRY(hipMemcpyAsync(d_pp.p, t_p.data(), (size_t)nt * 4, hipMemcpyHostToDevice, st));

			// A. speculative per-frame work
			r = demod_dev_energy(st, GMR1_HIP_NT3_FACCH, nt, sps, t_in_len, tch, d_off2.as<uint64_t>(), d_fs2.as<float>(),
			                     d_feb.as<int8_t>(), 104, d_fsid.as<int32_t>(), d_ftoa.as<float>(), d_en.as<float>(),
			                     d_frv.as<int32_t>());
			if (r) return r;
			r = demod_dev_energy(st, GMR1_HIP_NT3_SPEECH, nt, sps, t_in_len, tch, d_off2.as<uint64_t>(), d_fs2.as<float>(),
			                     d_seb.as<int8_t>(), 212, nullptr, nullptr, nullptr, d_srv.as<int32_t>());
			if (r) return r;
			{
				const int ids[2] = {GMR1_HIP_NT3_FACCH, GMR1_HIP_NT3_SPEECH};     // gmr1_rx.c:534-538
				r = gmr1_hip_detect_batch_dev(st, 2, ids, nt, sps, t_in_len, tch, d_off2.as<uint64_t>(), d_fs2.as<float>(),
				                              d_et.as<float>(), d_bt.as<int32_t>(), d_dsid.as<int32_t>(), d_dtoa.as<float>(),
				                              d_drv.as<int32_t>());
				if (r) return r;
			}
			r = gmr1_hip_dkab_demod_batch_dev(st, nt, sps, t_in_len, tch, d_off2.as<uint64_t>(), d_fs2.as<float>(),
			                                  d_pp.as<int32_t>(), nullptr, nullptr, d_krv.as<int32_t>());
			if (r) return r;
			std::vector<int8_t> h_feb((size_t)nt * 104), h_seb((size_t)nt * 212);
			std::vector<int32_t> h_fsid(nt), h_frv(nt), h_srv(nt), h_bt(nt), h_drv(nt), h_krv(nt);
			std::vector<float> h_en(nt);
			HIP_TRY(hipMemcpyAsync(h_feb.data(), d_feb.p, (size_t)nt * 104, hipMemcpyDeviceToHost, st));
			HIP_TRY(hipMemcpyAsync(h_seb.data(), d_seb.p, (size_t)nt * 212, hipMemcpyDeviceToHost, st));
			HIP_TRY(hipMemcpyAsync(h_fsid.data(), d_fsid.p, (size_t)nt * 4, hipMemcpyDeviceToHost, st));
			HIP_TRY(hipMemcpyAsync(h_frv.data(), d_frv.p, (size_t)nt * 4, hipMemcpyDeviceToHost, st));
			HIP_TRY(hipMemcpyAsync(h_srv.data(), d_srv.p, (size_t)nt * 4, hipMemcpyDeviceToHost, st));
			HIP_TRY(hipMemcpyAsync(h_bt.data(), d_bt.p, (size_t)nt * 4, hipMemcpyDeviceToHost, st));
			HIP_TRY(hipMemcpyAsync(h_drv.data(), d_drv.p, (size_t)nt * 4, hipMemcpyDeviceToHost, st));
			HIP_TRY(hipMemcpyAsync(h_krv.data(), d_krv.p, (size_t)nt * 4, hipMemcpyDeviceToHost, st));
			HIP_TRY(hipMemcpyAsync(h_en.data(), d_en.p, (size_t)nt * 4, hipMemcpyDeviceToHost, st));
			HIP_TRY(hipStreamSynchronize(st));


			// B. the state machine of rx_tch3 (gmr1_rx.c:531-600) and its helpers, chain by chain
			std::vector<TchJob> jobs;
			{
				int k = 0;
				while (k < nt) {
					const int ci = titems[k].chain_idx;
					RxChain &c = chains[ci];
					Tch3State ts;
					size_t ev = 0;
					auto flush = [&](int frame, int fn) {       // _rx_tch3_facch_flush, gmr1_rx.c:397-450 (decode deferred)
						TchJob j;
						j.chain_idx = ci; j.frame = frame; j.is_flush = 1; j.fn = fn; j.tn = ts.tn; j.item = -1;
						std::memcpy(j.ebits, ts.ebits, sizeof(j.ebits));
						std::memcpy(j.bi_fn, ts.bi_fn, sizeof(j.bi_fn));
						jobs.push_back(j);
						ts.sync_id ^= 1;
						ts.burst_cnt = 0;
						std::memset(ts.bi_fn, 0xff, sizeof(ts.bi_fn));
						std::memset(ts.ebits, 0, sizeof(ts.ebits));
					};
					for (; k < nt && titems[k].chain_idx == ci; k++) {
						const TchItem &ti = titems[k];
						// assignments taken in this frame or in skipped ones (rx_tch3_init, gmr1_rx.c:358-378)
						while (ev < c.events.size() && c.events[ev].frame <= ti.frame) {
							const AssEvt &e = c.events[ev++];
							ts.active = 1;
							ts.tn = e.tn; ts.p = e.p;
							ts.energy_burst = e.ref_energy * 0.75f;
							ts.energy_dkab = ts.energy_burst / 8.0f;
							ts.weak_cnt = 0;
							ts.sync_id = 0;
							std::memset(ts.ebits, 0, sizeof(ts.ebits));
						}
						if (!ts.active)
							continue;
						const int fn = c.log[ti.frame].fn;
						const float be = h_en[k];
						const float det = (ts.energy_dkab + ts.energy_burst) / 4.0f;
						if (be < det) {
							const int drv = h_krv[k];
							if (drv < 0)
								continue;
							if (drv == 1) {
								if (ts.weak_cnt++ > 8)
									ts.active = 0;
							} else
								ts.energy_dkab = (0.1f * be) + (0.9f * ts.energy_dkab);
							continue;
						}
						ts.weak_cnt = 0;
						ts.energy_burst = (0.1f * be) + (0.9f * ts.energy_burst);
						if (h_drv[k] < 0)
							continue;
						if (h_bt[k] == 0) {
							// _rx_tch3_facch, gmr1_rx.c:452-493
							if (h_frv[k] < 0)
								continue;
							const int bi = fn & 3;
							if (h_fsid[k] != ts.sync_id)
								flush(ti.frame, fn);
							std::memcpy(&ts.ebits[104 * bi], &h_feb[(size_t)k * 104], 104);
							ts.sync_id = h_fsid[k];
							ts.bi_fn[bi] = (uint32_t)fn;
							ts.burst_cnt += 1;
							if (ts.burst_cnt == 4)
								flush(ti.frame, fn);
						} else {
							// _rx_tch3_speech, gmr1_rx.c:495-529
							if (h_srv[k] < 0)
								continue;
							TchJob j;
							j.chain_idx = ci; j.frame = ti.frame; j.is_flush = 0; j.fn = fn; j.tn = ts.tn; j.item = k;
							jobs.push_back(j);
						}
					}
				}
			}

			// C. keystreams and decodes, plain and deciphered
			std::vector<int> sj, fj;
			for (size_t j = 0; j < jobs.size(); j++)
				(jobs[j].is_flush ? fj : sj).push_back((int)j);
			const int ns = (int)sj.size(), nf = (int)fj.size();
			std::vector<uint8_t> s_fr[2], f_l2[2];
			std::vector<int32_t> s_conv[2], f_crc[2], f_conv[2];
			if (ns) {
				std::vector<int8_t> eb((size_t)ns * 212);
				std::vector<uint8_t> keys((size_t)ns * 8, 0);
				std::vector<uint32_t> fns(ns);
				for (int i = 0; i < ns; i++) {
					const TchJob &j = jobs[sj[i]];
					std::memcpy(&eb[(size_t)i * 212], &h_seb[(size_t)j.item * 212], 212);
					if (kc) std::memcpy(&keys[(size_t)i * 8], kc + (size_t)chains[j.chain_idx].a * 8, 8);
					fns[i] = (uint32_t)j.fn;
				}
				ABuf d_eb, d_k, d_fn, d_ks, d_fr, d_cv;
				HIP_TRY(d_eb.alloc(eb.size())); HIP_TRY(d_k.alloc(keys.size())); HIP_TRY(d_fn.alloc((size_t)ns * 4));
				HIP_TRY(d_ks.alloc((size_t)ns * 208)); HIP_TRY(d_fr.alloc((size_t)ns * 20)); HIP_TRY(d_cv.alloc((size_t)ns * 8));
				HIP_TRY(hipMemcpyAsync(d_eb.p, eb.data(), eb.size(), hipMemcpyHostToDevice, st));
				HIP_TRY(hipMemcpyAsync(d_k.p, keys.data(), keys.size(), hipMemcpyHostToDevice, st));
				HIP_TRY(hipMemcpyAsync(d_fn.p, fns.data(), (size_t)ns * 4, hipMemcpyHostToDevice, st));
				r = gmr1_hip_a5_batch_dev(st, ns, 1, 208, d_k.as<uint8_t>(), d_fn.as<uint32_t>(), d_ks.as<uint8_t>(), nullptr);
				if (r) return r;
				for (int v = 0; v < 2; v++) {
					r = gmr1_hip_tch3_decode_batch_dev(st, ns, 0, d_eb.as<int8_t>(), v ? d_ks.as<uint8_t>() : nullptr,
					                                   d_fr.as<uint8_t>(), nullptr, d_cv.as<int32_t>());
					if (r) return r;
					s_fr[v].resize((size_t)ns * 20);
					s_conv[v].resize((size_t)ns * 2);
					HIP_TRY(hipMemcpyAsync(s_fr[v].data(), d_fr.p, (size_t)ns * 20, hipMemcpyDeviceToHost, st));
					HIP_TRY(hipMemcpyAsync(s_conv[v].data(), d_cv.p, (size_t)ns * 8, hipMemcpyDeviceToHost, st));
					HIP_TRY(hipStreamSynchronize(st));
				}
			}
			if (nf) {
				std::vector<int8_t> eb((size_t)nf * 416);
				std::vector<uint8_t> keys((size_t)nf * 4 * 8, 0);
				std::vector<uint32_t> fns((size_t)nf * 4);
				for (int i = 0; i < nf; i++) {
					const TchJob &j = jobs[fj[i]];
					std::memcpy(&eb[(size_t)i * 416], j.ebits, 416);
					for (int b = 0; b < 4; b++) {
						if (kc) std::memcpy(&keys[((size_t)i * 4 + b) * 8], kc + (size_t)chains[j.chain_idx].a * 8, 8);
						fns[(size_t)i * 4 + b] = j.bi_fn[b];
					}
				}
				ABuf d_eb, d_k, d_fn, d_ks, d_l2, d_crc, d_cv;
				HIP_TRY(d_eb.alloc(eb.size())); HIP_TRY(d_k.alloc(keys.size())); HIP_TRY(d_fn.alloc((size_t)nf * 16));
				HIP_TRY(d_ks.alloc((size_t)nf * 384)); HIP_TRY(d_l2.alloc((size_t)nf * 10)); HIP_TRY(d_crc.alloc((size_t)nf * 4));
				HIP_TRY(d_cv.alloc((size_t)nf * 4));
				HIP_TRY(hipMemcpyAsync(d_eb.p, eb.data(), eb.size(), hipMemcpyHostToDevice, st));
				HIP_TRY(hipMemcpyAsync(d_k.p, keys.data(), keys.size(), hipMemcpyHostToDevice, st));
				HIP_TRY(hipMemcpyAsync(d_fn.p, fns.data(), (size_t)nf * 16, hipMemcpyHostToDevice, st));
				// 4 x 96 keystream bits per message, one per burst's frame number (gmr1_rx.c:409-412)
				r = gmr1_hip_a5_batch_dev(st, nf * 4, 1, 96, d_k.as<uint8_t>(), d_fn.as<uint32_t>(), d_ks.as<uint8_t>(), nullptr);
				if (r) return r;
				for (int v = 0; v < 2; v++) {
					r = gmr1_hip_facch3_decode_batch_dev(st, nf, d_eb.as<int8_t>(), v ? d_ks.as<uint8_t>() : nullptr,
					                                     d_l2.as<uint8_t>(), nullptr, d_crc.as<int32_t>(), d_cv.as<int32_t>());
					if (r) return r;
					f_l2[v].resize((size_t)nf * 10);
					f_crc[v].resize(nf);
					f_conv[v].resize(nf);
					HIP_TRY(hipMemcpyAsync(f_l2[v].data(), d_l2.p, (size_t)nf * 10, hipMemcpyDeviceToHost, st));
					HIP_TRY(hipMemcpyAsync(f_crc[v].data(), d_crc.p, (size_t)nf * 4, hipMemcpyDeviceToHost, st));
					HIP_TRY(hipMemcpyAsync(f_conv[v].data(), d_cv.p, (size_t)nf * 4, hipMemcpyDeviceToHost, st));
					HIP_TRY(hipStreamSynchronize(st));
				}
			}


			// D. ciphering state and records, in the order things happened
			{
				std::vector<int> ciph(chains.size(), 0);
				int is = 0, iff = 0;
				for (size_t jj = 0; jj < jobs.size(); jj++) {
					const TchJob &j = jobs[jj];
					RxChain &c = chains[j.chain_idx];
					const uint16_t an = arfcn ? arfcn[c.a] : (uint16_t)c.a;
					int &cf = ciph[j.chain_idx];
					if (!j.is_flush) {
						const int v = cf ? 1 : 0;
						const int32_t *cv = &s_conv[v][(size_t)is * 2];
						emit(c, an, 0x10 /* GSMTAP_GMR1_TCH3 */, j.fn, j.tn, &s_fr[v][(size_t)is * 20],
						     (cv[0] & 0xffff) | (cv[1] << 16), j.frame, 20);
						is++;
					} else {
						int v = cf ? 1 : 0;
						int crc = f_crc[v][iff];
						if (!cf && crc) {                 // retry with ciphering (gmr1_rx.c:420-432)
							v = 1;
							crc = f_crc[1][iff];
							if (!crc)
								cf = 1;
						}
						if (!crc) {
							const uint8_t *m = &f_l2[v][(size_t)iff * 10];
							emit(c, an, 0x12 /* GSMTAP_GMR1_TCH3 | GSMTAP_GMR1_FACCH */, j.fn - 3, j.tn, m,
							     f_conv[v][iff], j.frame, 10);
							// ASSIGNMENT COMMAND 1 starts the TCH9 follow-up (gmr1_rx.c:248-258, 436-442)
							if (csd && m[3] == 0x06 && m[4] == 0x2e)
								c.events9.push_back({j.frame, ((m[5] & 0x03) << 3) | (m[6] >> 5), 0, 0.f});
						}
						iff++;
					}
				}
			}

			// frame order within each chain: BCCH / CCCH of a frame come before its TCH records
			for (RxChain &c : chains) {
				if (c.events.empty())
					continue;
				std::vector<size_t> order(c.rec.size());
				for (size_t i = 0; i < order.size(); i++) order[i] = i;
				std::stable_sort(order.begin(), order.end(),
				                 [&](size_t x, size_t y) { return c.rec_frame[x] < c.rec_frame[y]; });
				std::vector<gmr1_hip_rx_record> sorted(c.rec.size());
				for (size_t i = 0; i < order.size(); i++) sorted[i] = c.rec[order[i]];
				c.rec.swap(sorted);
			}
		}
	}

	return 0;
}

int RxRun::tch9_pass()
{
	// ---- TCH9 follow-up (rx_tch9, gmr1_rx.c:262-353) ----------------------------------------------
	// From the frame of a chain's first ASSIGNMENT COMMAND 1 on, every frame's NT9 burst on the assigned
	// timeslot of the CSD carrier: demodulate (sync sequence 0 = FACCH9, 1 = TCH9), decipher with A5/1 of the
	// frame number, decode.  Nothing feeds back, so it is one more batched pass: one demodulation launch, one
	// keystream launch, one FACCH9 launch, one TCH9 launch per interleaver run (a run starts at every
	// assignment; gmr1_deinterleave_inter only advances on TCH9 bursts).
	if (csd) {
		struct Nt9Item { int chain_idx, frame, tn; };
		std::vector<Nt9Item> items9;
		const int win9 = sps + (sps / 2), in_len9 = 351 * sps + win9, etoa9 = win9 >> 1;
		for (size_t ci = 0; ci < chains.size(); ci++) {
			RxChain &c = chains[ci];
			if (c.events9.empty())
				continue;
			size_t ev = 0;
			for (int f = c.events9[0].frame; f < (int)c.log.size(); f++) {
				while (ev + 1 < c.events9.size() && c.events9[ev + 1].frame <= f)
					ev++;
				const int tn = c.events9[ev].tn;
				const int64_t begin = (int64_t)c.log[f].align + sps * tn * 39 - etoa9;
				if (begin < 0 || begin + in_len9 > c.len)
					continue;
				items9.push_back({(int)ci, f, tn});
			}
		}
		const int n9 = (int)items9.size();
		if (n9) {
			// demodulation 682 B per frame, keystreams and decodes at most 662 + 8 + 4 + 658 + 64 B per frame
			Arena arena;
			if ((r = arena.init((size_t)n9 * 2300 + 64 * 1024))) return r;
			g_arena = &arena;
			struct Reset { ~Reset() { g_arena = nullptr; } } reset;
			std::vector<uint64_t> off9(n9);
			std::vector<float> fs9(n9);
			for (int k = 0; k < n9; k++) {
				const RxChain &c = chains[items9[k].chain_idx];
				const FrameCtx &x = c.log[items9[k].frame];
				off9[k] = c.base + (uint64_t)((int64_t)x.align + sps * items9[k].tn * 39 - etoa9);
				fs9[k] = -x.freq_err;
			}
			ABuf d_o, d_f, d_eb, d_sid, d_rv;
			HIP_TRY(d_o.alloc((size_t)n9 * 8)); HIP_TRY(d_f.alloc((size_t)n9 * 4)); HIP_TRY(d_eb.alloc((size_t)n9 * 662));
			HIP_TRY(d_sid.alloc((size_t)n9 * 4)); HIP_TRY(d_rv.alloc((size_t)n9 * 4));
			HIP_TRY(hipMemcpyAsync(d_o.p, off9.data(), (size_t)n9 * 8, hipMemcpyHostToDevice, st));
			HIP_TRY(hipMemcpyAsync(d_f.p, fs9.data(), (size_t)n9 * 4, hipMemcpyHostToDevice, st));
			r = demod_dev_energy(st, GMR1_HIP_NT9, n9, sps, in_len9, csd, d_o.as<uint64_t>(), d_f.as<float>(),
			                     d_eb.as<int8_t>(), 662, d_sid.as<int32_t>(), nullptr, nullptr, d_rv.as<int32_t>());
			if (r) return r;
			std::vector<int8_t> h_eb((size_t)n9 * 662);
			std::vector<int32_t> h_sid(n9), h_rv(n9);
			HIP_TRY(hipMemcpyAsync(h_eb.data(), d_eb.p, (size_t)n9 * 662, hipMemcpyDeviceToHost, st));
			HIP_TRY(hipMemcpyAsync(h_sid.data(), d_sid.p, (size_t)n9 * 4, hipMemcpyDeviceToHost, st));
			HIP_TRY(hipMemcpyAsync(h_rv.data(), d_rv.p, (size_t)n9 * 4, hipMemcpyDeviceToHost, st));
			HIP_TRY(hipStreamSynchronize(st));
			// classify; TCH9 bursts are laid out run after run (one run per interleaver life)
			std::vector<int> fj, tj;                 // item indices: FACCH9 jobs, TCH9 jobs (run-major)
			std::vector<int> run_len;
			{
				int k = 0;
				while (k < n9) {
					const int ci = items9[k].chain_idx;
					const RxChain &c = chains[ci];
					size_t ev = 0;
					int cur = 0;
					bool open = false;
					for (; k < n9 && items9[k].chain_idx == ci; k++) {
						// a (re-)assignment at or before this frame restarts the interleaver (rx_tch9_init)
						bool restart = !open;
						while (ev < c.events9.size() && c.events9[ev].frame <= items9[k].frame) { ev++; restart = true; }
						if (restart) {
							if (open && cur) run_len.push_back(cur);
							cur = 0;
							open = true;
						}
						if (h_rv[k])
							continue;                    // decision D8: a failed demodulation is no burst
						if (h_sid[k] == 0)
							fj.push_back(k);
						else {
							tj.push_back(k);
							cur++;
						}
					}
					if (cur) run_len.push_back(cur);
				}
			}
			const int nf = (int)fj.size(), nt9 = (int)tj.size(), nj = nf + nt9;
			if (nj) {
				std::vector<int8_t> eb((size_t)nj * 662);
				std::vector<uint8_t> keys((size_t)nj * 8, 0);
				std::vector<uint32_t> fns(nj);
				for (int i = 0; i < nj; i++) {
					const int k = i < nf ? fj[i] : tj[i - nf];
					const RxChain &c = chains[items9[k].chain_idx];
					std::memcpy(&eb[(size_t)i * 662], &h_eb[(size_t)k * 662], 662);
					if (kc) std::memcpy(&keys[(size_t)i * 8], kc + (size_t)c.a * 8, 8);
					fns[i] = (uint32_t)c.log[items9[k].frame].fn;
				}
				ABuf d_e2, d_k, d_fn, d_ks, d_l2f, d_crc, d_cvf, d_l2t, d_cvt;
				HIP_TRY(d_e2.alloc(eb.size())); HIP_TRY(d_k.alloc(keys.size())); HIP_TRY(d_fn.alloc((size_t)nj * 4));
				HIP_TRY(d_ks.alloc((size_t)nj * 658));
				HIP_TRY(d_l2f.alloc((size_t)nf * 38)); HIP_TRY(d_crc.alloc((size_t)nf * 4)); HIP_TRY(d_cvf.alloc((size_t)nf * 4));
				HIP_TRY(d_l2t.alloc((size_t)nt9 * 60)); HIP_TRY(d_cvt.alloc((size_t)nt9 * 4));
				HIP_TRY(hipMemcpyAsync(d_e2.p, eb.data(), eb.size(), hipMemcpyHostToDevice, st));
				HIP_TRY(hipMemcpyAsync(d_k.p, keys.data(), keys.size(), hipMemcpyHostToDevice, st));
				HIP_TRY(hipMemcpyAsync(d_fn.p, fns.data(), (size_t)nj * 4, hipMemcpyHostToDevice, st));
				r = gmr1_hip_a5_batch_dev(st, nj, 1, 658, d_k.as<uint8_t>(), d_fn.as<uint32_t>(), d_ks.as<uint8_t>(), nullptr);
				if (r) return r;
				std::vector<uint8_t> l2f((size_t)nf * 38), l2t((size_t)nt9 * 60);
				std::vector<int32_t> crcf(nf), cvf(nf), cvt(nt9);
				if (nf) {
					r = gmr1_hip_facch9_decode_batch_dev(st, nf, d_e2.as<int8_t>(), d_ks.as<uint8_t>(), d_l2f.as<uint8_t>(),
					                                     nullptr, nullptr, d_crc.as<int32_t>(), d_cvf.as<int32_t>());
					if (r) return r;
					HIP_TRY(hipMemcpyAsync(l2f.data(), d_l2f.p, l2f.size(), hipMemcpyDeviceToHost, st));
					HIP_TRY(hipMemcpyAsync(crcf.data(), d_crc.p, (size_t)nf * 4, hipMemcpyDeviceToHost, st));
					HIP_TRY(hipMemcpyAsync(cvf.data(), d_cvf.p, (size_t)nf * 4, hipMemcpyDeviceToHost, st));
				}
				std::vector<int32_t> pos((size_t)nt9);       // lives until the synchronisation below
				{
					// all runs in one launch: every burst knows its position in its own run
					if (nt9) {
						size_t i = 0;
						for (int len_run : run_len)
							for (int q = 0; q < len_run; q++)
								pos[i++] = q;
						ABuf d_pos;
						HIP_TRY(d_pos.alloc((size_t)nt9 * 4));
						HIP_TRY(hipMemcpyAsync(d_pos.p, pos.data(), (size_t)nt9 * 4, hipMemcpyHostToDevice, st));
						r = tch9_runs_dev_impl(st, 2 /* GMR1_TCH9_9k6, gmr1_rx.c:333 */, nt9, d_pos.as<int32_t>(),
						                       d_e2.as<int8_t>() + (size_t)nf * 662, d_ks.as<uint8_t>() + (size_t)nf * 658,
						                       d_l2t.as<uint8_t>(), d_cvt.as<int32_t>());
						if (r) return r;
					}
					if (nt9) {
						HIP_TRY(hipMemcpyAsync(l2t.data(), d_l2t.p, l2t.size(), hipMemcpyDeviceToHost, st));
						HIP_TRY(hipMemcpyAsync(cvt.data(), d_cvt.p, (size_t)nt9 * 4, hipMemcpyDeviceToHost, st));
					}
				}
				HIP_TRY(hipStreamSynchronize(st));
				// records in frame order per chain: merge the two job lists by item index
				int a9 = 0, b9 = 0;
				while (a9 < nf || b9 < nt9) {
					const bool take_f = b9 >= nt9 || (a9 < nf && fj[a9] < tj[b9]);
					const int k = take_f ? fj[a9] : tj[b9];
					RxChain &c = chains[items9[k].chain_idx];
					gmr1_hip_rx_big_record rec;
					std::memset(&rec, 0, sizeof(rec));
					rec.arfcn = arfcn ? arfcn[c.a] : (uint16_t)c.a;
					rec.chain = (uint8_t)c.chain;
					rec.fn = (uint32_t)c.log[items9[k].frame].fn;
					rec.tn = (uint8_t)items9[k].tn;
					if (take_f) {
						if (!crcf[a9]) {
							rec.type = 0x1a;     // GSMTAP_GMR1_TCH9 | GSMTAP_GMR1_FACCH
							rec.len = 38;
							rec.conv = cvf[a9];
							std::memcpy(rec.l2, &l2f[(size_t)a9 * 38], 38);
							c.big.push_back(rec);
						}
						a9++;
					} else {
						rec.type = 0x18;         // GSMTAP_GMR1_TCH9 (no CRC to check, gmr1_rx.c:336-339)
						rec.len = 60;
						rec.conv = cvt[b9];
						std::memcpy(rec.l2, &l2t[(size_t)b9 * 60], 60);
						c.big.push_back(rec);
						b9++;
					}
				}
			}
		}
	}

	return 0;
}

// gmr1_hip_rx_run_full_dev, plus (optional) how many of the records each carrier contributed
int rx_run_full_impl(void *stream_, int n_arfcn, int sps, const float *iq, const float *tch,
                     const float *csd, const uint64_t *offset, const uint64_t *length,
                     const uint16_t *arfcn, const uint8_t *kc,
                     struct gmr1_hip_rx_record *out, int max_records, int *n_records,
                     struct gmr1_hip_rx_big_record *big_out, int max_big, int *n_big,
                     int32_t *status, int32_t *n_chains, int32_t *rec_per_carrier)
{
	hipStream_t st = (hipStream_t)stream_;
	if (n_records) *n_records = 0;
	if (n_big) *n_big = 0;
	if (csd && (!tch || !n_big || max_big < 0 || (max_big > 0 && !big_out)))
		return fail(-EINVAL, "rx_run: the CSD carrier needs the traffic carrier and the big-record outputs");
	if (n_arfcn < 0 || !iq || !offset || !length || !n_records || (max_records > 0 && !out) || max_records < 0)
		return fail(-EINVAL, "rx_run: iq/offset/length/n_records (and out when max_records > 0) are required");
	if (sps < 1 || sps > 16)                  // gmr1_rx.c:919-922
		return fail(-EINVAL, "rx_run: sps=%d unsupported (1..16)", sps);
	DevState *ds;
	int r = dev_state(&ds);
	if (r) return r;
	if (n_arfcn == 0) return 0;
	// the whole call holds the device's workspace, the loop's side stream and its events: calls from other threads wait
	WsLease lease;
	if ((r = lease.acquire(ds, st))) return r;
	for (int i = 0; i < n_arfcn; i++)
		if (length[i] > 0x7fffffffull)
			return fail(-EINVAL, "rx_run: carrier %d longer than 2^31-1 samples", i);

	RxRun run;
	run.st = st; run.sps = sps; run.iq = iq; run.tch = tch; run.csd = csd;
	run.offset = offset; run.length = length; run.arfcn = arfcn; run.kc = kc;
	run.A = n_arfcn;
	run.out = out; run.max_records = max_records;
	run.flen = kFcchLen * sps;
	run.stat.assign(n_arfcn, 0); run.nch.assign(n_arfcn, 0);
	run.align.assign(n_arfcn, kStartDiscard); run.base_align.assign(n_arfcn, 0);
	run.ferr.assign(n_arfcn, 0.0f);
	// GMR1_HIP_RX_TIMING=1: wall time of the phases on stderr (profiling only)
	static const bool timing = profile_env("GMR1_HIP_RX_TIMING") != nullptr;
	auto now = [] { return std::chrono::steady_clock::now(); };
	const auto t0 = now();
	if ((r = run.acquire())) return r;
	const auto t1 = now();
	if ((r = run.frame_loop())) return r;
	const auto t2 = now();
	if (tch && (r = run.tch3_pass())) return r;
	if (csd && (r = run.tch9_pass())) return r;
	{
		auto us = [](auto a, auto b) { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() / 1e3; };
		const double chain = run.t_chain_us > 0 ? run.t_chain_us : run.t_loop_gpu_us;
		t_last_timing[0] = us(t0, t1);                              // FCCH acquisition incl. its decisions on the host
		t_last_timing[1] = chain;                                   // frame loop: launches until its kernels are through
		t_last_timing[2] = run.t_loop_gpu_us - chain;               // records to the caller's buffer
		t_last_timing[3] = us(t1, t2) - run.t_loop_gpu_us;          // host work around the loop (chains set up, states read)
		t_last_timing[4] = us(t2, now());                           // traffic-channel passes
	}
	if (timing) {
		auto us = [](auto a, auto b) { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() / 1e3; };
		fprintf(stderr, "rx_run: acquire %.0f us, frame loop %.0f us (launch+copy %.0f, collect %.0f), traffic passes %.0f us\n",
		        us(t0, t1), us(t1, t2), run.t_loop_gpu_us, us(t1, t2) - run.t_loop_gpu_us, us(t2, now()));
	}
	const std::vector<RxChain> &chains = run.chains;
	const std::vector<int32_t> &stat = run.stat, &nch = run.nch;
	const int A = n_arfcn;

	// ---- hand back: carriers in order, chains in order, frames in order -------------------------
	int total = 0;
	if (run.direct) {
		total = run.direct_total;            // already in the caller's buffer, in this very order (k_rx_pack)
	} else {
		for (const RxChain &c : chains) {       // chains were created carrier by carrier, chain by chain
			const int cnt = (int)c.rec.size();
			const int fit = std::max(0, std::min(cnt, max_records - total));
			if (fit)
				std::memcpy(out + total, c.rec.data(), (size_t)fit * sizeof(gmr1_hip_rx_record));
			total += cnt;
		}
	}
	*n_records = total;
	if (n_big) {
		int tb = 0;
		for (const RxChain &c : chains)
			for (const gmr1_hip_rx_big_record &rec : c.big) {
				if (tb < max_big)
					big_out[tb] = rec;
				tb++;
			}
		*n_big = tb;
	}
	for (int i = 0; i < A; i++) {
		if (status) status[i] = stat[i];
		if (n_chains) n_chains[i] = nch[i];
		if (rec_per_carrier) rec_per_carrier[i] = 0;
	}
	if (rec_per_carrier)
		for (const RxChain &c : chains)
			rec_per_carrier[c.a] += run.direct ? c.n_rec : (int)c.rec.size();
	return 0;
}

}  // namespace

namespace gmr1 {
int rx_run_dev_counted(void *stream, int n_arfcn, int sps, const float *iq, const uint64_t *offset, const uint64_t *length,
                       const uint16_t *arfcn, struct gmr1_hip_rx_record *out, int max_records, int *n_records,
                       int32_t *status, int32_t *n_chains, int32_t *rec_per_carrier)
{
	return rx_run_full_impl(stream, n_arfcn, sps, iq, nullptr, nullptr, offset, length, arfcn, nullptr, out, max_records, n_records,
	                        nullptr, 0, nullptr, status, n_chains, rec_per_carrier);
}
}  // namespace gmr1

extern "C" {

int gmr1_hip_rx_run_last_timing(double *us5)
{
	if (!us5)
		return -EINVAL;
	for (int i = 0; i < 5; i++)
		us5[i] = t_last_timing[i];
	return 0;
}

int gmr1_hip_rx_run_full_dev(void *stream_, int n_arfcn, int sps, const float *iq, const float *tch,
                             const float *csd, const uint64_t *offset, const uint64_t *length,
                             const uint16_t *arfcn, const uint8_t *kc,
                             struct gmr1_hip_rx_record *out, int max_records, int *n_records,
                             struct gmr1_hip_rx_big_record *big_out, int max_big, int *n_big,
                             int32_t *status, int32_t *n_chains)
{
	return rx_run_full_impl(stream_, n_arfcn, sps, iq, tch, csd, offset, length, arfcn, kc, out, max_records, n_records,
	                        big_out, max_big, n_big, status, n_chains, nullptr);
}

int gmr1_hip_rx_run_tch_dev(void *stream, int n_arfcn, int sps, const float *iq, const float *tch,
                            const uint64_t *offset, const uint64_t *length, const uint16_t *arfcn,
                            const uint8_t *kc,
                            struct gmr1_hip_rx_record *out, int max_records, int *n_records,
                            int32_t *status, int32_t *n_chains)
{
	return gmr1_hip_rx_run_full_dev(stream, n_arfcn, sps, iq, tch, nullptr, offset, length, arfcn, kc,
	                                out, max_records, n_records, nullptr, 0, nullptr, status, n_chains);
}

int gmr1_hip_rx_run_full(int n_arfcn, int sps, const float *iq, const float *tch, const float *csd, uint64_t iq_len,
                         const uint64_t *offset, const uint64_t *length, const uint16_t *arfcn, const uint8_t *kc,
                         struct gmr1_hip_rx_record *out, int max_records, int *n_records,
                         struct gmr1_hip_rx_big_record *big_out, int max_big, int *n_big,
                         int32_t *status, int32_t *n_chains)
{
	if (n_records) *n_records = 0;
	if (n_big) *n_big = 0;
	DevState *ds;
	int r = dev_state(&ds);
	if (r) return r;
	if (n_arfcn < 0 || !iq || !offset || !length)
		return fail(-EINVAL, "rx_run: iq/offset/length are required");
	for (int i = 0; i < n_arfcn; i++)
		if (offset[i] + length[i] > iq_len)
			return fail(-EINVAL, "rx_run: carrier %d runs past the end of iq", i);
	DBuf d_iq, d_tch, d_csd;
	HIP_TRY(d_iq.alloc(iq_len * 8));
	HIP_TRY(hipMemcpy(d_iq.p, iq, iq_len * 8, hipMemcpyHostToDevice));
	if (tch) {
		HIP_TRY(d_tch.alloc(iq_len * 8));
		HIP_TRY(hipMemcpy(d_tch.p, tch, iq_len * 8, hipMemcpyHostToDevice));
	}
	if (csd) {
		HIP_TRY(d_csd.alloc(iq_len * 8));
		HIP_TRY(hipMemcpy(d_csd.p, csd, iq_len * 8, hipMemcpyHostToDevice));
	}
	return gmr1_hip_rx_run_full_dev(nullptr, n_arfcn, sps, d_iq.as<float>(), tch ? d_tch.as<float>() : nullptr,
	                                csd ? d_csd.as<float>() : nullptr, offset, length, arfcn, kc, out, max_records,
	                                n_records, big_out, max_big, n_big, status, n_chains);
}

int gmr1_hip_rx_run_dev(void *stream, int n_arfcn, int sps, const float *iq,
                        const uint64_t *offset, const uint64_t *length, const uint16_t *arfcn,
                        struct gmr1_hip_rx_record *out, int max_records, int *n_records,
                        int32_t *status, int32_t *n_chains)
{
	return gmr1_hip_rx_run_tch_dev(stream, n_arfcn, sps, iq, nullptr, offset, length, arfcn, nullptr,
	                               out, max_records, n_records, status, n_chains);
}

int gmr1_hip_rx_run_tch(int n_arfcn, int sps, const float *iq, const float *tch, uint64_t iq_len,
                        const uint64_t *offset, const uint64_t *length, const uint16_t *arfcn, const uint8_t *kc,
                        struct gmr1_hip_rx_record *out, int max_records, int *n_records,
                        int32_t *status, int32_t *n_chains)
{
	if (n_records) *n_records = 0;
	DevState *ds;
	int r = dev_state(&ds);
	if (r) return r;
	if (n_arfcn < 0 || !iq || !offset || !length)
		return fail(-EINVAL, "rx_run: iq/offset/length are required");
	for (int i = 0; i < n_arfcn; i++)
		if (offset[i] + length[i] > iq_len)
			return fail(-EINVAL, "rx_run: carrier %d runs past the end of iq", i);
	DBuf d_iq, d_tch;
	HIP_TRY(d_iq.alloc(iq_len * 8));
	HIP_TRY(hipMemcpy(d_iq.p, iq, iq_len * 8, hipMemcpyHostToDevice));
	if (tch) {
		HIP_TRY(d_tch.alloc(iq_len * 8));
		HIP_TRY(hipMemcpy(d_tch.p, tch, iq_len * 8, hipMemcpyHostToDevice));
	}
	return gmr1_hip_rx_run_tch_dev(nullptr, n_arfcn, sps, d_iq.as<float>(), tch ? d_tch.as<float>() : nullptr,
	                               offset, length, arfcn, kc, out, max_records, n_records, status, n_chains);
}

int gmr1_hip_rx_run(int n_arfcn, int sps, const float *iq, uint64_t iq_len,
                    const uint64_t *offset, const uint64_t *length, const uint16_t *arfcn,
                    struct gmr1_hip_rx_record *out, int max_records, int *n_records,
                    int32_t *status, int32_t *n_chains)
{
	if (n_records) *n_records = 0;
	DevState *ds;
	int r = dev_state(&ds);
	if (r) return r;
	if (n_arfcn < 0 || !iq || !offset || !length)
		return fail(-EINVAL, "rx_run: iq/offset/length are required");
	for (int i = 0; i < n_arfcn; i++)
		if (offset[i] + length[i] > iq_len)
			return fail(-EINVAL, "rx_run: carrier %d runs past the end of iq", i);
	DBuf d_iq;
	HIP_TRY(d_iq.alloc(iq_len * 8));
	HIP_TRY(hipMemcpy(d_iq.p, iq, iq_len * 8, hipMemcpyHostToDevice));
	return gmr1_hip_rx_run_dev(nullptr, n_arfcn, sps, d_iq.as<float>(), offset, length, arfcn,
	                           out, max_records, n_records, status, n_chains);
}

// What gmr1_gsmtap_makemsg (reference src/gsmtap.c:43-71) puts on the wire for one decoded frame:
// the 16-byte struct gsmtap_hdr of libosmocore (version 2, hdr_len 4 words, type GMR1_UM = 0x0a,
// timeslot, arfcn BE16, signal_dbm, snr_db, frame_number BE32, sub_type, antenna_nr, sub_slot, res)
// followed by the L2 bytes.  The reference leaves the arfcn field 0; with_arfcn != 0 fills it.
// Host-only byte packing (the I/O sink itself -- the UDP socket -- stays with the caller).
static int gsmtap_pack_any(uint16_t arfcn, uint8_t type, uint32_t fn, uint8_t tn, const uint8_t *l2, int len,
                           int max_len, int with_arfcn, uint8_t *buf, int buf_len)
{
	const int total = 16 + len;
	if (len > max_len || buf_len < total)
		return fail(-EINVAL, "gsmtap_pack: need %d bytes, have %d", total, buf_len);
	std::memset(buf, 0, 16);
	buf[0] = 2;                       // GSMTAP_VERSION
	buf[1] = 4;                       // sizeof(struct gsmtap_hdr) / 4
	buf[2] = 0x0a;                    // GSMTAP_TYPE_GMR1_UM
	buf[3] = tn;
	if (with_arfcn) {
		buf[4] = (uint8_t)((arfcn >> 8) & 0x3f);     // 14-bit ARFCN, flags clear
		buf[5] = (uint8_t)(arfcn & 0xff);
	}
	buf[8] = (uint8_t)(fn >> 24);                    // htonl(fn)
	buf[9] = (uint8_t)(fn >> 16);
	buf[10] = (uint8_t)(fn >> 8);
	buf[11] = (uint8_t)fn;
	buf[12] = type;                   // GSMTAP_GMR1_BCCH 0x01, CCCH 0x02, TCH3 0x10 (| FACCH 0x02), TCH9 0x18 (| FACCH 0x02)
	std::memcpy(buf + 16, l2, (size_t)len);
	return total;
}

int gmr1_hip_gsmtap_pack(const struct gmr1_hip_rx_record *rec, int with_arfcn, uint8_t *buf, int buf_len)
{
	if (!rec || !buf)
		return fail(-EINVAL, "gsmtap_pack: rec / buf are required");
	return gsmtap_pack_any(rec->arfcn, rec->type, rec->fn, rec->tn, rec->l2, rec->len, 24, with_arfcn, buf, buf_len);
}

int gmr1_hip_gsmtap_pack_big(const struct gmr1_hip_rx_big_record *rec, int with_arfcn, uint8_t *buf, int buf_len)
{
	if (!rec || !buf)
		return fail(-EINVAL, "gsmtap_pack: rec / buf are required");
	return gsmtap_pack_any(rec->arfcn, rec->type, rec->fn, rec->tn, rec->l2, rec->len, 64, with_arfcn, buf, buf_len);
}

}  // extern "C"
