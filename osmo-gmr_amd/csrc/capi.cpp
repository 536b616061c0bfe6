// capi.cpp -- the C-ABI shim: reference-compatible per-burst calls and the
// batched entry points of include/gmr1_hip.h.  Host code only; every compute
// step is a HIP kernel (rx_kernels.hip).  There is no CPU fallback: without a
// HIP device every call returns -ENODEV.
#include <cerrno>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <strings.h>
#include <atomic>
#include <mutex>
#include <vector>

#include <hip/hip_runtime.h>

#include <osmocom/gmr1/l1/bcch.h>
#include <osmocom/gmr1/l1/ccch.h>

#include "capi_common.h"
#include "rx_server.h"
#include "rx_debug.h"

namespace gmr1 {

namespace {
thread_local char t_err[256] = "";
std::mutex g_mu;
constexpr int kMaxDevices = 16;
DevState g_dev[kMaxDevices];
bool g_host_types_ready = false;
FcchTables g_fcch_tables;
}  // namespace

DevBurst g_host_types[kNumTypes];

int fail(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(t_err, sizeof(t_err), fmt, ap);
	va_end(ap);
	return code;
}

const char *last_error() { return t_err; }

namespace {
std::atomic<int> g_conv_decoder{-1};       // -1: not chosen yet (the environment decides on first use)
}
int conv_acc()
{
	int v = g_conv_decoder.load(std::memory_order_relaxed);
	if (v < 0) {
		// default: what osmo_conv_decode() of every libosmocore since 0.10 (2017) runs for the K = 5 / 7, N <= 4 codes
		// (INTEGRATION.md "Which Viterbi decoder"); GMR1_HIP_CONV_DECODER=generic selects the older behaviour
		// Accepted spellings, case-insensitive: "generic" / "0", "acc" / "1"; anything else is reported once on stderr and
		// the default stands (a misspelt "Generic" must not silently select the other decoder).
		const char *e = getenv("GMR1_HIP_CONV_DECODER");
		v = GMR1_HIP_CONV_ACC;
		if (e && *e) {
			if (!strcasecmp(e, "generic") || !strcmp(e, "0"))
				v = GMR1_HIP_CONV_GENERIC;
			else if (strcasecmp(e, "acc") && strcmp(e, "1"))
				fprintf(stderr, "libgmr1_hip: GMR1_HIP_CONV_DECODER=\"%s\" is neither generic / 0 nor acc / 1: using acc\n", e);
		}
		int expect = -1;
		if (!g_conv_decoder.compare_exchange_strong(expect, v))
			v = expect;
	}
	return v == GMR1_HIP_CONV_ACC;
}

std::mutex &custom_slots_mutex()
{
	static std::mutex mu;
	return mu;
}

static int host_types_build();

int host_types()
{
	// first use may come from several threads at once (the legacy one-burst calls take no lock)
	static std::once_flag once;
	static int rv_once = 0;
	std::call_once(once, [] { rv_once = host_types_build(); });
	return rv_once;
}

static int host_types_build()
{
	tables_init();
	std::memset(g_host_types, 0, sizeof(g_host_types));
	for (int i = 0; i < GMR1_HIP_N_BURSTS; i++) {
		gmr1_hip_burst_flat f;
		int rv = flatten(kBuiltin[i], &f, kBuiltinName[i]);
		if (rv == 0)
			rv = to_dev(f, &g_host_types[i]);
		if (rv)
			return fail(rv, "built-in burst table %d is inconsistent", i);
	}
	fcch_tables_init(&g_fcch_tables);
	g_host_types_ready = true;
	return 0;
}

int dev_state(DevState **out)
{
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if (e != hipSuccess || count <= 0)
		return fail(-ENODEV, "no HIP device available (%s)", hipGetErrorString(e));
	int dev = 0;
	HIP_TRY(hipGetDevice(&dev));
	if (dev < 0 || dev >= kMaxDevices)
		return fail(-EINVAL, "device index %d out of range", dev);
	std::lock_guard<std::mutex> lk(g_mu);
	DevState &s = g_dev[dev];
	if (!s.ready) {
		int rv = host_types();
		if (rv)
			return rv;
		HIP_TRY(upload_types(g_host_types, 0, kNumTypes, nullptr));
		HIP_TRY(upload_fcch_tables(&g_fcch_tables, nullptr));
		HIP_TRY(hipStreamSynchronize(nullptr));
		s.ready = true;
	}
	*out = &s;
	return 0;
}

int WsLease::acquire(DevState *s, hipStream_t st)
{
	s->ws_mu.lock();
	s_ = s;
	st_ = st;
	if (s->ws_depth++ == 0 && s->ws_ev) {
		const hipError_t e = hipStreamWaitEvent(st, s->ws_ev, 0);
		if (e != hipSuccess)
			return fail(-EIO, "workspace lease: hipStreamWaitEvent: %s", hipGetErrorString(e));
	}
	return 0;
}

WsLease::~WsLease()
{
	if (!s_)
		return;
	if (--s_->ws_depth == 0) {
		if (!s_->ws_ev && hipEventCreateWithFlags(&s_->ws_ev, hipEventDisableTiming) != hipSuccess)
			s_->ws_ev = nullptr;
		if (s_->ws_ev)
			(void)hipEventRecord(s_->ws_ev, st_);
	}
	s_->ws_mu.unlock();
}

int dev_workspace(DevState *s, size_t bytes, void **out)
{
	std::lock_guard<std::mutex> lk(g_mu);
	if (s->ws_bytes < bytes) {
		if (s->ws) {
			HIP_TRY(hipDeviceSynchronize());
			HIP_TRY(hipFree(s->ws));
			s->ws = nullptr;
			s->ws_bytes = 0;
		}
		const size_t want = bytes + bytes / 4;
		HIP_TRY(hipMalloc(&s->ws, want));
		s->ws_bytes = want;
	}
	*out = s->ws;
	return 0;
}

namespace {
int window_len(int burst_len, int sps, int win) { return burst_len * sps + win; }
}  // namespace

}  // namespace gmr1

using namespace gmr1;

extern "C" {

// ---------------------------------------------------------------------------
// library / device
// ---------------------------------------------------------------------------
const char *gmr1_hip_version(void)
{
	// names the Viterbi decoder in force at the time of the call (gmr1_hip_set_conv_decoder)
	return conv_acc() ? "gmr1-hip 0.4 (gfx950; conv decoder: acc)" : "gmr1-hip 0.4 (gfx950; conv decoder: generic)";
}
const char *gmr1_hip_last_error(void) { return last_error(); }

int gmr1_hip_set_conv_decoder(int decoder)
{
	if (decoder != GMR1_HIP_CONV_GENERIC && decoder != GMR1_HIP_CONV_ACC)
		return fail(-EINVAL, "gmr1_hip_set_conv_decoder: %d is neither GMR1_HIP_CONV_GENERIC nor GMR1_HIP_CONV_ACC", decoder);
	g_conv_decoder.store(decoder);
	return 0;
}

int gmr1_hip_get_conv_decoder(void) { return conv_acc() ? GMR1_HIP_CONV_ACC : GMR1_HIP_CONV_GENERIC; }

int gmr1_hip_init(int device)
{
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if (e != hipSuccess || count <= 0)
		return fail(-ENODEV, "no HIP device available (%s)", hipGetErrorString(e));
	if (device < 0 || device >= count)
		return fail(-EINVAL, "device %d not in [0,%d)", device, count);
	HIP_TRY(hipSetDevice(device));
	DevState *s;
	return dev_state(&s);
}

int gmr1_hip_burst_info(int burst_id, struct gmr1_hip_burst_flat *out)
{
	if (burst_id < 0 || burst_id >= GMR1_HIP_N_BURSTS || !out)
		return fail(-EINVAL, "bad burst id %d", burst_id);
	return flatten(kBuiltin[burst_id], out, kBuiltinName[burst_id]);
}

// ---------------------------------------------------------------------------
// demod batch
// ---------------------------------------------------------------------------
static int dbg_stop_env()
{
	static int v = -1;
	if (v < 0) {
		const char *e = profile_env("GMR1_HIP_DBG_STOP");
		v = e ? atoi(e) : 0;
	}
	return v;
}

// the launch arguments of a demodulation batch, with the kernel chosen for it (RxArgs::impl)
static int demod_args(int type, const DevBurst &ht,
                      int n, int sps, int in_len, const float *iq, const uint64_t *offset,
                      const float *freq_shift, int8_t *ebits, int ebits_stride, int32_t *sync_id,
                      float *toa, float *freq_err, float *ssyms, int32_t *rv, float *energy, RxArgs *out)
{
	if (n < 0 || !iq || !offset || !rv)
		return fail(-EINVAL, "demod: n/iq/offset/rv are required");
	if (sps < 1 || sps > 16)
		return fail(-EINVAL, "demod: sps=%d out of range (1..16)", sps);
	const int w = in_len - ht.len * sps + 1;
	if (w < 1 || in_len > kMaxInLen)
		return fail(-EINVAL, "demod: window of %d samples gives %d lags (>= 1, <= %d samples supported)", in_len, w, kMaxInLen);
	if (ebits && ebits_stride < ht.ebits)
		return fail(-EINVAL, "demod: ebits_stride %d < %d", ebits_stride, ht.ebits);
	RxArgs a;
	std::memset(&a, 0, sizeof(a));
	a.n = n; a.sps = sps; a.in_len[0] = a.in_len[1] = in_len;
	a.fixed_type = type;
	a.ebits_stride = ebits_stride;
	a.ssyms_stride = ht.len;
	a.dbg_stop = dbg_stop_env();
	a.iq = reinterpret_cast<const float2 *>(iq);
	a.offset = offset; a.freq_shift = freq_shift;
	a.ebits = ebits; a.sync_id = sync_id; a.toa = toa; a.freq_err = freq_err; a.ssyms = ssyms; a.rv = rv;
	a.energy = energy;
	// Large batches of a simple format (one training sequence, QPSK, <= 3 sync chunks of <= 128 window samples,
	// <= 18 sync symbols, <= 256 symbols; sps 4: NT3 speech, DC2, BCCH, DC6) take the four-bursts-per-wave kernel, where the
	// serial phases of four bursts share their instructions (k_rx4g); everything else, and small batches, one burst per wave.
	static int gen_off = -1;                    // profiling only: GMR1_HIP_RX_GEN=0 keeps every batch on k_rx
	if (gen_off < 0) {
		const char *e = profile_env("GMR1_HIP_RX_GEN");
		gen_off = (e && atoi(e) == 0) ? 1 : 0;
	}
	if (!gen_off && n > 4096 && sps == 4 && ht.n_sync == 1 && ht.nbits == 2 && ht.n_chunks[0] >= 1 &&
	    ht.n_chunks[0] <= 3 && ht.sync_tl[0] <= 18 && ht.len <= 256 && in_len <= 1024 && w <= 128 && ht.ebits <= 432) {
		bool fits = true;
		int stage = 0;
		for (int c = 0; c < ht.n_chunks[0]; c++) {
			const int wl = ht.sync[0][c].len * sps + w - 1;
			fits &= wl <= 128;
			stage += wl;
		}
		if (fits) {
			// short formats with a single sync chunk (NT3 speech, DC2) have their own instantiation
			const bool small = in_len <= 512 && ht.len <= 128 && ht.n_chunks[0] == 1 && ht.sync_tl[0] <= 16 && stage <= 64;
			a.impl = small ? 3 : 2;
			a.stage_samples = stage;
		}
	}
	// The same kernel's variant for two training sequences of one chunk each at the same place, one bit per symbol (NT3 FACCH)
	if (!gen_off && a.impl == 0 && n > 4096 && sps == 4 && a.dbg_stop == 0 && ht.n_sync == 2 && ht.nbits == 1 &&
	    ht.n_chunks[0] == 1 && ht.n_chunks[1] == 1 && ht.sync[0][0].pos == ht.sync[1][0].pos &&
	    ht.sync[0][0].len == ht.sync[1][0].len && ht.sync_tl[0] <= 8 && ht.sync_tl[0] == ht.sync_tl[1] && in_len <= 512 &&
	    ht.len <= 128 && w <= 64 && ht.ebits <= 432) {
		const int stage = ht.sync[0][0].len * sps + w - 1;
		if (stage <= 64) {
			a.impl = 4;
			a.stage_samples = stage;
		}
	}
	*out = a;
	return 0;
}

static int demod_dev_impl(hipStream_t st, int type, const DevBurst &ht,
                          int n, int sps, int in_len, const float *iq, const uint64_t *offset,
                          const float *freq_shift, int8_t *ebits, int ebits_stride, int32_t *sync_id,
                          float *toa, float *freq_err, float *ssyms, int32_t *rv, float *energy = nullptr)
{
	RxArgs a;
	int r = demod_args(type, ht, n, sps, in_len, iq, offset, freq_shift, ebits, ebits_stride, sync_id, toa, freq_err, ssyms, rv,
	                   energy, &a);
	if (r) return r;
	HIP_TRY(launch_rx(a, false, in_len, st));
	return 0;
}

}  // extern "C" (closed for the shared implementation below)

namespace gmr1 {
// built-in burst type, device pointers, with the burst_energy() output the receive loop needs
int demod_dev_energy(hipStream_t st, int burst_id, int n, int sps, int in_len, const float *iq,
                     const uint64_t *offset, const float *freq_shift, int8_t *ebits, int ebits_stride,
                     int32_t *sync_id, float *toa, float *energy, int32_t *rv)
{
	if (burst_id < 0 || burst_id >= GMR1_HIP_N_BURSTS)
		return fail(-EINVAL, "bad burst id %d", burst_id);
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	return demod_dev_impl(st, burst_id, g_host_types[burst_id], n, sps, in_len, iq, offset, freq_shift,
	                      ebits, ebits_stride, sync_id, toa, nullptr, nullptr, rv, energy);
}
}  // namespace gmr1

extern "C" {

// What rx_tch3 does with a speech burst (gmr1_rx.c:551-587): gmr1_pi4cxpsk_demod of the NT3 speech format, then
// gmr1_tch3_decode of its 212 soft bits -- for a batch, in ONE launch where the four-bursts-per-wave demodulator applies
// (k_rx4g_tch3: the soft bits never leave LDS), otherwise as the two launches the separate entry points make.
int gmr1_hip_tch3_rx_batch_dev(void *stream, int n, int sps, int in_len,
                               const float *iq, const uint64_t *offset, const float *freq_shift,
                               int m, const uint8_t *ciph,
                               int8_t *ebits, int32_t *sync_id, float *toa, int32_t *rv,
                               uint8_t *frames, uint8_t *bits_s, int32_t *conv)
{
	if (n < 0 || !iq || !offset || !rv || !frames)
		return fail(-EINVAL, "tch3 rx: n/iq/offset/rv/frames are required");
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n == 0) return 0;
	const int type = GMR1_HIP_NT3_SPEECH;
	const DevBurst &ht = g_host_types[type];
	RxArgs a;
	r = demod_args(type, ht, n, sps, in_len, iq, offset, freq_shift, ebits, 212, sync_id, toa, nullptr, nullptr, rv, nullptr, &a);
	if (r) return r;
	Tch3Args t;
	t.n = n; t.m = m ? 1 : 0; t.conv_acc = conv_acc(); t.ebits = ebits; t.ciph = ciph; t.frames = frames; t.bits_s = bits_s;
	t.conv = conv;
	if (a.impl == 3 && ht.ebits == 212 && a.dbg_stop == 0) {
		HIP_TRY(launch_rx_tch3(a, t, (hipStream_t)stream));
		return 0;
	}
	// two launches; the soft bits pass through the caller's buffer or the library's workspace
	WsLease lease;
	if (!ebits) {
		void *ws;
		if ((r = lease.acquire(s, (hipStream_t)stream))) return r;
		r = dev_workspace(s, (size_t)n * 212, &ws);
		if (r) return r;
		a.ebits = reinterpret_cast<int8_t *>(ws);
		t.ebits = a.ebits;
	}
	HIP_TRY(launch_rx(a, false, in_len, (hipStream_t)stream));
	HIP_TRY(launch_tch3(t, (hipStream_t)stream));
	return 0;
}

int gmr1_hip_demod_batch_dev(void *stream, int burst_id, int n, int sps, int in_len,
                             const float *iq, const uint64_t *offset, const float *freq_shift,
                             int8_t *ebits, int ebits_stride, int32_t *sync_id,
                             float *toa, float *freq_err, float *ssyms, int32_t *rv)
{
	if (burst_id < 0 || burst_id >= GMR1_HIP_N_BURSTS)
		return fail(-EINVAL, "bad burst id %d", burst_id);
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	return demod_dev_impl((hipStream_t)stream, burst_id, g_host_types[burst_id], n, sps, in_len,
	                      iq, offset, freq_shift, ebits, ebits_stride, sync_id, toa, freq_err, ssyms, rv);
}

// host-pointer staging shared by the batch wrapper and the legacy call
static int demod_host_impl(int type, const DevBurst &ht, const DevBurst *custom,
                           int n, int sps, int in_len, const float *iq, uint64_t iq_len,
                           const uint64_t *offset, const float *freq_shift,
                           int8_t *ebits, int ebits_stride, int32_t *sync_id,
                           float *toa, float *freq_err, float *ssyms, int32_t *rv)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n <= 0)
		return 0;
	for (int i = 0; i < n; i++)
		if (offset[i] + (uint64_t)in_len > iq_len)
			return fail(-EINVAL, "burst %d runs past the end of iq", i);
	hipStream_t st = nullptr;
	// a caller-defined description occupies the one spare table slot for the duration of the call: two threads
	// demodulating different custom formats must not interleave upload and launch
	std::unique_lock<std::mutex> lk(custom_slots_mutex(), std::defer_lock);
	if (custom) {
		lk.lock();
		HIP_TRY(upload_types(custom, kCustomSlot, 1, st));
	}
	DBuf d_iq, d_off, d_fs, d_eb, d_sid, d_toa, d_fe, d_ss, d_rv;
	HIP_TRY(d_iq.alloc(iq_len * 8));
	HIP_TRY(d_off.alloc((size_t)n * 8));
	HIP_TRY(d_rv.alloc((size_t)n * 4));
	HIP_TRY(hipMemcpy(d_iq.p, iq, iq_len * 8, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(d_off.p, offset, (size_t)n * 8, hipMemcpyHostToDevice));
	if (freq_shift) {
		HIP_TRY(d_fs.alloc((size_t)n * 4));
		HIP_TRY(hipMemcpy(d_fs.p, freq_shift, (size_t)n * 4, hipMemcpyHostToDevice));
	}
	if (ebits) HIP_TRY(d_eb.alloc((size_t)n * ebits_stride));
	if (sync_id) HIP_TRY(d_sid.alloc((size_t)n * 4));
	if (toa) HIP_TRY(d_toa.alloc((size_t)n * 4));
	if (freq_err) HIP_TRY(d_fe.alloc((size_t)n * 4));
	if (ssyms) HIP_TRY(d_ss.alloc((size_t)n * ht.len * 4));
	r = demod_dev_impl(st, type, ht, n, sps, in_len, d_iq.as<float>(), d_off.as<uint64_t>(),
	                   freq_shift ? d_fs.as<float>() : nullptr, ebits ? d_eb.as<int8_t>() : nullptr,
	                   ebits_stride, sync_id ? d_sid.as<int32_t>() : nullptr,
	                   toa ? d_toa.as<float>() : nullptr, freq_err ? d_fe.as<float>() : nullptr,
	                   ssyms ? d_ss.as<float>() : nullptr, d_rv.as<int32_t>());
	if (r) return r;
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipMemcpy(rv, d_rv.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	if (ebits) HIP_TRY(hipMemcpy(ebits, d_eb.p, (size_t)n * ebits_stride, hipMemcpyDeviceToHost));
	if (sync_id) HIP_TRY(hipMemcpy(sync_id, d_sid.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	if (toa) HIP_TRY(hipMemcpy(toa, d_toa.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	if (freq_err) HIP_TRY(hipMemcpy(freq_err, d_fe.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	if (ssyms) HIP_TRY(hipMemcpy(ssyms, d_ss.p, (size_t)n * ht.len * 4, hipMemcpyDeviceToHost));
	return 0;
}

// host pointers: staged through HBM
int gmr1_hip_tch3_rx_batch(int n, int sps, int in_len,
                           const float *iq, uint64_t iq_len, const uint64_t *offset, const float *freq_shift,
                           int m, const uint8_t *ciph,
                           int8_t *ebits, int32_t *sync_id, float *toa, int32_t *rv,
                           uint8_t *frames, uint8_t *bits_s, int32_t *conv)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n <= 0)
		return 0;
	if (!iq || !offset || !rv || !frames)
		return fail(-EINVAL, "tch3 rx: n/iq/offset/rv/frames are required");
	for (int i = 0; i < n; i++)
		if (offset[i] + (uint64_t)in_len > iq_len)
			return fail(-EINVAL, "burst %d runs past the end of iq", i);
	DBuf d_iq, d_off, d_fs, d_ci, d_eb, d_sid, d_toa, d_rv, d_fr, d_s, d_conv;
	HIP_TRY(d_iq.alloc(iq_len * 8));
	HIP_TRY(d_off.alloc((size_t)n * 8));
	HIP_TRY(d_rv.alloc((size_t)n * 4));
	HIP_TRY(d_fr.alloc((size_t)n * 20));
	HIP_TRY(hipMemcpy(d_iq.p, iq, iq_len * 8, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(d_off.p, offset, (size_t)n * 8, hipMemcpyHostToDevice));
	if (freq_shift) {
		HIP_TRY(d_fs.alloc((size_t)n * 4));
		HIP_TRY(hipMemcpy(d_fs.p, freq_shift, (size_t)n * 4, hipMemcpyHostToDevice));
	}
	if (ciph) {
		HIP_TRY(d_ci.alloc((size_t)n * 208));
		HIP_TRY(hipMemcpy(d_ci.p, ciph, (size_t)n * 208, hipMemcpyHostToDevice));
	}
	if (ebits) HIP_TRY(d_eb.alloc((size_t)n * 212));
	if (sync_id) HIP_TRY(d_sid.alloc((size_t)n * 4));
	if (toa) HIP_TRY(d_toa.alloc((size_t)n * 4));
	if (bits_s) HIP_TRY(d_s.alloc((size_t)n * 4));
	if (conv) HIP_TRY(d_conv.alloc((size_t)n * 8));
	r = gmr1_hip_tch3_rx_batch_dev(nullptr, n, sps, in_len, d_iq.as<float>(), d_off.as<uint64_t>(),
	                               freq_shift ? d_fs.as<float>() : nullptr, m, ciph ? d_ci.as<uint8_t>() : nullptr,
	                               ebits ? d_eb.as<int8_t>() : nullptr, sync_id ? d_sid.as<int32_t>() : nullptr,
	                               toa ? d_toa.as<float>() : nullptr, d_rv.as<int32_t>(), d_fr.as<uint8_t>(),
	                               bits_s ? d_s.as<uint8_t>() : nullptr, conv ? d_conv.as<int32_t>() : nullptr);
	if (r) return r;
	HIP_TRY(hipStreamSynchronize(nullptr));
	HIP_TRY(hipMemcpy(rv, d_rv.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(frames, d_fr.p, (size_t)n * 20, hipMemcpyDeviceToHost));
	if (ebits) HIP_TRY(hipMemcpy(ebits, d_eb.p, (size_t)n * 212, hipMemcpyDeviceToHost));
	if (sync_id) HIP_TRY(hipMemcpy(sync_id, d_sid.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	if (toa) HIP_TRY(hipMemcpy(toa, d_toa.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	if (bits_s) HIP_TRY(hipMemcpy(bits_s, d_s.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	if (conv) HIP_TRY(hipMemcpy(conv, d_conv.p, (size_t)n * 8, hipMemcpyDeviceToHost));
	return 0;
}

int gmr1_hip_demod_batch(int burst_id, int n, int sps, int in_len,
                         const float *iq, uint64_t iq_len, const uint64_t *offset, const float *freq_shift,
                         int8_t *ebits, int ebits_stride, int32_t *sync_id,
                         float *toa, float *freq_err, float *ssyms, int32_t *rv)
{
	if (burst_id < 0 || burst_id >= GMR1_HIP_N_BURSTS)
		return fail(-EINVAL, "bad burst id %d", burst_id);
	int r = host_types();
	if (r) return r;
	return demod_host_impl(burst_id, g_host_types[burst_id], nullptr, n, sps, in_len, iq, iq_len, offset,
	                       freq_shift, ebits, ebits_stride, sync_id, toa, freq_err, ssyms, rv);
}

// Debugging aid: one burst demodulated exactly as gmr1_hip_demod_batch does, plus the four intermediate vectors the
// reference writes out under ENABLE_DEBUG_SIGNAL (include/osmocom/gmr1/sdr/defs.h:35-39; pi4cxpsk.c:251,345,545,582).
// Host pointers, blocking, one wave: for looking at ONE burst of a capture that decodes differently, not a data path.
int gmr1_hip_demod_taps(int burst_id, int sps, int in_len, const float *iq, float freq_shift,
                        float *corr, float *burst, float *align, float *final_,
                        int8_t *ebits, int32_t *sync_id, float *toa, float *freq_err, float *ssyms, int32_t *rv)
{
	if (burst_id < 0 || burst_id >= GMR1_HIP_N_BURSTS)
		return fail(-EINVAL, "bad burst id %d", burst_id);
	int r = host_types();
	if (r) return r;
	const DevBurst &ht = g_host_types[burst_id];
	if (!iq || !rv)
		return fail(-EINVAL, "demod taps: iq and rv are required");
	if (sps < 1 || sps > 16)
		return fail(-EINVAL, "demod taps: sps=%d out of range (1..16)", sps);
	const int w = in_len - ht.len * sps + 1;
	if (w < 1 || in_len > kMaxInLen)
		return fail(-EINVAL, "demod taps: window of %d samples gives %d lags (>= 1, <= %d samples supported)", in_len, w, kMaxInLen);
	DevState *s;
	r = dev_state(&s);
	if (r) return r;
	DBuf d_iq, d_off, d_fs, d_eb, d_sid, d_toa, d_fe, d_ss, d_rv, d_taps;
	const uint64_t zero = 0;
	const size_t n_taps = (size_t)w + 2 * (size_t)in_len + 4 * (size_t)ht.len;     // floats: corr, burst, align, final
	HIP_TRY(d_iq.alloc((size_t)in_len * 8));
	HIP_TRY(d_off.alloc(8));
	HIP_TRY(d_fs.alloc(4));
	HIP_TRY(d_eb.alloc(ht.ebits));
	HIP_TRY(d_sid.alloc(4));
	HIP_TRY(d_toa.alloc(4));
	HIP_TRY(d_fe.alloc(4));
	HIP_TRY(d_ss.alloc((size_t)ht.len * 4));
	HIP_TRY(d_rv.alloc(4));
	HIP_TRY(d_taps.alloc(n_taps * 4 + 16));
	HIP_TRY(hipMemcpy(d_iq.p, iq, (size_t)in_len * 8, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(d_off.p, &zero, 8, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(d_fs.p, &freq_shift, 4, hipMemcpyHostToDevice));
	RxArgs a;
	std::memset(&a, 0, sizeof(a));
	a.n = 1; a.sps = sps; a.in_len[0] = a.in_len[1] = in_len;
	a.fixed_type = burst_id;
	a.ebits_stride = ht.ebits;
	a.ssyms_stride = ht.len;
	a.iq = d_iq.as<float2>();
	a.offset = d_off.as<uint64_t>();
	a.freq_shift = d_fs.as<float>();
	a.ebits = d_eb.as<int8_t>();
	a.sync_id = d_sid.as<int32_t>();
	a.toa = d_toa.as<float>();
	a.freq_err = d_fe.as<float>();
	a.ssyms = d_ss.as<float>();
	a.rv = d_rv.as<int32_t>();
	// (burst / align / final are complex: they come first so that they sit on 8-byte boundaries)
	float *t = d_taps.as<float>();
	RxTapsOut o;
	o.burst = reinterpret_cast<float2 *>(t);
	o.align = o.burst + in_len;
	o.final_ = o.align + ht.len;
	o.corr = reinterpret_cast<float *>(o.final_ + ht.len);
	HIP_TRY(launch_rx_taps(a, o, nullptr));
	HIP_TRY(hipStreamSynchronize(nullptr));
	HIP_TRY(hipMemcpy(rv, d_rv.p, 4, hipMemcpyDeviceToHost));
	if (corr) HIP_TRY(hipMemcpy(corr, o.corr, (size_t)w * 4, hipMemcpyDeviceToHost));
	if (burst) HIP_TRY(hipMemcpy(burst, o.burst, (size_t)in_len * 8, hipMemcpyDeviceToHost));
	if (align) HIP_TRY(hipMemcpy(align, o.align, (size_t)ht.len * 8, hipMemcpyDeviceToHost));
	if (final_) HIP_TRY(hipMemcpy(final_, o.final_, (size_t)ht.len * 8, hipMemcpyDeviceToHost));
	if (ebits) HIP_TRY(hipMemcpy(ebits, d_eb.p, ht.ebits, hipMemcpyDeviceToHost));
	if (sync_id) HIP_TRY(hipMemcpy(sync_id, d_sid.p, 4, hipMemcpyDeviceToHost));
	if (toa) HIP_TRY(hipMemcpy(toa, d_toa.p, 4, hipMemcpyDeviceToHost));
	if (freq_err) HIP_TRY(hipMemcpy(freq_err, d_fe.p, 4, hipMemcpyDeviceToHost));
	if (ssyms) HIP_TRY(hipMemcpy(ssyms, d_ss.p, (size_t)ht.len * 4, hipMemcpyDeviceToHost));
	return 0;
}

// ---------------------------------------------------------------------------
// reference-compatible single burst demodulation (pi4cxpsk.h:101-105)
// ---------------------------------------------------------------------------
}  // extern "C" (closed for the one-burst machinery)

namespace {

// ---- the reference's one-burst calls without per-call allocations -----------------------------------------------
// An unchanged gmr1_rx.c makes ~1300 blocking calls per carrier-minute (gmr1_pi4cxpsk_demod, then gmr1_bcch_decode /
// gmr1_ccch_decode on what it returned).  Each used to cost a handful of hipMalloc / pageable hipMemcpy / hipFree
// round trips; now operands and results live in ONE pinned, device-mapped host block created on first use: the call
// copies its input there (<= 16 KB), launches on a private stream, the kernel reads and writes the block over the
// link (zero copy), and the call returns when the stream has drained.  For the BCCH and DC6 formats the demodulator
// call runs the fused kernel (the layer-1 chain on the soft bits it has just produced costs nothing extra) and
// remembers (soft bits -> L2, CRC, metric); the decode call that follows with those very soft bits -- compared byte
// by byte -- is answered from that memo, anything else is decoded on the GPU as before.  Process-wide, one call at a
// time (the reference's calls are not re-entrant either).
struct OneBurst {
	std::mutex mu;
	int dev = -1;
	hipStream_t st = nullptr;
	unsigned char *h = nullptr, *d = nullptr;      // the block: host address, device address
	// the resident server of the fused BCCH / DC6 call at 4 samples per symbol (rx_server.h)
	hipStream_t srv_st = nullptr;
	uint32_t seq = 0, gen = 0;
	int srv_acc = -1;
	bool memo = false;
	int memo_chain = 0, memo_n = 0, memo_acc = 0;
	int8_t memo_eb[432];
	uint8_t memo_l2[24];
	int32_t memo_crc = 0, memo_conv = 0;
};
OneBurst g_one;

constexpr size_t kOneIq = 0;                        // kMaxInLen complex samples
constexpr size_t kOneOff = (size_t)kMaxInLen * 8;   // uint64 offset (0), uint8 kind, float freq_shift
constexpr size_t kOneOut = kOneOff + 64;            // rv, sync_id, toa, freq_err, crc, conv | l2[24] at +32 | soft bits at +64
constexpr size_t kOneEb = kOneOut + 64;
constexpr size_t kOneMail = kOneEb + 1024;          // OneMail
constexpr size_t kOneBytes = kOneMail + 64;

// g_one.mu held.  0, or -errno; *usable = false when the context belongs to another device (caller takes the slow path)
int one_ready(bool *usable)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	int dev = 0;
	HIP_TRY(hipGetDevice(&dev));
	if (!g_one.h) {
		HIP_TRY(hipStreamCreateWithFlags(&g_one.st, hipStreamNonBlocking));
		void *h = nullptr, *d = nullptr;
		HIP_TRY(hipHostMalloc(&h, kOneBytes, hipHostMallocMapped));
		HIP_TRY(hipHostGetDevicePointer(&d, h, 0));
		std::memset(h, 0, kOneBytes);               // (the mailbox: no request, no answer, generation 0 = no server yet)
		g_one.h = static_cast<unsigned char *>(h);
		g_one.d = static_cast<unsigned char *>(d);
		g_one.dev = dev;
	}
	*usable = g_one.dev == dev;
	return 0;
}

template <typename T> T *one_h(size_t off) { return reinterpret_cast<T *>(g_one.h + off); }
template <typename T> T *one_d(size_t off) { return reinterpret_cast<T *>(g_one.d + off); }

// The fused one-burst call through the resident server (rx_server_kernels.inc): the request is in the block; post its number,
// start a server if none is alive (or the one alive decodes with the other Viterbi decoder), spin on the answer's number.
// GMR1_HIP_ONE_BURST_SERVER=0 in the environment keeps the launch per call.  g_one.mu held.  0, 1 = not taken (the caller
// launches as before), or -errno.
constexpr unsigned kServerIdleUs = 200, kServerLifeUs = 500000;
// the next server generation, on the servers' one stream: it starts when the last one has gone, so there is never more
// than one at work.  The generation number is written BEFORE any request it is to answer, and a server reads the request
// number before the generation: a superseded server cannot take a request posted behind the change.
int one_server_start(OneMail *mh)
{
	RxArgs a;
	int r = rx_fused_base_args(4, one_d<float>(kOneIq), &a);
	if (r) return r;
	a.n = 1;
	a.offset = one_d<uint64_t>(kOneOff); a.kind = one_d<uint8_t>(kOneOff + 8); a.freq_shift = one_d<float>(kOneOff + 12);
	a.l2 = one_d<uint8_t>(kOneOut + 32); a.crc = one_d<int32_t>(kOneOut + 16); a.conv = one_d<int32_t>(kOneOut + 20);
	a.toa = one_d<float>(kOneOut + 8); a.freq_err = one_d<float>(kOneOut + 12);
	a.ebits = one_d<int8_t>(kOneEb); a.rv = one_d<int32_t>(kOneOut);
	volatile uint32_t *v_ended = &mh->ended, *v_gen = &mh->gen;
	*v_gen = ++g_one.gen;
	std::atomic_thread_fence(std::memory_order_seq_cst);
	const hipError_t e = launch_one_server(a, one_d<OneMail>(kOneMail), g_one.gen, kServerIdleUs, kServerLifeUs, g_one.srv_st);
	if (e != hipSuccess) {
		*v_ended = g_one.gen;                     // it never ran: the next call starts another
		return fail(-EIO, "one-burst server: launch failed: %s", hipGetErrorString(e));
	}
	g_one.srv_acc = conv_acc();
	return 0;
}

int one_server_call()
{
	static const bool enabled = [] { const char *e = getenv("GMR1_HIP_ONE_BURST_SERVER"); return !(e && e[0] == '0'); }();
	if (!enabled)
		return 1;
	if (!g_one.srv_st)
		HIP_TRY(hipStreamCreateWithFlags(&g_one.srv_st, hipStreamNonBlocking));
	OneMail *mh = one_h<OneMail>(kOneMail);
	volatile uint32_t *v_req = &mh->req, *v_done = &mh->done, *v_ended = &mh->ended;
	int launches = 0, r;
	// a server that decodes with the other Viterbi decoder is retired before the request exists
	if (g_one.gen != 0 && g_one.srv_acc != conv_acc()) {
		launches++;
		if ((r = one_server_start(mh))) return r;
	}
	const uint32_t seq = ++g_one.seq;
	std::atomic_thread_fence(std::memory_order_release);
	*v_req = seq;
	// Giving up on the server must not leave a live request behind: a server that starts late (queued behind a long kernel)
	// would serve it after this call has returned and the lock is released -- into a block the next call is rewriting.  So
	// the generation is retired first (a server reads the request number BEFORE the generation: none of an older generation
	// takes the request any more), the servers' stream is drained (one that was in the middle of the request finishes; every
	// server ends by itself), and the call goes on as a launch per call (return 1) instead of failing.
	auto give_up = [&](const char *why) -> int {
		volatile uint32_t *v_gen = &mh->gen;
		*v_gen = ++g_one.gen;
		std::atomic_thread_fence(std::memory_order_seq_cst);
		const hipError_t e = hipStreamSynchronize(g_one.srv_st);
		*v_ended = g_one.gen;                         // nothing is alive: the next call starts a server of its own
		*v_done = seq;                                // (and no later server may mistake the abandoned request for a new one)
		if (e != hipSuccess)
			return fail(-EIO, "one-burst server: %s, and its stream does not drain: %s", why, hipGetErrorString(e));
		return 1;
	};
	const auto t0 = std::chrono::steady_clock::now();
	for (unsigned spins = 0;; spins++) {
		if (*v_done == seq)
			break;
		if (*v_ended == g_one.gen) {
			// the current generation has ended (idle, lifetime; or none was ever started: both numbers 0): the next one
			// finds the request waiting
			if (launches++ >= 4)
				return give_up("ends without answering");
			if ((r = one_server_start(mh))) return r;
		}
		if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5))
			return give_up("no answer within 5 s");
#if defined(__x86_64__)
		__builtin_ia32_pause();
#endif
	}
	std::atomic_thread_fence(std::memory_order_acquire);
	return 0;
}

}  // namespace

extern "C" {

int gmr1_pi4cxpsk_demod(struct gmr1_pi4cxpsk_burst *burst_type,
                        struct osmo_cxvec *burst_in, int sps, float freq_shift,
                        sbit_t *ebits, int *sync_id_p, float *toa_p, float *freq_err_p)
{
	if (!burst_type || !burst_in || !burst_in->data || !ebits)
		return fail(-EINVAL, "gmr1_pi4cxpsk_demod: NULL argument");
	int r = host_types();
	if (r) return r;
	int type = -1;
	for (int i = 0; i < GMR1_HIP_N_BURSTS; i++)
		if (burst_type == kBuiltin[i])
			type = i;
	DevBurst custom;
	const DevBurst *cp = nullptr;
	if (type < 0) {
		gmr1_hip_burst_flat f;
		r = flatten(burst_type, &f, "custom");
		if (r == 0) r = to_dev(f, &custom);
		if (r) return fail(r, "gmr1_pi4cxpsk_demod: unsupported burst description");
		type = kCustomSlot;
		cp = &custom;
	}
	const DevBurst &ht = cp ? custom : g_host_types[type];
	if (!cp && burst_in->len >= 1 && burst_in->len <= kMaxInLen) {
		std::lock_guard<std::mutex> lk(g_one.mu);
		bool usable = false;
		r = one_ready(&usable);
		if (r) return r;
		if (usable) {
			const int in_len = burst_in->len;
			std::memcpy(one_h<unsigned char>(kOneIq), burst_in->data, (size_t)in_len * 8);
			*one_h<uint64_t>(kOneOff) = 0;
			*one_h<float>(kOneOff + 12) = freq_shift;
			int32_t *o = one_h<int32_t>(kOneOut);
			float *of = one_h<float>(kOneOut);
			g_one.memo = false;
			// the fused kernel takes the two formats of rx_bcch / rx_ccch at the window lengths they use (gmr1_rx.c:759, 809)
			const int kind = type == GMR1_HIP_BCCH ? 0 : (type == GMR1_HIP_DC6 ? 1 : -1);
			const bool fused = kind >= 0 && sps >= 4 && sps <= 8 && in_len == window_len(234, sps, (kind ? 10 : 20) * sps);
			bool served = false;
			if (fused && sps == 4) {
				*one_h<uint8_t>(kOneOff + 8) = (uint8_t)kind;
				r = one_server_call();
				if (r < 0) return r;
				served = r == 0;
				r = 0;
				o[1] = 0;
			}
			if (served) {
				// (answered by the resident server)
			} else if (fused) {
				*one_h<uint8_t>(kOneOff + 8) = (uint8_t)kind;
				r = rx_bcch_ccch_dev_impl(g_one.st, 1, sps, one_d<float>(kOneIq), one_d<uint64_t>(kOneOff),
				                          one_d<uint8_t>(kOneOff + 8), one_d<float>(kOneOff + 12), one_d<uint8_t>(kOneOut + 32),
				                          one_d<int32_t>(kOneOut + 16), one_d<int32_t>(kOneOut + 20), one_d<float>(kOneOut + 8),
				                          one_d<float>(kOneOut + 12), nullptr, one_d<int8_t>(kOneEb), nullptr,
				                          one_d<int32_t>(kOneOut));
				o[1] = 0;                                      // one training sequence: sync_id 0 when found
			} else {
				r = demod_dev_impl(g_one.st, type, ht, 1, sps, in_len, one_d<float>(kOneIq), one_d<uint64_t>(kOneOff),
				                   one_d<float>(kOneOff + 12), one_d<int8_t>(kOneEb), ht.ebits, one_d<int32_t>(kOneOut + 4),
				                   one_d<float>(kOneOut + 8), one_d<float>(kOneOut + 12), nullptr, one_d<int32_t>(kOneOut));
			}
			if (r) return r;
			if (!served)
				HIP_TRY(hipStreamSynchronize(g_one.st));
			if (o[0]) return o[0];
			std::memcpy(ebits, one_h<int8_t>(kOneEb), (size_t)ht.ebits);
			if (sync_id_p) *sync_id_p = o[1];
			if (toa_p) *toa_p = of[2];
			if (freq_err_p) *freq_err_p = of[3];
			if (fused) {
				g_one.memo = true;
				g_one.memo_chain = kind ? kChainCcch : kChainBcch;
				g_one.memo_acc = conv_acc();
				g_one.memo_n = ht.ebits;
				std::memcpy(g_one.memo_eb, one_h<int8_t>(kOneEb), (size_t)ht.ebits);
				std::memcpy(g_one.memo_l2, one_h<uint8_t>(kOneOut + 32), 24);
				g_one.memo_crc = o[4];
				g_one.memo_conv = o[5];
			}
			return 0;
		}
	}
	const uint64_t off = 0;
	int32_t rv = 0, sid = -1;
	float toa = 0.f, fe = 0.f;
	r = demod_host_impl(type, ht, cp, 1, sps, burst_in->len, reinterpret_cast<const float *>(burst_in->data),
	                    (uint64_t)burst_in->len, &off, &freq_shift, reinterpret_cast<int8_t *>(ebits), ht.ebits,
	                    &sid, &toa, &fe, nullptr, &rv);
	if (r) return r;
	if (rv) return rv;
	if (sync_id_p) *sync_id_p = sid;
	if (toa_p) *toa_p = toa;
	if (freq_err_p) *freq_err_p = fe;
	return 0;
}

// ---------------------------------------------------------------------------
// layer 1
// ---------------------------------------------------------------------------
static int l1_dev(hipStream_t st, int chain, int n, const int8_t *ebits, uint8_t *l2, int32_t *crc, int32_t *conv)
{
	if (n < 0 || !ebits || !l2 || !crc || !conv)
		return fail(-EINVAL, "l1 decode: NULL argument");
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	L1Args a;
	a.n = n; a.chain = chain; a.conv_acc = conv_acc(); a.ebits = ebits; a.l2 = l2; a.crc = crc; a.conv = conv;
	HIP_TRY(launch_l1(a, st));
	return 0;
}

static int l1_host(int chain, int n, const int8_t *ebits, uint8_t *l2, int32_t *crc, int32_t *conv)
{
	const int neb = chain == kChainCcch ? 432 : 424;
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n <= 0) return 0;
	DBuf d_eb, d_l2, d_crc, d_conv;
	HIP_TRY(d_eb.alloc((size_t)n * neb));
	HIP_TRY(d_l2.alloc((size_t)n * 24));
	HIP_TRY(d_crc.alloc((size_t)n * 4));
	HIP_TRY(d_conv.alloc((size_t)n * 4));
	HIP_TRY(hipMemcpy(d_eb.p, ebits, (size_t)n * neb, hipMemcpyHostToDevice));
	r = l1_dev(nullptr, chain, n, d_eb.as<int8_t>(), d_l2.as<uint8_t>(), d_crc.as<int32_t>(), d_conv.as<int32_t>());
	if (r) return r;
	HIP_TRY(hipStreamSynchronize(nullptr));
	HIP_TRY(hipMemcpy(l2, d_l2.p, (size_t)n * 24, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(crc, d_crc.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(conv, d_conv.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	return 0;
}

int gmr1_hip_bcch_decode_batch_dev(void *stream, int n, const int8_t *ebits, uint8_t *l2, int32_t *crc, int32_t *conv)
{
	return l1_dev((hipStream_t)stream, kChainBcch, n, ebits, l2, crc, conv);
}

int gmr1_hip_ccch_decode_batch_dev(void *stream, int n, const int8_t *ebits, uint8_t *l2, int32_t *crc, int32_t *conv)
{
	return l1_dev((hipStream_t)stream, kChainCcch, n, ebits, l2, crc, conv);
}

int gmr1_hip_bcch_decode_batch(int n, const int8_t *ebits, uint8_t *l2, int32_t *crc, int32_t *conv)
{
	return l1_host(kChainBcch, n, ebits, l2, crc, conv);
}

int gmr1_hip_ccch_decode_batch(int n, const int8_t *ebits, uint8_t *l2, int32_t *crc, int32_t *conv)
{
	return l1_host(kChainCcch, n, ebits, l2, crc, conv);
}

// reference-compatible single-burst decoders (bcch.h:38, ccch.h:38).  A device
// failure cannot be reported through the reference's "crc result" return value
// without being mistaken for a CRC verdict, so it is returned as -errno (< 0).
static int decode_one(int chain, uint8_t *l2, const sbit_t *bits_e, int *conv_rv)
{
	if (!l2 || !bits_e)
		return fail(-EINVAL, "decode: NULL argument");
	const int neb = chain == kChainCcch ? 432 : 424;
	int32_t crc = 0, conv = 0;
	{
		std::lock_guard<std::mutex> lk(g_one.mu);
		bool usable = false;
		int r = one_ready(&usable);
		if (r) return r;
		if (usable) {
			if (g_one.memo && g_one.memo_chain == chain && g_one.memo_n == neb && g_one.memo_acc == conv_acc() &&
			    !std::memcmp(g_one.memo_eb, bits_e, (size_t)neb)) {
				// these very soft bits were decoded by the demodulator call that produced them
				std::memcpy(l2, g_one.memo_l2, 24);
				if (conv_rv) *conv_rv = g_one.memo_conv;
				return g_one.memo_crc;
			}
			std::memcpy(one_h<int8_t>(kOneEb), bits_e, (size_t)neb);
			r = l1_dev(g_one.st, chain, 1, one_d<int8_t>(kOneEb), one_d<uint8_t>(kOneOut + 32), one_d<int32_t>(kOneOut + 16),
			           one_d<int32_t>(kOneOut + 20));
			if (r) return r;
			HIP_TRY(hipStreamSynchronize(g_one.st));
			std::memcpy(l2, one_h<uint8_t>(kOneOut + 32), 24);
			if (conv_rv) *conv_rv = one_h<int32_t>(kOneOut)[5];
			return one_h<int32_t>(kOneOut)[4];
		}
	}
	int r = l1_host(chain, 1, reinterpret_cast<const int8_t *>(bits_e), l2, &crc, &conv);
	if (r) return r;
	if (conv_rv) *conv_rv = conv;
	return crc;
}

int gmr1_bcch_decode(uint8_t *l2, const sbit_t *bits_e, int *conv_rv) { return decode_one(kChainBcch, l2, bits_e, conv_rv); }

int gmr1_ccch_decode(uint8_t *l2, const sbit_t *bits_e, int *conv_rv) { return decode_one(kChainCcch, l2, bits_e, conv_rv); }

// ---------------------------------------------------------------------------
// fused BCCH / CCCH receive
// ---------------------------------------------------------------------------
}  // extern "C" (closed for the shared implementation below)

namespace gmr1 {
namespace {
// what every burst of the fused BCCH / CCCH path shares
int rx_base_args(int sps, const float *iq, RxArgs *out, int min_sps = 4)
{
	if (sps < min_sps || sps > 16)
		return fail(-EINVAL, "rx_bcch_ccch: sps=%d unsupported (%d..16)", sps, min_sps);
	RxArgs a;
	std::memset(&a, 0, sizeof(a));
	a.sps = sps;
	a.conv_acc = conv_acc();
	a.in_len[0] = window_len(234, sps, 20 * sps);   // gmr1_rx.c:759
	a.in_len[1] = window_len(234, sps, 10 * sps);   // gmr1_rx.c:809
	a.fixed_type = -1;
	a.ebits_stride = 432;
	a.ssyms_stride = 234;
	a.iq = reinterpret_cast<const float2 *>(iq);
	if (a.in_len[0] > kMaxInLen)
		return fail(-EINVAL, "rx_bcch_ccch: window too long");
	// samples of the sync-chunk windows: sum over chunks of len*sps + w - 1
	const int ty[2] = {GMR1_HIP_BCCH, GMR1_HIP_DC6};
	a.stage_samples = 0;
	// the fused kernels unroll the sync correlation for these two formats (corr_fixed, rx_kernels.hip)
	// ... and carry both formats' geometry and training symbols as constants (Fmt<false>, rx_kernels.hip; nb.c:36-62, 94-120)
	static const int kTaps[2][3] = {{11, 3, 3}, {7, 3, 3}};
	static const int kPos[3] = {28, 119, 197};
	static const uint8_t kSyms[2][17] = {{0, 2, 2, 0, 0, 0, 2, 0, 2, 2, 2, 2, 2, 0, 2, 2, 0}, {0, 0, 0, 2, 2, 0, 2, 0, 3, 0, 3, 1, 1}};
	for (int k = 0; k < 2; k++) {
		const DevBurst &bt = g_host_types[ty[k]];
		bool ok = bt.n_sync == 1 && bt.n_chunks[0] == 3 && bt.len == 234 && bt.nbits == 2 && bt.rotation == (float)M_PI / 4.0f &&
		          bt.sync_tl[0] == kTaps[k][0] + 6;
		for (int c = 0, n = 0; ok && c < 3; c++) {
			ok = bt.sync[0][c].len == kTaps[k][c] && bt.sync[0][c].pos == kPos[c];
			for (int j = 0; ok && j < kTaps[k][c]; j++, n++)
				ok = bt.sync[0][c].syms[j] == kSyms[k][n];
		}
		if (!ok)
			return fail(-EINVAL, "rx_bcch_ccch: burst table %d does not have the training layout the kernel is built for", ty[k]);
		const int w = a.in_len[k] - bt.len * sps + 1;
		int tot = 0;
		for (int c = 0; c < bt.n_chunks[0]; c++)
			tot += bt.sync[0][c].len * sps + w - 1;
		if (tot > a.stage_samples) a.stage_samples = tot;
	}
	*out = a;
	return 0;
}
}  // namespace

int rx_fused_base_args(int sps, const float *iq, RxArgs *out) { return rx_base_args(sps, iq, out, 4); }

int rx_bcch_ccch_dev_impl(hipStream_t stream, int n, int sps,
                          const float *iq, const uint64_t *offset, const uint8_t *kind,
                          const float *freq_shift,
                          uint8_t *l2, int32_t *crc, int32_t *conv,
                          float *toa, float *freq_err, float *energy,
                          int8_t *ebits, float *ssyms, int32_t *rv, long long plane_stride)
{
	if (n < 0 || !iq || !offset || !kind || !l2 || !crc || !conv || !rv)
		return fail(-EINVAL, "rx_bcch_ccch: iq/offset/kind/l2/crc/conv/rv are required");
	if (plane_stride < 0 || (plane_stride && (sps != 4 || energy)))
		return fail(-EINVAL, "rx_bcch_ccch: the polyphase-planar sample layout exists at 4 samples per symbol (sps=%d)", sps);
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	RxArgs a;
	r = rx_base_args(sps, iq, &a, 1);
	if (r) return r;
	a.n = n;
	a.plane_stride = plane_stride;
	a.dbg_stop = dbg_stop_env();
	{
		static int impl = -1;
		if (impl < 0) {
			const char *e = profile_env("GMR1_HIP_RX_IMPL");
			impl = e ? atoi(e) : 0;
		}
		a.impl = impl;
		// below four samples per symbol the reference delays the burst by a fraction of a sample with a 21-tap sinc
		// (pi4cxpsk.c:298-343) instead of picking samples: the one-burst-at-a-time body has that branch, the row-batched one not
		if (sps < 4)
			a.impl = 1;
		if (plane_stride)
			a.impl = 0;
	}
	a.offset = offset; a.kind = kind; a.freq_shift = freq_shift;
	a.l2 = l2; a.crc = crc; a.conv = conv; a.toa = toa; a.freq_err = freq_err;
	a.ebits = ebits; a.ssyms = ssyms; a.rv = rv;
	a.energy = energy;
	HIP_TRY(launch_rx(a, true, a.in_len[0], stream));
	return 0;
}

// process_bcch of n_chains chains (launch_rx_loop: k_rx_chain, k_rx4, k_rx_merge); every pointer in `la` is device memory
int rx_loop_dev_impl(hipStream_t stream, int n_chains, int sps, const float *iq, const RxLoopArgs &la)
{
	if (n_chains < 0 || !iq || !la.state || !la.rec || !la.n_rounds || !la.n_rec || !la.n_frames || la.max_rounds < 1 ||
	    la.rec_stride < 1 || (la.rec_frame && !la.rec_minen) || (la.flog && la.flog_stride < 1) || !la.rounds || !la.n_ccch || !la.fin || !la.slice_end ||
	    la.c_stride < 4 || (la.c_stride & 3) || !la.c_off || !la.c_fs || !la.c_kind || !la.c_meta || !la.c_l2 || !la.c_crc ||
	    !la.c_conv || !la.c_rv || !la.c_en)
		return fail(-EINVAL, "rx_loop: bad arguments");
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	RxArgs a;
	r = rx_base_args(sps, iq, &a, 1);
	if (r) return r;
	HIP_TRY(launch_rx_loop(a, la, n_chains, stream));
	return 0;
}
}  // namespace gmr1

extern "C" {

int gmr1_hip_rx_bcch_ccch_batch_dev(void *stream, int n, int sps,
                                    const float *iq, const uint64_t *offset, const uint8_t *kind,
                                    const float *freq_shift,
                                    uint8_t *l2, int32_t *crc, int32_t *conv,
                                    float *toa, float *freq_err,
                                    int8_t *ebits, float *ssyms, int32_t *rv)
{
	return rx_bcch_ccch_dev_impl((hipStream_t)stream, n, sps, iq, offset, kind, freq_shift, l2, crc, conv,
	                             toa, freq_err, nullptr, ebits, ssyms, rv, 0);
}

int gmr1_hip_rx_bcch_ccch_batch_planar_dev(void *stream, int n, int sps,
                                           const float *iq_planes, uint64_t plane_stride,
                                           const uint64_t *offset, const uint8_t *kind,
                                           const float *freq_shift,
                                           uint8_t *l2, int32_t *crc, int32_t *conv,
                                           float *toa, float *freq_err,
                                           int8_t *ebits, float *ssyms, int32_t *rv)
{
	if (plane_stride == 0 || plane_stride > (uint64_t)1 << 40)
		return fail(-EINVAL, "rx_bcch_ccch planar: plane_stride is required");
	return rx_bcch_ccch_dev_impl((hipStream_t)stream, n, sps, iq_planes, offset, kind, freq_shift, l2, crc, conv,
	                             toa, freq_err, nullptr, ebits, ssyms, rv, (long long)plane_stride);
}

int gmr1_hip_iq_to_planar_dev(void *stream, int sps, uint64_t n_samples, const float *iq,
                              float *iq_planes, uint64_t plane_stride)
{
	if (sps < 1 || sps > 16 || !iq || !iq_planes || plane_stride < (n_samples + (uint64_t)sps - 1) / (uint64_t)sps)
		return fail(-EINVAL, "iq_to_planar: sps 1..16, both arrays, plane_stride >= ceil(n_samples / sps)");
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	HIP_TRY(launch_to_planar(reinterpret_cast<const float2 *>(iq), reinterpret_cast<float2 *>(iq_planes), n_samples, sps,
	                         (long long)plane_stride, (hipStream_t)stream));
	return 0;
}

int gmr1_hip_rx_bcch_ccch_batch(int n, int sps,
                                const float *iq, uint64_t iq_len, const uint64_t *offset, const uint8_t *kind,
                                const float *freq_shift,
                                uint8_t *l2, int32_t *crc, int32_t *conv,
                                float *toa, float *freq_err,
                                int8_t *ebits, float *ssyms, int32_t *rv)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n <= 0) return 0;
	if (!iq || !offset || !kind || !l2 || !crc || !conv || !rv)
		return fail(-EINVAL, "rx_bcch_ccch: iq/offset/kind/l2/crc/conv/rv are required");
	for (int i = 0; i < n; i++) {
		const uint64_t len = (uint64_t)window_len(234, sps, (kind[i] ? 10 : 20) * sps);
		if (offset[i] + len > iq_len)
			return fail(-EINVAL, "burst %d runs past the end of iq", i);
	}
	DBuf d_iq, d_off, d_kind, d_fs, d_l2, d_crc, d_conv, d_toa, d_fe, d_eb, d_ss, d_rv;
	HIP_TRY(d_iq.alloc(iq_len * 8));
	HIP_TRY(d_off.alloc((size_t)n * 8));
	HIP_TRY(d_kind.alloc((size_t)n));
	HIP_TRY(d_l2.alloc((size_t)n * 24));
	HIP_TRY(d_crc.alloc((size_t)n * 4));
	HIP_TRY(d_conv.alloc((size_t)n * 4));
	HIP_TRY(d_toa.alloc((size_t)n * 4));
	HIP_TRY(d_fe.alloc((size_t)n * 4));
	HIP_TRY(d_rv.alloc((size_t)n * 4));
	if (freq_shift) HIP_TRY(d_fs.alloc((size_t)n * 4));
	if (ebits) HIP_TRY(d_eb.alloc((size_t)n * 432));
	if (ssyms) HIP_TRY(d_ss.alloc((size_t)n * 234 * 4));
	HIP_TRY(hipMemcpy(d_iq.p, iq, iq_len * 8, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(d_off.p, offset, (size_t)n * 8, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(d_kind.p, kind, (size_t)n, hipMemcpyHostToDevice));
	if (freq_shift) HIP_TRY(hipMemcpy(d_fs.p, freq_shift, (size_t)n * 4, hipMemcpyHostToDevice));
	r = gmr1_hip_rx_bcch_ccch_batch_dev(nullptr, n, sps, d_iq.as<float>(), d_off.as<uint64_t>(), d_kind.as<uint8_t>(),
	                                    freq_shift ? d_fs.as<float>() : nullptr,
	                                    d_l2.as<uint8_t>(), d_crc.as<int32_t>(), d_conv.as<int32_t>(),
	                                    d_toa.as<float>(), d_fe.as<float>(),
	                                    ebits ? d_eb.as<int8_t>() : nullptr, ssyms ? d_ss.as<float>() : nullptr,
	                                    d_rv.as<int32_t>());
	if (r) return r;
	HIP_TRY(hipStreamSynchronize(nullptr));
	HIP_TRY(hipMemcpy(l2, d_l2.p, (size_t)n * 24, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(crc, d_crc.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(conv, d_conv.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(rv, d_rv.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	if (toa) HIP_TRY(hipMemcpy(toa, d_toa.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	if (freq_err) HIP_TRY(hipMemcpy(freq_err, d_fe.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	if (ebits) HIP_TRY(hipMemcpy(ebits, d_eb.p, (size_t)n * 432, hipMemcpyDeviceToHost));
	if (ssyms) HIP_TRY(hipMemcpy(ssyms, d_ss.p, (size_t)n * 234 * 4, hipMemcpyDeviceToHost));
	return 0;
}

}  // extern "C"
