// capi_detect.cpp -- C-ABI entry points of burst-type detection and modulation-order estimation.
#include "capi_common.h"

#include <mutex>
#include <vector>

using namespace gmr1;

extern "C" {

// descriptor-table slots of caller-defined candidate types (gmr1_pi4cxpsk_detect); kCustomSlot stays the demodulator's
static constexpr int kDetectSlot0 = kCustomSlot - 4;
static_assert(kDetectSlot0 >= GMR1_HIP_N_BURSTS, "descriptor table too small");

// hts[i]: host copy of candidate i; slots[i]: its descriptor-table slot when it is built in, -1 when it is the caller's own
// (customs[i]): those are uploaded into the four spare slots, four candidates per launch.  Any number of candidates, as in
// the reference (a NULL-terminated list, pi4cxpsk.c:617-682): lists of more than four run as several launches that hand
// the best candidate so far on through the outputs and a scratch array of powers.  `scratch`: 4 x n words of device memory
// for outputs the caller did not ask for (only needed for lists of more than four).
static int detect_dev_impl(hipStream_t stream, int n_types, const int *slots, const DevBurst *const *hts,
                           const DevBurst *const *customs, int n, int sps,
                           int in_len, const float *iq, const uint64_t *offset, const float *freq_shift,
                           const float *e_toa, int32_t *bt_id, int32_t *sync_id, float *toa, int32_t *rv)
{
	if (n < 0 || !slots || !iq || !offset || !rv)
		return fail(-EINVAL, "detect: NULL argument");
	if (n_types < 1)
		return fail(-EINVAL, "detect: no candidate burst types");
	if (sps < 1 || sps > 16)
		return fail(-EINVAL, "detect: sps=%d out of range (1..16)", sps);
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	int max_lags = 0;
	for (int i = 0; i < n_types; i++) {
		const DevBurst &ht = *hts[i];
		const int w = in_len - ht.len * sps + 1;
		if (w < 1 || in_len > kMaxInLen)
			return fail(-EINVAL, "detect: window of %d samples (<= %d supported) gives %d lags", in_len, kMaxInLen, w);
		if (w > max_lags) max_lags = w;
	}
	DetectArgs a;
	std::memset(&a, 0, sizeof(a));
	a.n = n; a.sps = sps; a.in_len = in_len;
	a.max_lags = max_lags;
	a.rot0 = hts[0]->rotation;
	a.iq = reinterpret_cast<const float2 *>(iq);
	a.offset = offset; a.freq_shift = freq_shift; a.e_toa = e_toa;
	a.bt_id = bt_id; a.sync_id = sync_id; a.toa = toa; a.rv = rv;
	WsLease lease;
	if (n_types > 4) {
		// the carry needs every output: the library's workspace stands in for the ones the caller left out
		void *ws = nullptr;
		if ((r = lease.acquire(s, stream))) return r;
		r = dev_workspace(s, (size_t)(n > 0 ? n : 1) * 16, &ws);
		if (r) return r;
		int32_t *w32 = static_cast<int32_t *>(ws);
		a.best_pwr = reinterpret_cast<float *>(w32);
		if (!a.bt_id) a.bt_id = w32 + n;
		if (!a.sync_id) a.sync_id = w32 + 2 * (size_t)n;
		if (!a.toa) a.toa = reinterpret_cast<float *>(w32 + 3 * (size_t)n);
	}
	for (int c0 = 0; c0 < n_types; c0 += 4) {
		a.n_types = n_types - c0 < 4 ? n_types - c0 : 4;
		a.first = c0;
		a.carry = c0 > 0;
		for (int j = 0; j < a.n_types; j++) {
			const int i = c0 + j;
			if (customs && customs[i]) {
				a.types[j] = kDetectSlot0 + j;
				HIP_TRY(upload_types(customs[i], a.types[j], 1, stream));     // stream-ordered behind the launch before
			} else {
				a.types[j] = slots[i];
			}
		}
		HIP_TRY(launch_detect(a, stream));
	}
	return 0;
}

static int builtin_slots(int n_types, const int *burst_ids, std::vector<const DevBurst *> &hts)
{
	if (!burst_ids || n_types < 1)
		return fail(-EINVAL, "detect: no candidate burst types");
	int r = host_types();
	if (r) return r;
	hts.resize((size_t)n_types);
	for (int i = 0; i < n_types; i++) {
		if (burst_ids[i] < 0 || burst_ids[i] >= GMR1_HIP_N_BURSTS)
			return fail(-EINVAL, "detect: bad burst id %d", burst_ids[i]);
		hts[i] = &g_host_types[burst_ids[i]];
	}
	return 0;
}

int gmr1_hip_detect_batch_dev(void *stream, int n_types, const int *burst_ids, int n, int sps, int in_len,
                              const float *iq, const uint64_t *offset, const float *freq_shift,
                              const float *e_toa, int32_t *bt_id, int32_t *sync_id, float *toa, int32_t *rv)
{
	std::vector<const DevBurst *> hts;
	int r = builtin_slots(n_types, burst_ids, hts);
	if (r) return r;
	return detect_dev_impl((hipStream_t)stream, n_types, burst_ids, hts.data(), nullptr, n, sps, in_len, iq, offset, freq_shift,
	                       e_toa, bt_id, sync_id, toa, rv);
}

// host-pointer staging; customs[i] != nullptr: candidate i is caller-defined
static int detect_host_impl(int n_types, const int *slots, const DevBurst *const *hts, const DevBurst *const *customs,
                            int n, int sps, int in_len, const float *iq, uint64_t iq_len, const uint64_t *offset,
                            const float *freq_shift, const float *e_toa, int32_t *bt_id, int32_t *sync_id, float *toa,
                            int32_t *rv)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n <= 0) return 0;
	if (!iq || !offset || !rv)
		return fail(-EINVAL, "detect: NULL argument");
	for (int i = 0; i < n; i++)
		if (offset[i] + (uint64_t)in_len > iq_len)
			return fail(-EINVAL, "burst %d runs past the end of iq", i);
	DBuf d_iq, d_off, d_fs, d_et, d_bt, d_sid, d_toa, d_rv;
	HIP_TRY(d_iq.alloc(iq_len * 8));
	HIP_TRY(d_off.alloc((size_t)n * 8));
	HIP_TRY(d_bt.alloc((size_t)n * 4));
	HIP_TRY(d_sid.alloc((size_t)n * 4));
	HIP_TRY(d_toa.alloc((size_t)n * 4));
	HIP_TRY(d_rv.alloc((size_t)n * 4));
	HIP_TRY(hipMemcpy(d_iq.p, iq, iq_len * 8, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(d_off.p, offset, (size_t)n * 8, hipMemcpyHostToDevice));
	if (freq_shift) {
		HIP_TRY(d_fs.alloc((size_t)n * 4));
		HIP_TRY(hipMemcpy(d_fs.p, freq_shift, (size_t)n * 4, hipMemcpyHostToDevice));
	}
	if (e_toa) {
		HIP_TRY(d_et.alloc((size_t)n * 4));
		HIP_TRY(hipMemcpy(d_et.p, e_toa, (size_t)n * 4, hipMemcpyHostToDevice));
	}
	// caller-defined descriptors share the spare table slots process-wide: uploads, launches and completion under one lock
	bool any_custom = false;
	for (int i = 0; i < n_types; i++)
		any_custom |= customs && customs[i];
	std::unique_lock<std::mutex> lk(custom_slots_mutex(), std::defer_lock);
	if (any_custom)
		lk.lock();
	r = detect_dev_impl(nullptr, n_types, slots, hts, customs, n, sps, in_len, d_iq.as<float>(), d_off.as<uint64_t>(),
	                    freq_shift ? d_fs.as<float>() : nullptr, e_toa ? d_et.as<float>() : nullptr,
	                    d_bt.as<int32_t>(), d_sid.as<int32_t>(), d_toa.as<float>(), d_rv.as<int32_t>());
	if (r) return r;
	HIP_TRY(hipStreamSynchronize(nullptr));
	if (lk.owns_lock())
		lk.unlock();
	HIP_TRY(hipMemcpy(rv, d_rv.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	if (bt_id) HIP_TRY(hipMemcpy(bt_id, d_bt.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	if (sync_id) HIP_TRY(hipMemcpy(sync_id, d_sid.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	if (toa) HIP_TRY(hipMemcpy(toa, d_toa.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	return 0;
}

int gmr1_hip_detect_batch(int n_types, const int *burst_ids, int n, int sps, int in_len,
                          const float *iq, uint64_t iq_len, const uint64_t *offset, const float *freq_shift,
                          const float *e_toa, int32_t *bt_id, int32_t *sync_id, float *toa, int32_t *rv)
{
	std::vector<const DevBurst *> hts;
	int r = builtin_slots(n_types, burst_ids, hts);
	if (r) return r;
	return detect_host_impl(n_types, burst_ids, hts.data(), nullptr, n, sps, in_len, iq, iq_len, offset, freq_shift, e_toa,
	                        bt_id, sync_id, toa, rv);
}

int gmr1_hip_mod_order_batch_dev(void *stream, int n, int sps, int in_len,
                                 const float *iq, const uint64_t *offset, const float *freq_shift, int32_t *order)
{
	if (n < 0 || !iq || !offset || !order)
		return fail(-EINVAL, "mod_order: NULL argument");
	if (sps < 1 || sps > 16 || in_len < 1 || in_len > kMaxInLen)
		return fail(-EINVAL, "mod_order: sps / window out of range");
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	ModOrderArgs a;
	std::memset(&a, 0, sizeof(a));
	a.n = n; a.sps = sps; a.in_len = in_len;
	a.iq = reinterpret_cast<const float2 *>(iq);
	a.offset = offset; a.freq_shift = freq_shift; a.order = order;
	HIP_TRY(launch_mod_order(a, (hipStream_t)stream));
	return 0;
}

int gmr1_hip_mod_order_batch(int n, int sps, int in_len, const float *iq, uint64_t iq_len,
                             const uint64_t *offset, const float *freq_shift, int32_t *order)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n <= 0) return 0;
	if (!iq || !offset || !order)
		return fail(-EINVAL, "mod_order: NULL argument");
	for (int i = 0; i < n; i++)
		if (offset[i] + (uint64_t)in_len > iq_len)
			return fail(-EINVAL, "burst %d runs past the end of iq", i);
	DBuf d_iq, d_off, d_fs, d_o;
	HIP_TRY(d_iq.alloc(iq_len * 8));
	HIP_TRY(d_off.alloc((size_t)n * 8));
	HIP_TRY(d_o.alloc((size_t)n * 4));
	HIP_TRY(hipMemcpy(d_iq.p, iq, iq_len * 8, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(d_off.p, offset, (size_t)n * 8, hipMemcpyHostToDevice));
	if (freq_shift) {
		HIP_TRY(d_fs.alloc((size_t)n * 4));
		HIP_TRY(hipMemcpy(d_fs.p, freq_shift, (size_t)n * 4, hipMemcpyHostToDevice));
	}
	r = gmr1_hip_mod_order_batch_dev(nullptr, n, sps, in_len, d_iq.as<float>(), d_off.as<uint64_t>(),
	                                 freq_shift ? d_fs.as<float>() : nullptr, d_o.as<int32_t>());
	if (r) return r;
	HIP_TRY(hipStreamSynchronize(nullptr));
	HIP_TRY(hipMemcpy(order, d_o.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	return 0;
}

// ---- reference-compatible single calls (pi4cxpsk.h:107-113) -------------------------------
int gmr1_pi4cxpsk_detect(struct gmr1_pi4cxpsk_burst **burst_types, float e_toa,
                         struct osmo_cxvec *burst_in, int sps, float freq_shift,
                         int *bt_id_p, int *sync_id_p, float *toa_p)
{
	if (!burst_types || !burst_types[0] || !burst_in || !burst_in->data)
		return fail(-EINVAL, "gmr1_pi4cxpsk_detect: NULL argument");
	int r = host_types();
	if (r) return r;
	// any NULL-terminated list of burst descriptions, as in the reference (pi4cxpsk.c:617-682): the built-in ones by
	// their table slot, caller-defined ones flattened and uploaded into spare slots for the call; the index returned
	// in *bt_id_p is the position in the caller's list either way
	int nt = 0;
	while (burst_types[nt])
		nt++;
	std::vector<int> slots((size_t)nt);
	std::vector<DevBurst> custom((size_t)nt);
	std::vector<const DevBurst *> hts((size_t)nt), cps((size_t)nt, nullptr);
	for (int k = 0; k < nt; k++) {
		int id = -1;
		for (int i = 0; i < GMR1_HIP_N_BURSTS; i++)
			if (burst_types[k] == kBuiltin[i])
				id = i;
		if (id < 0) {
			gmr1_hip_burst_flat f;
			r = flatten(burst_types[k], &f, "custom");
			if (r == 0) r = to_dev(f, &custom[k]);
			if (r) return fail(r, "gmr1_pi4cxpsk_detect: unsupported burst description (candidate %d)", k);
			cps[k] = &custom[k];
		}
		slots[k] = id;
		hts[k] = cps[k] ? &custom[k] : &g_host_types[id];
	}
	const uint64_t off = 0;
	int32_t bt = -1, sid = -1, rv = 0;
	float toa = 0.f;
	r = detect_host_impl(nt, slots.data(), hts.data(), cps.data(), 1, sps, burst_in->len, reinterpret_cast<const float *>(burst_in->data),
	                     (uint64_t)burst_in->len, &off, &freq_shift, &e_toa, &bt, &sid, &toa, &rv);
	if (r) return r;
	if (rv) return rv;
	if (bt_id_p) *bt_id_p = bt;
	if (sync_id_p) *sync_id_p = sid;
	if (toa_p) *toa_p = toa;
	return 0;
}

int gmr1_pi4cxpsk_mod_order(struct osmo_cxvec *burst_in, int sps, float freq_shift)
{
	if (!burst_in || !burst_in->data)
		return fail(-EINVAL, "gmr1_pi4cxpsk_mod_order: NULL argument");
	const uint64_t off = 0;
	int32_t order = 0;
	int r = gmr1_hip_mod_order_batch(1, sps, burst_in->len, reinterpret_cast<const float *>(burst_in->data),
	                                 (uint64_t)burst_in->len, &off, &freq_shift, &order);
	if (r) return r;
	return order;
}

}  // extern "C"
