// capi_chan.cpp -- C ABI of the wideband -> per-ARFCN channelizer (reference utils/gmr1_rx_sdr.py:391-602:
// PFBBase, PFBOutputParameters, PFBOutputBranch, and the GNU Radio blocks they configure).  The filter design (firdes.low_pass, firdes.root_raised_cosine) runs on the host
// in double precision, once per (sample rate, sps); everything per sample runs on the GPU.

#include "capi_common.h"

#include <cmath>
#include <deque>
#include <mutex>
#include <vector>

#include "../../include/gmr1_hip.h"

using namespace gmr1;

namespace {

constexpr double kChanWidth = 31250.0;     // GMR-1 carrier raster (gmr1_rx_sdr.py)
constexpr int kSymRate = 23400;
constexpr int kNfilt = 32;

// gr::filter::firdes::low_pass, Hamming window
std::vector<float> design_low_pass(double gain, double fs, double cutoff, double tw)
{
	int ntaps = (int)(53.0 * fs / (22.0 * tw));
	if ((ntaps & 1) == 0)
		ntaps++;
	const int M = (ntaps - 1) / 2;
	std::vector<double> t(ntaps);
	const double fw = 2.0 * M_PI * cutoff / fs;
	for (int n = -M; n <= M; n++) {
		const double w = 0.54 - 0.46 * std::cos(2.0 * M_PI * (double)(n + M) / (double)(ntaps - 1));
		t[n + M] = (n == 0 ? fw / M_PI : std::sin(n * fw) / (n * M_PI)) * w;
	}
	double fmax = t[M];
	double tail = 0.0;
	for (int n = 1; n <= M; n++)
		tail += t[n + M];
	fmax += 2.0 * tail;
	std::vector<float> out(ntaps);
	for (int i = 0; i < ntaps; i++)
		out[i] = (float)(t[i] * (gain / fmax));
	return out;
}

// gr::filter::firdes::root_raised_cosine
std::vector<float> design_rrc(double gain, double fs, double sym_rate, double alpha, int ntaps)
{
	ntaps |= 1;
	const double spb = fs / sym_rate;
	std::vector<double> t(ntaps, 0.0);
	double scale = 0.0;
	for (int i = 0; i < ntaps; i++) {
		const double xindx = i - ntaps / 2;
		const double x1 = M_PI * xindx / spb;
		double x2 = 4.0 * alpha * xindx / spb;
		double x3 = x2 * x2 - 1.0;
		double num, den;
		if (std::fabs(x3) >= 0.000001) {
			if (i != ntaps / 2)
				num = std::cos((1 + alpha) * x1) + std::sin((1 - alpha) * x1) / (4 * alpha * xindx / spb);
			else
				num = std::cos((1 + alpha) * x1) + (1 - alpha) * M_PI / (4 * alpha);
			den = x3 * M_PI;
		} else {
			if (alpha == 1) {
				t[i] = -1;
				continue;
			}
			x3 = (1 - alpha) * x1;
			x2 = (1 + alpha) * x1;
			num = std::sin(x2) * (1 + alpha) * M_PI - std::cos(x3) * ((1 - alpha) * M_PI * spb) / (4 * alpha * xindx) +
			      std::sin(x3) * spb * spb / (4 * alpha * xindx * xindx);
			den = -32 * M_PI * alpha * alpha * xindx / spb;
		}
		t[i] = 4 * alpha * num / den;
		scale += t[i];
	}
	std::vector<float> out(ntaps);
	for (int i = 0; i < ntaps; i++)
		out[i] = (float)(t[i] * gain / scale);
	return out;
}

long long gcdll(long long a, long long b) { return b ? gcdll(b, a % b) : a; }

struct ChanPlan {
	double samp_rate = 0;
	int sps = 0;
	int device = -1;
	int n_chans = 0, n_blocks = 0, ntaps = 0;
	int tpf = 0, j0 = 0;
	long long num = 0, den = 1;
	float *d_taps = nullptr;
	float2 *d_bank = nullptr;
};

std::mutex g_plan_mu;
std::deque<ChanPlan> g_plans;       // addresses stay valid as plans are added

// PFBBase.__init__ :393-437 and PFBOutputParameters.__init__ :497-531 for width-1 ARFCNs
int get_plan(double samp_rate, int sps, const ChanPlan **out)
{
	int dev = 0;
	HIP_TRY(hipGetDevice(&dev));
	std::lock_guard<std::mutex> lk(g_plan_mu);
	for (const ChanPlan &p : g_plans)
		if (p.samp_rate == samp_rate && p.sps == sps && p.device == dev) {
			*out = &p;
			return 0;
		}
	ChanPlan p;
	p.samp_rate = samp_rate; p.sps = sps; p.device = dev;
	p.n_chans = ((int)std::ceil(samp_rate / kChanWidth) + 1) & ~1;
	const double resamp = (p.n_chans * kChanWidth) / samp_rate;
	if (std::fabs(resamp - 1.0) >= 1e-5)
		return fail(-EINVAL, "channelize: sample rate %.1f is not n_chans x 31250 Hz (the pre-resampler is not built)", samp_rate);
	if (p.n_chans > kPfbMaxChans)
		return fail(-EINVAL, "channelize: %d channels (at most %d)", p.n_chans, kPfbMaxChans);
	const std::vector<float> taps = design_low_pass(1.0, samp_rate, kChanWidth * 0.5, kChanWidth * 0.25);
	p.ntaps = (int)taps.size();
	p.n_blocks = (p.ntaps + p.n_chans - 1) / p.n_chans + 1;
	if (p.n_chans == 64 && p.n_blocks > kPfbMaxBlocks)
		return fail(-EINVAL, "channelize: prototype filter too long (%d taps)", p.ntaps);
	const double chan_rate2 = kChanWidth * 2.0;                    // 2x oversampled channel rate
	const std::vector<float> rrc = design_rrc(32.0, 32.0 * chan_rate2, kSymRate, 0.35,
	                                          (int)(11.0 * 32 * chan_rate2 / kSymRate));
	const int nt = (int)rrc.size();
	p.tpf = (nt + kNfilt - 1) / kNfilt;
	p.j0 = (nt / 2) % kNfilt;
	// phase step nfilt / rate, rate = sym_rate sps / chan_rate2, as a reduced fraction
	long long num = (long long)kNfilt * (long long)chan_rate2, den = (long long)kSymRate * sps;
	const long long g = gcdll(num, den);
	p.num = num / g; p.den = den / g;
	std::vector<float2> bank((size_t)kNfilt * p.tpf, make_float2(0.f, 0.f));
	for (int j = 0; j < kNfilt; j++)
		for (int k = 0; k < p.tpf; k++) {
			const int i = j + k * kNfilt;
			const float b = i < nt ? rrc[i] : 0.0f;
			const float d = (i + 1 < nt) ? (rrc[i + 1] - rrc[i]) : 0.0f;     // first difference, last one 0
			bank[(size_t)j * p.tpf + k] = make_float2(b, d);
		}
	HIP_TRY(hipMalloc(&p.d_taps, taps.size() * sizeof(float)));
	HIP_TRY(hipMalloc(&p.d_bank, bank.size() * sizeof(float2)));
	HIP_TRY(hipMemcpy(p.d_taps, taps.data(), taps.size() * sizeof(float), hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(p.d_bank, bank.data(), bank.size() * sizeof(float2), hipMemcpyHostToDevice));
	g_plans.push_back(p);
	*out = &g_plans.back();
	return 0;
}

}  // namespace

extern "C" {

int gmr1_hip_channelize_plan(double samp_rate, int sps, uint64_t n_in,
                             int32_t *n_chans, uint64_t *n_mid, uint64_t *n_out)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (sps < 1 || sps > 16)
		return fail(-EINVAL, "channelize: sps=%d out of range (1..16)", sps);
	const ChanPlan *p;
	r = get_plan(samp_rate, sps, &p);
	if (r) return r;
	const uint64_t T = n_in / (uint64_t)(p->n_chans / 2);
	if (n_chans) *n_chans = p->n_chans;
	if (n_mid) *n_mid = T;
	if (n_out) {
		const long long v = ((long long)T * kNfilt - p->j0) * p->den;
		*n_out = v > 0 ? (uint64_t)(v / p->num) : 0;
	}
	return 0;
}

int gmr1_hip_channelize_dev(void *stream, double samp_rate, int sps, const float *wide, uint64_t n_in,
                            float rotation, int n_sel, const int32_t *chan_idx,
                            float *out, uint64_t out_stride, uint64_t *n_out_p)
{
	if (!wide || n_sel < 0 || (n_sel && (!chan_idx || !out)))
		return fail(-EINVAL, "channelize: wide / chan_idx / out are required");
	int32_t nch;
	uint64_t T, n_out;
	int r = gmr1_hip_channelize_plan(samp_rate, sps, n_in, &nch, &T, &n_out);
	if (r) return r;
	if (n_out_p) *n_out_p = n_out;
	if (n_sel == 0 || n_out == 0)
		return 0;
	if (out_stride < n_out)
		return fail(-EINVAL, "channelize: out_stride %llu < %llu output samples per channel",
		            (unsigned long long)out_stride, (unsigned long long)n_out);
	const ChanPlan *p;
	r = get_plan(samp_rate, sps, &p);
	if (r) return r;
	std::vector<int32_t> slot(nch, -1);
	for (int i = 0; i < n_sel; i++) {
		if (chan_idx[i] < 0 || chan_idx[i] >= nch)
			return fail(-EINVAL, "channelize: channel index %d outside 0..%d", chan_idx[i], nch - 1);
		if (slot[chan_idx[i]] >= 0)
			return fail(-EINVAL, "channelize: channel %d selected twice", chan_idx[i]);
		slot[chan_idx[i]] = i;
	}
	hipStream_t st = (hipStream_t)stream;
	DevState *s;
	r = dev_state(&s);
	if (r) return r;
	// scratch: slot table + the 2x oversampled channel streams
	const size_t slot_bytes = 2 * kPfbMaxChans * 4;          // slot[n_chans], then sel[n_sel]
	void *ws;
	r = dev_workspace(s, slot_bytes + (size_t)n_sel * T * sizeof(float2), &ws);
	if (r) return r;
	int32_t *d_slot = static_cast<int32_t *>(ws);
	float2 *d_mid = reinterpret_cast<float2 *>(static_cast<char *>(ws) + slot_bytes);
	int32_t *d_sel = d_slot + kPfbMaxChans;
	HIP_TRY(hipMemcpyAsync(d_slot, slot.data(), (size_t)nch * 4, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(d_sel, chan_idx, (size_t)n_sel * 4, hipMemcpyHostToDevice, st));
	HIP_TRY(hipStreamSynchronize(st));      // slot[] is a host temporary
	PfbArgs pa;
	std::memset(&pa, 0, sizeof(pa));
	pa.n_chans = nch; pa.n_blocks = p->n_blocks; pa.ntaps = p->ntaps;
	pa.n_in = (long long)n_in; pa.T = (long long)T; pa.rotation = rotation;
	pa.x = reinterpret_cast<const float2 *>(wide); pa.taps = p->d_taps; pa.slot = d_slot; pa.y = d_mid;
	pa.sel = d_sel; pa.n_sel = n_sel;
	HIP_TRY(launch_pfb(pa, st));
	ResampArgs ra;
	std::memset(&ra, 0, sizeof(ra));
	ra.n_slots = n_sel; ra.nfilt = kNfilt; ra.tpf = p->tpf; ra.j0 = p->j0; ra.num = p->num; ra.den = p->den;
	ra.T = (long long)T; ra.n_out = (long long)n_out; ra.out_stride = (long long)out_stride;
	ra.y = d_mid; ra.bank = p->d_bank; ra.out = reinterpret_cast<float2 *>(out);
	HIP_TRY(launch_resamp(ra, st));
	return 0;
}

int gmr1_hip_channelize(double samp_rate, int sps, const float *wide, uint64_t n_in, float rotation,
                        int n_sel, const int32_t *chan_idx, float *out, uint64_t out_stride, uint64_t *n_out_p)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (!wide || n_sel < 0 || (n_sel && (!chan_idx || !out)))
		return fail(-EINVAL, "channelize: wide / chan_idx / out are required");
	uint64_t n_out;
	r = gmr1_hip_channelize_plan(samp_rate, sps, n_in, nullptr, nullptr, &n_out);
	if (r) return r;
	if (n_out_p) *n_out_p = n_out;
	if (!n_sel || !n_out) return 0;
	if (out_stride < n_out)
		return fail(-EINVAL, "channelize: out_stride too small");
	DBuf d_w, d_o;
	HIP_TRY(d_w.alloc(n_in * 8));
	HIP_TRY(d_o.alloc((size_t)n_sel * out_stride * 8));
	HIP_TRY(hipMemcpy(d_w.p, wide, n_in * 8, hipMemcpyHostToDevice));
	r = gmr1_hip_channelize_dev(nullptr, samp_rate, sps, d_w.as<float>(), n_in, rotation, n_sel, chan_idx,
	                            d_o.as<float>(), out_stride, nullptr);
	if (r) return r;
	HIP_TRY(hipStreamSynchronize(nullptr));
	HIP_TRY(hipMemcpy(out, d_o.p, (size_t)n_sel * out_stride * 8, hipMemcpyDeviceToHost));
	return 0;
}

}  // extern "C"
