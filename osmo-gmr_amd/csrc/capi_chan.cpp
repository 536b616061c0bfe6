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
size_t up_to(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct ChanPlan {
	double samp_rate = 0;
	int sps = 0;
	int device = -1;
	int n_chans = 0, n_blocks = 0, ntaps = 0;
	int tpf = 0, j0 = 0;
	long long num = 0, den = 1;
	float *d_taps = nullptr;
	float2 *d_bank = nullptr;
	// off the 31.25 kHz grid: the 32-phase pre-resampler to n_chans x 31250 Hz in front of the filterbank
	bool pre = false;
	int pre_j0 = 0;
	long long pre_num = 0, pre_den = 1;      // phase step 32 / rate in 1 / 32 input samples, reduced
	float2 *d_pre_bank = nullptr;            // 32 x 30 (tap, derivative tap)
	// samples the filterbank sees for n_in wideband samples
	uint64_t mid_samples(uint64_t n_in) const
	{
		if (!pre)
			return n_in;
		const long long v = ((long long)n_in * kNfilt - pre_j0) * pre_den;
		return v > 0 ? (uint64_t)(v / pre_num) : 0;
	}
};

// The prototype of the pre-resampler (utils/gmr1_rx_sdr.py:453-461: pfb.arb_resampler_ccf(rate, taps=None, flt_size=32)).
// With taps=None GNU Radio designs it itself, for rates >= 1 -- the only case here: n_chans x 31250 >= samp_rate -- with
// its Parks-McClellan routine (pass band to 0.4 of the input rate, stop band from 0.6, 100 dB), which cannot be
// restated without gr-filter.  OWN DESIGN to the same specification by the window method, at 32 x the input rate, gain 32
// (round 5: the Blackman-Harris design of round 4 had its -6 dB point at 0.5 and was only 36 dB down at 0.6 -- found by
// the test that now holds the specification).  oracle/orc_chan.py: pre_resampler_taps.
std::vector<float> design_pre_resampler(int nfilt)
{
	// Kaiser-windowed sinc, 30 taps per phase (what k_resamp<30, ...> holds), beta = 11, -6 dB point at 0.482 of the input
	// rate: within 0.06 dB up to 0.4 and more than 107 dB down from 0.6 (tests/test_oracle_chan.py checks the specification
	// GNU Radio runs its Parks-McClellan design with: 0.1 dB / 100 dB at those edges)
	const double fs = nfilt, cutoff = 0.482, beta = 11.0;
	const int ntaps = 30 * nfilt - 1;
	const int M = (ntaps - 1) / 2;
	auto bessel_i0 = [](double x) {
		double sum = 1.0, term = 1.0;
		const double q = x * x / 4.0;
		for (int k = 1; k < 200; k++) {
			term *= q / ((double)k * (double)k);
			sum += term;
			if (term < 1e-18 * sum)
				break;
		}
		return sum;
	};
	std::vector<double> t(ntaps);
	const double fw = 2.0 * M_PI * cutoff / fs, i0b = bessel_i0(beta);
	double sum = 0.0;
	for (int n = -M; n <= M; n++) {
		const double r = (double)n / (double)M;
		const double w = bessel_i0(beta * std::sqrt(std::max(0.0, 1.0 - r * r))) / i0b;
		t[n + M] = (n == 0 ? fw / M_PI : std::sin(n * fw) / (n * M_PI)) * w;
		sum += t[n + M];
	}
	std::vector<float> out(ntaps);
	for (int i = 0; i < ntaps; i++)
		out[i] = (float)(t[i] * ((double)nfilt / sum));
	return out;
}

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
	if (p.n_chans > kPfbMaxChans)
		return fail(-EINVAL, "channelize: %d channels (at most %d)", p.n_chans, kPfbMaxChans);
	// Off the grid the script resamples the capture to n_chans x 31250 Hz first and designs the filterbank's prototype
	// for ceil(samp_rate / 31250) x 31250 Hz (:413-417; as written that branch reads self.samp_rate before anything
	// sets it and cannot run -- built to its evident intent, the constructor's argument)
	double mid_rate = samp_rate;
	std::vector<float2> pre_bank;
	if (std::fabs(resamp - 1.0) >= 1e-5) {
		if ((double)std::llround(samp_rate) != samp_rate)
			return fail(-EINVAL, "channelize: the sample rate must be a whole number of Hz");
		p.pre = true;
		mid_rate = std::ceil(samp_rate / kChanWidth) * kChanWidth;
		const std::vector<float> pt = design_pre_resampler(kNfilt);
		const int nt = (int)pt.size();
		const int tpf = (nt + kNfilt - 1) / kNfilt;
		if (tpf > 30)
			return fail(-EINVAL, "channelize: pre-resampler of %d taps per phase", tpf);
		p.pre_j0 = (nt / 2) % kNfilt;
		long long num = (long long)kNfilt * std::llround(samp_rate), den = (long long)p.n_chans * (long long)kChanWidth;
		const long long g = gcdll(num, den);
		p.pre_num = num / g; p.pre_den = den / g;
		pre_bank.assign((size_t)kNfilt * 30, make_float2(0.f, 0.f));
		for (int j = 0; j < kNfilt; j++)
			for (int k = 0; k < tpf; k++) {
				const int i = j + k * kNfilt;
				const float b = i < nt ? pt[i] : 0.0f;
				const float d = (i + 1 < nt) ? (pt[i + 1] - pt[i]) : 0.0f;
				pre_bank[(size_t)j * 30 + k] = make_float2(b, d);
			}
	}
	const std::vector<float> taps = design_low_pass(1.0, mid_rate, kChanWidth * 0.5, kChanWidth * 0.25);
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
	if (p.pre) {
		HIP_TRY(hipMalloc(&p.d_pre_bank, pre_bank.size() * sizeof(float2)));
		HIP_TRY(hipMemcpy(p.d_pre_bank, pre_bank.data(), pre_bank.size() * sizeof(float2), hipMemcpyHostToDevice));
	}
	g_plans.push_back(p);
	*out = &g_plans.back();
	return 0;
}

// ---- direct mode (utils/gmr1_rx_sdr.py:605-807) ------------------------------------------------------------------
struct DdcPlan {
	double samp_rate = 0;
	int sps = 0, device = -1;
	int d1 = 1, d2 = 1;
	double resamp = 1.0;
	std::vector<float> taps1, taps2;         // host copies (stage 1 taps are turned per carrier at call time)
	int tpf = 0, j0 = 0;
	int bank_tpf = 30;                       // row length of the bank on the device: the kernel instantiation's 30 or 96 taps
	long long num = 0, den = 1;
	float2 *d_taps2 = nullptr;               // real taps as (t, 0)
	float2 *d_bank = nullptr;                // resampler bank, rows padded to bank_tpf taps
};
std::deque<DdcPlan> g_ddc_plans;

// DirectOutputParameters._factor / _score :637-650
std::vector<int> ddc_factor(int decim)
{
	const int d_ideal = (int)std::lround(std::sqrt((double)decim));
	for (int i = d_ideal; i > 1; i--)
		if (decim % i == 0)
			return {decim / i, i};
	return {decim};
}
double ddc_score(const std::vector<int> &f)
{
	if (f.size() == 1)
		return f[0];
	return ((double)f[0] * f[0] * f[1]) / (1.0 + (double)f[0] / f[1]);
}

int get_ddc_plan(double samp_rate, int sps, const DdcPlan **out)
{
	int dev = 0;
	HIP_TRY(hipGetDevice(&dev));
	std::lock_guard<std::mutex> lk(g_plan_mu);
	for (const DdcPlan &p : g_ddc_plans)
		if (p.samp_rate == samp_rate && p.sps == sps && p.device == dev) {
			*out = &p;
			return 0;
		}
	DdcPlan p;
	p.samp_rate = samp_rate; p.sps = sps; p.device = dev;
	const double out_rate = (double)kSymRate * sps;
	if (std::fmod(samp_rate, out_rate) == 0.0)
		return fail(-EINVAL, "ddc: %.1f is an exact multiple of %d x %d: the reference's own direct mode cannot run there "
		                     "(_select_decim returns its factors without storing them, gmr1_rx_sdr.py:652-655)", samp_rate, kSymRate, sps);
	// _select_decim :657-680
	const int decim_max = (int)std::floor(samp_rate / (2.0 * kSymRate));
	const int decim_min = (int)std::ceil(samp_rate / (3.0 * kSymRate));
	std::vector<int> best;
	double best_score = -1.0;
	for (int i = decim_min; i <= decim_max; i++) {
		const std::vector<int> f = ddc_factor(i);
		const double sc = ddc_score(f);
		if (sc > best_score) { best_score = sc; best = f; }     // the first of equal scores, like Python's stable sort
	}
	if (best.empty())
		return fail(-EINVAL, "ddc: sample rate %.1f too low for a direct branch", samp_rate);
	if (best.size() == 1)
		best.push_back(1);
	const int decim = best[0] * best[1];
	double resamp = (out_rate * decim) / samp_rate;
	if (best[1] <= 4) {
		resamp /= best[1];
		best[1] = 1;
	}
	p.d1 = best[0]; p.d2 = best[1]; p.resamp = resamp;
	if (resamp == 1.0)
		return fail(-EINVAL, "ddc: resampling rate 1 (no resampler stage) is not built");
	// _generate_taps :682-749: root-raised cosine in the resampler, plain low-passes before it
	const double fs2 = samp_rate / (p.d1 * p.d2);
	const std::vector<float> rrc = design_rrc(32.0, 32.0 * fs2, kSymRate, 0.35, (int)(11.0 * 32 * fs2 / kSymRate));
	if (p.d2 != 1)
		p.taps2 = design_low_pass(1.0, 1.0, 0.45 / p.d2, 0.10 / p.d2);
	p.taps1 = design_low_pass(1.0, 1.0, 0.3 / p.d1, 0.3 / p.d1);
	if ((int)p.taps1.size() > kDdcMaxTaps || (int)p.taps2.size() > kDdcMaxTaps)
		return fail(-EINVAL, "ddc: filters of %zu / %zu taps (at most %d)", p.taps1.size(), p.taps2.size(), kDdcMaxTaps);
	const int nt = (int)rrc.size();
	p.tpf = (nt + kNfilt - 1) / kNfilt;
	// what k_resamp keeps in registers per phase: 30 taps, or 96 in its long instantiation (plans that resample down:
	// 1.0 Msps -> rate 0.468, 95 taps per phase)
	const int kBankTaps = p.tpf <= 30 ? 30 : 96;
	if (p.tpf > kBankTaps)
		return fail(-EINVAL, "ddc: the resampler of this plan (rate %.4f, %d taps per phase) is longer than the %d the kernel "
		                     "holds", resamp, p.tpf, kBankTaps);
	p.bank_tpf = kBankTaps;
	p.j0 = (nt / 2) % kNfilt;
	// phase step nfilt / rate = nfilt * samp_rate / (d1 d2 out_rate), as a reduced fraction
	long long num = (long long)kNfilt * (long long)std::llround(samp_rate), den = (long long)p.d1 * p.d2 * (long long)out_rate;
	if ((double)std::llround(samp_rate) != samp_rate)
		return fail(-EINVAL, "ddc: the sample rate must be a whole number of Hz");
	const long long g = gcdll(num, den);
	p.num = num / g; p.den = den / g;
	std::vector<float2> bank((size_t)kNfilt * kBankTaps, make_float2(0.f, 0.f));
	for (int j = 0; j < kNfilt; j++)
		for (int k = 0; k < p.tpf; k++) {
			const int i = j + k * kNfilt;
			const float b = i < nt ? rrc[i] : 0.0f;
			const float d = (i + 1 < nt) ? (rrc[i + 1] - rrc[i]) : 0.0f;
			bank[(size_t)j * kBankTaps + k] = make_float2(b, d);
		}
	std::vector<float2> t2(p.taps2.size());
	for (size_t i = 0; i < t2.size(); i++)
		t2[i] = make_float2(p.taps2[i], 0.f);
	if (!t2.empty()) {
		HIP_TRY(hipMalloc(&p.d_taps2, t2.size() * sizeof(float2)));
		HIP_TRY(hipMemcpy(p.d_taps2, t2.data(), t2.size() * sizeof(float2), hipMemcpyHostToDevice));
	}
	HIP_TRY(hipMalloc(&p.d_bank, bank.size() * sizeof(float2)));
	HIP_TRY(hipMemcpy(p.d_bank, bank.data(), bank.size() * sizeof(float2), hipMemcpyHostToDevice));
	g_ddc_plans.push_back(p);
	*out = &g_ddc_plans.back();
	return 0;
}

}  // namespace

extern "C" {

int gmr1_hip_ddc_plan(double samp_rate, int sps, uint64_t n_in, int32_t *decim1, int32_t *decim2, double *resamp,
                      uint64_t *n_out)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (sps < 1 || sps > 16)
		return fail(-EINVAL, "ddc: sps=%d out of range (1..16)", sps);
	const DdcPlan *p;
	r = get_ddc_plan(samp_rate, sps, &p);
	if (r) return r;
	if (decim1) *decim1 = p->d1;
	if (decim2) *decim2 = p->d2;
	if (resamp) *resamp = p->resamp;
	if (n_out) {
		const uint64_t n2 = n_in / (uint64_t)p->d1 / (uint64_t)p->d2;
		const long long v = ((long long)n2 * kNfilt - p->j0) * p->den;
		*n_out = v > 0 ? (uint64_t)(v / p->num) : 0;
	}
	return 0;
}

int gmr1_hip_ddc_dev(void *stream, double samp_rate, int sps, const float *wide, uint64_t n_in, int n_sel,
                     const double *freq_hz, float *out, uint64_t out_stride, uint64_t *n_out_p)
{
	if (!wide || n_sel < 0 || (n_sel && (!freq_hz || !out)))
		return fail(-EINVAL, "ddc: wide / freq_hz / out are required");
	uint64_t n_out;
	int r = gmr1_hip_ddc_plan(samp_rate, sps, n_in, nullptr, nullptr, nullptr, &n_out);
	if (r) return r;
	if (n_out_p) *n_out_p = n_out;
	if (n_sel == 0 || n_out == 0)
		return 0;
	if (out_stride < n_out)
		return fail(-EINVAL, "ddc: out_stride %llu < %llu output samples per carrier", (unsigned long long)out_stride,
		            (unsigned long long)n_out);
	const DdcPlan *p;
	r = get_ddc_plan(samp_rate, sps, &p);
	if (r) return r;
	hipStream_t st = (hipStream_t)stream;
	DevState *s;
	r = dev_state(&s);
	if (r) return r;
	const long long n1 = (long long)(n_in / (uint64_t)p->d1), n2 = n1 / p->d2;
	const int nt1 = (int)p->taps1.size();
	// scratch: per-carrier stage-1 taps and rotation steps, then the two intermediate streams
	const size_t b_t1 = up_to(sizeof(float2) * (size_t)n_sel * nt1, 256), b_rot = up_to(sizeof(double) * (size_t)n_sel, 256);
	const size_t b_y1 = up_to(sizeof(float2) * (size_t)n_sel * (size_t)n1, 256);
	const size_t b_y2 = p->d2 > 1 ? up_to(sizeof(float2) * (size_t)n_sel * (size_t)n2, 256) : 0;
	void *ws;
	WsLease lease;
	if ((r = lease.acquire(s, st))) return r;
	r = dev_workspace(s, b_t1 + b_rot + b_y1 + b_y2, &ws);
	if (r) return r;
	char *w = static_cast<char *>(ws);
	float2 *d_t1 = reinterpret_cast<float2 *>(w);
	double *d_rot = reinterpret_cast<double *>(w + b_t1);
	float2 *d_y1 = reinterpret_cast<float2 *>(w + b_t1 + b_rot);
	float2 *d_y2 = p->d2 > 1 ? reinterpret_cast<float2 *>(w + b_t1 + b_rot + b_y1) : d_y1;
	// freq_xlating_fir_filter_ccc: the low-pass turned up to the carrier, taps[k] e^{+j 2 pi f k / fs}; the output is
	// turned back by e^{-j 2 pi f m d1 / fs}
	std::vector<float2> t1((size_t)n_sel * nt1);
	std::vector<double> rot((size_t)n_sel);
	for (int c = 0; c < n_sel; c++) {
		const double fn = freq_hz[c] / samp_rate;
		for (int k = 0; k < nt1; k++) {
			const double ph = 2.0 * M_PI * std::fmod(fn * k, 1.0);
			t1[(size_t)c * nt1 + k] = make_float2((float)(p->taps1[k] * std::cos(ph)), (float)(p->taps1[k] * std::sin(ph)));
		}
		rot[c] = fn * p->d1;
	}
	HIP_TRY(hipMemcpyAsync(d_t1, t1.data(), t1.size() * sizeof(float2), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(d_rot, rot.data(), rot.size() * sizeof(double), hipMemcpyHostToDevice, st));
	HIP_TRY(hipStreamSynchronize(st));       // t1 / rot are host temporaries
	DdcFirArgs f1;
	std::memset(&f1, 0, sizeof(f1));
	f1.n_sel = n_sel; f1.decim = p->d1; f1.ntaps = nt1; f1.n_in = (long long)n_in; f1.n_out = n1; f1.in_stride = 0;
	f1.x = reinterpret_cast<const float2 *>(wide); f1.taps = d_t1; f1.rot = d_rot; f1.y = d_y1;
	HIP_TRY(launch_ddc_fir(f1, st));
	if (p->d2 > 1) {
		// the same real taps for every carrier: a stride of zero taps rows is not expressible, so the row is repeated
		// by launching carrier by carrier on the one row
		for (int c = 0; c < n_sel; c++) {
			DdcFirArgs f2;
			std::memset(&f2, 0, sizeof(f2));
			f2.n_sel = 1; f2.decim = p->d2; f2.ntaps = (int)p->taps2.size(); f2.n_in = n1; f2.n_out = n2; f2.in_stride = 0;
			f2.x = d_y1 + (size_t)c * n1; f2.taps = p->d_taps2; f2.rot = nullptr; f2.y = d_y2 + (size_t)c * n2;
			HIP_TRY(launch_ddc_fir(f2, st));
		}
	}
	ResampArgs ra;
	std::memset(&ra, 0, sizeof(ra));
	ra.n_slots = n_sel; ra.nfilt = kNfilt; ra.tpf = p->bank_tpf; ra.j0 = p->j0; ra.num = p->num; ra.den = p->den;
	ra.T = n2; ra.n_out = (long long)n_out; ra.out_stride = (long long)out_stride;
	ra.y = d_y2; ra.bank = p->d_bank; ra.out = reinterpret_cast<float2 *>(out);
	HIP_TRY(launch_resamp(ra, st));
	return 0;
}

int gmr1_hip_ddc(double samp_rate, int sps, const float *wide, uint64_t n_in, int n_sel, const double *freq_hz,
                 float *out, uint64_t out_stride, uint64_t *n_out_p)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (!wide || n_sel < 0 || (n_sel && (!freq_hz || !out)))
		return fail(-EINVAL, "ddc: wide / freq_hz / out are required");
	uint64_t n_out;
	r = gmr1_hip_ddc_plan(samp_rate, sps, n_in, nullptr, nullptr, nullptr, &n_out);
	if (r) return r;
	if (n_out_p) *n_out_p = n_out;
	if (!n_sel || !n_out) return 0;
	if (out_stride < n_out)
		return fail(-EINVAL, "ddc: out_stride too small");
	DBuf d_w, d_o;
	HIP_TRY(d_w.alloc(n_in * 8));
	HIP_TRY(d_o.alloc((size_t)n_sel * out_stride * 8));
	HIP_TRY(hipMemcpy(d_w.p, wide, n_in * 8, hipMemcpyHostToDevice));
	r = gmr1_hip_ddc_dev(nullptr, samp_rate, sps, d_w.as<float>(), n_in, n_sel, freq_hz, d_o.as<float>(), out_stride, nullptr);
	if (r) return r;
	HIP_TRY(hipStreamSynchronize(nullptr));
	HIP_TRY(hipMemcpy(out, d_o.p, (size_t)n_sel * out_stride * 8, hipMemcpyDeviceToHost));
	return 0;
}

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
	const uint64_t T = p->mid_samples(n_in) / (uint64_t)(p->n_chans / 2);
	if (n_chans) *n_chans = p->n_chans;
	if (n_mid) *n_mid = T;
	if (n_out) {
		const long long v = ((long long)T * kNfilt - p->j0) * p->den;
		*n_out = v > 0 ? (uint64_t)(v / p->num) : 0;
	}
	return 0;
}

static int channelize_dev_impl(void *stream, double samp_rate, int sps, const float *wide, uint64_t n_in,
                               float rotation, int n_sel, const int32_t *chan_idx,
                               float *out, uint64_t out_stride, uint64_t plane_stride, bool planar, uint64_t *n_out_p)
{
	if (!wide || n_sel < 0 || (n_sel && (!chan_idx || !out)))
		return fail(-EINVAL, "channelize: wide / chan_idx / out are required");
	if (planar && plane_stride < ((uint64_t)n_sel * out_stride + (uint64_t)sps - 1) / (uint64_t)(sps > 0 ? sps : 1))
		return fail(-EINVAL, "channelize: plane_stride %llu < ceil(n_sel x out_stride / sps)", (unsigned long long)plane_stride);
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
	// scratch: slot table + the 2x oversampled channel streams (+ the pre-resampled capture, off the grid)
	const size_t slot_bytes = 2 * kPfbMaxChans * 4;          // slot[n_chans], then sel[n_sel]
	const uint64_t n_pre = p->mid_samples(n_in);
	const size_t mid_bytes = up_to((size_t)n_sel * T * sizeof(float2), 256);
	void *ws;
	WsLease lease;
	if ((r = lease.acquire(s, st))) return r;
	r = dev_workspace(s, slot_bytes + mid_bytes + (p->pre ? (size_t)n_pre * sizeof(float2) : 0), &ws);
	if (r) return r;
	int32_t *d_slot = static_cast<int32_t *>(ws);
	float2 *d_mid = reinterpret_cast<float2 *>(static_cast<char *>(ws) + slot_bytes);
	float2 *d_pre = reinterpret_cast<float2 *>(static_cast<char *>(ws) + slot_bytes + mid_bytes);
	int32_t *d_sel = d_slot + kPfbMaxChans;
	HIP_TRY(hipMemcpyAsync(d_slot, slot.data(), (size_t)nch * 4, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(d_sel, chan_idx, (size_t)n_sel * 4, hipMemcpyHostToDevice, st));
	HIP_TRY(hipStreamSynchronize(st));      // slot[] is a host temporary
	const float2 *pfb_in = reinterpret_cast<const float2 *>(wide);
	if (p->pre) {
		// rotator (if any), then pfb.arb_resampler_ccf to n_chans x 31250 Hz (gmr1_rx_sdr.py:444-461): one stream
		ResampArgs rp;
		std::memset(&rp, 0, sizeof(rp));
		rp.n_slots = 1; rp.nfilt = kNfilt; rp.tpf = 30; rp.j0 = p->pre_j0; rp.num = p->pre_num; rp.den = p->pre_den;
		rp.T = (long long)n_in; rp.n_out = (long long)n_pre; rp.out_stride = (long long)n_pre;
		rp.y = pfb_in; rp.bank = p->d_pre_bank; rp.out = d_pre; rp.rotation = rotation;
		HIP_TRY(launch_resamp(rp, st));
		pfb_in = d_pre;
		rotation = 0.0f;
	}
	PfbArgs pa;
	std::memset(&pa, 0, sizeof(pa));
	pa.n_chans = nch; pa.n_blocks = p->n_blocks; pa.ntaps = p->ntaps;
	pa.n_in = (long long)n_pre; pa.T = (long long)T; pa.rotation = rotation;
	pa.x = pfb_in; pa.taps = p->d_taps; pa.slot = d_slot; pa.y = d_mid;
	pa.sel = d_sel; pa.n_sel = n_sel;
	HIP_TRY(launch_pfb(pa, st));
	ResampArgs ra;
	std::memset(&ra, 0, sizeof(ra));
	ra.n_slots = n_sel; ra.nfilt = kNfilt; ra.tpf = p->tpf; ra.j0 = p->j0; ra.num = p->num; ra.den = p->den;
	ra.T = (long long)T; ra.n_out = (long long)n_out; ra.out_stride = (long long)out_stride;
	ra.y = d_mid; ra.bank = p->d_bank; ra.out = reinterpret_cast<float2 *>(out);
	if (planar) {
		ra.planar_sps = sps;
		ra.plane_stride = (long long)plane_stride;
	}
	HIP_TRY(launch_resamp(ra, st));
	return 0;
}

int gmr1_hip_channelize_dev(void *stream, double samp_rate, int sps, const float *wide, uint64_t n_in,
                            float rotation, int n_sel, const int32_t *chan_idx,
                            float *out, uint64_t out_stride, uint64_t *n_out_p)
{
	return channelize_dev_impl(stream, samp_rate, sps, wide, n_in, rotation, n_sel, chan_idx, out, out_stride, 0, false, n_out_p);
}

int gmr1_hip_channelize_planar_dev(void *stream, double samp_rate, int sps, const float *wide, uint64_t n_in,
                                   float rotation, int n_sel, const int32_t *chan_idx,
                                   float *out_planes, uint64_t out_stride, uint64_t plane_stride, uint64_t *n_out_p)
{
	return channelize_dev_impl(stream, samp_rate, sps, wide, n_in, rotation, n_sel, chan_idx, out_planes, out_stride, plane_stride,
	                           true, n_out_p);
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
