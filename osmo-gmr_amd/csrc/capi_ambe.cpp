// capi_ambe.cpp -- C ABI of the AMBE speech decoder: the reference's codec object (include/osmocom/gmr1/codec/codec.h:37-45:
// gmr1_codec_alloc / _release / _decode_frame / _decode_dtx) and batched forms over many voice channels
// (gmr1_hip_codec_*).  All samples are produced on the GPU (ambe_kernels.hip); there is no CPU path.
//
// What the host contributes is tables: the codebooks (ambe_tables.cpp), and every value the reference obtains from
// libm on an argument that can be enumerated - computed here with the same libm calls on the same float arguments,
// so they are the numbers the reference itself would use on this machine.

#include "capi_common.h"

#include <cmath>
#include <new>

#include "ambe_dev.h"
#include "ambe_tables.h"

#include "../../include/gmr1_hip.h"
#include "../../include/osmocom/gmr1/codec/codec.h"

using namespace gmr1;

namespace {

constexpr float kPi = 3.141592653589793f;       // src/codec/private.h:117

// libm through pointers the optimiser cannot see through: no pow(2, x) -> exp2(x) rewriting, no folding
float (*volatile p_powf)(float, float) = powf;
float (*volatile p_cosf)(float) = cosf;
float (*volatile p_log2f)(float) = log2f;
float (*volatile p_exp2f)(float) = exp2f;

float word(uint32_t w)
{
	float f;
	std::memcpy(&f, &w, 4);
	return f;
}

// frame.c:79-118
float f0log_sf0(float before, float now, int rule)
{
	if (now != before) {
		switch (rule) {
		case 0: return now;
		case 1: return (0.65f * now) + (0.35f * before);
		case 2: return (now + before) / 2.0f;
		default: return before;
		}
	}
	const float step = 4.2672e-2f;
	switch (rule) {
	case 0:
	case 1: return now;
	case 2: return now + step;
	default: return now - step;
	}
}

AmbeTab *g_tab;              // host image
std::once_flag g_tab_once;

void build_tab()
{
	AmbeTab *t = new (std::nothrow) AmbeTab;
	if (!t)
		return;
	std::memset(t, 0, sizeof(*t));
	for (int i = 0; i < 1024; i++)
		t->cosv[i] = p_cosf((kPi * i) / 512.0f);                           // math.c:45-52
	for (int i = 0; i < 121; i++) {
		// synth.c:36-54 lists 0.000f, 0.025f, ...: the correctly rounded quotient of the exact integers is that literal
		const int k = i < 40 ? i : i > 80 ? 120 - i : 40;
		t->win[i] = (float)(25 * k) / 1000.0f;
	}
	float f0log[129];
	for (int p = 0; p < 128; p++) {
		f0log[p] = -4.312f - 2.1336e-2f * p;                               // frame.c:300
		t->f0_sf1[p] = p_powf(2.0f, f0log[p]);
	}
	f0log[128] = 0.0f;                                                      // a decoder that has seen no speech yet (ambe.c:41)
	for (int before = 0; before < 129; before++)
		for (int p = 0; p < 128; p++)
			for (int rule = 0; rule < 4; rule++)
				t->f0_sf0[(before * 128 + p) * 4 + rule] = p_powf(2.0f, f0log_sf0(f0log[before], f0log[p], rule));
	for (int L = 1; L < 64; L++)
		t->log2_L[L] = p_log2f((float)L);                                   // frame.c:238, 279
	for (int a = 0; a < 256; a++)
		t->tone_ampl[a] = (int)(32767.0f * p_exp2f(((float)a - 255.0f) / 17.0f));   // tone.c:146
	// synth.c:98-110 stepped i + 1 times: x -> 171 x + 11213 (mod 53125)
	uint64_t mul = 1, add = 0;
	for (int i = 0; i < 121; i++) {
		mul = (mul * 171) % 53125;
		add = (add * 171 + 11213) % 53125;
		t->lcg_mul[i] = (uint32_t)mul;
		t->lcg_add[i] = (uint32_t)add;
	}
	for (int i = 0; i < 512; i++) t->gain[i] = word(ambe::k_gain[i]);
	for (int i = 0; i < 256; i++) t->prba12[i] = word(ambe::k_prba12[i]);
	for (int i = 0; i < 128; i++) t->prba34[i] = word(ambe::k_prba34[i]);
	for (int i = 0; i < 384; i++) t->prba57[i] = word(ambe::k_prba57[i]);
	for (int i = 0; i < 512; i++) t->hoc[0][i] = word(ambe::k_hoc0[i]);
	for (int i = 0; i < 256; i++) {
		t->hoc[1][i] = word(ambe::k_hoc1[i]);
		t->hoc[2][i] = word(ambe::k_hoc2[i]);
		t->hoc[3][i] = word(ambe::k_hoc3[i]);
		t->perr14[i] = word(ambe::k_sf0_perr14[i]);
	}
	for (int i = 0; i < 4; i++) t->interp[i] = word(ambe::k_sf0_interp[i]);
	for (int i = 0; i < 128; i++) t->perr58[i] = word(ambe::k_sf0_perr58[i]);
	for (int i = 0; i < 56; i++) t->rho[i] = word(ambe::k_rho[i]);
	for (int i = 0; i < 64; i++) t->vuv[i] = ambe::k_vuv[i];
	for (int i = 0; i < 192; i++) t->hpg[i] = ambe::k_hpg[i];
	g_tab = t;
}

const AmbeTab *host_tab()
{
	std::call_once(g_tab_once, build_tab);
	return g_tab;
}

// device copy of the tables, one per device
constexpr int kMaxDev = 16;
std::mutex g_mu;
AmbeTab *g_dev_tab[kMaxDev];
AmbeBig *g_dev_big[kMaxDev];
ambe_libm::LibmTab *g_dev_libm[kMaxDev];

long double (*volatile p_exp2l)(long double) = exp2l;

const ambe_libm::LibmTab *host_libm()
{
	static ambe_libm::LibmTab t;
	static std::once_flag once;
	std::call_once(once, [] {
		for (int i = 0; i < 32; i++)
			t.exp2_tab[i] = ambe_libm::bits_of((double)p_exp2l((long double)i / 32.0L)) - ((uint64_t)i << 47);
		for (int i = 0; i < 16; i++)
			ambe_libm::log2_entry(i, t.invc[i], t.logc[i]);
	});
	return &t;
}

int dev_tab(const AmbeTab **out, const AmbeBig **big, const ambe_libm::LibmTab **libm)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	const AmbeTab *h = host_tab();
	if (!h)
		return fail(-ENOMEM, "codec: no memory for the tables");
	int dev = 0;
	HIP_TRY(hipGetDevice(&dev));
	if (dev < 0 || dev >= kMaxDev)
		return fail(-EINVAL, "codec: device index %d out of range", dev);
	std::lock_guard<std::mutex> lk(g_mu);
	if (!g_dev_tab[dev]) {
		void *d = nullptr;
		HIP_TRY(hipMalloc(&d, sizeof(AmbeTab)));
		HIP_TRY(hipMemcpy(d, h, sizeof(AmbeTab), hipMemcpyHostToDevice));
		g_dev_tab[dev] = static_cast<AmbeTab *>(d);
		// the two tables the device builds for itself (ambe_dev.h: AmbeBig), once per device
		void *b = nullptr;
		HIP_TRY(hipMalloc(&b, sizeof(AmbeBig)));
		HIP_TRY(launch_ambe_big(g_dev_tab[dev], static_cast<AmbeBig *>(b), nullptr));
		HIP_TRY(hipStreamSynchronize(nullptr));
		g_dev_big[dev] = static_cast<AmbeBig *>(b);
		void *l = nullptr;
		HIP_TRY(hipMalloc(&l, sizeof(ambe_libm::LibmTab)));
		HIP_TRY(hipMemcpy(l, host_libm(), sizeof(ambe_libm::LibmTab), hipMemcpyHostToDevice));
		g_dev_libm[dev] = static_cast<ambe_libm::LibmTab *>(l);
	}
	*out = g_dev_tab[dev];
	*big = g_dev_big[dev];
	*libm = g_dev_libm[dev];
	return 0;
}

int decode_dev(hipStream_t st, int n_ch, int n_frames, const uint8_t *frames, int16_t *pcm, int pcm_stride, int32_t *rv,
               void *state, int tone_n)
{
	if (n_ch < 0 || n_frames < 0)
		return fail(-EINVAL, "codec: negative channel or frame count");
	const AmbeTab *t;
	const AmbeBig *big;
	const ambe_libm::LibmTab *libm;
	int r = dev_tab(&t, &big, &libm);
	if (r) return r;
	if (n_ch == 0 || n_frames == 0)
		return 0;
	if (!frames || !pcm || !state)
		return fail(-EINVAL, "codec: frames / pcm / state are required");
	if ((reinterpret_cast<uintptr_t>(pcm) & 1u) || (reinterpret_cast<uintptr_t>(state) & 15u))
		return fail(-EINVAL, "codec: pcm must be 2-byte and state 16-byte aligned");
	AmbeArgs a;
	a.n_ch = n_ch;
	a.n_frames = n_frames;
	a.frames = frames;
	a.pcm = pcm;
	a.pcm_stride = pcm_stride;
	a.rv = rv;
	a.state = static_cast<AmbeState *>(state);
	a.tab = t;
	a.big = big;
	a.libm = libm;
	a.tone_n = tone_n;
	a.dbg = 0;
	HIP_TRY(launch_ambe(a, st));
	return 0;
}

int init_dev(hipStream_t st, int n_ch, void *state, int flags)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n_ch < 0 || (n_ch > 0 && !state))
		return fail(-EINVAL, "codec: state is required");
	if (flags & ~GMR1_HIP_CODEC_CLEARED)
		return fail(-EINVAL, "codec: unknown flag bits 0x%x", flags);
	if (reinterpret_cast<uintptr_t>(state) & 15u)
		return fail(-EINVAL, "codec: state must be 16-byte aligned");
	HIP_TRY(launch_ambe_init(static_cast<AmbeState *>(state), n_ch, flags, st));
	return 0;
}

}  // namespace

// the reference's opaque decoder (src/codec/codec.c:37-40): here a device-resident state plus a block of pinned
// host memory the kernel reads the frame from and writes the samples to
struct gmr1_codec {
	int dev;
	hipStream_t st;
	AmbeState *state;
	unsigned char *h, *d;      // mapped block: [0, 16) frame, [16, 20) rv, [32, ...) samples
	size_t samples;            // capacity of the sample area
};

namespace {

constexpr size_t kCodecPcmOff = 32;

int codec_block(gmr1_codec *c, size_t samples)
{
	if (c->h && c->samples >= samples)
		return 0;
	if (c->h) {
		HIP_TRY(hipHostFree(c->h));
		c->h = c->d = nullptr;
	}
	void *h = nullptr, *d = nullptr;
	HIP_TRY(hipHostMalloc(&h, kCodecPcmOff + samples * 2, hipHostMallocMapped));
	HIP_TRY(hipHostGetDevicePointer(&d, h, 0));
	c->h = static_cast<unsigned char *>(h);
	c->d = static_cast<unsigned char *>(d);
	c->samples = samples;
	return 0;
}

}  // namespace

extern "C" {

size_t gmr1_hip_codec_state_bytes(void)
{
	return sizeof(AmbeState);
}

int gmr1_hip_codec_host_tables(const void **image, size_t *bytes)
{
	const AmbeTab *t = host_tab();
	if (!t)
		return fail(-ENOMEM, "codec: no memory for the tables");
	if (image) *image = t;
	if (bytes) *bytes = sizeof(AmbeTab);
	return 0;
}

int gmr1_hip_codec_libm_check(int which, int n, const float *x, float *out)
{
	// the device's restatement of glibc's powf / cosf (ambe_libm.h), run on the host: which = 0: powf(2, x[i]);
	// 1: powf(x[i], 0.25f); 2: cosf(x[i]).
	// out[i] = the result, or NaN where the argument is outside what is restated (the kernel then evaluates in double)
	if (n < 0 || (n > 0 && (!x || !out)) || which < 0 || which > 2)
		return fail(-EINVAL, "gmr1_hip_codec_libm_check: bad argument");
	const ambe_libm::LibmTab &T = *host_libm();
	for (int i = 0; i < n; i++) {
		bool ok;
		if (which == 2) {
			out[i] = ambe_libm::cosf_glibc(x[i]);
			continue;
		}
		const float v = which == 0 ? ambe_libm::pow2f(T, x[i], &ok) : ambe_libm::powf_pos(T, x[i], 0.25f, &ok);
		out[i] = ok ? v : std::nanf("");
	}
	return 0;
}

int gmr1_hip_codec_init_dev(void *stream, int n_ch, void *state, int flags)
{
	return init_dev((hipStream_t)stream, n_ch, state, flags);
}

int gmr1_hip_codec_decode_batch_dev(void *stream, int n_ch, int n_frames, const uint8_t *frames, int16_t *pcm,
                                    int32_t *rv, void *state)
{
	return decode_dev((hipStream_t)stream, n_ch, n_frames, frames, pcm, kAmbeFrameSamples, rv, state, kAmbeFrameSamples);
}

int gmr1_hip_codec_decode_batch(int n_ch, int n_frames, const uint8_t *frames, int16_t *pcm, int32_t *rv, void *state,
                                int flags)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n_ch < 0 || n_frames < 0)
		return fail(-EINVAL, "codec: negative channel or frame count");
	if (flags & ~(GMR1_HIP_CODEC_CLEARED | GMR1_HIP_CODEC_FRESH))
		return fail(-EINVAL, "codec: unknown flag bits 0x%x", flags);
	if (n_ch == 0 || n_frames == 0)
		return 0;
	if (!frames || !pcm)
		return fail(-EINVAL, "codec: frames / pcm are required");
	const size_t nf = (size_t)n_ch * n_frames;
	DBuf d_fr, d_pcm, d_rv, d_st;
	HIP_TRY(d_fr.alloc(nf * kAmbeFrameBytes));
	HIP_TRY(d_pcm.alloc(nf * kAmbeFrameSamples * 2));
	HIP_TRY(d_rv.alloc(nf * 4));
	HIP_TRY(d_st.alloc((size_t)n_ch * sizeof(AmbeState)));
	HIP_TRY(hipMemcpy(d_fr.p, frames, nf * kAmbeFrameBytes, hipMemcpyHostToDevice));
	if (state && !(flags & GMR1_HIP_CODEC_FRESH))
		HIP_TRY(hipMemcpy(d_st.p, state, (size_t)n_ch * sizeof(AmbeState), hipMemcpyHostToDevice));
	else {
		r = init_dev(nullptr, n_ch, d_st.p, flags & GMR1_HIP_CODEC_CLEARED);
		if (r) return r;
	}
	r = decode_dev(nullptr, n_ch, n_frames, d_fr.as<uint8_t>(), d_pcm.as<int16_t>(), kAmbeFrameSamples, d_rv.as<int32_t>(),
	               d_st.p, kAmbeFrameSamples);
	if (r) return r;
	HIP_TRY(hipStreamSynchronize(nullptr));
	HIP_TRY(hipMemcpy(pcm, d_pcm.p, nf * kAmbeFrameSamples * 2, hipMemcpyDeviceToHost));
	if (rv) HIP_TRY(hipMemcpy(rv, d_rv.p, nf * 4, hipMemcpyDeviceToHost));
	if (state) HIP_TRY(hipMemcpy(state, d_st.p, (size_t)n_ch * sizeof(AmbeState), hipMemcpyDeviceToHost));
	return 0;
}

// ---- the reference's own calls (include/osmocom/gmr1/codec/codec.h:37-45) ----

struct gmr1_codec *gmr1_codec_alloc(void)
{
	DevState *s;
	if (dev_state(&s))
		return nullptr;                 // no device: the reference returns NULL when it cannot allocate
	gmr1_codec *c = new (std::nothrow) gmr1_codec();
	if (!c)
		return nullptr;
	void *st = nullptr;
	if (hipGetDevice(&c->dev) != hipSuccess ||
	    hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) != hipSuccess ||
	    hipMalloc(&st, sizeof(AmbeState)) != hipSuccess) {
		delete c;
		return nullptr;
	}
	c->state = static_cast<AmbeState *>(st);
	if (codec_block(c, 1024) || init_dev(c->st, 1, c->state, 0) || hipStreamSynchronize(c->st) != hipSuccess) {
		gmr1_codec_release(c);
		return nullptr;
	}
	return c;
}

void gmr1_codec_release(struct gmr1_codec *codec)
{
	if (!codec)
		return;
	if (codec->st) {
		(void)hipStreamSynchronize(codec->st);
		(void)hipStreamDestroy(codec->st);
	}
	if (codec->state) (void)hipFree(codec->state);
	if (codec->h) (void)hipHostFree(codec->h);
	delete codec;
}

int gmr1_codec_decode_frame(struct gmr1_codec *codec, int16_t *audio, int N, const uint8_t *frame, int bad)
{
	(void)bad;                           // unused by the reference as well (ambe.c:77-81)
	if (!codec || !audio || !frame)
		return fail(-EINVAL, "gmr1_codec_decode_frame: NULL argument");
	if (N < 0)
		return fail(-EINVAL, "gmr1_codec_decode_frame: N < 0");
	int dev = -1;
	HIP_TRY(hipGetDevice(&dev));
	if (dev != codec->dev)
		return fail(-EINVAL, "gmr1_codec_decode_frame: the decoder lives on device %d, the caller is on %d", codec->dev, dev);
	// speech and silence frames fill 160 samples whatever N says, tone frames fill N (ambe.c:110-126)
	const bool tone = (frame[0] & 0xfc) == 0xfc;
	const size_t span = tone ? (size_t)N : (size_t)kAmbeFrameSamples;
	const size_t stride = span > (size_t)kAmbeFrameSamples ? span : (size_t)kAmbeFrameSamples;
	int r = codec_block(codec, stride);
	if (r) return r;
	std::memcpy(codec->h, frame, kAmbeFrameBytes);
	r = decode_dev(codec->st, 1, 1, codec->d, reinterpret_cast<int16_t *>(codec->d + kCodecPcmOff), (int)stride,
	               reinterpret_cast<int32_t *>(codec->d + 16), codec->state, N);
	if (r) return r;
	HIP_TRY(hipStreamSynchronize(codec->st));
	std::memcpy(audio, codec->h + kCodecPcmOff, span * 2);
	return *reinterpret_cast<int32_t *>(codec->h + 16);
}

int gmr1_codec_decode_dtx(struct gmr1_codec *codec, int16_t *audio, int N)
{
	// the reference writes N zeros and leaves the decoder alone (ambe.c:130-141): nothing for the GPU to do
	if (!codec || !audio)
		return fail(-EINVAL, "gmr1_codec_decode_dtx: NULL argument");
	if (N < 0)
		return fail(-EINVAL, "gmr1_codec_decode_dtx: N < 0");
	std::memset(audio, 0, sizeof(int16_t) * (size_t)N);
	return 0;
}

}  // extern "C"
