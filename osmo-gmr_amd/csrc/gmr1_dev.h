// gmr1_dev.h -- device-side data layout shared by the kernels and the host shim.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gmr1_hip.h"
#include "rx_loop.h"

#include "profile_env.h"

namespace gmr1 {

constexpr int kMaxSync = GMR1_HIP_MAX_SYNC;
constexpr int kMaxChunks = GMR1_HIP_MAX_CHUNKS;
constexpr int kMaxSyncSyms = GMR1_HIP_MAX_SYNC_SYMS;
constexpr int kMaxWindow = GMR1_HIP_MAX_WINDOW;
constexpr int kMaxInLen = GMR1_HIP_MAX_IN_LEN;
constexpr int kMaxLen = 480;            // symbols per burst (DC12: 468)
constexpr int kMaxCoef = 128;          // sync symbols of one sequence (RACH: 99)
constexpr int kNumTypes = 16;          // descriptor table slots (10 built in + custom)
constexpr int kCustomSlot = 15;

// Burst format as the kernels read it (uniform / scalar loads).
struct DevChunk {
	int16_t pos, len;
	uint8_t syms[kMaxSyncSyms];
};

struct DevBurst {
	float rotation;
	int32_t nbits, len, ebits;
	int32_t n_sync;
	int32_t n_chunks[kMaxSync];
	int32_t sync_tl[kMaxSync];          // total sync symbols of each sequence
	DevChunk sync[kMaxSync][kMaxChunks];
	int32_t n_data;
	int16_t dpos[kMaxChunks], dlen[kMaxChunks], dcum[kMaxChunks];
	int16_t ord_of_sym[kMaxLen];        // data-symbol ordinal of each symbol, -1 if not data
};

// L1 chain selector for the fused kernel / decode kernels
enum Chain : int { kChainNone = 0, kChainBcch = 1, kChainCcch = 2 };

struct RxArgs {
	int n;                 // bursts
	int sps;
	int in_len[2];         // window length per kind (fused) ; [0] used when fixed_type >= 0
	int fixed_type;        // >= 0: every burst has this type, demod only
	int ebits_stride;
	int ssyms_stride;
	int dbg_stop;          // profiling aid: 0 = run everything, N = stop after phase N
	int stage_samples;     // fused path: LDS samples for the sync-chunk windows (max over BCCH / DC6)
	int impl;              // fused path: 0 = k_rx4 (row-batched serial phases), 1 = k_rx (one burst at a time);
	                       // demod only: 2 / 3 = k_rx4g (four bursts per wave; 3: its small-format variant), else k_rx
	int conv_acc;          // fused path: 1 = libosmocore's accelerated Viterbi decoder (GMR1_HIP_CONV_ACC), 0 = its generic one
	int seg_stride;        // fused path, with seg_count: the bursts are listed in segments of seg_stride slots (a multiple of 4) ...
	const int32_t *seg_count;   // ... of which the first seg_count[s] are in use (the receive loop's CCCH lists); NULL: all n
	const int32_t *seg_first;   // optional, with seg_count: only the slots from seg_first[s] (rounded up to a multiple of 4) on are this launch's
	int seg_groups;             // with seg_first: groups of four slots the launch covers per segment (its grid is n_segments x seg_groups)
	const float2 *iq;
	long long plane_stride;     // 0: iq is the interleaved sample array.  > 0 (fused batch kernel, sps 4): polyphase-planar,
	                            // sample s at iq[(s & 3) * plane_stride + (s >> 2)]
	const uint64_t *offset;
	const uint8_t *kind;
	const float *freq_shift;
	// outputs (device, optional unless noted)
	uint8_t *l2;           // n x 24
	int32_t *crc, *conv;
	float *toa, *freq_err;
	float *energy;             // optional: burst_energy() of each window (gmr1_rx.c:172-182)
	int8_t *ebits;
	float *ssyms;
	int32_t *rv;           // required
	int32_t *sync_id;
};

// The receive loop of all chains (rx_kernels.hip), three launches:
//   k_rx_chain  one wavefront per chain walks the feedback chain -- list a round, demodulate and decode its BCCH burst,
//               apply the result -- and only LISTS the round's CCCH bursts (nothing feeds back from them);
//   k_rx4       the listed CCCH bursts of all chains, four per wavefront (the burst kernel at its throughput shape);
//   k_rx_merge  one wavefront per chain writes the records in frame order, exactly as rx_bcch / rx_ccch emit them.
// `a` carries what every burst shares (iq, sps, window lengths, staging size).
struct RxLoopRound {               // what a round leaves for k_rx_merge
	int32_t c_first, c_n;          // its CCCH bursts in the chain's list
	float minen;                   // the CCCH energy gate level the round started with (gmr1_rx.c:813)
	int32_t b_emit;                // 1: its BCCH burst was found and passed the CRC (a record)
	int32_t b_fn, b_tn;            // ... after the SI1 alignment
	int32_t b_conv, b_frame;
	uint8_t b_l2[24];
};
struct RxLoopCcch { int32_t fn, tn, frame; };    // what a CCCH record needs besides the burst kernel's outputs

constexpr int kLoopSlices = 4;     // time slices the chains are walked in (see launch_rx_loop)
struct RxLoopArgs {
	RxLoopState *state;            // n_chains: starting states in, final states out
	gmr1_hip_rx_record *rec;       // n_chains x rec_stride records, frame order per chain
	int32_t *rec_frame;            // optional, parallel to rec: index of the record's frame in the chain's frame log
	float *rec_minen;              // optional, parallel to rec: the CCCH energy gate level of the record's round
	RxLoopFrame *flog;             // optional, n_chains x flog_stride: (align, freq_err, fn) of every frame
	int rec_stride, flog_stride;
	int max_rounds;
	int32_t *n_rounds, *n_rec, *n_frames;   // n_chains each
	// optional: the records of all chains back to back in chain order (the order gmr1_hip_rx_run hands them back in),
	// chains that outgrew their buffers left out; *n_packed = how many
	gmr1_hip_rx_record *packed;
	int32_t *n_packed;
	// between the three launches (device scratch)
	RxLoopRound *rounds;           // n_chains x max_rounds
	int c_stride;                  // CCCH list slots per chain (a multiple of 4, >= the frames of the longest chain + 4 per time slice)
	int32_t *n_ccch;               // n_chains: bursts listed
	int32_t *fin;                  // n_chains: 1 once the chain has reached the end of its capture
	int32_t *slice_end;            // (kLoopSlices + 1) x n_chains: list length at the end of each time slice ([0] = zeros)
	uint64_t *c_off;               // n_chains x c_stride each: the burst kernel's operands ...
	float *c_fs;
	uint8_t *c_kind;
	RxLoopCcch *c_meta;
	uint8_t *c_l2;                 // ... and results (x 24)
	int32_t *c_crc, *c_conv, *c_rv;
	float *c_en;
};

struct DetectArgs {
	int n, sps, in_len;
	int n_types;
	int types[4];          // candidate burst types of THIS launch (same length / modulation family): table slots
	int max_lags;          // most lags any candidate of this launch searches (sizes the correlation accumulator when > 256)
	int first;             // position of types[0] in the caller's list (longer lists run as several launches)
	int carry;             // 1: rv / bt_id / sync_id / toa / best_pwr hold the outcome of the earlier candidates
	float rot0;            // rotation of the caller's FIRST candidate: the window is normalised with it once (pi4cxpsk.c:629)
	float *best_pwr;       // n: weighted power of the best candidate so far (required when carry is used)
	const float2 *iq;
	const uint64_t *offset;
	const float *freq_shift;
	const float *e_toa;    // optional expected TOA per burst (< 0 or NULL: not used)
	int32_t *bt_id, *sync_id, *rv;
	float *toa;
};

struct ModOrderArgs {
	int n, sps, in_len;
	const float2 *iq;
	const uint64_t *offset;
	const float *freq_shift;
	int32_t *order;        // 2 (BPSK) or 4 (QPSK)
};

struct L1Args {
	int n;
	int chain;             // kChainBcch / kChainCcch
	int conv_acc;          // 1 = libosmocore's accelerated decoder, 0 = its generic one
	const int8_t *ebits;   // n x (424|432)
	uint8_t *l2;
	int32_t *crc, *conv;
};

// ---- FCCH -------------------------------------------------------------------
constexpr int kFcchTabs = 3;          // gmr1_fcch_burst, gmr1_fcch3_lband_burst, gmr1_fcch3_sband_burst
constexpr int kFcchMaxLen = 480;      // symbols (117 / 468)

struct FcchTables {
	float  dual[kFcchTabs][kFcchMaxLen];    // sqrt(2) cos(phi)                 fcch.c:167-193
	float2 up[kFcchTabs][kFcchMaxLen];      // sqrt(2)/2 e^{+j phi}             fcch.c:92-121
	float2 shift[kFcchTabs][kFcchMaxLen];   // e^{j 2 pi (len/2) i / len}       fcch.c:575-580
	float2 twid[kFcchTabs][kFcchMaxLen];    // e^{-2 pi j k / len}
	float  freq[kFcchTabs];
	int    len[kFcchTabs];
};

struct FcchRoughArgs {
	int n, len, sps, tab;
	const float2 *iq;
	const uint64_t *offset;
	const float *freq_shift;
	float2 *dec;  size_t dec_stride;        // decimated samples per stream
	float *partial;  int n_stat_tiles;      // 4 floats per (stream, tile)
	float *tile_best;  int n_lag_tiles;     // 8 floats per (stream, lag tile)
	float *energy;  size_t energy_stride;   // optional |corr|^2 per lag
	int32_t *toa, *rv;
	// the folded sweep (k_fcch_sweep<NT, true>): every tile's statistics partial is a 32-byte record {sum re, epoch, sum im, epoch,
	// sum |x|^2, epoch, -, -} in a buffer only these kernels write: a word whose epoch is THIS launch's was written by this launch
	float *fold_partial;  uint32_t epoch;
	int fold_polls;                          // how often a tile looks for the stream's records before it gives up
};

struct FcchMultiArgs {
	int n, sps, burst_len, N;
	int nlags, Lw, Lp;
	const float *energy;  size_t energy_stride;
	int32_t *toa;          // n x N
	int32_t *count;        // n : peaks found, or -EINVAL
};

struct FcchFineArgs {
	int n, sps, tab, mode;                  // mode 0 fine, 1 snr
	const float2 *iq;
	const uint64_t *offset;
	const float *freq_shift;
	int32_t *toa;
	float *freq_err, *snr;
};

hipError_t upload_fcch_tables(const FcchTables *host, hipStream_t stream);
int fcch_stat_tiles(int len);
bool fcch_one_pass();      // the rough sweep as k_fcch_sweep + k_fcch_energy (profiling build: GMR1_HIP_FCCH_TWO_PASS selects the old pair)
int fcch_lag_tiles(int nlags);
hipError_t launch_fcch_rough(const FcchRoughArgs &a, int ntaps, hipStream_t stream);
hipError_t launch_fcch_multi(const FcchMultiArgs &a, hipStream_t stream);
hipError_t launch_fcch_fine(const FcchFineArgs &a, int nsym, hipStream_t stream);

// Acquisition of the receive loop (gmr1_rx.c:605-744) as ONE dependent chain on the stream: what the host used to do
// between two sweeps (add the found offset, check the bounds, lay out the next sweep's windows) runs in k_acq_glue, so
// the five sweeps need no host round trip between them.  Entries are per carrier of the sweep list (`k`), candidate
// slots per carrier are kAcqPeaks wide; a slot that is not live gets the carrier's first sample as a harmless window.
constexpr int kAcqPeaks = 16;
struct AcqArgs {
	int n;                       // carriers in the list
	int sps, flen;               // samples of an FCCH burst
	int wl3;                     // samples of the 650 ms sweep
	const uint64_t *base;        // n: first sample of the carrier
	const uint64_t *len;         // n: samples of the carrier
	int32_t *stat;               // n: 0 alive, else the status the carrier ends with
	int32_t *align, *base_align; // n
	float *ferr;                 // n
	const int32_t *can3;         // n: the carrier is long enough for the 650 ms sweep at all
	// sweep outputs the glue reads
	const int32_t *toa1, *rv1;   // rough
	const int32_t *ftoa; const float *fe;           // fine
	const int32_t *peaks, *count;                   // rough_multi: n x kAcqPeaks, n
	const int32_t *ctoa; const float *cfe;          // fine over the candidate slots
	// next sweep's inputs the glue writes
	uint64_t *off;               // n or n x kAcqPeaks
	float *fs;
	int32_t *live;               // n x kAcqPeaks: the slot holds a candidate
};
hipError_t launch_acq_glue(int step, const AcqArgs &a, hipStream_t stream);

// ---- traffic-channel layer 1 (l1_kernels.hip) -------------------------------
struct Facch3Args {
	int n;                     // frames (each = 4 bursts x 104 soft bits)
	int conv_acc;              // 1 = libosmocore's accelerated decoder, 0 = its generic one
	const int8_t *ebits;       // n x 416
	const uint8_t *ciph;       // optional n x 384 keystream bits
	uint8_t *l2;               // n x 10
	uint8_t *bits_s;           // optional n x 32 status bits
	int32_t *crc, *conv;
};

struct Tch3Args {
	int n;                     // bursts (212 soft bits each, two speech frames)
	int m;                     // multiplexing mode 0 / 1
	int conv_acc;              // 1 = libosmocore's accelerated decoder, 0 = its generic one
	const int8_t *ebits;       // n x 212
	const uint8_t *ciph;       // optional n x 208 keystream bits
	uint8_t *frames;           // n x 2 x 10
	uint8_t *bits_s;           // optional n x 4
	int32_t *conv;             // optional n x 2
};

struct DkabArgs {
	int n, sps, in_len;
	const float2 *iq;
	const uint64_t *offset;
	const float *freq_shift;   // optional, rad/symbol
	const int32_t *p;          // DKAB position per burst
	int8_t *ebits;             // optional n x 8
	float *toa;                // optional
	int32_t *rv;               // 0 found, 1 not found, < 0 error
};

struct A5Args {
	int n;                     // (key, fn) pairs
	int alg;                   // 0: zeros, 1: A5/1
	int nbits;
	const uint8_t *keys;       // n x 8
	const uint32_t *fn;
	uint8_t *dl, *ul;          // optional n x nbits ubits each
};

// wideband -> per-ARFCN channelizer (chan_kernels.hip)
constexpr int kPfbMaxBlocks = 11;        // prototype taps / n_chans, rounded up, + 1
static constexpr int kPfbMaxChans = 256;
struct PfbArgs {
	int n_chans;               // even; 64 runs the lane-FFT kernel, anything else (<= kPfbMaxChans) the generic one
	int n_blocks;              // taps per polyphase branch (<= kPfbMaxBlocks)
	int ntaps;
	long long n_in;            // wideband samples
	long long T;               // output instants = n_in / (n_chans / 2)
	float rotation;            // optional pre-rotation, rad / sample
	const float2 *x;           // wideband capture
	const float *taps;         // prototype low-pass, ntaps floats
	const int32_t *slot;       // n_chans entries: output slot of channel k or -1
	float2 *y;                 // n_slots x T, 2x oversampled channel streams
	const int32_t *sel;        // n_sel selected channel indices (slot order), used by the generic kernel
	int n_sel;
};
struct ResampArgs {
	int n_slots;
	int nfilt, tpf;            // 32 filters x tpf taps
	int j0;                    // starting filter
	long long num, den;        // phase step nfilt / rate = num / den, in 1 / nfilt input samples
	long long T;               // input samples per stream
	long long n_out;           // outputs per stream
	long long out_stride;      // complex samples between output streams
	const float2 *y;           // n_slots x T
	const float2 *bank;        // nfilt x tpf (tap, derivative tap) pairs
	float2 *out;
	float rotation;            // != 0: input sample s is multiplied by e^{j rotation s} as it is read (the pre-resampler
	                           // of an off-grid capture: the script rotates first, utils/gmr1_rx_sdr.py:444-461)
	int planar_sps;            // > 0: `out` is polyphase-planar -- sample g = slot out_stride + n of the flat output array
	long long plane_stride;    // goes to out[(g % planar_sps) * plane_stride + g / planar_sps] (include/gmr1_hip.h)
};
hipError_t launch_pfb(const PfbArgs &a, hipStream_t stream);
hipError_t launch_resamp(const ResampArgs &a, hipStream_t stream);

// direct mode of the recorder script (utils/gmr1_rx_sdr.py:605-807): per selected carrier a frequency-translating
// decimating FIR, then a second decimating FIR (k_ddc_fir, chan_kernels.hip); the arbitrary resampler above follows
constexpr int kDdcMaxTaps = 256;
struct DdcFirArgs {
	int n_sel;                 // carriers
	int decim, ntaps;
	long long n_in, n_out;     // samples per stream in / out (n_out = n_in / decim)
	long long in_stride;       // complex samples between input streams; 0: every carrier reads the same (wideband) stream
	const float2 *x;
	const float2 *taps;        // n_sel x ntaps complex taps (stage 1: the low-pass turned to the carrier; stage 2: real taps, im = 0)
	const double *rot;         // optional, n_sel: output m is multiplied by exp(-j 2 pi frac(m rot[s])) (stage 1)
	float2 *y;                 // n_sel x n_out
};
hipError_t launch_ddc_fir(const DdcFirArgs &a, hipStream_t stream);

hipError_t launch_dkab(const DkabArgs &a, hipStream_t stream);
hipError_t launch_a5(const A5Args &a, hipStream_t stream);
// NT9 bursts: FACCH9 and the three TCH9 modes share one decoder kernel (nt9_kernels.hip)
struct Nt9Args {
	int n;                     // bursts
	int seq_len;               // TCH9: consecutive bursts per channel (inter-burst de-interleaver), FACCH9: 1
	const int32_t *seq_pos;    // optional: position of every burst in its run (runs of unequal length, back to back)
	int kind;                  // 0 2k4, 1 4k8, 2 9k6 (enum gmr1_tch9_mode), 3 FACCH9
	int N;                     // coded bits per input bit (5, 3, 2, 2)
	int conv_acc;              // 1 = libosmocore's accelerated decoder where it applies (N <= 4), 0 = its generic one
	int len;                   // data bits (144, 240, 480, 316)
	const uint32_t *map;       // (len + 4) x N descriptors of the coded bits (see nt9_kernels.hip)
	const int8_t *ebits;       // n x 662
	const uint8_t *ciph;       // optional n x 658
	uint8_t *l2;               // n x l2_bytes
	int l2_bytes;              // 18, 30, 60, 38
	int8_t *sacch;             // optional n x 10 soft bits
	int8_t *status;            // optional n x 4 soft bits
	int32_t *crc;              // FACCH9 only
	int32_t *conv;             // optional
};
hipError_t launch_nt9(const Nt9Args &a, hipStream_t stream);
// RACH (nt9_kernels.hip, same K = 5 trellis) and xCH over DC12 (xch_kernels.hip, K = 9)
struct RachArgs {
	int n;
	int conv_acc;              // 1 = libosmocore's accelerated decoder, 0 = its generic one
	const int8_t *ebits;       // n x 494
	const uint8_t *sb_mask;    // n
	uint8_t *rach;             // n x 18
	int32_t *rv;               // n: 0 = both CRCs pass
	int32_t *conv;             // optional n
	int32_t *crc;              // optional n x 2 (CRC8, CRC12)
};
hipError_t launch_rach(const RachArgs &a, hipStream_t stream);
struct XchArgs {
	int n;
	const int8_t *ebits;       // n x 432
	uint8_t *l2;               // n x 24
	int32_t *crc;              // n
	int32_t *conv;             // optional n
};
hipError_t launch_xch(const XchArgs &a, hipStream_t stream);

// ---- transmit direction (tx_kernels.hip): the channel encoders and the 1-sample-per-symbol modulator ------
// Every GMR-1 channel coder is a fixed GF(2)-linear map from payload bits to burst bits (CRC, feed-forward
// convolutional code, puncturing, interleavers, scrambler, multiplexing).  The host writes that map down once
// per channel as an EncPlan (capi_tx.cpp); one generic kernel evaluates it, one unit (burst, or group of four
// FACCH3 bursts) per wave.
constexpr int kEncMaxIn = 64;          // payload bytes of one unit (TCH9 9k6: 60)
constexpr int kEncMaxExt = 512;        // bits of the extended information word (TCH9 9k6: 480 + 8)
constexpr int kEncMaxOut = 672;        // burst bits of one unit (NT9: 662)
struct EncPlan {
	int32_t n_in0, n_in1;              // payload bytes per unit: first / second input array
	int32_t n_ext;                     // extended information word: [K-1 preset bits | info + CRC | flush zeros] ...
	int32_t n_out;                     // burst bits written per unit
	int32_t n_aux0, n_aux1;            // multiplexed-in bits per unit (status / SACCH): first / second array
	int32_t n_ciph;                    // keystream bits per unit
	int32_t depth;                     // 1, or 3: inter-burst interleaver (bursts n, n-1, n-2 of a run)
	uint32_t poly[8];                  // window masks over K consecutive bits of the extended word
	uint16_t crc_tab[kEncMaxIn * 8];   // what payload bit b adds to CRC register A (the CRCs are linear: init 0, no final xor)
	uint16_t crc_tab2[kEncMaxIn * 8];  // the same for CRC register B (RACH: CRC12), else zeros
	uint16_t ext_src[kEncMaxExt];      // where every bit of the extended word comes from (see tx_kernels.hip)
	uint32_t out[kEncMaxOut];          // descriptor of every burst bit (see tx_kernels.hip)
};
struct EncArgs {
	int n;                             // units
	int seq_len;                       // depth 3: units per interleaver run (state starts empty at each run)
	const EncPlan *plan;               // device
	const uint8_t *in0, *in1;          // n x n_in0, n x n_in1 payload bytes
	const uint8_t *aux0, *aux1;        // n x n_aux0, n x n_aux1 ubits
	const uint8_t *ciph;               // optional n x n_ciph ubits
	uint8_t *ebits;                    // n x n_out ubits
};
hipError_t launch_encode(const EncArgs &a, hipStream_t stream);

struct ModArgs {
	int n, len, nbits, n_ebits;
	float rotation;
	const int16_t *plan;               // len entries: -1 guard, 0..3 sync symbol, 4 + k: data symbol from ebits[k ...]
	const float2 *rot;                 // len entries: e^{j rotation i}, evaluated on the host as osmo_cxvec_rotate does
	const uint8_t *ebits;              // n x n_ebits ubits
	float2 *out;                       // n x len symbols
};
hipError_t launch_mod(const ModArgs &a, hipStream_t stream);

// the stand-alone layer-1 primitives (scrambler, intra- / inter-burst interleaver) as one gather kernel:
// out[i] = in[perm ? perm[i] : i], then xor (ubits) or negate (sbits) where mask[i] is set
struct BitMapArgs {
	int n;
	int soft;                          // 1: int8 soft bits (mask negates), 0: ubits / bytes (mask xors bit 0)
	const uint8_t *in;
	const int32_t *perm;               // optional
	const uint8_t *mask;               // optional
	uint8_t *out;
};
hipError_t launch_bitmap(const BitMapArgs &a, hipStream_t stream);

hipError_t launch_rx_loop(const RxArgs &a, const RxLoopArgs &la, int n_chains, hipStream_t stream);
hipError_t launch_facch3(const Facch3Args &a, hipStream_t stream);
hipError_t launch_tch3(const Tch3Args &a, hipStream_t stream);
hipError_t launch_rx_tch3(const RxArgs &a, const Tch3Args &t, hipStream_t stream);   // a.impl == 3, NT3 speech

// launchers (rx_kernels.hip)
// descriptors live in __constant__ memory of the current device
hipError_t upload_types(const DevBurst *host, int first, int count, hipStream_t stream);
hipError_t launch_rx(const RxArgs &a, bool decode, int max_in_len, hipStream_t stream);
// interleaved sample array -> polyphase-planar (sample s to out[(s % sps) * plane_stride + s / sps])
hipError_t launch_to_planar(const float2 *in, float2 *out, unsigned long long n, int sps, long long plane_stride, hipStream_t stream);
hipError_t launch_l1(const L1Args &a, hipStream_t stream);
hipError_t launch_detect(const DetectArgs &a, hipStream_t stream);
hipError_t launch_mod_order(const ModOrderArgs &a, hipStream_t stream);
size_t rx_lds_bytes(int max_in_len);

}  // namespace gmr1
