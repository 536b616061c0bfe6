/*
 * gmr1_hip.h -- C ABI of the MI355X-native GMR-1 receive hot path.
 *
 * Two families of entry points live in libgmr1_hip.so:
 *
 *  1. The reference's own per-burst C API (declared in include/osmocom/gmr1/
 *     {sdr,l1}/.h of this repo, same names / argument meaning / return values
 *     as osmocom/osmo-gmr) -- each call is blocking: H2D, one kernel, D2H.  (The BCCH / DC6 demodulation call at
 *     4 samples per symbol is answered by a resident one-wave kernel fed from a mailbox in pinned memory instead of
 *     a launch per call; it ends by itself 200 us after the last call.  GMR1_HIP_ONE_BURST_SERVER=0 in the
 *     environment keeps the launch per call.  INTEGRATION.md, "What the unchanged application costs".)
 *
 *  2. The batched entry points below, which are what a high-rate caller binds.
 *     "_dev" variants take DEVICE pointers (inputs already resident in HBM) and
 *     a hipStream_t passed as void*; they only enqueue work.  The variants
 *     without "_dev" take HOST pointers and stage through HBM themselves.
 *
 * Every function returns 0 on success or a negative errno; the per-burst
 * status (the reference's own return value) is written to rv[].  There is NO
 * CPU fallback: without a usable HIP device every call fails with -ENODEV.
 *
 * Reference interfaces replaced (osmocom/osmo-gmr, paths relative to the
 * reference tree):
 *   gmr1_hip_demod_batch*        -> gmr1_pi4cxpsk_demod   include/osmocom/gmr1/sdr/pi4cxpsk.h:101-105
 *   gmr1_hip_detect_batch*       -> gmr1_pi4cxpsk_detect  include/osmocom/gmr1/sdr/pi4cxpsk.h:107-110
 *   gmr1_hip_mod_order_batch*    -> gmr1_pi4cxpsk_mod_order include/osmocom/gmr1/sdr/pi4cxpsk.h:112-113
 *   gmr1_hip_bcch_decode_batch*  -> gmr1_bcch_decode      include/osmocom/gmr1/l1/bcch.h:38
 *   gmr1_hip_ccch_decode_batch*  -> gmr1_ccch_decode      include/osmocom/gmr1/l1/ccch.h:38
 *   gmr1_hip_facch3_decode_batch*-> gmr1_facch3_decode    include/osmocom/gmr1/l1/facch3.h:39-40
 *   gmr1_hip_tch3_decode_batch*  -> gmr1_tch3_decode      include/osmocom/gmr1/l1/tch3.h:40-42
 *   gmr1_hip_rx_bcch_ccch_batch* -> rx_bcch / rx_ccch     src/gmr1_rx.c:746-850 (demod + decode of one burst)
 *   gmr1_hip_fcch_rough_batch*   -> gmr1_fcch_rough       include/osmocom/gmr1/sdr/fcch.h:47-49
 *   gmr1_hip_fcch_rough_multi_batch* -> gmr1_fcch_rough_multi include/osmocom/gmr1/sdr/fcch.h:51-53
 *   gmr1_hip_fcch_fine_batch*    -> gmr1_fcch_fine        include/osmocom/gmr1/sdr/fcch.h:55-57
 *   gmr1_hip_fcch_snr_batch*     -> gmr1_fcch_snr         include/osmocom/gmr1/sdr/fcch.h:59-61
 */
#ifndef GMR1_HIP_H
#define GMR1_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GMR1_HIP_MAX_SYNC       4
#define GMR1_HIP_MAX_CHUNKS     8
#define GMR1_HIP_MAX_SYNC_SYMS  32
#define GMR1_HIP_MAX_WINDOW     256   /* lags the demod kernels keep room for by default; wider search windows size their own */
#define GMR1_HIP_MAX_IN_LEN     4096  /* max samples per burst window (BCCH at 16 samples per symbol: 4 064) */

/* burst type ids (order of include/osmocom/gmr1/sdr/nb.h:37-46) */
enum gmr1_hip_burst_id {
	GMR1_HIP_BCCH = 0, GMR1_HIP_DC2, GMR1_HIP_DC6, GMR1_HIP_DC12,
	GMR1_HIP_NT3_SPEECH, GMR1_HIP_NT3_FACCH, GMR1_HIP_NT6, GMR1_HIP_NT9,
	GMR1_HIP_RACH, GMR1_HIP_SDCCH, GMR1_HIP_N_BURSTS
};

/* flat, pointer-free copy of a struct gmr1_pi4cxpsk_burst */
struct gmr1_hip_chunk {
	int32_t pos, len;
	uint8_t syms[GMR1_HIP_MAX_SYNC_SYMS];
};

struct gmr1_hip_burst_flat {
	char    name[16];
	float   rotation;
	int32_t nbits, guard_pre, guard_post, len, ebits;
	int32_t n_sync;
	int32_t n_sync_chunks[GMR1_HIP_MAX_SYNC];
	struct gmr1_hip_chunk sync[GMR1_HIP_MAX_SYNC][GMR1_HIP_MAX_CHUNKS];
	int32_t n_data;
	struct gmr1_hip_chunk data[GMR1_HIP_MAX_CHUNKS];
};

/* ---- library / device ----------------------------------------------------
 * Threads: gmr1_hip_last_error() is per thread; the burst-level _batch / _batch_dev calls keep no state
 * between calls and may run from several threads on different streams.  The calls that use the library's
 * grow-only per-device workspace -- gmr1_hip_fcch_rough*_batch*, gmr1_hip_channelize*, gmr1_hip_ddc*,
 * gmr1_hip_rx_run*, gmr1_hip_detect_batch* with more than four candidates, and gmr1_hip_tch3_rx_batch* where it
 * runs as two launches without a caller's soft-bit buffer -- may ALSO be called from several threads and streams:
 * they take turns.  The host part of such a call runs under a per-device lock (a second thread waits), and a call
 * on another stream first makes its stream wait, on the device, for the previous user's kernels; nothing is
 * refused and no result depends on the interleaving (tests/test_gpu_threads.py).  They do not run in PARALLEL
 * with each other on one device: one receiver per GPU, as in the reference (one process per capture). */
int         gmr1_hip_init(int device);          /* optional; selects the HIP device     */
const char *gmr1_hip_last_error(void);
const char *gmr1_hip_version(void);
int         gmr1_hip_burst_info(int burst_id, struct gmr1_hip_burst_flat *out);
/* Measurement aid (no counterpart in the reference): the shader clock the device holds right now, in MHz -- one wave on
 * `stream` compares the shader-clock counter with the constant-rate wall counter over `micros` microseconds (blocking).
 * Launched behind a timed region it tells what clock that region ran at.  wall_mhz (optional): the wall counter's rate. */
int         gmr1_hip_clock_probe_dev(void *stream, int micros, double *core_mhz, double *wall_mhz);

/* ---- which Viterbi decoder of libosmocore the layer-1 chains reproduce ------
 * The reference hands every channel to osmo_conv_decode() (src/l1/bcch.c:94, ccch.c:98, facch3.c:160, tch3.c:174,
 * facch9.c:134, tch9.c:170, rach.c:167, xch_dc12.c:97).  Which arithmetic that is depends on the libosmocore the
 * integrator links: its generic decoder (every version; src/conv.c), or -- since 2017, for codes with K in {5, 7} and
 * N in {2, 3, 4}: every chain above except xCH (K = 9) and TCH9 2k4 (N = 5) -- osmo_conv_decode_acc (src/conv_acc.c).
 * The two differ in metric quantisation, start states, flush steps and the tail-biting end state, so near the decoding
 * threshold they return different frames, and the accelerated one returns 0 where the generic one returns the path
 * metric (`conv_rv`).  Pick the one the replaced build used.  Process-wide; takes effect with the next call.  The
 * DEFAULT is GMR1_HIP_CONV_ACC -- what a build against any libosmocore released since 2017 ran (configure.ac:23 only
 * asks for >= 0.4.1, but no distribution has shipped anything older than 1.x for years) -- unless the environment says
 * GMR1_HIP_CONV_DECODER=generic when the library is first used; gmr1_hip_version() names the decoder in force. */
enum gmr1_hip_conv_decoder {
	GMR1_HIP_CONV_GENERIC = 0,   /* osmo_conv_decode's generic path for every code */
	GMR1_HIP_CONV_ACC     = 1,   /* osmo_conv_decode_acc where libosmocore >= 0.10 dispatches to it, generic elsewhere */
};
int gmr1_hip_set_conv_decoder(int decoder);     /* 0, or -EINVAL */
int gmr1_hip_get_conv_decoder(void);

/* ---- normal-burst demodulation (any burst type, one type per call) ------- */
/* iq: interleaved float32 I/Q; burst i occupies iq[offset[i] .. offset[i]+in_len) (complex samples).
 * Optional outputs may be NULL.  ebits is n x ebits_stride int8, ssyms n x bt.len float. */
int gmr1_hip_demod_batch_dev(void *stream, int burst_id, int n, int sps, int in_len,
                             const float *iq, const uint64_t *offset, const float *freq_shift,
                             int8_t *ebits, int ebits_stride, int32_t *sync_id,
                             float *toa, float *freq_err, float *ssyms, int32_t *rv);
int gmr1_hip_demod_batch(int burst_id, int n, int sps, int in_len,
                         const float *iq, uint64_t iq_len, const uint64_t *offset, const float *freq_shift,
                         int8_t *ebits, int ebits_stride, int32_t *sync_id,
                         float *toa, float *freq_err, float *ssyms, int32_t *rv);
/* Debugging aid (the reference's ENABLE_DEBUG_SIGNAL dumps, include/osmocom/gmr1/sdr/defs.h:35-39): ONE burst,
 * demodulated as above (host pointers, blocking), plus the four intermediate vectors of gmr1_pi4cxpsk_demod --
 *   corr   [in_len - symbols*sps + 1] float    "pi4cxpsk_corr"  (pi4cxpsk.c:251)  sync correlation magnitude per lag
 *                                              (formats with several training sequences: summed over them)
 *   burst  [in_len]  complex                   "pi4cxpsk_burst" (pi4cxpsk.c:545)  normalised, de-rotated window
 *   align  [symbols] complex                   "pi4cxpsk_align" (pi4cxpsk.c:345)  one sample per symbol at the found timing
 *   final  [symbols] complex                   "pi4cxpsk_final" (pi4cxpsk.c:582)  after fine-frequency + carrier correction
 * Any output but rv may be NULL.  The production kernels work on phases and never form three of these; this entry
 * rebuilds them from the same demodulation (rx_debug_kernels.inc), so ssyms / ebits / toa here ARE the batch entry's. */
int gmr1_hip_demod_taps(int burst_id, int sps, int in_len, const float *iq, float freq_shift,
                        float *corr, float *burst, float *align, float *final_,
                        int8_t *ebits, int32_t *sync_id, float *toa, float *freq_err, float *ssyms, int32_t *rv);

/* ---- burst type detection / modulation order -------------------------------
 * burst_ids: any number of candidate types (same length / modulation family; more than four run as several launches
 * that hand the best so far on); e_toa optional. */
int gmr1_hip_detect_batch_dev(void *stream, int n_types, const int *burst_ids, int n, int sps, int in_len,
                              const float *iq, const uint64_t *offset, const float *freq_shift,
                              const float *e_toa, int32_t *bt_id, int32_t *sync_id, float *toa, int32_t *rv);
int gmr1_hip_detect_batch(int n_types, const int *burst_ids, int n, int sps, int in_len,
                          const float *iq, uint64_t iq_len, const uint64_t *offset, const float *freq_shift,
                          const float *e_toa, int32_t *bt_id, int32_t *sync_id, float *toa, int32_t *rv);
int gmr1_hip_mod_order_batch_dev(void *stream, int n, int sps, int in_len,
                                 const float *iq, const uint64_t *offset, const float *freq_shift, int32_t *order);
int gmr1_hip_mod_order_batch(int n, int sps, int in_len, const float *iq, uint64_t iq_len,
                             const uint64_t *offset, const float *freq_shift, int32_t *order);

/* ---- layer-1 channel decoding (soft bits in, L2 out) --------------------- */
int gmr1_hip_bcch_decode_batch_dev(void *stream, int n, const int8_t *ebits /* n x 424 */,
                                   uint8_t *l2 /* n x 24 */, int32_t *crc, int32_t *conv);
int gmr1_hip_ccch_decode_batch_dev(void *stream, int n, const int8_t *ebits /* n x 432 */,
                                   uint8_t *l2 /* n x 24 */, int32_t *crc, int32_t *conv);
int gmr1_hip_bcch_decode_batch(int n, const int8_t *ebits, uint8_t *l2, int32_t *crc, int32_t *conv);
int gmr1_hip_ccch_decode_batch(int n, const int8_t *ebits, uint8_t *l2, int32_t *crc, int32_t *conv);

/* ---- fused BCCH / CCCH receive: demod + descramble + deinterleave + Viterbi + CRC
 * kind[i]: 0 = BCCH burst (window 234*sps + 20*sps), 1 = CCCH on a DC6 burst
 * (window 234*sps + 10*sps), the windows gmr1_rx.c:759,809 cut.
 * crc[i]: 0 pass, 1 fail, -1 when the demodulator found no sync (rv[i] != 0). */
int gmr1_hip_rx_bcch_ccch_batch_dev(void *stream, int n, int sps,
                                    const float *iq, const uint64_t *offset, const uint8_t *kind,
                                    const float *freq_shift,
                                    uint8_t *l2 /* n x 24 */, int32_t *crc, int32_t *conv,
                                    float *toa, float *freq_err,
                                    int8_t *ebits /* n x 432, optional */,
                                    float *ssyms /* n x 234, optional */, int32_t *rv);
int gmr1_hip_rx_bcch_ccch_batch(int n, int sps,
                                const float *iq, uint64_t iq_len, const uint64_t *offset, const uint8_t *kind,
                                const float *freq_shift,
                                uint8_t *l2, int32_t *crc, int32_t *conv,
                                float *toa, float *freq_err,
                                int8_t *ebits, float *ssyms, int32_t *rv);

/* The same call on a POLYPHASE-PLANAR sample array (opt-in, 4 samples per symbol only): sample s of the flat array the
 * offsets count in is stored at iq_planes[(s % sps) * plane_stride + s / sps] (complex samples; plane_stride >=
 * ceil(total samples / sps)).  Nothing but addresses changes -- every output is bit-identical to the interleaved call's
 * on the same samples -- but the samples gmr1_pi4cxpsk_demod keeps after timing recovery (one per symbol from sample
 * round(toa) on, src/sdr/pi4cxpsk.c:292-295) are then 234 CONSECUTIVE samples of one plane instead of every fourth
 * sample of the whole window, which cuts the second read of a burst from every 128-byte line of its window to 15 of
 * them.  gmr1_hip_iq_to_planar_dev converts an interleaved array (any sps 1..16); the channelizer writes the layout
 * directly when asked (gmr1_hip_channelize_planar_dev). */
int gmr1_hip_rx_bcch_ccch_batch_planar_dev(void *stream, int n, int sps,
                                           const float *iq_planes, uint64_t plane_stride,
                                           const uint64_t *offset, const uint8_t *kind,
                                           const float *freq_shift,
                                           uint8_t *l2 /* n x 24 */, int32_t *crc, int32_t *conv,
                                           float *toa, float *freq_err,
                                           int8_t *ebits /* n x 432, optional */,
                                           float *ssyms /* n x 234, optional */, int32_t *rv);
int gmr1_hip_iq_to_planar_dev(void *stream, int sps, uint64_t n_samples, const float *iq /* interleaved, device */,
                              float *iq_planes /* device, sps x plane_stride complex samples */, uint64_t plane_stride);

/* ---- traffic channel layer 1 -------------------------------------------------
 * FACCH3: n frames, each 4 bursts x 104 soft bits (n x 416) -> l2 n x 10, optional
 * bits_s n x 32, crc / conv per frame; ciph optional n x 384 keystream bits.
 * TCH3: n speech bursts x 212 soft bits -> frames n x 2 x 10 (frame0 then frame1),
 * optional bits_s n x 4, conv n x 2; ciph optional n x 208; m = multiplexing mode. */
int gmr1_hip_facch3_decode_batch_dev(void *stream, int n, const int8_t *ebits, const uint8_t *ciph,
                                     uint8_t *l2, uint8_t *bits_s, int32_t *crc, int32_t *conv);
int gmr1_hip_facch3_decode_batch(int n, const int8_t *ebits, const uint8_t *ciph,
                                 uint8_t *l2, uint8_t *bits_s, int32_t *crc, int32_t *conv);
int gmr1_hip_tch3_decode_batch_dev(void *stream, int n, int m, const int8_t *ebits, const uint8_t *ciph,
                                   uint8_t *frames, uint8_t *bits_s, int32_t *conv);
int gmr1_hip_tch3_decode_batch(int n, int m, const int8_t *ebits, const uint8_t *ciph,
                               uint8_t *frames, uint8_t *bits_s, int32_t *conv);
/* What rx_tch3 does with a speech burst (src/gmr1_rx.c:551-587): gmr1_pi4cxpsk_demod of the NT3 speech format
 * (include/osmocom/gmr1/sdr/pi4cxpsk.h:101-105, src/sdr/nb.c gmr1_nt3_speech_burst), then gmr1_tch3_decode
 * (include/osmocom/gmr1/l1/tch3.h:35-37) of its 212 soft bits -- n bursts from samples to 2 x 10-byte speech frames.  The
 * outputs are those of gmr1_hip_demod_batch* (ebits n x 212, sync_id, toa, rv: optional except rv) followed by those of
 * gmr1_hip_tch3_decode_batch* (frames n x 20, bits_s n x 4, conv n x 2) and identical to calling the two in sequence; large
 * batches at 4 samples per symbol run as ONE launch with the soft bits kept on chip. */
int gmr1_hip_tch3_rx_batch_dev(void *stream, int n, int sps, int in_len,
                               const float *iq, const uint64_t *offset, const float *freq_shift,
                               int m, const uint8_t *ciph,
                               int8_t *ebits, int32_t *sync_id, float *toa, int32_t *rv,
                               uint8_t *frames, uint8_t *bits_s, int32_t *conv);
int gmr1_hip_tch3_rx_batch(int n, int sps, int in_len,
                           const float *iq, uint64_t iq_len, const uint64_t *offset, const float *freq_shift,
                           int m, const uint8_t *ciph,
                           int8_t *ebits, int32_t *sync_id, float *toa, int32_t *rv,
                           uint8_t *frames, uint8_t *bits_s, int32_t *conv);

/* ---- TCH3 follow-up pieces --------------------------------------------------
 * DKAB demodulation (gmr1_dkab_demod, include/osmocom/gmr1/sdr/dkab.h:39-41): n windows of in_len
 * samples (117 * sps + search window), p[i] = DKAB position; rv[i] = 0 found / 1 not found;
 * ebits n x 8 (zero where not found), toa in samples.
 * A5 keystream (gmr1_a5, include/osmocom/gmr1/l1/a5.h:37-41): one (key, frame number) per item,
 * keys n x 8 bytes, dl / ul n x nbits ubits (either may be NULL); alg 0 = zeros, 1 = A5/1. */
int gmr1_hip_dkab_demod_batch_dev(void *stream, int n, int sps, int in_len,
                                  const float *iq, const uint64_t *offset, const float *freq_shift,
                                  const int32_t *p, int8_t *ebits, float *toa, int32_t *rv);
int gmr1_hip_dkab_demod_batch(int n, int sps, int in_len,
                              const float *iq, uint64_t iq_len, const uint64_t *offset, const float *freq_shift,
                              const int32_t *p, int8_t *ebits, float *toa, int32_t *rv);
int gmr1_hip_a5_batch_dev(void *stream, int n, int alg, int nbits,
                          const uint8_t *keys, const uint32_t *fn, uint8_t *dl, uint8_t *ul);
int gmr1_hip_a5_batch(int n, int alg, int nbits, const uint8_t *keys, const uint32_t *fn, uint8_t *dl, uint8_t *ul);

/* ---- NT9 bursts (662 soft bits): FACCH9 and TCH9 ---------------------------------
 * FACCH9 (gmr1_facch9_decode, include/osmocom/gmr1/l1/facch9.h:39-41): n bursts -> l2 n x 38, sacch n x 10 and
 * status n x 4 soft bits (optional), crc (0 = pass), conv.  ciph: optional n x 658 keystream bits.
 * TCH9 (gmr1_tch9_decode, include/osmocom/gmr1/l1/tch9.h:40-53): mode 0 2k4 / 1 4k8 / 2 9k6
 * (enum gmr1_tch9_mode) -> l2 of 18 / 30 / 60 bytes per burst.  The bursts are n_chan sequences of seq_len
 * consecutive bursts each (channel after channel): the depth-3 inter-burst de-interleaver
 * (gmr1_deinterleave_inter, l1/interleave.h:52-56) runs along every sequence from an all-zero state, so
 * burst i of a sequence yields the block sent two bursts earlier.  No CRC (the reference has none). */
int gmr1_hip_facch9_decode_batch_dev(void *stream, int n, const int8_t *ebits, const uint8_t *ciph,
                                     uint8_t *l2, int8_t *sacch, int8_t *status, int32_t *crc, int32_t *conv);
int gmr1_hip_facch9_decode_batch(int n, const int8_t *ebits, const uint8_t *ciph,
                                 uint8_t *l2, int8_t *sacch, int8_t *status, int32_t *crc, int32_t *conv);
int gmr1_hip_tch9_decode_batch_dev(void *stream, int n_chan, int seq_len, int mode, const int8_t *ebits,
                                   const uint8_t *ciph, uint8_t *l2, int8_t *sacch, int8_t *status, int32_t *conv);
int gmr1_hip_tch9_decode_batch(int n_chan, int seq_len, int mode, const int8_t *ebits, const uint8_t *ciph,
                               uint8_t *l2, int8_t *sacch, int8_t *status, int32_t *conv);

/* ---- the two layer-1 decoders gmr1_rx does not call ----------------------------------------------------
 * xCH over DC12 (gmr1_xch_dc12_decode, include/osmocom/gmr1/l1/xch_dc12.h:38): ebits n x 432 -> l2 n x 24,
 * crc[n] (0 = pass), conv[n] (optional).  K = 9 rate 1/3 tail-biting, 256 states.
 * RACH (gmr1_rach_decode, include/osmocom/gmr1/l1/rach.h:38-39): ebits n x 494, sb_mask[n] -> rach n x 18,
 * rv[n] (0 = both CRCs pass), conv[n] and crc[n x 2] = {CRC8, CRC12} (both optional).
 * _dev: device pointers (ebits 4-byte, l2 2-byte aligned), asynchronous on `stream`. */
int gmr1_hip_xch_dc12_decode_batch_dev(void *stream, int n, const int8_t *ebits, uint8_t *l2, int32_t *crc,
                                       int32_t *conv);
int gmr1_hip_xch_dc12_decode_batch(int n, const int8_t *ebits, uint8_t *l2, int32_t *crc, int32_t *conv);
int gmr1_hip_rach_decode_batch_dev(void *stream, int n, const int8_t *ebits, const uint8_t *sb_mask,
                                   uint8_t *rach, int32_t *rv, int32_t *conv, int32_t *crc);
int gmr1_hip_rach_decode_batch(int n, const int8_t *ebits, const uint8_t *sb_mask, uint8_t *rach, int32_t *rv,
                               int32_t *conv, int32_t *crc);

/* ---- transmit direction: channel encoders and modulator -------------------------------------------------
 * Batch forms of gmr1_{bcch,ccch,xch_dc12,facch3,tch3,facch9,tch9,rach}_encode (the headers under include/osmocom/gmr1/l1) and
 * of gmr1_pi4cxpsk_mod (sdr/pi4cxpsk.h:115-117).  Payloads are packed bytes, everything else one ubit per byte:
 *   bcch / xch_dc12: l2 n x 24 -> ebits n x 424 / 432;  ccch: l2 n x 24 -> n x 432
 *   facch3: l2 n x 10, bits_s n x 32, ciph n x 384 (optional) -> ebits n x 416 (four bursts of 104)
 *   tch3:   frames n x 2 x 10, bits_s n x 4, ciph n x 208 (optional), m -> ebits n x 212
 *   facch9: l2 n x 38, sacch n x 10, status n x 4, ciph n x 658 (optional) -> ebits n x 662
 *   tch9:   l2 n x 18 / 30 / 60 by mode; n = whole runs of seq_len consecutive bursts of one channel, the
 *           inter-burst interleaver starts empty at every run (as after gmr1_interleaver_init)
 *   rach:   rach n x 18, sb_mask n -> ebits n x 494
 *   mod:    ebits n x (ebits of the burst format) -> out n x len complex float32 symbols, sync sequence sync_id
 * _dev: device pointers, asynchronous on `stream` (mod_dev returns after its launch has finished). */
int gmr1_hip_bcch_encode_batch_dev(void *stream, int n, const uint8_t *l2, uint8_t *ebits);
int gmr1_hip_bcch_encode_batch(int n, const uint8_t *l2, uint8_t *ebits);
int gmr1_hip_ccch_encode_batch_dev(void *stream, int n, const uint8_t *l2, uint8_t *ebits);
int gmr1_hip_ccch_encode_batch(int n, const uint8_t *l2, uint8_t *ebits);
int gmr1_hip_xch_dc12_encode_batch_dev(void *stream, int n, const uint8_t *l2, uint8_t *ebits);
int gmr1_hip_xch_dc12_encode_batch(int n, const uint8_t *l2, uint8_t *ebits);
int gmr1_hip_facch3_encode_batch_dev(void *stream, int n, const uint8_t *l2, const uint8_t *bits_s, const uint8_t *ciph,
                                     uint8_t *ebits);
int gmr1_hip_facch3_encode_batch(int n, const uint8_t *l2, const uint8_t *bits_s, const uint8_t *ciph, uint8_t *ebits);
int gmr1_hip_tch3_encode_batch_dev(void *stream, int n, int m, const uint8_t *frames, const uint8_t *bits_s,
                                   const uint8_t *ciph, uint8_t *ebits);
int gmr1_hip_tch3_encode_batch(int n, int m, const uint8_t *frames, const uint8_t *bits_s, const uint8_t *ciph,
                               uint8_t *ebits);
int gmr1_hip_facch9_encode_batch_dev(void *stream, int n, const uint8_t *l2, const uint8_t *sacch, const uint8_t *status,
                                     const uint8_t *ciph, uint8_t *ebits);
int gmr1_hip_facch9_encode_batch(int n, const uint8_t *l2, const uint8_t *sacch, const uint8_t *status,
                                 const uint8_t *ciph, uint8_t *ebits);
int gmr1_hip_tch9_encode_batch_dev(void *stream, int mode, int n, int seq_len, const uint8_t *l2, const uint8_t *sacch,
                                   const uint8_t *status, const uint8_t *ciph, uint8_t *ebits);
int gmr1_hip_tch9_encode_batch(int mode, int n, int seq_len, const uint8_t *l2, const uint8_t *sacch,
                               const uint8_t *status, const uint8_t *ciph, uint8_t *ebits);
int gmr1_hip_rach_encode_batch_dev(void *stream, int n, const uint8_t *rach, const uint8_t *sb_mask, uint8_t *ebits);
int gmr1_hip_rach_encode_batch(int n, const uint8_t *rach, const uint8_t *sb_mask, uint8_t *ebits);
int gmr1_hip_mod_batch_dev(void *stream, int burst_id, int sync_id, int n, const uint8_t *ebits, float *out);
int gmr1_hip_mod_batch(int burst_id, int sync_id, int n, const uint8_t *ebits, float *out);
/* The position map (payload bits -> burst bits) the encoder kernel evaluates for one chain, copied into buf as
 * struct EncPlan of osmo-gmr_amd/csrc/gmr1_dev.h; returns its size (buf may be NULL) or -EINVAL.  Host-only,
 * works without a GPU: tests/test_tx_plan.py evaluates it bit by bit against the CPU oracle. */
enum gmr1_hip_enc_chain {
	GMR1_HIP_ENC_BCCH = 0, GMR1_HIP_ENC_CCCH, GMR1_HIP_ENC_FACCH3, GMR1_HIP_ENC_TCH3_M0, GMR1_HIP_ENC_TCH3_M1,
	GMR1_HIP_ENC_FACCH9, GMR1_HIP_ENC_TCH9_2K4, GMR1_HIP_ENC_TCH9_4K8, GMR1_HIP_ENC_TCH9_9K6, GMR1_HIP_ENC_RACH,
	GMR1_HIP_ENC_XCH_DC12, GMR1_HIP_ENC__COUNT
};
int gmr1_hip_encoder_plan(int chain, void *buf, int buf_len);

/* ---- FCCH acquisition ------------------------------------------------------
 * fcch_type: 0 gmr1_fcch_burst, 1 gmr1_fcch3_lband_burst, 2 gmr1_fcch3_sband_burst.
 * rough: n search windows of `len` samples each -> toa[i] (samples), rv[i] (0 / -errno).
 * fine / snr: n bursts of exactly burst_len*sps samples each.
 * The rough sweep keeps a grow-only device scratch per GPU (decimated streams +
 * partials, about len/sps*8 bytes per stream). */
int gmr1_hip_fcch_rough_batch_dev(void *stream, int fcch_type, int n, int sps, int len,
                                  const float *iq, const uint64_t *offset, const float *freq_shift,
                                  int32_t *toa, int32_t *rv);
int gmr1_hip_fcch_rough_batch(int fcch_type, int n, int sps, int len,
                              const float *iq, uint64_t iq_len, const uint64_t *offset,
                              const float *freq_shift, int32_t *toa, int32_t *rv);
/* rough_multi: peaks_toa is n x N, count[i] = peaks found or -EINVAL (fcch.c:425-427) */
int gmr1_hip_fcch_rough_multi_batch_dev(void *stream, int fcch_type, int n, int sps, int len,
                                        const float *iq, const uint64_t *offset, const float *freq_shift,
                                        int32_t *peaks_toa, int N, int32_t *count);
int gmr1_hip_fcch_rough_multi_batch(int fcch_type, int n, int sps, int len,
                                    const float *iq, uint64_t iq_len, const uint64_t *offset,
                                    const float *freq_shift, int32_t *peaks_toa, int N, int32_t *count);
int gmr1_hip_fcch_fine_batch_dev(void *stream, int fcch_type, int n, int sps,
                                 const float *iq, const uint64_t *offset, const float *freq_shift,
                                 int32_t *toa, float *freq_error);
int gmr1_hip_fcch_fine_batch(int fcch_type, int n, int sps,
                             const float *iq, uint64_t iq_len, const uint64_t *offset,
                             const float *freq_shift, int32_t *toa, float *freq_error);
int gmr1_hip_fcch_snr_batch_dev(void *stream, int fcch_type, int n, int sps,
                                const float *iq, const uint64_t *offset, const float *freq_shift,
                                float *snr);
int gmr1_hip_fcch_snr_batch(int fcch_type, int n, int sps,
                            const float *iq, uint64_t iq_len, const uint64_t *offset,
                            const float *freq_shift, float *snr);

/* Direct mode of the recorder script (utils/gmr1_rx_sdr.py:605-807, DirectOutputParameters / DirectOutputBranch): a few
 * carriers straight from the wideband stream, no filterbank.  Per carrier at freq_hz[i] from the centre:
 * filter.freq_xlating_fir_filter_ccc(decim1, low_pass(1, 1, .3 / decim1, .3 / decim1), f, fs), filter.fir_filter_ccc(decim2,
 * low_pass(1, 1, .45 / decim2, .1 / decim2)), pfb.arb_resampler_ccf(resamp, root_raised_cosine(32, ..., 0.35), 32) -- the
 * split chosen as the script's _select_decim does (2.0 Msps: 7, 6, 1.9656).  out: n_sel streams of *n_out complex64 at
 * 23400 x sps, out_stride complex samples apart.  -EINVAL for rates the reference itself cannot run (an exact multiple
 * of 23400 x sps) or whose resampler is longer than the kernel holds (e.g. 1.0 Msps: rate 0.468). */
int gmr1_hip_ddc_plan(double samp_rate, int sps, uint64_t n_in, int32_t *decim1, int32_t *decim2, double *resamp,
                      uint64_t *n_out);
int gmr1_hip_ddc_dev(void *stream, double samp_rate, int sps, const float *wide, uint64_t n_in, int n_sel,
                     const double *freq_hz, float *out, uint64_t out_stride, uint64_t *n_out);
int gmr1_hip_ddc(double samp_rate, int sps, const float *wide, uint64_t n_in, int n_sel, const double *freq_hz,
                 float *out, uint64_t out_stride, uint64_t *n_out);

/* ------------------------------------------------------------------------
 * Wideband capture -> per-ARFCN streams at sym_rate x sps (reference utils/gmr1_rx_sdr.py:391-602:
 * PFBBase = 2x oversampled polyphase channelizer over n_chans = (ceil(fs / 31.25 kHz) + 1) & ~1 channels
 * with firdes.low_pass(1, fs, 15.625 k, 7.8125 k); PFBOutputBranch = per-channel arbitrary resampler to
 * 23.4 k x sps with a 32-phase root-raised-cosine bank, alpha 0.35, 11 symbols).
 * Channel k is the carrier k x 31.25 kHz above the centre (k >= n_chans / 2: below, freq2index :478-485).
 * rotation: optional pre-rotation in rad / sample (:444-448).  chan_idx: n_sel channel numbers (host);
 * out: n_sel streams, out_stride complex samples apart; *n_out = samples written per stream.
 * n_chans even and <= 256 (64 channels / 2.0 Msps is the fast path).  A sample rate (whole Hz) that is not
 * n_chans x 31.25 kHz is first resampled to that rate by a 32-phase arbitrary resampler (:413-417, :453-461:
 * pfb.arb_resampler_ccf(rate, taps=None, flt_size=32)) -- a branch that cannot run in the reference as written
 * (it reads self.samp_rate before anything sets it); built to its evident intent, with the library's own
 * prototype for the resampler (GNU Radio's default design is not restatable: csrc/capi_chan.cpp,
 * design_pre_resampler).  The multi-ARFCN synthesizer (:567-576) is not built.
 * gmr1_hip_channelize_plan reports sizes: n_mid = 2x oversampled samples per channel.
 * gmr1_hip_channelize_planar_dev writes the n_sel streams POLYPHASE-PLANAR, the layout
 * gmr1_hip_rx_bcch_ccch_batch_planar_dev reads: sample m of stream i is flat sample g = i x out_stride + m and
 * goes to out_planes[(g % sps) * plane_stride + g / sps]  (plane_stride >= ceil(n_sel x out_stride / sps)). */
int gmr1_hip_channelize_plan(double samp_rate, int sps, uint64_t n_in,
                             int32_t *n_chans, uint64_t *n_mid, uint64_t *n_out);
int gmr1_hip_channelize_dev(void *stream, double samp_rate, int sps, const float *wide, uint64_t n_in,
                            float rotation, int n_sel, const int32_t *chan_idx,
                            float *out, uint64_t out_stride, uint64_t *n_out);
int gmr1_hip_channelize(double samp_rate, int sps, const float *wide, uint64_t n_in, float rotation,
                        int n_sel, const int32_t *chan_idx, float *out, uint64_t out_stride, uint64_t *n_out);
int gmr1_hip_channelize_planar_dev(void *stream, double samp_rate, int sps, const float *wide, uint64_t n_in,
                                   float rotation, int n_sel, const int32_t *chan_idx,
                                   float *out_planes, uint64_t out_stride, uint64_t plane_stride, uint64_t *n_out);

/* ------------------------------------------------------------------------
 * The gmr1_rx receive loop over many BCCH carriers (reference src/gmr1_rx.c:605-895 and
 * main() :897-975, one process per capture file there).
 *
 * Carrier i is the complex64 stream iq[offset[i] .. offset[i]+length[i]) at sps samples per
 * symbol (offset / length / arfcn / out / status / n_chains are HOST arrays; iq is a device
 * pointer for _dev, a host pointer otherwise).  For every carrier: FCCH acquisition
 * (fcch_single_init), multi-FCCH survivor selection (fcch_multi_process), then per chain the
 * BCCH / CCCH frame loop with its time / frequency / SI1 TDMA feedback (process_bcch, rx_bcch,
 * rx_ccch, bcch_tdma_align).  Every frame whose CRC passed comes back as one record -- what the
 * reference hands to gsmtap_sendmsg (gmr1_rx.c:793-795, 845-847) -- ordered by carrier, chain,
 * time.  n_records = records found (only the first max_records are stored).
 * status[i] = 0 or the negative value main() would have exited with for that carrier;
 * n_chains[i] = FCCH chains followed.  TCH follow-up after IMM.ASS is not performed.
 * arfcn may be NULL (records then carry the carrier index).
 * `out` of gmr1_hip_rx_run_dev may be pageable host memory, pinned / registered host memory (hipHostMalloc,
 * hipHostRegister) or DEVICE memory: the records are closed up on the device in the order above and copied once,
 * exactly as many as there are -- straight into a pinned or device buffer (no staging copy; a device buffer is what
 * gmr1_hip_rx_run_sharded sends from), through the library's pinned block for pageable memory. */
struct gmr1_hip_rx_record {
	uint16_t arfcn;
	uint8_t  chain;      /* FCCH chain within the carrier          */
	uint8_t  type;       /* 1 = GSMTAP_GMR1_BCCH, 2 = GSMTAP_GMR1_CCCH */
	uint32_t fn;
	uint8_t  tn;
	uint8_t  crc;        /* always 0: only frames whose CRC passed */
	uint8_t  len;        /* 24                                     */
	uint8_t  pad;
	int32_t  conv;       /* Viterbi path metric, as logged by the reference */
	uint8_t  l2[24];
};

int gmr1_hip_rx_run_dev(void *stream, int n_arfcn, int sps, const float *iq,
                        const uint64_t *offset, const uint64_t *length, const uint16_t *arfcn,
                        struct gmr1_hip_rx_record *out, int max_records, int *n_records,
                        int32_t *status, int32_t *n_chains);
int gmr1_hip_rx_run(int n_arfcn, int sps, const float *iq, uint64_t iq_len,
                    const uint64_t *offset, const uint64_t *length, const uint16_t *arfcn,
                    struct gmr1_hip_rx_record *out, int max_records, int *n_records,
                    int32_t *status, int32_t *n_chains);

/* The same with the TCH3 follow-up (gmr1_rx's optional tch.cfile and key arguments, gmr1_rx.c:355-600,
 * 897-975): tch holds, for every carrier, the traffic carrier an IMMEDIATE ASSIGNMENT on its CCCH points
 * to -- same offset[] / length[] layout and timing as iq; kc = n_arfcn x 8 key bytes (NULL: the all-zero
 * key the reference starts with).  After an assignment every frame's burst on the assigned timeslot is
 * classified (energy -> DKAB | burst; burst -> FACCH3 | speech by sync detection), FACCH3 bursts are
 * grouped by sync sequence and decoded (type 0x12 = GSMTAP_GMR1_TCH3 | GSMTAP_GMR1_FACCH, 10 bytes,
 * fn = fn of the flush - 3), with A5/1 deciphering once a message only decodes ciphered.  Speech bursts
 * come back as type 0x10 records of 20 bytes (two 10-byte frames; conv = conv0 | conv1 << 16) -- the
 * reference decodes them but only logs them.  tch = NULL is gmr1_hip_rx_run*. */
int gmr1_hip_rx_run_tch_dev(void *stream, int n_arfcn, int sps, const float *iq, const float *tch,
                            const uint64_t *offset, const uint64_t *length, const uint16_t *arfcn,
                            const uint8_t *kc,
                            struct gmr1_hip_rx_record *out, int max_records, int *n_records,
                            int32_t *status, int32_t *n_chains);
int gmr1_hip_rx_run_tch(int n_arfcn, int sps, const float *iq, const float *tch, uint64_t iq_len,
                        const uint64_t *offset, const uint64_t *length, const uint16_t *arfcn, const uint8_t *kc,
                        struct gmr1_hip_rx_record *out, int max_records, int *n_records,
                        int32_t *status, int32_t *n_chains);

/* The whole application: gmr1_rx with all its optional arguments (tch.cfile, key, tch_csd.cfile; gmr1_rx.c:897-975).
 * csd holds, per carrier, the carrier of the TCH9 (circuit switched data) channel an ASSIGNMENT COMMAND 1
 * on the FACCH3 points to -- same layout and timing as iq / tch.  From that message on every frame's NT9 burst
 * on the assigned timeslot is demodulated, deciphered (A5/1, always) and decoded: sync sequence 0 -> FACCH9
 * (type 0x1a = GSMTAP_GMR1_TCH9 | GSMTAP_GMR1_FACCH, 38 bytes, reported when the CRC passes), sync sequence 1 ->
 * TCH9 9k6 (type 0x18, 60 bytes, always reported -- no CRC; gmr1_rx.c:262-353).  Their payloads do not fit
 * the 40-byte record, so they come back as big records, ordered like the others. */
struct gmr1_hip_rx_big_record {
	uint16_t arfcn;
	uint8_t  chain, type;
	uint32_t fn;
	uint8_t  tn, crc, len, pad;
	int32_t  conv;
	uint8_t  l2[64];
};
/* Measurement aid: wall time, in microseconds, of the phases of the calling thread's last gmr1_hip_rx_run* call --
 * [0] FCCH acquisition (main() -> fcch_single_init / fcch_multi_process, gmr1_rx.c:605-744) incl. its decisions on the host,
 * [1] the frame loop (process_bcch, :852-895) from its first launch until its kernels are through, [2] the records to the
 * caller's buffer, [3] host work around the loop, [4] the traffic-channel passes. */
int gmr1_hip_rx_run_last_timing(double *us5);

int gmr1_hip_rx_run_full_dev(void *stream, int n_arfcn, int sps, const float *iq, const float *tch,
                             const float *csd, const uint64_t *offset, const uint64_t *length,
                             const uint16_t *arfcn, const uint8_t *kc,
                             struct gmr1_hip_rx_record *out, int max_records, int *n_records,
                             struct gmr1_hip_rx_big_record *big_out, int max_big, int *n_big,
                             int32_t *status, int32_t *n_chains);
int gmr1_hip_rx_run_full(int n_arfcn, int sps, const float *iq, const float *tch, const float *csd, uint64_t iq_len,
                         const uint64_t *offset, const uint64_t *length, const uint16_t *arfcn, const uint8_t *kc,
                         struct gmr1_hip_rx_record *out, int max_records, int *n_records,
                         struct gmr1_hip_rx_big_record *big_out, int max_big, int *n_big,
                         int32_t *status, int32_t *n_chains);

/* ---- AMBE speech decoder over many voice channels (reference: one `struct gmr1_codec` per channel, frames one call
 * at a time: include/osmocom/gmr1/codec/codec.h:37-45, src/codec) ----
 * frames [n_ch][n_frames][10] bytes, pcm [n_ch][n_frames][160] samples, rv (optional) [n_ch][n_frames]: 0, or -EINVAL
 * for a tone frame with an unassigned code.  One wavefront per channel walks that channel's frames in order.
 * `state`: gmr1_hip_codec_state_bytes() per channel, 16-byte aligned, opaque, carried between calls so a channel can
 * be decoded piecewise; gmr1_hip_codec_init_dev makes fresh decoders (the reference's gmr1_codec_alloc).
 * GMR1_HIP_CODEC_CLEARED: per-harmonic voicing above the harmonic count reads 0 instead of what the same subframe of
 * an earlier frame left there (DESIGN.md decision D9: the reference leaves it to its stack; the default is what its
 * own program gmr1_ambe_decode computes).
 * The host form copies frames in and samples out; state == NULL or GMR1_HIP_CODEC_FRESH: fresh decoders; otherwise
 * `state` (host memory) is read before and written after. */
#define GMR1_HIP_CODEC_CLEARED 1
#define GMR1_HIP_CODEC_FRESH   2
size_t gmr1_hip_codec_state_bytes(void);
int gmr1_hip_codec_init_dev(void *stream, int n_ch, void *state, int flags);
int gmr1_hip_codec_decode_batch_dev(void *stream, int n_ch, int n_frames, const uint8_t *frames, int16_t *pcm,
                                    int32_t *rv, void *state);
int gmr1_hip_codec_decode_batch(int n_ch, int n_frames, const uint8_t *frames, int16_t *pcm, int32_t *rv, void *state,
                                int flags);
/* The tables the library computes with the host's libm when it loads (cosine table, 2^f0log per pitch history, ...):
 * for tests; works without a GPU. */
int gmr1_hip_codec_host_tables(const void **image, size_t *bytes);
/* The kernel's restatement of glibc's powf and cosf, evaluated on the host (for tests; works without a GPU): which = 0:
 * out[i] = powf(2, x[i]); 1: powf(x[i], 0.25f) (NaN where the kernel would fall back to double precision); 2: cosf(x[i]). */
int gmr1_hip_codec_libm_check(int which, int n, const float *x, float *out);

/* The GSMTAP packet gmr1_gsmtap_makemsg (reference src/gsmtap.c:43-71, include/osmocom/gmr1/gsmtap.h:35-37)
 * builds for one record: 16-byte gsmtap_hdr + L2.  Returns the packet length (16 + rec->len) or
 * -EINVAL.  Host-only; works without a GPU.  with_arfcn = 0 leaves the arfcn field 0 as the reference does. */
int gmr1_hip_gsmtap_pack(const struct gmr1_hip_rx_record *rec, int with_arfcn, uint8_t *buf, int buf_len);
int gmr1_hip_gsmtap_pack_big(const struct gmr1_hip_rx_big_record *rec, int with_arfcn, uint8_t *buf, int buf_len);

#ifdef __cplusplus
}
#endif

#endif /* GMR1_HIP_H */
