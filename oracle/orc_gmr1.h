/*
 * oracle/orc_gmr1.h -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the reference's GMR-1 receive hot path (SURVEY.md
 * section 8a rows a1-a21).  Function names are the reference's with the
 * gmr1_ prefix replaced by orc_; each definition cites the reference
 * file:line it follows.  PARITY UNPINNED (see orc_3p.h): the reference
 * has no tests or vectors and cannot be built in this image because its
 * arithmetic lives in libosmocore / libosmo-dsp / FFTW3f, which are absent.
 */
#ifndef ORC_GMR1_H
#define ORC_GMR1_H

#include "orc_3p.h"

/* ---- sdr: burst descriptions (reference include/osmocom/gmr1/sdr/pi4cxpsk.h:43-98) */

#define ORC_MAX_SYNC      4
#define ORC_MAX_CHUNKS    8
#define ORC_MAX_SYNC_SYMS 32

struct orc_chunk { int pos, len; uint8_t syms[ORC_MAX_SYNC_SYMS]; };

struct orc_burst {
	const char *name;
	float rotation;     /* per-symbol rotation (rad)      */
	int nbits;          /* ebits per symbol (1 or 2)      */
	int guard_pre, guard_post;
	int len;            /* symbols incl. guard            */
	int ebits;
	int n_sync;         /* number of alternative sync sequences */
	int n_sync_chunks[ORC_MAX_SYNC];
	struct orc_chunk sync[ORC_MAX_SYNC][ORC_MAX_CHUNKS];
	int n_data;
	struct orc_chunk data[ORC_MAX_CHUNKS];   /* syms unused */
};

enum orc_burst_id {
	ORC_BURST_BCCH = 0, ORC_BURST_DC2, ORC_BURST_DC6, ORC_BURST_DC12,
	ORC_BURST_NT3_SPEECH, ORC_BURST_NT3_FACCH, ORC_BURST_NT6, ORC_BURST_NT9,
	ORC_BURST_RACH, ORC_BURST_SDCCH, ORC_BURST__COUNT
};

const struct orc_burst *orc_burst_get(int id);

int orc_pi4cxpsk_demod(const struct orc_burst *bt, const orc_cf *in, int in_len,
                       int sps, float freq_shift, orc_sbit_t *ebits,
                       int *sync_id_p, float *toa_p, float *freq_err_p,
                       float *ssyms_out /* optional: bt->len soft symbols */);
int orc_pi4cxpsk_detect(const struct orc_burst *const *bts, int n_bts, float e_toa,
                        const orc_cf *in, int in_len, int sps, float freq_shift,
                        int *bt_id_p, int *sync_id_p, float *toa_p);
int orc_pi4cxpsk_mod_order(const orc_cf *in, int in_len, int sps, float freq_shift);
int orc_pi4cxpsk_mod(const struct orc_burst *bt, const orc_ubit_t *ebits, int sync_id, orc_cf *out);

/* ---- sdr: FCCH (reference include/osmocom/gmr1/sdr/fcch.h:36-61) */

struct orc_fcch_burst { float freq; int len; };
extern const struct orc_fcch_burst orc_fcch_burst, orc_fcch3_lband_burst, orc_fcch3_sband_burst;

int orc_fcch_rough(const struct orc_fcch_burst *bt, const orc_cf *in, int in_len,
                   int sps, float freq_shift, int *toa);
int orc_fcch_rough_multi(const struct orc_fcch_burst *bt, const orc_cf *in, int in_len,
                         int sps, float freq_shift, int *peaks_toa, int N);
int orc_fcch_fine(const struct orc_fcch_burst *bt, const orc_cf *in, int in_len,
                  int sps, float freq_shift, int *toa, float *freq_error);
int orc_fcch_snr(const struct orc_fcch_burst *bt, const orc_cf *in, int in_len,
                 int sps, float freq_shift, float *snr);

/* ---- l1 primitives */

void orc_scramble_sbit(orc_sbit_t *out, const orc_sbit_t *in, int len);
void orc_scramble_ubit(orc_ubit_t *out, const orc_ubit_t *in, int len);
void orc_interleave_intra(void *out, const void *in, int N);
void orc_deinterleave_intra(void *out, const void *in, int N);

/* ---- l1 channel codecs */

void orc_bcch_encode(orc_ubit_t *bits_e, const uint8_t *l2);
int  orc_bcch_decode(uint8_t *l2, const orc_sbit_t *bits_e, int *conv_rv);
void orc_ccch_encode(orc_ubit_t *bits_e, const uint8_t *l2);
int  orc_ccch_decode(uint8_t *l2, const orc_sbit_t *bits_e, int *conv_rv);
void orc_facch3_encode(orc_ubit_t *bits_e, const uint8_t *l2,
                       const orc_ubit_t *bits_s, const orc_ubit_t *ciph);
int  orc_facch3_decode(uint8_t *l2, orc_ubit_t *bits_s, const orc_sbit_t *bits_e,
                       const orc_ubit_t *ciph, int *conv_rv);
/* NOTE: the reference's tch3 ENCODER passes its conv in/out swapped
 * (tch3.c:81); orc_tch3_encode is the evidently intended encoder. */
void orc_tch3_encode(orc_ubit_t *bits_e, const uint8_t *frame0, const uint8_t *frame1,
                     const orc_ubit_t *bits_s, const orc_ubit_t *ciph, int m);
void orc_tch3_decode(uint8_t *frame0, uint8_t *frame1, orc_ubit_t *bits_s,
                     const orc_sbit_t *bits_e, const orc_ubit_t *ciph, int m,
                     int *conv0_rv, int *conv1_rv);

/* ---- TCH3 follow-up pieces: DKAB demodulator (reference include/osmocom/gmr1/sdr/dkab.h:39-41) and
 *      A5 keystream (reference include/osmocom/gmr1/l1/a5.h:37-41) */
int  orc_dkab_demod(const orc_cf *in, int in_len, int sps, float freq_shift, int p,
                    orc_sbit_t *ebits, float *toa_p);
void orc_a5(int n, const uint8_t *key, uint32_t fn, int nbits, orc_ubit_t *dl, orc_ubit_t *ul);
void orc_a5_1(const uint8_t *key, uint32_t fn, int nbits, orc_ubit_t *dl, orc_ubit_t *ul);

/* ---- NT9 codecs: FACCH9 (reference include/osmocom/gmr1/l1/facch9.h:36-41), TCH9 (l1/tch9.h:40-53),
 *      inter-burst interleaver (l1/interleave.h:40-56) */
enum orc_tch9_mode { ORC_TCH9_2k4 = 0, ORC_TCH9_4k8, ORC_TCH9_9k6 };   /* l1/tch9.h:40-45 */
#define ORC_IL_MAXK 648
struct orc_interleaver { int N, K, n; uint8_t bits_cpp[3 * ORC_IL_MAXK]; };
void orc_interleaver_init(struct orc_interleaver *il, int N, int K);
void orc_interleave_inter(struct orc_interleaver *il, void *bits_epp, const void *bits_ep);
void orc_deinterleave_inter(struct orc_interleaver *il, void *bits_ep, const void *bits_epp);
void orc_facch9_encode(orc_ubit_t *bits_e, const uint8_t *l2, const orc_ubit_t *bits_sacch,
                       const orc_ubit_t *bits_status, const orc_ubit_t *ciph);
int  orc_facch9_decode(uint8_t *l2, orc_sbit_t *bits_sacch, orc_sbit_t *bits_status,
                       const orc_sbit_t *bits_e, const orc_ubit_t *ciph, int *conv_rv);
void orc_tch9_encode(orc_ubit_t *bits_e, const uint8_t *l2, int mode, const orc_ubit_t *bits_sacch,
                     const orc_ubit_t *bits_status, const orc_ubit_t *ciph, struct orc_interleaver *il);
void orc_tch9_decode(uint8_t *l2, orc_sbit_t *bits_sacch, orc_sbit_t *bits_status, const orc_sbit_t *bits_e,
                     int mode, const orc_ubit_t *ciph, struct orc_interleaver *il, int *conv_rv);
int  orc_tch9_punct(int mode, int *idx);

/* ---- xCH over DC12 (reference include/osmocom/gmr1/l1/xch_dc12.h:37-38) and RACH (l1/rach.h:37-39) */
void orc_xch_dc12_encode(orc_ubit_t *bits_e, const uint8_t *l2);
int  orc_xch_dc12_decode(uint8_t *l2, const orc_sbit_t *bits_e, int *conv_rv);
void orc_rach_encode(orc_ubit_t *bits_e, const uint8_t *rach, uint8_t sb_mask);
int  orc_rach_decode(uint8_t *rach, const orc_sbit_t *bits_e, uint8_t sb_mask, int *conv_rv, int *crc_rv);

/* ---- the specialised convolutional code (trellis + punctured positions) each chain uses, for the table tests */
const struct orc_conv_code *orc_l1_code(int which);    /* 0 BCCH / CCCH, 1 FACCH3, 2 TCH3 speech */
const struct orc_conv_code *orc_nt9_code(int which);   /* 0 FACCH9, 1 TCH9 2k4, 2 TCH9 4k8, 3 TCH9 9k6 */
const struct orc_conv_code *orc_xch_code(int which);   /* 0 xCH over DC12, 1 RACH */

/* ---- batch drivers used by tests and by bench.py's cpu_baseline leg only */

/* kind: 0 = BCCH (orc_burst BCCH + bcch_decode), 1 = CCCH (DC6 + ccch_decode) */
void orc_demod_decode_batch(int n, const orc_cf *iq, const uint64_t *offset,
                            const uint8_t *kind, int sps, const float *freq_shift,
                            uint8_t *l2 /* n*24 */, int32_t *crc, int32_t *conv,
                            float *toa, float *freq_err,
                            orc_sbit_t *ebits /* optional n*432 */,
                            float *ssyms /* optional n*234 */, int32_t *rv);

/* ---- receive control loop of one BCCH carrier (reference src/gmr1_rx.c:605-895) */

#define ORC_RX_TYPE_BCCH 1      /* GSMTAP_GMR1_BCCH */
#define ORC_RX_TYPE_CCCH 2      /* GSMTAP_GMR1_CCCH */
#define ORC_RX_TYPE_TCH3 0x10       /* GSMTAP_GMR1_TCH3: two 10-byte speech frames (the reference only logs them) */
#define ORC_RX_TYPE_TCH3_FACCH 0x12 /* GSMTAP_GMR1_TCH3 | GSMTAP_GMR1_FACCH, 10 bytes, fn = last burst's fn - 3 */

struct orc_rx_record {          /* what gmr1_rx hands to GSMTAP for a frame whose CRC passed */
	uint16_t arfcn;
	uint8_t  chain, type;
	uint32_t fn;
	uint8_t  tn, crc, len, pad;
	int32_t  conv;
	uint8_t  l2[24];
};

int orc_rx_run(const orc_cf *iq, int len, int sps, int arfcn,
               struct orc_rx_record *out, int max_records, int *n_records, int *n_chains);
int orc_rx_run_tch(const orc_cf *iq, const orc_cf *tch, int len, int sps, int arfcn, const uint8_t *kc,
                   struct orc_rx_record *out, int max_records, int *n_records, int *n_chains);

#define ORC_RX_TYPE_TCH9 0x18       /* GSMTAP_GMR1_TCH9: one 60-byte 9k6 block */
#define ORC_RX_TYPE_TCH9_FACCH 0x1a /* GSMTAP_GMR1_TCH9 | GSMTAP_GMR1_FACCH, 38 bytes */
struct orc_rx_big_record {          /* NT9 payloads do not fit the 24 payload bytes of orc_rx_record */
	uint16_t arfcn;
	uint8_t  chain, type;
	uint32_t fn;
	uint8_t  tn, crc, len, pad;
	int32_t  conv;
	uint8_t  l2[64];
};
int orc_rx_run_full(const orc_cf *iq, const orc_cf *tch, const orc_cf *csd, int len, int sps, int arfcn,
                    const uint8_t *kc, struct orc_rx_record *out, int max_records, int *n_records,
                    struct orc_rx_big_record *big, int max_big, int *n_big, int *n_chains);

#endif
