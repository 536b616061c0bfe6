/*
 * FCCH (frequency correction channel) acquisition -- C API kept identical to
 * osmocom/osmo-gmr include/osmocom/gmr1/sdr/fcch.h:36-61; implemented by
 * libgmr1_hip.so on an MI355X (blocking H2D / kernels / D2H per call).
 */
#ifndef __OSMO_GMR1_SDR_FCCH_H__
#define __OSMO_GMR1_SDR_FCCH_H__

#include <osmocom/gmr1/compat.h>

#ifdef __cplusplus
extern "C" {
#endif

struct gmr1_fcch_burst {
	float freq;   /* chirp sweep range           */
	int len;      /* burst duration in symbols   */
};

extern const struct gmr1_fcch_burst gmr1_fcch_burst;
extern const struct gmr1_fcch_burst gmr1_fcch3_lband_burst;
extern const struct gmr1_fcch_burst gmr1_fcch3_sband_burst;

/* coarse timing: position (samples) of the strongest FCCH in the window; 0 / -errno */
int gmr1_fcch_rough(const struct gmr1_fcch_burst *burst_type,
                    struct osmo_cxvec *search_win_in, int sps, float freq_shift,
                    int *toa);

/* multi-FCCH detection on >= 650 ms of signal: up to N positions ranked by power, duplicates one
 * BCCH period apart removed.  Returns the number found (>= 0) or -errno (-EINVAL: window too short
 * or no consistent 320 ms periodicity) */
int gmr1_fcch_rough_multi(const struct gmr1_fcch_burst *burst_type,
                          struct osmo_cxvec *search_win_in, int sps, float freq_shift,
                          int *toa, int N);

/* fine timing + frequency error (rad/symbol) on exactly len*sps samples; -EINVAL otherwise */
int gmr1_fcch_fine(const struct gmr1_fcch_burst *burst_type,
                   struct osmo_cxvec *burst_in, int sps, float freq_shift,
                   int *toa, float *freq_error);

/* SNR estimate (top-2 over bins 5,6 of the dual-chirp spectrum) on exactly len*sps samples */
int gmr1_fcch_snr(const struct gmr1_fcch_burst *burst_type,
                  struct osmo_cxvec *burst_in, int sps, float freq_shift,
                  float *snr);

#ifdef __cplusplus
}
#endif

#endif
