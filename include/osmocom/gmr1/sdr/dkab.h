/* DKAB (dual keep-alive burst) demodulation (API of osmocom/osmo-gmr include/osmocom/gmr1/sdr/dkab.h:39-41) */
#ifndef __OSMO_GMR1_SDR_DKAB_H__
#define __OSMO_GMR1_SDR_DKAB_H__

#include <stdint.h>
#include <osmocom/gmr1/compat.h>

#define GMR1_DKAB_SYMS (39*3)

#ifdef __cplusplus
extern "C" {
#endif

/* burst_in: GMR1_DKAB_SYMS * sps samples plus a search window; freq_shift in rad/symbol; p = DKAB
 * position.  Returns 0 (found: 8 soft bits in ebits), 1 (not found), -errno.  *toa_p is written
 * whenever the search ran. */
int gmr1_dkab_demod(struct osmo_cxvec *burst_in, int sps, float freq_shift, int p,
                    sbit_t *ebits, float *toa_p);

#ifdef __cplusplus
}
#endif

#endif
