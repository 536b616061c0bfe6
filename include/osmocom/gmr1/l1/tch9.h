/* TCH9 channel decoding (API of osmocom/osmo-gmr include/osmocom/gmr1/l1/tch9.h:40-53) */
#ifndef __OSMO_GMR1_L1_TCH9_H__
#define __OSMO_GMR1_L1_TCH9_H__

#include <stdint.h>
#include <osmocom/gmr1/compat.h>

#ifdef __cplusplus
extern "C" {
#endif

struct gmr1_interleaver;

enum gmr1_tch9_mode {
	GMR1_TCH9_2k4,
	GMR1_TCH9_4k8,
	GMR1_TCH9_9k6,
	GMR1_TCH9_MAX
};

/* 662 soft bits of one NT9 burst -> the block sent two bursts earlier (18 / 30 / 60 bytes by mode, LSB
 * first), 10 SACCH and 4 status soft bits.  ciph: optional 658 keystream bits.  il carries the previous
 * bursts of the channel (gmr1_interleaver_init(il, 3, 648)).  Returns void like the reference: a device
 * failure leaves l2 zeroed and is reported through gmr1_hip_last_error(). */
void gmr1_tch9_decode(uint8_t *l2, sbit_t *bits_sacch, sbit_t *bits_status,
                      const sbit_t *bits_e, enum gmr1_tch9_mode mode,
                      const ubit_t *ciph, struct gmr1_interleaver *il,
                      int *conv_rv);

/* tch9.h:47-49: 18 / 30 / 60 bytes -> 662 burst bits through the depth-3 inter-burst interleaver `il`
 * (gmr1_interleaver_init(il, 3, 648); one object per direction and channel) */
void gmr1_tch9_encode(ubit_t *bits_e, const uint8_t *l2, enum gmr1_tch9_mode mode,
                      const ubit_t *bits_sacch, const ubit_t *bits_status,
                      const ubit_t *ciph, struct gmr1_interleaver *il);

#ifdef __cplusplus
}
#endif

#endif
