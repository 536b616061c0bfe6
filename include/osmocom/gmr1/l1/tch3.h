/* TCH3 speech channel decoding (API of osmocom/osmo-gmr include/osmocom/gmr1/l1/tch3.h:40-42) */
#ifndef __OSMO_GMR1_L1_TCH3_H__
#define __OSMO_GMR1_L1_TCH3_H__

#include <stdint.h>
#include <osmocom/gmr1/compat.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 212 soft bits of one NT3 speech burst -> two 10-byte speech frames (MSB first) and 4 status bits.
 * ciph: optional 208 keystream bits; m: multiplexing mode (0 interleaved, 1 sequential). */
void gmr1_tch3_decode(uint8_t *frame0, uint8_t *frame1, ubit_t *bits_s,
                      const sbit_t *bits_e, const ubit_t *ciph, int m,
                      int *conv0_rv, int *conv1_rv);

#ifdef __cplusplus
}
#endif

#endif
