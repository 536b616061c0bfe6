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

/* tch3.h:37-39: two 10-byte speech frames -> 212 burst bits.  NOTE: the reference's encoder calls osmo_conv_encode with
 * input and output swapped (src/l1/tch3.c:81), so its own output is not a TCH3 burst; this is the encoder
 * gmr1_tch3_decode inverts (48 class-1 bits through the tail-biting K=7 code, 32 class-2 bits as they are). */
void gmr1_tch3_encode(ubit_t *bits_e, const uint8_t *frame0, const uint8_t *frame1,
                      const ubit_t *bits_s, const ubit_t *ciph, int m);

#ifdef __cplusplus
}
#endif

#endif
