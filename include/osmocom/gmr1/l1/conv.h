/* GMR-1 convolutional codes (API of osmocom/osmo-gmr include/osmocom/gmr1/l1/conv.h:36-44).
 *
 * The trellis descriptions libosmocore's encoder / decoder take.  The GPU codecs of this library do not read them
 * (their trellises are built from the same generator polynomials at compile time); the objects are exported for the
 * other consumers of libgmr1-l1.  Every table equals the one printed in the reference's src/l1/conv.c (mechanically
 * compared in tests/test_ref_tables.py) -- including gmr1_conv_k9_14, whose printed table implements g3 without the D^5
 * term its comment lists. */
#ifndef __OSMO_GMR1_L1_CONV_H__
#define __OSMO_GMR1_L1_CONV_H__

#include <osmocom/gmr1/compat.h>
#ifdef GMR1_HIP_USE_SYSTEM_OSMOCOM
#include <osmocom/core/conv.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* .len = 0 and .term as in the reference: users copy the struct and specialise it (bcch.c:44-50, tch3.c:42-49) */
extern const struct osmo_conv_code gmr1_conv_k5_12;   /* K = 5, rate 1/2: BCCH, CCCH, FACCH9, TCH9 9k6      */
extern const struct osmo_conv_code gmr1_conv_k5_13;   /* K = 5, rate 1/3: TCH9 4k8                          */
extern const struct osmo_conv_code gmr1_conv_k5_14;   /* K = 5, rate 1/4: FACCH3, RACH                      */
extern const struct osmo_conv_code gmr1_conv_k5_15;   /* K = 5, rate 1/5: TCH9 2k4                          */
extern const struct osmo_conv_code gmr1_conv_k6_14;   /* K = 6, rate 1/4                                    */
extern const struct osmo_conv_code gmr1_conv_k9_12;   /* K = 9, rate 1/2                                    */
extern const struct osmo_conv_code gmr1_conv_k9_13;   /* K = 9, rate 1/3: xCH over DC12                     */
extern const struct osmo_conv_code gmr1_conv_k9_14;   /* K = 9, rate 1/4                                    */
extern const struct osmo_conv_code gmr1_conv_tch3;    /* K = 7, rate 1/2, tail-biting: TCH3 speech          */

#ifdef __cplusplus
}
#endif

#endif
