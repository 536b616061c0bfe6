/*
 * osmocom/gmr1/compat.h -- the few libosmocore / libosmo-dsp TYPES that appear
 * in the GMR-1 C API signatures (sbit_t, ubit_t, struct osmo_cxvec).
 *
 * When the real libraries are installed, build with
 * -DGMR1_HIP_USE_SYSTEM_OSMOCOM and their headers are used instead; the
 * layouts below follow them (libosmocore include/osmocom/core/bits.h,
 * libosmo-dsp include/osmocom/dsp/cxvec.h) so the ABI is the same either way.
 */
#ifndef OSMO_GMR1_COMPAT_H
#define OSMO_GMR1_COMPAT_H

#include <stdint.h>

#ifdef GMR1_HIP_USE_SYSTEM_OSMOCOM
#include <osmocom/core/bits.h>
#include <osmocom/dsp/cxvec.h>
/* (l1/conv.h and l1/crc.h pull in <osmocom/core/conv.h> / <osmocom/core/crcgen.h> themselves in this mode) */
#ifdef __cplusplus
typedef struct { float re, im; } gmr1_cfloat;
#else
#include <complex.h>
typedef float complex gmr1_cfloat;
#endif
#else /* own definitions */

typedef int8_t  sbit_t;   /* soft bit: +127 = confident 0, -127 = confident 1, 0 = unknown */
typedef uint8_t ubit_t;   /* unpacked bit, 0 or 1 */
typedef uint8_t pbit_t;   /* packed bits */

#ifdef __cplusplus
typedef struct { float re, im; } gmr1_cfloat;   /* same layout as C99 float complex */
#else
#include <complex.h>
typedef float complex gmr1_cfloat;
#endif

#define CXVEC_FLG_REAL_ONLY (1 << 0)

struct osmo_cxvec {
	int len;            /* valid samples              */
	int max_len;        /* capacity                   */
	int flags;          /* CXVEC_FLG_*                */
	gmr1_cfloat *data;  /* samples (may point at _data) */
	gmr1_cfloat _data[0];
};

/* libosmocore's code description types, as the reference's l1/conv.h, l1/punct.h and l1/crc.h use them
 * (include/osmocom/core/conv.h, crcgen.h) [3P-recollection: field order as of libosmocore >= 0.4.1, where `term`
 * follows `len`; the reference only ever names the fields, conv.c:138-145] */
enum osmo_conv_term {
	CONV_TERM_FLUSH = 0,      /* K - 1 zero bits appended                       */
	CONV_TERM_TRUNCATION,     /* stops after the last data bit                  */
	CONV_TERM_TAIL_BITING     /* register preset with the last K - 1 data bits  */
};

struct osmo_conv_code {
	int N;                               /* coded bits per data bit                */
	int K;                               /* constraint length                      */
	int len;                             /* data bits                              */
	enum osmo_conv_term term;
	const uint8_t (*next_output)[2];     /* [state][bit] -> N-bit word, MSB = g0   */
	const uint8_t (*next_state)[2];      /* [state][bit]                           */
	const uint8_t *next_term_output;     /* flush transitions of recursive codes   */
	const uint8_t *next_term_state;
	const int *puncture;                 /* ascending punctured positions, -1 ends */
};

struct osmo_crc8gen_code  { int bits; uint8_t  poly, init, remainder; };
struct osmo_crc16gen_code { int bits; uint16_t poly, init, remainder; };

#endif /* GMR1_HIP_USE_SYSTEM_OSMOCOM */

#endif /* OSMO_GMR1_COMPAT_H */
