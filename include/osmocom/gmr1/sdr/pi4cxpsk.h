/*
 * pi/2-CBPSK, pi/4-CBPSK and pi/4-CQPSK burst modem -- C API kept identical to
 * osmocom/osmo-gmr include/osmocom/gmr1/sdr/pi4cxpsk.h:43-117 (same struct
 * layouts, names, argument meaning, return values); implemented by
 * libgmr1_hip.so on an MI355X (one blocking H2D / kernel / D2H per call).
 */
#ifndef __OSMO_GMR1_SDR_PI4CXPSK_H__
#define __OSMO_GMR1_SDR_PI4CXPSK_H__

#include <stdint.h>
#include <osmocom/gmr1/compat.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GMR1_MAX_SYM_EBITS  2   /* encoded bits carried by one symbol   */
#define GMR1_MAX_SYNC       4   /* alternative sync sequences per burst */
#define GMR1_MAX_SYNC_SYMS  32  /* symbols in one sync chunk            */

struct gmr1_pi4cxpsk_symbol {
	short  idx;                       /* symbol number                    */
	ubit_t data[GMR1_MAX_SYM_EBITS];  /* bits it encodes                  */
	float  mod_phase;                 /* modulating phase                 */
	gmr1_cfloat mod_val;              /* e^(j mod_phase)                  */
};

struct gmr1_pi4cxpsk_modulation {
	float rotation;                      /* continuous rotation per symbol */
	int nbits;                           /* encoded bits per symbol        */
	struct gmr1_pi4cxpsk_symbol *syms;   /* indexed by symbol number       */
	struct gmr1_pi4cxpsk_symbol *bits;   /* indexed by bit pattern         */
};

extern struct gmr1_pi4cxpsk_modulation gmr1_pi2cbpsk;
extern struct gmr1_pi4cxpsk_modulation gmr1_pi4cbpsk;
extern struct gmr1_pi4cxpsk_modulation gmr1_pi4cqpsk;

struct gmr1_pi4cxpsk_sync {
	int pos;                             /* first symbol (-1 terminates a list) */
	int len;                             /* symbols                             */
	uint8_t syms[GMR1_MAX_SYNC_SYMS];    /* symbol numbers                      */
	struct osmo_cxvec *_ref;             /* unused by this implementation       */
};

struct gmr1_pi4cxpsk_data {
	int pos;                             /* first symbol (-1 terminates a list) */
	int len;                             /* symbols                             */
};

struct gmr1_pi4cxpsk_burst {
	struct gmr1_pi4cxpsk_modulation *mod;
	int guard_pre;
	int guard_post;
	int len;                             /* symbols, guards included            */
	int ebits;                           /* encoded bits carried                */
	struct gmr1_pi4cxpsk_sync *sync[GMR1_MAX_SYNC];
	struct gmr1_pi4cxpsk_data *data;
};

/* 0 on success, -errno otherwise (-1 when no sync sequence correlates) */
int gmr1_pi4cxpsk_demod(struct gmr1_pi4cxpsk_burst *burst_type,
                        struct osmo_cxvec *burst_in, int sps, float freq_shift,
                        sbit_t *ebits, int *sync_id_p, float *toa_p, float *freq_err_p);

/* which of the NULL-terminated candidate types (same length and modulation family) is this
 * burst?  e_toa >= 0 weights the decision by 1/|e_toa - toa|.  0 on success, -errno otherwise */
int gmr1_pi4cxpsk_detect(struct gmr1_pi4cxpsk_burst **burst_types, float e_toa,
                         struct osmo_cxvec *burst_in, int sps, float freq_shift,
                         int *bt_id_p, int *sync_id_p, float *toa_p);

/* 2 for BPSK, 4 for QPSK (x^2 vs x^4 line power), < 0 on error */
int gmr1_pi4cxpsk_mod_order(struct osmo_cxvec *burst_in, int sps, float freq_shift);

/* pi4cxpsk.h:115-117: burst bits -> burst_type->len symbols at one sample per symbol (guard = 0, training sequence
 * sync_id, pi/4 rotation applied); -ENOMEM when burst_out->max_len is too short, 0 on success */
int gmr1_pi4cxpsk_mod(struct gmr1_pi4cxpsk_burst *burst_type,
                      ubit_t *ebits, int sync_id, struct osmo_cxvec *burst_out);

#ifdef __cplusplus
}
#endif

#endif
