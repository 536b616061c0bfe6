/* oracle/orc_ambe.h -- TEST INFRASTRUCTURE ONLY: the CPU oracle of the AMBE speech decoder (see orc_ambe.c). */
#pragma once
#include <stddef.h>
#include <stdint.h>

/* one 10 ms subframe's parameters (reference src/codec/private.h:70-82) */
struct orc_ambe_sub {
	float f0, f0log, w0;
	int L, Lb[4];
	int band_v[8];          /* voicing of the eight bands */
	int V[56];              /* per harmonic */
	float gain;
	float Mlog[56], M[56];
};

/* decoder state carried from frame to frame (private.h:84-111) */
struct orc_ambe_dec {
	float tone_ph1, tone_ph2;
	struct orc_ambe_sub prev;
	int16_t u_last;
	float uw_last[121];
	float psi1;
	float phi[56];
	float SE;
	/* not in the reference's struct: see decision D9 in orc_ambe.c */
	int cleared;
	int V_slot[2][56];
};

void orc_ambe_init(struct orc_ambe_dec *d);
void orc_ambe_set_cleared(struct orc_ambe_dec *d, int on);
int orc_ambe_decode_frame(struct orc_ambe_dec *d, int16_t *pcm, int N, const uint8_t *frame, int bad);
int orc_ambe_decode_dtx(struct orc_ambe_dec *d, int16_t *pcm, int N);
int orc_ambe_decode_stream(struct orc_ambe_dec *d, const uint8_t *frames, int n, int16_t *pcm, int *rv);
size_t orc_ambe_state_size(void);

/* pieces the host-table tests of the product compare against */
void orc_ambe_unpack(const uint8_t *frame, unsigned out[14]);
float orc_ambe_f0log_sf0(float before, float now, int rule);
float orc_ambe_f0log_sf1(int pitch);
int orc_ambe_harmonics(float f0);
int orc_ambe_tone_ampl(int log_ampl);
float orc_ambe_cos_entry(int i);
float orc_ambe_pow2(float x);
float orc_ambe_log2_int(int L);
void orc_ambe_powf_array(int n, const float *x, float y, int x_is_base, float *out);
void orc_ambe_cosf_array(int n, const float *x, float *out);
