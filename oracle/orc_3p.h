/*
 * oracle/orc_3p.h -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the third-party arithmetic the reference's hot path
 * delegates to libraries that are NOT vendored under /root/reference and are
 * not installed in this image:
 *
 *   - libosmocore (>= 0.4.1, only a floor is given: configure.ac:23):
 *       osmo_conv_encode / osmo_conv_decode (generic decoder, src/conv.c),
 *       osmo_crc16gen_{set,check}_bits, osmo_{p,u}bit2{u,p}bit[_ext]
 *   - libosmo-dsp (no version constraint: configure.ac:24):
 *       osmo_cxvec_sig_normalize / correlate / peak_energy_find / peaks_scan /
 *       rotate / scale / convolve, osmo_sinc
 *   - FFTW3f (>= 3.2.0: configure.ac:25): plain forward DFT (fcch.c:583-589)
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
 * (SURVEY.md section 4 / 8c), and none of these libraries can be built or run
 * here, so these are restatements of the published algorithms written from
 * their documented behaviour.  Every place where a detail had to be decided
 * is marked "DECISION Dn" below and listed in DESIGN.md.
 *
 * All symbols carry the orc_ prefix so the oracle can be loaded next to the
 * product library in one process without symbol clashes.
 */
#ifndef ORC_3P_H
#define ORC_3P_H

#include <complex.h>
#include <stdint.h>

typedef int8_t  orc_sbit_t;  /* soft bit: +127 ~ sure 0, -127 ~ sure 1, 0 = erasure */
typedef uint8_t orc_ubit_t;  /* unpacked hard bit 0/1 */
typedef float complex orc_cf;

/* ---- convolutional codes (libosmocore conv.h semantics) ---------------- */

enum orc_conv_term { ORC_TERM_FLUSH = 0, ORC_TERM_TRUNCATION, ORC_TERM_TAIL_BITING };

struct orc_conv_code {
	int N;                       /* output bits per input bit            */
	int K;                       /* constraint length                    */
	int len;                     /* number of data bits                  */
	enum orc_conv_term term;
	uint8_t next_output[256][2]; /* [state][bit] -> N-bit word, MSB = g0 */
	uint8_t next_state[256][2];
	int n_punct;                 /* number of punctured positions        */
	int punct[1024];             /* ascending, -1 terminated             */
};

/* Build a feed-forward code from generator polynomials (bit i = D^i). */
void orc_conv_make(struct orc_conv_code *c, int N, int K, int len,
                   enum orc_conv_term term, const unsigned *polys);
int  orc_conv_output_length(const struct orc_conv_code *c);
void orc_conv_encode(const struct orc_conv_code *c, const orc_ubit_t *in, orc_ubit_t *out);
int  orc_conv_decode(const struct orc_conv_code *c, const orc_sbit_t *in, orc_ubit_t *out);
/* decision D1b (orc_3p_acc.c): libosmocore's accelerated decoder for K in {5,7}, N in {2,3,4}.  orc_conv_set_mode(1)
 * makes orc_conv_decode use it for those codes (process-wide, tests only); the default 0 is D1 for every code. */
void orc_conv_set_mode(int mode);
int  orc_conv_get_mode(void);
int  orc_conv_acc_applies(const struct orc_conv_code *c);
int  orc_conv_decode_acc(const struct orc_conv_code *c, const orc_sbit_t *in, orc_ubit_t *out);

/* ---- CRC / bit packing ------------------------------------------------- */

struct orc_crc_code { int bits; uint32_t poly, init, remainder; };
uint32_t orc_crc_compute_bits(const struct orc_crc_code *c, const orc_ubit_t *in, int len);
void     orc_crc_set_bits(const struct orc_crc_code *c, const orc_ubit_t *in, int len, orc_ubit_t *crc);
int      orc_crc_check_bits(const struct orc_crc_code *c, const orc_ubit_t *in, int len, const orc_ubit_t *crc);

void orc_pbit2ubit_lsb(orc_ubit_t *out, const uint8_t *in, int n);  /* ..._ext(lsb_mode=1) */
void orc_ubit2pbit_lsb(uint8_t *out, const orc_ubit_t *in, int n);
void orc_pbit2ubit_msb(orc_ubit_t *out, const uint8_t *in, int n);
void orc_ubit2pbit_msb(uint8_t *out, const orc_ubit_t *in, int n);

/* ---- complex vector math (libosmo-dsp cxvec_math semantics) ------------ */

enum orc_peak_alg { ORC_PEAK_WEIGH_WIN = 0, ORC_PEAK_EARLY_LATE = 2 };

float orc_sinc(float x);
/* out has len/decim entries; returns that count */
int   orc_sig_normalize(const orc_cf *sig, int len, int decim, float freq_shift, orc_cf *out);
/* out has g_len - f_len*step + 1 entries; returns that count */
int   orc_correlate(const orc_cf *f, int f_len, const orc_cf *g, int g_len, int step, orc_cf *out);
orc_cf orc_interpolate_point(const orc_cf *cv, int len, float pos);
void  orc_peak_set_stop_shift(int steps);   /* tests only: D3's bisection runs `steps` halvings longer (< 0: shorter) */
float orc_peak_energy_find(const orc_cf *cv, int len, int win, enum orc_peak_alg alg, orc_cf *peak_val);
void  orc_peaks_scan(const orc_cf *cv, int len, int *idx, int N);
void  orc_rotate(orc_cf *v, int len, float rps);
void  orc_scale(orc_cf *v, int len, orc_cf s);
/* CONV_NO_DELAY convolution, real taps; out same length as g */
void  orc_convolve_nodelay_real(const float *f, int f_len, const orc_cf *g, int g_len, orc_cf *out);
/* unnormalised forward DFT, in place (FFTW_FORWARD) */
void  orc_dft_forward(orc_cf *v, int len);

#endif
