/* GMR-1 puncturing schemes (API of osmocom/osmo-gmr include/osmocom/gmr1/l1/punct.h:37-106).
 *
 * A scheme is a mask over L trellis steps of an N-output code, 0 = the coded bit is not sent (r of them);
 * gmr1_puncturer_generate() expands pre / main / post schemes into the ascending list of punctured positions that
 * struct osmo_conv_code carries.  Host code only: the GPU codecs fold the same positions into their gather maps. */
#ifndef __OSMO_GMR1_L1_PUNCT_H__
#define __OSMO_GMR1_L1_PUNCT_H__

#include <stdint.h>
#include <osmocom/gmr1/compat.h>

#ifdef __cplusplus
extern "C" {
#endif

struct gmr1_puncturer {
	int r;                  /* punctured bits in the mask            */
	int L;                  /* trellis steps the mask covers         */
	int N;                  /* outputs per step of the code (1/N)    */
	const uint8_t mask[];   /* L * N entries, 0 = punctured          */
};

struct osmo_conv_code;

/* punct.c:48-133.  Sets code->puncture to a malloc'd array (the caller frees it): positions of punct_pre's zeros over
 * the first block, punct_main's repeated `repeat` times (0: as often as fits), punct_post's over the last block;
 * -1 terminated.  Returns 0, -EINVAL when a scheme's N differs from the code's, -ENOMEM. */
int gmr1_puncturer_generate(struct osmo_conv_code *code,
                            const struct gmr1_puncturer *punct_pre,
                            const struct gmr1_puncturer *punct_main,
                            const struct gmr1_puncturer *punct_post,
                            int repeat);

/* the schemes of GMR-1 05.003, named <code>_<scheme> as in the reference */
extern const struct gmr1_puncturer gmr1_punct_k5_12_P23;
extern const struct gmr1_puncturer gmr1_punct_k5_12_P25;
extern const struct gmr1_puncturer gmr1_punct_k5_12_Ps25;
extern const struct gmr1_puncturer gmr1_punct_k5_12_P311;
extern const struct gmr1_puncturer gmr1_punct_k5_12_P412;
extern const struct gmr1_puncturer gmr1_punct_k5_12_Ps412;
extern const struct gmr1_puncturer gmr1_punct_k5_12_P12;
extern const struct gmr1_puncturer gmr1_punct_k5_12_Ps12;
extern const struct gmr1_puncturer gmr1_punct_k5_12_A;
extern const struct gmr1_puncturer gmr1_punct_k5_12_B;
extern const struct gmr1_puncturer gmr1_punct_k5_12_C;
extern const struct gmr1_puncturer gmr1_punct_k5_12_D;
extern const struct gmr1_puncturer gmr1_punct_k5_12_E;
extern const struct gmr1_puncturer gmr1_punct_k5_12_P38;
extern const struct gmr1_puncturer gmr1_punct_k5_12_P26;
extern const struct gmr1_puncturer gmr1_punct_k5_12_P37;
extern const struct gmr1_puncturer gmr1_punct_k5_13_P16;
extern const struct gmr1_puncturer gmr1_punct_k5_13_P25;
extern const struct gmr1_puncturer gmr1_punct_k5_13_P15;
extern const struct gmr1_puncturer gmr1_punct_k5_13_Ps15;
extern const struct gmr1_puncturer gmr1_punct_k5_13_P78;
extern const struct gmr1_puncturer gmr1_punct_k5_15_P23;
extern const struct gmr1_puncturer gmr1_punct_k5_15_P53;
extern const struct gmr1_puncturer gmr1_punct_k5_15_Ps53;
extern const struct gmr1_puncturer gmr1_punct_k7_12_P23;
extern const struct gmr1_puncturer gmr1_punct_k7_12_P410;
extern const struct gmr1_puncturer gmr1_punct_k7_12_P512;
extern const struct gmr1_puncturer gmr1_punct_k7_12_P116;
extern const struct gmr1_puncturer gmr1_punct_k7_12_P148;
extern const struct gmr1_puncturer gmr1_punct_k7_12_P184;
extern const struct gmr1_puncturer gmr1_punct_k7_12_P1152;
extern const struct gmr1_puncturer gmr1_punct_k7_12_P45;
extern const struct gmr1_puncturer gmr1_punct_k7_12_P245;
extern const struct gmr1_puncturer gmr1_punct_k9_12_P13;
extern const struct gmr1_puncturer gmr1_punct_k9_12_P47;
extern const struct gmr1_puncturer gmr1_punct_k9_12_P34;
extern const struct gmr1_puncturer gmr1_punct_k9_12_P17;
extern const struct gmr1_puncturer gmr1_punct_k9_12_P19;
extern const struct gmr1_puncturer gmr1_punct_k9_12_P26;
extern const struct gmr1_puncturer gmr1_punct_k9_12_P110;
extern const struct gmr1_puncturer gmr1_punct_k9_12_P14;
extern const struct gmr1_puncturer gmr1_punct_k9_12_P45;
extern const struct gmr1_puncturer gmr1_punct_k9_12_P234;
extern const struct gmr1_puncturer gmr1_punct_k6_14_P45;
extern const struct gmr1_puncturer gmr1_punct_k9_14_P148;
extern const struct gmr1_puncturer gmr1_punct_k9_14_P65;
extern const struct gmr1_puncturer gmr1_punct_k9_13_P12;
extern const struct gmr1_puncturer gmr1_punct_k9_13_P1213;
extern const struct gmr1_puncturer gmr1_punct_k9_13_P44;
extern const struct gmr1_puncturer gmr1_punct_k9_13_P33;
extern const struct gmr1_puncturer gmr1_punct_k9_13_P65;

#ifdef __cplusplus
}
#endif

#endif
