/* A C99 translation unit that uses the boundary the way a maintainer of the reference would:
 * every public header must be plain C, the structs must have the documented sizes, and the program
 * must link against libgmr1_hip.so and run without a GPU as far as non-compute calls go. */
#include <stdio.h>
#include <string.h>

#include <gmr1_hip.h>
#include <osmocom/gmr1/sdr/defs.h>
#include <osmocom/gmr1/sdr/pi4cxpsk.h>
#include <osmocom/gmr1/sdr/nb.h>
#include <osmocom/gmr1/sdr/fcch.h>
#include <osmocom/gmr1/sdr/dkab.h>
#include <osmocom/gmr1/l1/bcch.h>
#include <osmocom/gmr1/l1/ccch.h>
#include <osmocom/gmr1/l1/facch3.h>
#include <osmocom/gmr1/l1/facch9.h>
#include <osmocom/gmr1/l1/tch3.h>
#include <osmocom/gmr1/l1/tch9.h>
#include <osmocom/gmr1/l1/interleave.h>
#include <osmocom/gmr1/l1/rach.h>
#include <osmocom/gmr1/l1/xch_dc12.h>
#include <osmocom/gmr1/l1/a5.h>
#include <osmocom/gmr1/l1/scramb.h>
#include <osmocom/gmr1/l1/conv.h>
#include <osmocom/gmr1/l1/punct.h>
#include <osmocom/gmr1/l1/crc.h>
#include <osmocom/gmr1/gsmtap.h>
#include <osmocom/gmr1/codec/codec.h>
#include <stdlib.h>

int main(void)
{
	struct gmr1_hip_rx_record rec;
	struct gmr1_hip_rx_big_record big;
	uint8_t pkt[96];
	int n;

	if (sizeof(rec) != 40 || sizeof(big) != 80) {
		fprintf(stderr, "record sizes %zu / %zu\n", sizeof(rec), sizeof(big));
		return 1;
	}
	/* the exported burst descriptions are ordinary data objects, as in the reference (sdr/nb.h) */
	if (gmr1_bcch_burst.len != 234 || gmr1_dc6_burst.len != 234 || gmr1_nt3_speech_burst.len != 117 ||
	    gmr1_nt9_burst.len != 351 || gmr1_fcch_burst.len != 117) {
		fprintf(stderr, "burst tables\n");
		return 2;
	}
	memset(&rec, 0, sizeof(rec));
	rec.type = 1; rec.fn = 0x01020304; rec.tn = 7; rec.len = 24;
	n = gmr1_hip_gsmtap_pack(&rec, 0, pkt, (int)sizeof(pkt));     /* host-only call */
	if (n != 40 || pkt[0] != 2 || pkt[2] != 0x0a || pkt[3] != 7 || pkt[8] != 1 || pkt[11] != 4) {
		fprintf(stderr, "gsmtap pack %d\n", n);
		return 3;
	}
	{
		/* the inter-burst interleaver state is a caller-owned struct of the reference's layout (gmr1_rx.c:90) */
		struct gmr1_interleaver il;
		if (gmr1_interleaver_init(&il, 3, 648) != 0 || il.N != 3 || il.K != 648 || il.n != 0 || !il.bits_cpp) {
			fprintf(stderr, "interleaver init\n");
			return 4;
		}
		gmr1_interleaver_fini(&il);
		if (il.bits_cpp != NULL || gmr1_interleaver_init(&il, 4, 648) == 0)
			return 5;
		if (GMR1_TCH9_9k6 != 2 || sizeof(struct gmr1_interleaver) != 3 * sizeof(int) + sizeof(void *) + (sizeof(void *) - sizeof(int)))
			return 6;
	}
	{
		/* the way the reference's tch3.c:42-49 specialises a code: copy the description, set len, attach the
		 * puncturing of P(1;2) -- plain C, flexible array member and all */
		struct osmo_conv_code code;
		int k = 0;
		memcpy(&code, &gmr1_conv_tch3, sizeof(code));
		code.len = 48;
		if (gmr1_puncturer_generate(&code, NULL, &gmr1_punct_k5_12_P12, NULL, 0) != 0 || !code.puncture)
			return 7;
		while (code.puncture[k] >= 0) {
			if (code.puncture[k] != 4 * k + 3)
				return 8;
			k++;
		}
		if (k != 24 || code.K != 7 || code.term != CONV_TERM_TAIL_BITING || code.next_output[1][0] != 1 ||
		    gmr1_punct_k5_12_P12.mask[3] != 0 || gmr1_crc16.poly != 0x1021 || gmr1_crc8.bits != 8)
			return 9;
		free((void *)code.puncture);
	}
	{
		/* the vocoder object as src/gmr1_ambe_decode.c uses it; without a device the allocation fails like a calloc would */
		struct gmr1_codec *codec = gmr1_codec_alloc();
		int16_t audio[160];
		const void *img = NULL;
		size_t bytes = 0;
		memset(audio, 0x55, sizeof(audio));
		if (codec) {
			uint8_t silence[10] = {0xf8};
			if (gmr1_codec_decode_frame(codec, audio, 160, silence, 0) != 0 || audio[0] != 0 || audio[159] != 0)
				return 10;
			if (gmr1_codec_decode_dtx(codec, audio, 160) != 0)
				return 11;
		}
		gmr1_codec_release(codec);                  /* NULL is allowed (src/codec/codec.c:64-66) */
		if (gmr1_hip_codec_state_bytes() % 16 || gmr1_hip_codec_host_tables(&img, &bytes) != 0 || !img || bytes < 200000)
			return 12;
	}
	printf("%s\n", gmr1_hip_version());
	return 0;
}
