/* oracle/ref_codec_shim.c -- TEST INFRASTRUCTURE ONLY, linked next to the reference's unmodified src/codec into
 * oracle/_ref/libgmr1_codec_ref.so (`make -C oracle ref`).
 *
 * The reference's speech path keeps the two subframes of a frame in a local array it never clears
 * (src/codec/ambe.c:81-83) and its voiced synthesiser counts `Vl[l]` over max(L, L_prev) harmonics
 * (src/codec/synth.c:231-233) - when the previous subframe had more harmonics, the entries from L upwards are whatever
 * the stack held.  To compare against it at all, that memory has to hold something definite: this entry point clears
 * the stack region the call is about to use and then calls the reference, so those entries read 0 ("not voiced"), which
 * is what the oracle and the GPU kernel define them to be (decision D9).  Nothing of the reference is
 * changed or replaced. */
#include <stdint.h>
#include <string.h>

struct gmr1_codec;
int gmr1_codec_decode_frame(struct gmr1_codec *codec, int16_t *audio, int N, const uint8_t *frame, int bad);

static __attribute__((noinline)) void clear_stack_below(void)
{
	volatile unsigned char pad[32768];
	memset((void *)pad, 0, sizeof(pad));
	__asm__ volatile("" ::: "memory");
}

__attribute__((noinline)) int ref_codec_decode_frame_clean(struct gmr1_codec *codec, int16_t *audio, int N,
                                                           const uint8_t *frame, int bad)
{
	clear_stack_below();
	int rv = gmr1_codec_decode_frame(codec, audio, N, frame, bad);
	__asm__ volatile("" ::: "memory");
	return rv;
}

/* n frames of 10 bytes -> n x 160 samples */
int ref_codec_decode_stream(struct gmr1_codec *codec, const uint8_t *frames, int n, int16_t *pcm, int *rv)
{
	int bad = 0;
	for (int i = 0; i < n; i++) {
		int r = ref_codec_decode_frame_clean(codec, pcm + 160 * (size_t)i, 160, frames + 10 * (size_t)i, 0);
		if (rv)
			rv[i] = r;
		bad += r != 0;
	}
	return bad;
}
