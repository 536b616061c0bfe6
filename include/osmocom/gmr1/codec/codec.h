/* GMR-1 AMBE speech decoder (API of osmocom/osmo-gmr include/osmocom/gmr1/codec/codec.h:37-45).
 *
 * One `struct gmr1_codec` is one voice channel: its 10-byte frames must be given in order.  Samples are produced on
 * the GPU (osmo-gmr_amd/csrc/ambe_kernels.hip); a decoder belongs to the HIP device that was current when it was
 * allocated.  For many channels at once see gmr1_hip_codec_decode_batch* in gmr1_hip.h. */
#ifndef __OSMO_GMR1_CODEC_H__
#define __OSMO_GMR1_CODEC_H__

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct gmr1_codec;

/* codec.h:39-40.  NULL when there is no memory - or no HIP device */
struct gmr1_codec *gmr1_codec_alloc(void);
void gmr1_codec_release(struct gmr1_codec *codec);

/* codec.h:42-44.  One frame (speech, silence indication or tone) -> 8 kHz PCM.  Speech and silence frames write 160
 * samples whatever N is, tone frames write N (src/codec/ambe.c:110-126); `bad` is ignored, as in the reference.
 * Returns 0, or -EINVAL for a tone frame with an unassigned tone code (src/codec/tone.c:197-201). */
int gmr1_codec_decode_frame(struct gmr1_codec *codec, int16_t *audio, int N, const uint8_t *frame, int bad);

/* codec.h:45.  A frame that never arrived: N zeros, decoder untouched (src/codec/ambe.c:130-141) */
int gmr1_codec_decode_dtx(struct gmr1_codec *codec, int16_t *audio, int N);

#ifdef __cplusplus
}
#endif

#endif
