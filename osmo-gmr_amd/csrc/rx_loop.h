// rx_loop.h -- the integer control logic of process_bcch (reference src/gmr1_rx.c:852-895) with its helpers
// burst_map (:149-170) and bcch_tdma_align (:194-233).  The device runs it inside k_rx_chain (one work-group
// walks one chain through ALL of its frames, no host round trip between them); it is host-callable too
// (tests, sizing).  No signal arithmetic here.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIP__)
#define GMR1_HD __host__ __device__ inline
#else
#define GMR1_HD inline
#endif

namespace gmr1 {

constexpr int kLoopPerRound = 8;           // bursts one chain contributes to a round: <= 7 CCCH, then its BCCH
struct RxLoopState {                       // the part of struct chan_desc the frame loop reads and writes
	uint64_t base;                         // first sample of the carrier in iq
	int len;                               // samples of the carrier
	int align;
	float freq_err;
	int fn, delay, stn;
	int done;
	float bcch_energy;                     // burst_energy() of the last BCCH burst found (NaN: none yet)
	uint16_t arfcn;                        // what the chain's records carry
	uint16_t chain;
};

struct RxLoopFrame { int align; float freq_err; int fn; };     // what rx_tch3 sees in a frame

struct RxLoopItem {                        // one burst of a round
	int begin;                             // window start, samples from the carrier's first
	int is_bcch;
	int fn, tn, e_toa;
	int frames_before;                     // frames of this round completed before the burst's frame
};

// gmr1_rx.c:149-170 (begin < 0 is an out-of-bounds read in the reference; refused here and in the oracle)
GMR1_HD int rx_loop_burst_map(const RxLoopState &c, int sps, int burst_len, int tn, int win, int *begin)
{
	const int etoa = win >> 1;
	const int b = c.align + sps * tn * 39 - etoa;
	const int l = burst_len * sps + win;
	if (b < 0 || b + l > c.len)
		return -1;
	*begin = b;
	return etoa;
}

GMR1_HD void rx_loop_advance(RxLoopState &c, int sps)
{
	const int frame_len = sps * 24 * 39;
	c.fn++;
	c.align += frame_len;
	if (c.align + 2 * frame_len > c.len)
		c.done = 1;
}

// gmr1_rx.c:194-233: SI1 "Segment 2A bis" -> TDMA position
GMR1_HD void rx_loop_tdma_align(RxLoopState &c, int sps, const uint8_t *l2)
{
	if ((l2[0] & 0xf8) != 0x08)
		return;
	if ((l2[9] & 0xfc) != 0x80)
		return;
	const int delay = (l2[10] >> 3) & 0x0f;
	const int stn = ((l2[10] << 2) & 0x1c) | (l2[11] >> 6);
	const int superframe = ((l2[11] & 0x3f) << 7) | (l2[12] >> 1);
	const int multiframe = ((l2[12] & 0x01) << 1) | (l2[13] >> 7);
	const int mffn_hi = (l2[13] & 0x40) >> 6;
	const int fn = (superframe << 6) | (multiframe << 4) | (mffn_hi << 3) | ((2 + delay) & 7);
	c.align += (c.stn - stn) * 39 * sps;
	c.fn = fn;
	c.delay = delay;
	c.stn = stn;
}

// The frames from the chain's position up to and including its next BCCH burst (process_bcch's sirfn
// schedule, gmr1_rx.c:873-878): CCCH bursts are independent of each other, the BCCH burst ends the round
// because its result moves align / freq_err / fn.  Frames that end inside the round are advanced over
// (on_frame is told first); a BCCH item leaves the chain AT its frame until rx_loop_bcch_result.
template <class F>
GMR1_HD int rx_loop_build_round(RxLoopState &c, int sps, RxLoopItem *items, F on_frame)
{
	int n = 0, cnt = 0, frames = 0;
	while (!c.done && cnt < kLoopPerRound - 1) {
		const int m = (c.fn - c.delay) & 7;          // sirfn % 8 with sirfn = (fn - delay) & 63, gmr1_rx.c:870-877
		int begin = 0;
		if (m == 2) {
			const int e = rx_loop_burst_map(c, sps, 234, c.stn, 20 * sps, &begin);
			if (e >= 0) {
				items[n++] = {begin, 1, c.fn, c.stn, e, frames};
				break;                     // the frame completes once the burst's result is known
			}
		} else if (m != 0) {
			const int e = rx_loop_burst_map(c, sps, 234, c.stn, 10 * sps, &begin);
			if (e >= 0) {
				items[n++] = {begin, 0, c.fn, c.stn, e, frames};
				cnt++;
			}
		}
		on_frame(c);
		rx_loop_advance(c, sps);
		frames++;
	}
	return n;
}

// rx_bcch's feedback (gmr1_rx.c:782-791): only a burst that was found AND passed its CRC moves the chain.
// Returns 1 if it did.  The caller completes the frame (on_frame, rx_loop_advance) afterwards.
GMR1_HD int rx_loop_bcch_result(RxLoopState &c, int sps, int rv, int crc, float toa, float freq_err,
                                const uint8_t *l2, int e_toa)
{
	if (rv || crc)
		return 0;
	c.align += (int)roundf(toa) - e_toa;       // roundf is exact on both sides
	c.freq_err += freq_err;
	rx_loop_tdma_align(c, sps, l2);
	return 1;
}

}  // namespace gmr1
