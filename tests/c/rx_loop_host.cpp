// Host-side exercise of osmo-gmr_amd/csrc/rx_loop.h (the receive loop's integer control logic, which the
// device runs inside k_rx_chain): walks a chain through a capture and prints every round's bursts, applying a
// scripted BCCH feedback.  tests/test_rx_loop_host.py compares the output with a Python model of
// process_bcch (reference src/gmr1_rx.c:852-895).
//   usage: rx_loop_host sps len align fn delay stn toa_step
#include <cstdio>
#include <cstdlib>

#include "rx_loop.h"

using namespace gmr1;

int main(int argc, char **argv)
{
	if (argc < 8)
		return 2;
	const int sps = atoi(argv[1]);
	RxLoopState s{};
	s.base = 0;
	s.len = atoi(argv[2]);
	s.align = atoi(argv[3]);
	s.freq_err = 0.f;
	s.fn = atoi(argv[4]);
	s.delay = atoi(argv[5]);
	s.stn = atoi(argv[6]);
	const int toa_step = atoi(argv[7]);      // every BCCH burst reports e_toa + toa_step (and passes its CRC)
	int frames = 0, round = 0;
	auto on_frame = [&](const RxLoopState &) { frames++; };
	for (;; round++) {
		RxLoopItem items[kLoopPerRound];
		const int n = rx_loop_build_round(s, sps, items, on_frame);
		if (!n)
			break;
		for (int k = 0; k < n; k++)
			printf("%d %d %d %d %d %d %d\n", round, items[k].is_bcch, items[k].begin, items[k].fn, items[k].tn,
			       items[k].e_toa, items[k].frames_before);
		if (items[n - 1].is_bcch) {
			// an SI1 "Segment 2A bis" every fourth BCCH: it moves the slot and the frame number (gmr1_rx.c:194-233)
			uint8_t l2[24] = {0};
			if ((round & 3) == 1) {
				const int delay = (round >> 2) & 7, stn = (3 * round) % 24, sf = round & 0x1fff, mf = round & 3;
				l2[0] = 0x08;
				l2[9] = 0x80;
				l2[10] = (uint8_t)((delay << 3) | (stn >> 2));
				l2[11] = (uint8_t)(((stn & 3) << 6) | (sf >> 7));
				l2[12] = (uint8_t)(((sf & 0x7f) << 1) | (mf >> 1));
				l2[13] = (uint8_t)(((mf & 1) << 7) | 0x40);
			}
			rx_loop_bcch_result(s, sps, 0, 0, (float)(items[n - 1].e_toa + toa_step), 0.001f, l2, items[n - 1].e_toa);
			on_frame(s);
			rx_loop_advance(s, sps);
		}
	}
	printf("end %d %d %d %d %d %d\n", round, frames, s.align, s.fn, s.delay, s.stn);
	return 0;
}
