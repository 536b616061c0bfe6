// tx_kernels.hip -- transmit direction for gfx950: the layer-1 channel encoders (reference src/l1/bcch.c:60-81,
// ccch.c:60-83, facch3.c:65-116, tch3.c:60-118, facch9.c:60-104, tch9.c:81-137, rach.c:78-136, xch_dc12.c:64-84)
// and the pi/4-CxPSK modulator (reference src/sdr/pi4cxpsk.c:741-799).
//
// k_encode: one unit per wave, four units per workgroup.  Nothing in a GMR-1 coder depends on the data: CRC,
// convolutional code (all feed-forward), puncturing, intra- / inter-burst interleaving, scrambling, ciphering
// and multiplexing are GF(2)-linear position maps, so the host folds each chain into an EncPlan once and the
// kernel has three steps:
//   1. CRC registers = xor over the set payload bits of their table entries (lanes stride the bits, DPP-free
//      wave reduction with ds_swizzle-class shuffles);
//   2. the extended information word [K-1 preset bits | info, CRC | flush zeros] as a bit array in LDS, 64 bits
//      per ballot (tail-biting codes get their last K-1 bits as the preset, so a trellis step is always a
//      window of K consecutive bits);
//   3. every burst bit = parity(window(t) & poly) ^ scramble ^ keystream, or a multiplexed-in bit, written as a
//      ubit byte (coalesced 64-byte stores).
// The inter-burst interleaver of TCH9 (depth 3) is a delay per burst bit: the wave builds the extended words of
// bursts n, n-1, n-2 of its run (empty interleaver memory = all-zero word, since the code is linear).
//
// ext_src[i] (uint16): 0xffff constant 0 | 0x0000-0x3fff payload bit (byte * 8 + bit from the LSB; the second
//   input array follows the first) | 0x4000 + b bit b of CRC register A | 0x8000 + b bit b of CRC register B.
// out[e] (uint32): bits 0-9 window offset t in the extended word, or index of the multiplexed-in bit;
//   10-12 poly slot; 13-14 kind (0 coded, 1 constant 0, 2 multiplexed-in); 15 scrambler bit;
//   16-25 keystream index + 1 (0 = never ciphered); 26-27 delay in bursts.
//
// Bound: HBM (payload in, one byte per burst bit out: BCCH 24 + 424 B per burst); the kernel issues ~25 VALU per
// burst bit, far below the rate of the decoders it feeds.
#include <hip/hip_runtime.h>

#include "gmr1_dev.h"

namespace gmr1 {

#define WAVE_SYNC()                                               \
	do {                                                          \
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");    \
		__builtin_amdgcn_wave_barrier();                          \
	} while (0)

constexpr int kEncWaves = 4;
constexpr int kExtWords = kEncMaxExt / 32 + 2;

__global__ __launch_bounds__(64 * kEncWaves) void k_encode(EncArgs a)
{
	__shared__ uint32_t s_ext[kEncWaves][3][kExtWords];
	__shared__ uint8_t s_pay[kEncWaves][kEncMaxIn];
	const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const long long u = (long long)blockIdx.x * kEncWaves + w;
	if (u >= a.n)
		return;                               // waves are independent: no workgroup barrier below
	const EncPlan &P = *a.plan;
	const int n_in = P.n_in0 + P.n_in1;
	const int pos = a.seq_len > 1 ? (int)(u % a.seq_len) : 0;

	for (int d = 0; d < P.depth; d++) {
		uint32_t *ext = s_ext[w][d];
		if (d > pos) {                        // before the run started: empty interleaver memory
			if (lane < kExtWords)
				ext[lane] = 0u;
			continue;
		}
		const long long ud = u - d;
		if (lane < n_in)
			s_pay[w][lane] = lane < P.n_in0 ? a.in0[ud * P.n_in0 + lane] : a.in1[ud * P.n_in1 + (lane - P.n_in0)];
		WAVE_SYNC();
		uint32_t crc = 0;                     // register A in the low half, B in the high half
		for (int b = lane; b < n_in * 8; b += 64)
			if ((s_pay[w][b >> 3] >> (b & 7)) & 1)
				crc ^= (uint32_t)P.crc_tab[b] | ((uint32_t)P.crc_tab2[b] << 16);
		for (int m = 32; m >= 1; m >>= 1)
			crc ^= (uint32_t)__shfl_xor((int)crc, m, 64);
		for (int base = 0; base < P.n_ext + 64; base += 64) {     // one spare ballot: the window of the last step reads past n_ext
			const int i = base + lane;
			const uint32_t src = i < P.n_ext ? P.ext_src[i] : 0xffffu;
			uint32_t bit = 0;
			if (src < 0x4000u)
				bit = (s_pay[w][src >> 3] >> (src & 7)) & 1u;
			else if (src < 0x8000u)
				bit = (crc >> (src & 15u)) & 1u;
			else if (src != 0xffffu)
				bit = (crc >> (16u + (src & 15u))) & 1u;
			const unsigned long long bal = __ballot(bit != 0);
			if (lane == 0 && (base >> 5) + 1 < kExtWords) {
				ext[base >> 5] = (uint32_t)bal;
				ext[(base >> 5) + 1] = (uint32_t)(bal >> 32);
			}
		}
		WAVE_SYNC();                          // s_pay is rewritten by the next d
	}
	WAVE_SYNC();

	uint8_t *dst = a.ebits + u * P.n_out;
	auto burst_bit = [&](uint32_t ds) -> uint32_t {
		const uint32_t t = ds & 1023u, kind = (ds >> 13) & 3u, ci = (ds >> 16) & 1023u, d = (ds >> 26) & 3u;
		uint32_t bit = 0;
		if (kind == 0) {
			const uint32_t *ext = s_ext[w][d];
			const unsigned long long two = ((unsigned long long)ext[(t >> 5) + 1] << 32) | ext[t >> 5];
			bit = (uint32_t)__popc((uint32_t)(two >> (t & 31u)) & P.poly[(ds >> 10) & 7u]) & 1u;
		} else if (kind == 2) {
			bit = ((int)t < P.n_aux0 ? a.aux0[u * P.n_aux0 + t] : a.aux1[u * P.n_aux1 + (t - P.n_aux0)]) & 1u;
		}
		bit ^= (ds >> 15) & 1u;
		if (ci && a.ciph)
			bit ^= a.ciph[u * P.n_ciph + (ci - 1)] & 1u;
		return bit;
	};
	if ((P.n_out & 3) == 0 && (reinterpret_cast<uintptr_t>(a.ebits) & 3u) == 0) {
		// four burst bits per lane, one dword store (every unit starts on a dword when n_out is a multiple of four)
		const uint4 *dsc = reinterpret_cast<const uint4 *>(P.out);
		uint32_t *dst4 = reinterpret_cast<uint32_t *>(dst);
		for (int e4 = lane; e4 < (P.n_out >> 2); e4 += 64) {
			const uint4 d4 = dsc[e4];
			dst4[e4] = burst_bit(d4.x) | (burst_bit(d4.y) << 8) | (burst_bit(d4.z) << 16) | (burst_bit(d4.w) << 24);
		}
	} else {
		for (int e = lane; e < P.n_out; e += 64)
			dst[e] = (uint8_t)burst_bit(P.out[e]);
	}
}

hipError_t launch_encode(const EncArgs &a, hipStream_t stream)
{
	if (a.n <= 0)
		return hipSuccess;
	const unsigned blocks = (unsigned)(((long long)a.n + kEncWaves - 1) / kEncWaves);
	hipLaunchKernelGGL(k_encode, dim3(blocks), dim3(64 * kEncWaves), 0, stream, a);
	return hipGetLastError();
}

// ---- modulator: pi4cxpsk.c:741-799.  One wave per burst; the constellation points are +-1 / +-j exactly,
// so the "multiply by the rotation" of osmo_cxvec_rotate is written out as the complex product the CPU computes
// (the zeros of the operand matter only for the sign of a zero); the rotation phasors e^{j rotation i} are a
// per-format table the host evaluates with the CPU's own cosf / sinf, so the symbols match a CPU build bit for bit.
__global__ __launch_bounds__(256) void k_mod(ModArgs a)
{
	const long long u = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);   // one burst per wave, symbols strided over its lanes
	if (u >= a.n)
		return;
	const uint8_t *ebu = a.ebits + u * a.n_ebits;
	float2 *outu = a.out + u * a.len;
	for (int i = threadIdx.x & 63; i < a.len; i += 64) {
		const int p = a.plan[i];
		float2 v = make_float2(0.f, 0.f);
		if (p >= 0) {
			int sym = p;
			if (p >= 4) {
				const uint8_t *eb = ebu + (p - 4);
				if (a.nbits == 2) {
					const int bv = ((eb[0] & 1) << 1) | (eb[1] & 1);
					sym = bv ^ (bv >> 1);             // 00 01 10 11 -> 0 1 3 2 (the '.bits' table of pi4cxpsk.c:87-107)
				} else {
					sym = eb[0] & 1;
				}
			}
			if (a.nbits == 2)
				v = make_float2((sym & 1) ? 0.f : ((sym & 2) ? -1.f : 1.f), (sym & 1) ? ((sym & 2) ? -1.f : 1.f) : 0.f);
			else
				v = make_float2((sym & 1) ? -1.f : 1.f, 0.f);
		}
		const float2 r = a.rot[i];
		const float c = r.x, s = r.y;
		outu[i] = make_float2(v.x * c - v.y * s, v.x * s + v.y * c);
	}
}

hipError_t launch_mod(const ModArgs &a, hipStream_t stream)
{
	if (a.n <= 0 || a.len <= 0)
		return hipSuccess;
	hipLaunchKernelGGL(k_mod, dim3((unsigned)((a.n + 3) / 4)), dim3(256), 0, stream, a);
	return hipGetLastError();
}

// ---- stand-alone primitives: gmr1_scramble_{sbit,ubit} (scramb.c:62-93), gmr1_{de,}interleave_intra
// (interleave.c:48-87), gmr1_{de,}interleave_inter (interleave.c:128-190): a gather plus an optional sign / bit flip
__global__ __launch_bounds__(256) void k_bitmap(BitMapArgs a)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= a.n)
		return;
	uint8_t v = a.in[a.perm ? a.perm[i] : i];
	if (a.mask && a.mask[i])
		v = a.soft ? (uint8_t)(int8_t)(-(int)(int8_t)v) : (uint8_t)(v ^ 1u);
	a.out[i] = v;
}

hipError_t launch_bitmap(const BitMapArgs &a, hipStream_t stream)
{
	if (a.n <= 0)
		return hipSuccess;
	hipLaunchKernelGGL(k_bitmap, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, stream, a);
	return hipGetLastError();
}

}  // namespace gmr1
