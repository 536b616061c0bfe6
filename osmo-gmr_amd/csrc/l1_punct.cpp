// l1_punct.cpp -- gmr1_puncturer_generate (reference src/l1/punct.c:48-133): host code, no GPU involved.
#include <errno.h>
#include <stdlib.h>

#include <vector>

#include <osmocom/gmr1/l1/conv.h>
#include <osmocom/gmr1/l1/punct.h>

namespace {

// coded bits of the unpunctured stream (libosmocore's osmo_conv_get_output_length(code, 0) [3P]): one N-bit word per
// data bit, K - 1 more for a flushed code, less whatever an already attached puncture list removes
int coded_length(const struct osmo_conv_code *code)
{
	const int steps = code->len + (code->term == CONV_TERM_FLUSH ? code->K - 1 : 0);
	int n = steps * code->N;
	if (code->puncture)
		for (const int *p = code->puncture; *p >= 0; p++)
			n--;
	return n;
}

}  // namespace

extern "C" int gmr1_puncturer_generate(struct osmo_conv_code *code, const struct gmr1_puncturer *punct_pre,
                                       const struct gmr1_puncturer *punct_main, const struct gmr1_puncturer *punct_post,
                                       int repeat)
{
	if (!code || !punct_main)
		return -EINVAL;
	const int N = code->N;
	if ((punct_pre && punct_pre->N != N) || punct_main->N != N || (punct_post && punct_post->N != N))
		return -EINVAL;

	const int total = coded_length(code);
	// the stretch the main scheme covers: what the first and last blocks leave
	int body_end = total;
	if (punct_post)
		body_end -= punct_post->L * N;
	if (!repeat) {
		int body = body_end - (punct_pre ? punct_pre->L * N : 0);
		const int d = punct_main->L * N;
		repeat = (body + d - 1) / d;
	}

	std::vector<int> out;
	int pos = 0;
	if (punct_pre)
		for (int k = 0; k < punct_pre->L * N && pos < total; k++, pos++)
			if (punct_pre->mask[k] == 0)
				out.push_back(pos);
	for (int r = 0; r < repeat; r++)
		for (int k = 0; k < punct_main->L * N && pos < body_end; k++, pos++)
			if (punct_main->mask[k] == 0)
				out.push_back(pos);
	if (punct_post) {
		// the last block starts at body_end wherever the main scheme stopped (the reference's loop guard here is
		// `position > 0`, punct.c:121-124: kept)
		pos = body_end;
		for (int k = 0; k < punct_post->L * N && pos > 0; k++, pos++)
			if (punct_post->mask[k] == 0)
				out.push_back(pos);
	}

	int *p = static_cast<int *>(malloc((out.size() + 1) * sizeof(int)));
	if (!p)
		return -ENOMEM;
	for (size_t i = 0; i < out.size(); i++)
		p[i] = out[i];
	p[out.size()] = -1;
	code->puncture = p;
	return 0;
}
