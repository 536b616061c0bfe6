/*
 * oracle/orc_rx.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the receive control loop of
 * the reference's gmr1_rx application for one BCCH carrier (reference src/gmr1_rx.c): FCCH single
 * acquisition, multi-FCCH survivor selection, then the frame-by-frame BCCH / CCCH loop with its
 * tracking feedback (align, freq_err, TDMA position from SI1), and the TCH3 follow-up after an
 * IMMEDIATE ASSIGNMENT (DKAB / FACCH3 / speech on the traffic carrier, A5/1).  The TCH9 follow-up,
 * GSMTAP transport and stderr logging are outside the scope;
 * what GSMTAP would have carried is returned as records.  PARITY UNPINNED, see orc_3p.h.
 */
#include "orc_gmr1.h"

#include <errno.h>
#include <math.h>
#include <string.h>

#define START_DISCARD 8000          /* gmr1_rx.c:52 */
#define SYM_RATE 23400
#define PIf 3.14159265358979323846f

struct tch3_state {                 /* gmr1_rx.c:59-78 */
	int active;
	int tn, p, ciph;
	float energy_dkab, energy_burst;
	int weak_cnt;
	orc_sbit_t ebits[104 * 4];
	uint32_t bi_fn[4];
	int sync_id, burst_cnt;
};

struct tch9_state {                 /* gmr1_rx.c:82-91 */
	int active;
	int tn;
	struct orc_interleaver il;
};

struct chan_desc {                  /* gmr1_rx.c:93-115, the fields this scope uses */
	const orc_cf *iq;
	const orc_cf *tch;              /* traffic carrier, same length and timing as iq (may be NULL) */
	const orc_cf *csd;              /* carrier of the TCH9 (circuit switched data) channel (may be NULL) */
	struct tch3_state tch3_state;
	struct tch9_state tch9_state;
	uint8_t kc[8];
	int len;
	int sps;
	int align;
	float freq_err;
	int fn;
	int sa_sirfn_delay;
	int sa_bcch_stn;
};

struct sink {
	struct orc_rx_record *out;
	int max, n;
	int arfcn, chain;
	struct orc_rx_big_record *big;  /* NT9 payloads (38 / 60 bytes) do not fit the 40-byte record */
	int max_big, n_big;
};

static void emit_big(struct sink *s, int type, int fn, int tn, const uint8_t *l2, int n, int conv)
{
	if (s->big && s->n_big < s->max_big) {
		struct orc_rx_big_record *r = &s->big[s->n_big];
		memset(r, 0, sizeof(*r));
		r->arfcn = (uint16_t)s->arfcn;
		r->chain = (uint8_t)s->chain;
		r->type = (uint8_t)type;
		r->fn = (uint32_t)fn;
		r->tn = (uint8_t)tn;
		r->len = (uint8_t)n;
		r->conv = conv;
		memcpy(r->l2, l2, (size_t)n);
	}
	s->n_big++;
}

static void emit_n(struct sink *s, int type, int fn, int tn, const uint8_t *l2, int n, int conv)
{
	if (s->n < s->max) {
		struct orc_rx_record *r = &s->out[s->n];
		memset(r, 0, sizeof(*r));
		r->arfcn = (uint16_t)s->arfcn;
		r->chain = (uint8_t)s->chain;
		r->type = (uint8_t)type;
		r->fn = (uint32_t)fn;
		r->tn = (uint8_t)tn;
		r->crc = 0;
		r->len = (uint8_t)n;
		r->conv = conv;
		memcpy(r->l2, l2, (size_t)n);
	}
	s->n++;
}

static void emit(struct sink *s, int type, int fn, int tn, const uint8_t *l2, int conv)
{
	if (s->n < s->max) {
		struct orc_rx_record *r = &s->out[s->n];
		memset(r, 0, sizeof(*r));
		r->arfcn = (uint16_t)s->arfcn;
		r->chain = (uint8_t)s->chain;
		r->type = (uint8_t)type;
		r->fn = (uint32_t)fn;
		r->tn = (uint8_t)tn;
		r->crc = 0;
		r->len = 24;
		r->conv = conv;
		memcpy(r->l2, l2, 24);
	}
	s->n++;
}

static float to_hz(float f_rps) { return (SYM_RATE * f_rps) / (2.0f * PIf); }

/* gmr1_rx.c:149-170 */
static int burst_map(const struct chan_desc *cd, int burst_len, int tn, int win, int *begin_o, int *len_o)
{
	int etoa = win >> 1;
	int begin = cd->align + (cd->sps * tn * 39) - etoa;
	int len = (burst_len * cd->sps) + win;
	/* begin < 0 reads before the capture in the reference (undefined); refused here (decision D6) */
	if (begin < 0 || (begin + len) > cd->len)
		return -EIO;
	*begin_o = begin;
	*len_o = len;
	return etoa;
}

/* gmr1_rx.c:172-182 */
static float burst_energy(const orc_cf *b, int len)
{
	float e = 0.0f;
	int bd = len >> 5;
	for (int i = bd; i < len - bd; i++)
		e += crealf(b[i]) * crealf(b[i]) + cimagf(b[i]) * cimagf(b[i]);
	e /= len;
	return e;
}

/* gmr1_rx.c:194-233 */
static void bcch_tdma_align(struct chan_desc *cd, const uint8_t *l2)
{
	int delay, stn, superframe, multiframe, mffn_hi, fn;
	if ((l2[0] & 0xf8) != 0x08)
		return;
	if ((l2[9] & 0xfc) != 0x80)
		return;
	delay = (l2[10] >> 3) & 0x0f;
	stn = ((l2[10] << 2) & 0x1c) | (l2[11] >> 6);
	superframe = ((l2[11] & 0x3f) << 7) | (l2[12] >> 1);
	multiframe = ((l2[12] & 0x01) << 1) | (l2[13] >> 7);
	mffn_hi = ((l2[13] & 0x40) >> 6);
	fn = (superframe << 6) | (multiframe << 4) | (mffn_hi << 3) | ((2 + delay) & 7);
	cd->align += (cd->sa_bcch_stn - stn) * 39 * cd->sps;
	cd->fn = fn;
	cd->sa_sirfn_delay = delay;
	cd->sa_bcch_stn = stn;
}

/* ---- TCH9 follow-up (gmr1_rx.c:248-353) ------------------------------------------------------- */

static int facch3_is_ass_cmd_1(const uint8_t *l2) { return (l2[3] == 0x06) && (l2[4] == 0x2e); }

static void rx_tch9_init(struct chan_desc *cd, const uint8_t *ass_cmd)
{
	cd->tch9_state.active = 1;
	cd->tch9_state.tn = ((ass_cmd[5] & 0x03) << 3) | (ass_cmd[6] >> 5);      /* facch3_ass_cmd_1_parse, :254-258 */
	orc_interleaver_init(&cd->tch9_state.il, 3, 648);
}

static int burst_map(const struct chan_desc *cd, int burst_len, int tn, int win, int *begin_o, int *len_o);

/* gmr1_rx.c:276-353.  Decision D8: the reference uses sync_id / ebits even when the demodulator failed
 * (it never looks at rv before); a failed demodulation is "no burst" here and in the product. */
static int rx_tch9(struct chan_desc *cd, struct sink *s)
{
	const struct orc_burst *bt = orc_burst_get(ORC_BURST_NT9);
	orc_sbit_t ebits[662], sacch[10], status[4];
	orc_ubit_t ciph[658];
	int begin, len, e_toa, rv, sync_id, crc, conv;
	float toa;

	if (!cd->tch9_state.active)
		return 0;
	if (!cd->csd)
		return -EINVAL;
	e_toa = burst_map(cd, bt->len, cd->tch9_state.tn, cd->sps + (cd->sps / 2), &begin, &len);
	if (e_toa < 0)
		return e_toa;
	rv = orc_pi4cxpsk_demod(bt, cd->csd + begin, len, cd->sps, -cd->freq_err, ebits, &sync_id, &toa, NULL, NULL);
	if (rv)
		return rv;
	orc_a5(1, cd->kc, (uint32_t)cd->fn, 658, ciph, NULL);
	if (!sync_id) {     /* FACCH9 */
		uint8_t l2[38];
		crc = orc_facch9_decode(l2, sacch, status, ebits, ciph, &conv);
		if (!crc)
			emit_big(s, ORC_RX_TYPE_TCH9_FACCH, cd->fn, cd->tch9_state.tn, l2, 38, conv);
	} else {            /* TCH9, always decoded as 9k6 (gmr1_rx.c:333) */
		uint8_t l2[60];
		orc_tch9_decode(l2, sacch, status, ebits, ORC_TCH9_9k6, ciph, &cd->tch9_state.il, &conv);
		emit_big(s, ORC_RX_TYPE_TCH9, cd->fn, cd->tch9_state.tn, l2, 60, conv);
	}
	return rv;
}

/* ---- TCH3 follow-up (gmr1_rx.c:235-246, 355-600) -------------------------------------------- */

static int ccch_is_imm_ass(const uint8_t *l2) { return (l2[1] == 0x06) && (l2[2] == 0x3f); }

static void ccch_imm_ass_parse(const uint8_t *l2, int *rx_tn, int *p)
{
	*p = (l2[8] & 0xfc) >> 2;
	*rx_tn = ((l2[8] & 0x03) << 3) | (l2[9] >> 5);
}

/* gmr1_rx.c:358-378: note what it does NOT reset (ciph, burst_cnt, bi_fn) */
static void rx_tch3_init(struct chan_desc *cd, const uint8_t *imm_ass, float ref_energy)
{
	struct tch3_state *st = &cd->tch3_state;
	st->active = 1;
	ccch_imm_ass_parse(imm_ass, &st->tn, &st->p);
	st->energy_burst = ref_energy * 0.75f;
	st->energy_dkab = st->energy_burst / 8.0f;
	st->weak_cnt = 0;
	st->sync_id = 0;
	memset(st->ebits, 0x00, sizeof(st->ebits));
}

/* gmr1_rx.c:397-450 */
static void rx_tch3_facch_flush(struct chan_desc *cd, struct sink *s)
{
	struct tch3_state *st = &cd->tch3_state;
	orc_ubit_t ciph_buf[96 * 4], *ciph;
	uint8_t l2[10];
	orc_ubit_t sbits[8 * 4];
	int i, crc, conv;

	if (st->ciph) {
		ciph = ciph_buf;
		for (i = 0; i < 4; i++)
			orc_a5(1, cd->kc, st->bi_fn[i], 96, ciph + (96 * i), NULL);
	} else
		ciph = NULL;
	crc = orc_facch3_decode(l2, sbits, st->ebits, ciph, &conv);
	if (!st->ciph && crc) {
		ciph = ciph_buf;
		for (i = 0; i < 4; i++)
			orc_a5(1, cd->kc, st->bi_fn[i], 96, ciph + (96 * i), NULL);
		crc = orc_facch3_decode(l2, sbits, st->ebits, ciph, &conv);
		if (!crc)
			st->ciph = 1;
	}
	if (!crc)
		emit_n(s, ORC_RX_TYPE_TCH3_FACCH, cd->fn - 3, st->tn, l2, 10, conv);
	/* ASSIGNMENT COMMAND 1 starts the TCH9 follow-up when a CSD capture is given (gmr1_rx.c:436-442) */
	if (!crc && facch3_is_ass_cmd_1(l2) && cd->csd)
		rx_tch9_init(cd, l2);
	st->sync_id ^= 1;
	st->burst_cnt = 0;
	memset(st->bi_fn, 0xff, sizeof(st->bi_fn));
	memset(st->ebits, 0x00, sizeof(st->ebits));
}

/* gmr1_rx.c:452-493 */
static int rx_tch3_facch(struct chan_desc *cd, const orc_cf *burst, int len, struct sink *s)
{
	struct tch3_state *st = &cd->tch3_state;
	orc_sbit_t ebits[104];
	int rv, bi, sync_id;
	float toa;

	bi = cd->fn & 3;
	rv = orc_pi4cxpsk_demod(orc_burst_get(ORC_BURST_NT3_FACCH), burst, len, cd->sps, -cd->freq_err,
	                        ebits, &sync_id, &toa, NULL, NULL);
	if (rv < 0)
		return rv;
	if (sync_id != st->sync_id)
		rx_tch3_facch_flush(cd, s);
	memcpy(&st->ebits[104 * bi], ebits, sizeof(orc_sbit_t) * 104);
	st->sync_id = sync_id;
	st->bi_fn[bi] = (uint32_t)cd->fn;
	st->burst_cnt += 1;
	if (st->burst_cnt == 4)
		rx_tch3_facch_flush(cd, s);
	return 0;
}

/* gmr1_rx.c:495-529; the reference only logs the two speech frames, here they are returned too */
static int rx_tch3_speech(struct chan_desc *cd, const orc_cf *burst, int len, struct sink *s)
{
	orc_sbit_t ebits[212];
	orc_ubit_t sbits[4], ciph[208];
	uint8_t fr[20];
	int rv, conv[2];
	float toa;

	rv = orc_pi4cxpsk_demod(orc_burst_get(ORC_BURST_NT3_SPEECH), burst, len, cd->sps, -cd->freq_err,
	                        ebits, NULL, &toa, NULL, NULL);
	if (rv < 0)
		return rv;
	orc_a5(cd->tch3_state.ciph, cd->kc, (uint32_t)cd->fn, 208, ciph, NULL);
	orc_tch3_decode(fr, fr + 10, sbits, ebits, ciph, 0, &conv[0], &conv[1]);
	emit_n(s, ORC_RX_TYPE_TCH3, cd->fn, cd->tch3_state.tn, fr, 20, (conv[0] & 0xffff) | (conv[1] << 16));
	return 0;
}

/* gmr1_rx.c:531-600 */
static int rx_tch3(struct chan_desc *cd, struct sink *s)
{
	const struct orc_burst *bts[2] = { orc_burst_get(ORC_BURST_NT3_FACCH), orc_burst_get(ORC_BURST_NT3_SPEECH) };
	struct tch3_state *st = &cd->tch3_state;
	const orc_cf *burst;
	int begin, len, e_toa, rv, btid, sid;
	float be, det, toa;

	if (!st->active)
		return 0;
	if (!cd->tch)
		return -EINVAL;
	e_toa = burst_map(cd, bts[0]->len, st->tn, cd->sps + (cd->sps / 2), &begin, &len);
	if (e_toa < 0)
		return e_toa;
	burst = cd->tch + begin;

	be = burst_energy(burst, len);
	det = (st->energy_dkab + st->energy_burst) / 4.0f;
	if (be < det) {
		orc_sbit_t ebits[8];
		rv = orc_dkab_demod(burst, len, cd->sps, -cd->freq_err, st->p, ebits, &toa);
		if (rv < 0)
			return rv;
		else if (rv == 1) {
			if (st->weak_cnt++ > 8)
				st->active = 0;
		} else
			st->energy_dkab = (0.1f * be) + (0.9f * st->energy_dkab);
		return 0;
	} else
		st->weak_cnt = 0;
	st->energy_burst = (0.1f * be) + (0.9f * st->energy_burst);

	rv = orc_pi4cxpsk_detect(bts, 2, (float)e_toa, burst, len, cd->sps, -cd->freq_err, &btid, &sid, &toa);
	if (rv < 0)
		return rv;
	if (btid == 0)
		rv = rx_tch3_facch(cd, burst, len, s);
	else
		rv = rx_tch3_speech(cd, burst, len, s);
	return rv;
}

/* gmr1_rx.c:746-798 */
static void rx_bcch(struct chan_desc *cd, float *energy, struct sink *s)
{
	const struct orc_burst *bt = orc_burst_get(ORC_BURST_BCCH);
	orc_sbit_t ebits[424];
	uint8_t l2[24];
	float freq_err, toa;
	int begin, len, rv, crc, conv, e_toa;

	e_toa = burst_map(cd, bt->len, cd->sa_bcch_stn, 20 * cd->sps, &begin, &len);
	if (e_toa < 0)
		return;
	rv = orc_pi4cxpsk_demod(bt, cd->iq + begin, len, cd->sps, -cd->freq_err, ebits, NULL, &toa, &freq_err, NULL);
	if (rv)
		return;
	*energy = burst_energy(cd->iq + begin, len);
	crc = orc_bcch_decode(l2, ebits, &conv);
	if (!crc) {
		cd->align += ((int)roundf(toa)) - e_toa;
		cd->freq_err += freq_err;
		bcch_tdma_align(cd, l2);
		emit(s, ORC_RX_TYPE_BCCH, cd->fn, cd->sa_bcch_stn, l2, conv);
	}
}

/* gmr1_rx.c:800-850 */
static void rx_ccch(struct chan_desc *cd, float min_energy, struct sink *s)
{
	const struct orc_burst *bt = orc_burst_get(ORC_BURST_DC6);
	orc_sbit_t ebits[432];
	uint8_t l2[24];
	int begin, len, rv, crc, conv, e_toa;

	e_toa = burst_map(cd, bt->len, cd->sa_bcch_stn, 10 * cd->sps, &begin, &len);
	if (e_toa < 0)
		return;
	if (burst_energy(cd->iq + begin, len) < min_energy)
		return;
	rv = orc_pi4cxpsk_demod(bt, cd->iq + begin, len, cd->sps, -cd->freq_err, ebits, NULL, NULL, NULL, NULL);
	if (rv)
		return;
	crc = orc_ccch_decode(l2, ebits, &conv);
	if (!crc) {
		if (ccch_is_imm_ass(l2))
			rx_tch3_init(cd, l2, min_energy);
		emit(s, ORC_RX_TYPE_CCCH, cd->fn, cd->sa_bcch_stn, l2, conv);
	}
}

/* gmr1_rx.c:852-895 */
static void process_bcch(struct chan_desc *cd, struct sink *s)
{
	int frame_len = cd->sps * 24 * 39;
	float bcch_energy = nanf("inf");
	while (1) {
		int sirfn = (cd->fn - cd->sa_sirfn_delay) & 63;
		if (sirfn % 8 == 2)
			rx_bcch(cd, &bcch_energy, s);
		if ((sirfn % 8 != 0) && (sirfn % 8 != 2))
			rx_ccch(cd, bcch_energy / 2.0f, s);
		rx_tch3(cd, s);
		rx_tch9(cd, s);
		cd->fn++;
		cd->align += frame_len;
		if ((cd->align + 2 * frame_len) > cd->len)
			break;
	}
}

int orc_rx_run(const orc_cf *iq, int len, int sps, int arfcn,
               struct orc_rx_record *out, int max_records, int *n_records, int *n_chains)
{
	return orc_rx_run_tch(iq, NULL, len, sps, arfcn, NULL, out, max_records, n_records, n_chains);
}

/* main() with the optional tch.cfile and key arguments (gmr1_rx.c:897-975); kc NULL = the all-zero key
 * the reference starts with */
int orc_rx_run_tch(const orc_cf *iq, const orc_cf *tch, int len, int sps, int arfcn, const uint8_t *kc,
                   struct orc_rx_record *out, int max_records, int *n_records, int *n_chains)
{
	int n_big = 0;
	return orc_rx_run_full(iq, tch, NULL, len, sps, arfcn, kc, out, max_records, n_records, NULL, 0, &n_big, n_chains);
}

/* main() with all its optional arguments: tch.cfile, key, tch_csd.cfile (gmr1_rx.c:897-975) */
int orc_rx_run_full(const orc_cf *iq, const orc_cf *tch, const orc_cf *csd, int len, int sps, int arfcn,
                    const uint8_t *kc, struct orc_rx_record *out, int max_records, int *n_records,
                    struct orc_rx_big_record *big, int max_big, int *n_big, int *n_chains)
{
	struct chan_desc cd;
	struct sink s = { out, max_records, 0, arfcn, 0, big, max_big, 0 };
	int rv, toa, base_align, mtoa[16], n_fcch, i, j;
	float ref_snr = 0.0f, ref_freq_err = 0.0f;
	const struct orc_fcch_burst *ft = &orc_fcch_burst;

	*n_records = 0;
	if (n_big) *n_big = 0;
	if (n_chains) *n_chains = 0;
	memset(&cd, 0, sizeof(cd));
	cd.iq = iq; cd.tch = tch; cd.csd = csd; cd.len = len; cd.sps = sps;
	if (kc)
		memcpy(cd.kc, kc, 8);
	cd.align = START_DISCARD;                       /* gmr1_rx.c:906-909 */

	/* fcch_single_init, gmr1_rx.c:605-639 */
	{
		int wl = (330 * SYM_RATE * sps) / 1000;
		if (cd.align + wl > len)
			return -1;
		rv = orc_fcch_rough(ft, iq + cd.align, wl, sps, 0.0f, &toa);
		if (rv)
			return rv;
		cd.align += toa;
		/* the reference does not check this win_map (gmr1_rx.c:627); out of samples = error here */
		if (cd.align + ft->len * sps > len)
			return -1;
		rv = orc_fcch_fine(ft, iq + cd.align, ft->len * sps, sps, 0.0f, &toa, &cd.freq_err);
		if (rv)
			return rv;
		cd.align += toa;
	}

	/* fcch_multi_process, gmr1_rx.c:643-744 */
	base_align = cd.align - ft->len * sps;
	if (base_align < 0)
		base_align = 0;
	{
		int wl = (650 * SYM_RATE * sps) / 1000;
		if (base_align + wl > len)
			return -1;
		rv = orc_fcch_rough_multi(ft, iq + base_align, wl, sps, -cd.freq_err, mtoa, 16);
		if (rv < 0)
			return rv;
		n_fcch = rv;
	}
	for (i = 0, j = 0; i < n_fcch; i++) {
		float freq_err, snr = 0.0f;
		int ftoa;
		if (base_align + mtoa[i] < 0 || base_align + mtoa[i] + ft->len * sps > len)
			return -1;
		rv = orc_fcch_fine(ft, iq + base_align + mtoa[i], ft->len * sps, sps, -cd.freq_err, &ftoa, &freq_err);
		if (rv)
			return rv;
		if (base_align + mtoa[i] + ftoa < 0 || base_align + mtoa[i] + ftoa + ft->len * sps > len)
			return -1;
		orc_fcch_snr(ft, iq + base_align + mtoa[i] + ftoa, ft->len * sps, sps, -(cd.freq_err + freq_err), &snr);
		if (i == 0) {
			ref_snr = snr;
			ref_freq_err = freq_err;
		} else {
			if (snr < 2.0f)
				continue;
			if (snr < (ref_snr / 6.0f))
				continue;
			if (to_hz((float)fabs(ref_freq_err - freq_err)) > 500.0f)
				continue;
		}
		mtoa[j++] = mtoa[i] + ftoa;
	}
	n_fcch = j;
	if (n_chains) *n_chains = n_fcch;

	for (i = 0; i < n_fcch; i++) {
		struct chan_desc cdl = cd;
		cdl.align = base_align + mtoa[i];
		s.chain = i;
		process_bcch(&cdl, &s);
	}
	*n_records = s.n;
	if (n_big) *n_big = s.n_big;
	return 0;
}
