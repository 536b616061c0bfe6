/* TEST-ONLY stand-ins for the eleven libosmocore / libosmo-dsp functions the reference's application
 * (src/gmr1_rx.c:32,139-190,900-991 and src/gsmtap.c:43-71) calls besides the gmr1_* API -- file loading, a message
 * buffer and the GSMTAP "socket".  None of them holds receive-path arithmetic, so this pins nothing and is never part
 * of the product or of the oracle: it exists so that the UNCHANGED program can be run on a synthetic capture against
 * libgmr1_hip.so (tests/test_gpu_ref_program.py) and its GSMTAP messages compared with gmr1_hip_rx_run's records.
 *
 * gsmtap_sendmsg() appends [u32 length][message bytes] to the file named by GMR1_TEST_GSMTAP_OUT (instead of a UDP
 * datagram to 127.0.0.1:4729) and frees the message, as the library does after a successful send. */
#include <complex.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* layouts as in the declaration-only headers the program is compiled against (tests/tp_headers.py) */
struct osmo_cxvec { int len, max_len, flags; float complex *data; float complex _data[0]; };
struct cfile { float complex *data; unsigned int len; unsigned int _blen; };
struct msgb { unsigned int size, len; unsigned char *data; };
struct gsmtap_inst { FILE *out; };

struct cfile *cfile_load(const char *filename)
{
	FILE *f = fopen(filename, "rb");
	if (!f)
		return NULL;
	fseek(f, 0, SEEK_END);
	long bytes = ftell(f);
	fseek(f, 0, SEEK_SET);
	struct cfile *cf = calloc(1, sizeof(*cf));
	cf->len = (unsigned int)(bytes / (long)sizeof(float complex));
	cf->_blen = (unsigned int)bytes;
	cf->data = malloc(bytes > 0 ? (size_t)bytes : 1);
	if (fread(cf->data, sizeof(float complex), cf->len, f) != cf->len) {
		fclose(f);
		free(cf->data);
		free(cf);
		return NULL;
	}
	fclose(f);
	return cf;
}

void cfile_release(struct cfile *cf)
{
	if (cf) {
		free(cf->data);
		free(cf);
	}
}

void osmo_cxvec_init_from_data(struct osmo_cxvec *cv, float complex *data, int len)
{
	cv->len = cv->max_len = len;
	cv->flags = 0;
	cv->data = data;
}

int osmo_hexparse(const char *str, uint8_t *b, int max_len)
{
	int n = 0, have = 0;
	unsigned v = 0;
	memset(b, 0, (size_t)max_len);
	for (; *str; str++) {
		int d;
		if (*str >= '0' && *str <= '9') d = *str - '0';
		else if (*str >= 'a' && *str <= 'f') d = *str - 'a' + 10;
		else if (*str >= 'A' && *str <= 'F') d = *str - 'A' + 10;
		else if (*str == ' ' || *str == '\t' || *str == '\n' || *str == '\r') continue;
		else return -1;
		v = (v << 4) | (unsigned)d;
		if (++have == 2) {
			if (n >= max_len)
				return -1;
			b[n++] = (uint8_t)v;
			have = 0;
			v = 0;
		}
	}
	return have ? -1 : n;
}

char *osmo_hexdump_nospc(const unsigned char *buf, int len)
{
	static char out[4096];
	int o = 0;
	for (int i = 0; i < len && o + 3 < (int)sizeof(out); i++)
		o += sprintf(out + o, "%02x", buf[i]);
	out[o] = 0;
	return out;
}

struct msgb *msgb_alloc(uint16_t size, const char *name)
{
	(void)name;
	struct msgb *m = calloc(1, sizeof(*m));
	m->size = size;
	m->data = calloc(1, size ? size : 1);
	return m;
}

void msgb_free(struct msgb *m)
{
	if (m) {
		free(m->data);
		free(m);
	}
}

unsigned char *msgb_put(struct msgb *m, unsigned int len)
{
	if (m->len + len > m->size)
		abort();
	unsigned char *p = m->data + m->len;
	m->len += len;
	return p;
}

struct gsmtap_inst *gsmtap_source_init(const char *host, uint16_t port, int ofd_wq_mode)
{
	(void)host; (void)port; (void)ofd_wq_mode;
	static struct gsmtap_inst gti;
	const char *path = getenv("GMR1_TEST_GSMTAP_OUT");
	gti.out = path ? fopen(path, "wb") : NULL;
	return &gti;
}

int gsmtap_source_add_sink(struct gsmtap_inst *gti)
{
	(void)gti;
	return 0;
}

int gsmtap_sendmsg(struct gsmtap_inst *gti, struct msgb *msg)
{
	if (!gti || !msg)
		return -1;
	if (gti->out) {
		uint32_t n = msg->len;
		fwrite(&n, 4, 1, gti->out);
		fwrite(msg->data, 1, n, gti->out);
		fflush(gti->out);
	}
	msgb_free(msg);
	return 0;
}
