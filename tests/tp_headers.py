"""Declaration-only headers for the libosmocore / libosmo-dsp names the reference's APPLICATION (src/gmr1_rx.c, src/gsmtap.c)
uses -- just those two files' needs: types, constants, prototypes.  Written into a scratch directory so that the
unchanged sources compile in an image that has neither library (tests/test_link_gmr1_rx.py: the link census;
tests/ref_rx_program.py: the program that is actually run).  No arithmetic lives here."""
import os
import textwrap

HEADERS = {
    "osmocom/core/bits.h": """
        #include <stdint.h>
        typedef int8_t sbit_t; typedef uint8_t ubit_t; typedef uint8_t pbit_t;
    """,
    "osmocom/core/utils.h": """
        #include <stdint.h>
        int osmo_hexparse(const char *str, uint8_t *b, int max_len);
        char *osmo_hexdump_nospc(const unsigned char *buf, int len);
    """,
    "osmocom/core/msgb.h": """
        #include <stdint.h>
        struct msgb;
        struct msgb *msgb_alloc(uint16_t size, const char *name);
        void msgb_free(struct msgb *m);
        unsigned char *msgb_put(struct msgb *msgb, unsigned int len);
    """,
    "osmocom/core/gsmtap.h": """
        #include <stdint.h>
        #define GSMTAP_VERSION 0x02
        #define GSMTAP_UDP_PORT 4729
        #define GSMTAP_TYPE_GMR1_UM 0x0a
        #define GSMTAP_GMR1_BCCH 0x01
        #define GSMTAP_GMR1_CCCH 0x02
        #define GSMTAP_GMR1_TCH3 0x10
        #define GSMTAP_GMR1_TCH9 0x18
        #define GSMTAP_GMR1_FACCH 0x02
        struct gsmtap_hdr {
            uint8_t version, hdr_len, type, timeslot; uint16_t arfcn; int8_t signal_dbm, snr_db;
            uint32_t frame_number; uint8_t sub_type, antenna_nr, sub_slot, res;
        } __attribute__((packed));
    """,
    "osmocom/core/gsmtap_util.h": """
        #include <stdint.h>
        #include <osmocom/core/msgb.h>
        struct gsmtap_inst;
        struct gsmtap_inst *gsmtap_source_init(const char *host, uint16_t port, int ofd_wq_mode);
        int gsmtap_source_add_sink(struct gsmtap_inst *gti);
        int gsmtap_sendmsg(struct gsmtap_inst *gti, struct msgb *msg);
    """,
    "osmocom/dsp/cxvec.h": """
        #include <complex.h>
        #define CXVEC_FLG_REAL_ONLY (1 << 0)
        struct osmo_cxvec { int len, max_len, flags; float complex *data; float complex _data[0]; };
        void osmo_cxvec_init_from_data(struct osmo_cxvec *cv, float complex *data, int len);
    """,
    "osmocom/dsp/cxvec_math.h": """
        #include <complex.h>
        #include <math.h>
        #define M_PIf ((float)M_PI)
        static inline float osmo_normsqf(float complex c) { return crealf(c) * crealf(c) + cimagf(c) * cimagf(c); }
    """,
    "osmocom/dsp/cfile.h": """
        #include <complex.h>
        struct cfile { float complex *data; unsigned int len; unsigned int _blen; };
        struct cfile *cfile_load(const char *filename);
        void cfile_release(struct cfile *cf);
    """,
}



def write(root):
    """Write the headers under `root` (one include directory); returns it."""
    for rel, txt in HEADERS.items():
        p = os.path.join(str(root), rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        guard = rel.replace("/", "_").replace(".", "_").upper()
        with open(p, "w") as fh:
            fh.write(f"#ifndef {guard}\n#define {guard}\n{textwrap.dedent(txt)}\n#endif\n")
    return str(root)
