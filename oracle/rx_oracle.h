/*
 * CPU oracle for the 802.11 OFDM RX hot path (signal -> demod/demod2 -> decode).
 *
 * TEST INFRASTRUCTURE ONLY.  This is a scalar C restatement of the reference
 * algorithm, used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg as the checker / CPU baseline.  The product path (the HIP library) never
 * links, loads or calls anything in this directory.
 *
 * Each function cites the reference file:line it restates (paths relative to the
 * reference tree root).
 */
#ifndef RX_ORACLE_H
#define RX_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_F_L 0
#define ORC_F_HT 1
#define ORC_F_VHT 2

#define ORC_MAX_TRELLIS 32782 /* lib/decode_impl.h:36 */
#define ORC_MAX_PSDU 4095     /* lib/decode_impl.h:35 */

/* Why a frame stopped before producing LLRs / a PDU. */
enum {
    ORC_OK = 0,
    ORC_DROP_LSIG = 1,       /* L-SIG parity/rate/length check failed (signal_impl.cc:156-160) */
    ORC_DROP_LEN_CHECK = 2,  /* HT/VHT length vs L-SIG duration sanity failed (demod_impl.cc:159,186) */
    ORC_DROP_UNSUPPORTED = 3,/* nSS>2, invalid MCS, MU frame on two antennas, two data streams on one antenna */
    ORC_DROP_DECODE_LIMIT = 4,/* len>4095 or trellis>32782 (decode_impl.cc:93) */
    ORC_NDP = 5              /* VHT sounding NDP on the one-antenna chain: channel message instead of a PSDU
                                (demod_impl.cc:238-249,397-398, decode_impl.cc:100-121) */
};

/* Modulation descriptor: the fields of c8p_mod (lib/cloud80211phy.h:58-98) the path uses. */
typedef struct {
    int format, ampdu, nSym, nSymSamp, nSD, nSS, nLTF;
    int mcs, len, mod, cr, nBPSCS, nDBPS, nCBPS, nCBPSS;
} orc_mod_t;

/* Everything the three blocks hand to each other for one frame (tag contract). */
typedef struct {
    int status;          /* ORC_OK or ORC_DROP_* */
    /* signal -> demod tags (signal_impl.cc:135-152) */
    int lsig_mcs, lsig_len, nsamp;
    float h_l[128];      /* chan: 64 complex, re/im interleaved */
    uint8_t lsig_bits[24];
    /* demod -> decode tags (demod_impl.cc:224-253) */
    orc_mod_t m;
    int trellis, total;
    float sssnr0, sssnr1;
    uint8_t sig_bits[48]; /* decoded HT-SIG or VHT-SIG-A bits */
    uint8_t sigb_bits[26];
    float mu2x1chan[256]; /* NDP: the 64 + 64 raw VHT-LTF samples, re/im interleaved (demod_impl.cc:397-398) */
    int sumu;             /* 1: VHT MU-MIMO frame (group id 1..62), cloud80211phy.h:62 */
    int mu_pos;           /* in: this station's position in the MU group, demod::make(mupos, mugid) (demod_impl.cc:28-39) */
} orc_frame_t;

void orc_init(void);

/* 64-point forward DFT, unnormalised, natural bin order (stands for gr::fft::fft_complex_fwd(64,1)). */
void orc_fft64(const float *in_ri, float *out_ri);

/* signal / signal2 (lib/signal_impl.cc:108-162): `in` points at the sync index of antenna 1. */
int orc_signal(const float *in_ri, int navail, float rad, orc_frame_t *f);

/* demod / demod2 (lib/demod_impl.cc:59-342, lib/demod2_impl.cc:58-349).
 * in0/in1 point at the sync index of each antenna (in1 == NULL: SISO demod block).
 * Writes f->total LLRs to llr (capacity llr_cap floats).  Returns f->status. */
int orc_demod(const float *in0_ri, const float *in1_ri, int navail, float rad,
              orc_frame_t *f, float *llr, int llr_cap);

/* decode (lib/decode_impl.cc:164-520): LLRs + tags -> concatenated PDU blobs
 * [format][len lo][len hi][MPDU...][mcs] for every MPDU that passes the FCS check.
 * bits_out (nullable, >= trellis bytes) receives the descrambled bit array v_unCodedBits.
 * Returns number of PDUs written; *out_len = total bytes. */
int orc_decode(const float *llr, int nllr, int format, int len, int cr, int mcs, int ampdu,
               int trellis, uint8_t *out, int out_cap, int *out_len, uint8_t *bits_out);

/* Same with the A-MPDU rule selectable: strict != 0 = the standard's de-aggregation instead of the reference's. */
int orc_decode_ex(const float *llr, int nllr, int format, int len, int cr, int mcs, int ampdu,
                  int trellis, int strict, uint8_t *out, int out_cap, int *out_len, uint8_t *bits_out);

/* Whole chain for one frame. */
int orc_rx_frame(const float *in0_ri, const float *in1_ri, int navail, float rad,
                 orc_frame_t *f, float *llr, int llr_cap,
                 uint8_t *out, int out_cap, int *out_len, int *n_pdu);

/* Same with the station's MU group position (orc_rx_frame = position 0). */
int orc_rx_frame_mu(const float *in0_ri, const float *in1_ri, int navail, float rad, int mu_pos,
                    orc_frame_t *f, float *llr, int llr_cap,
                    uint8_t *out, int out_cap, int *out_len, int *n_pdu);

int orc_rx_frame_ex(const float *in0_ri, const float *in1_ri, int navail, float rad, int mu_pos, int strict,
                    orc_frame_t *f, float *llr, int llr_cap,
                    uint8_t *out, int out_cap, int *out_len, int *n_pdu);

/* Test hook: until it is called again with NULL, orc_demod (and the whole-chain calls) of this thread copy the
 * equalised constellation points of every data symbol to buf_ri ([symbol][stream][nSD] complex); orc_captured_points
 * = complex values written by the calls since. */
void orc_capture_points(float *buf_ri, int cap_complex);
int orc_captured_points(void);

/* Small pieces exported for table / known-answer tests. */
void orc_sig_viterbi(const float *llr, uint8_t *bits, int trellis);      /* cloud80211phy.cc:2001-2088 */
int orc_viterbi_bits(const float *llr, int nllr, int cr, int trellis, uint8_t *scram_bits); /* decode_impl.cc:205-302 */
void orc_get_table(const char *name, int *out, int cap, int *n);
uint32_t orc_crc32(const uint8_t *p, int n);
void orc_bcc_encode(const uint8_t *in, uint8_t *out, int n);            /* cloud80211phy.cc:2622-2644 */
void orc_crc8_bits(const uint8_t *in, int n, uint8_t *out8);            /* cloud80211phy.cc:1329-1365 */

#ifdef __cplusplus
}
#endif
#endif
