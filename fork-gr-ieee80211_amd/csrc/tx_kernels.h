// Synthetic transmitter (internal to the library).
#pragma once

#include "rx_common.h"

namespace gr80211 {

// Modulation and length bookkeeping of one frame to transmit (tools/phy80211header.py:433-536).
struct TxMod {
    int fmt, mcs, nss, nbpscs, cr, nsd, ncbpss, ncbps, ndbps, nltf, nsym;
    int psdu;      // PSDU bytes on air (VHT: A-MPDU plus padding)
    int npad_eof;  // VHT: EOF padding delimiters appended to the A-MPDU
    int lsig_len;  // L-SIG LENGTH field
    int npre;      // samples before the data field
    int nsamp;     // samples of the whole frame
};

__host__ __device__ bool tx_mod(int fmt, int mcs, int nss, int len, TxMod &m);

int upload_tx_tables(const Tables &t);
void launch_tx(const uint8_t *psdu, uint64_t psdu_bytes, const gr80211_tx_desc *desc, int n, void *iq0, void *iq1,
               uint64_t arena_len, int32_t *n_samples, hipStream_t s);

void launch_tx_mu(const uint8_t *psdu, uint64_t psdu_bytes, const gr80211_tx_mu_desc *desc, int n, const void *qmat, int n_q,
                  void *iq0, void *iq1, uint64_t arena_len, int32_t *n_samples, hipStream_t s);

}  // namespace gr80211
