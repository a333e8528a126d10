// Descrambler, (A-)MPDU assembly and FCS check of one decoded frame: shared by the decoder kernels.
#pragma once

#include "rx_common.h"

namespace gr80211 {

// CRC-8 of the SIG fields: x^8 + x^2 + x + 1 (0x07), register preset to ones, field bits fed in order
// (IEEE 802.11-2016 19.3.9.4.4; genCrc8Bits / checkBitCrc8, lib/cloud80211phy.cc:1329-1403).
__device__ __forceinline__ unsigned crc8_step(unsigned c, unsigned bit)
{
    return ((c << 1) & 0xffu) ^ ((((c >> 7) ^ bit) & 1u) ? 0x07u : 0u);
}

__device__ __forceinline__ uint32_t crc_byte(uint32_t c, uint32_t byte, const uint32_t *tab)
{
    return tab[(c ^ byte) & 0xff] ^ (c >> 8);
}

// A-MPDU delimiter check (IEEE 802.11-2016 9.7.1): CRC-8 of its first 16 bits in bits 16..23 (first transmitted CRC
// bit = bit 16) and the signature 0x4E in bits 24..31.
__device__ __forceinline__ bool delimiter_ok(uint32_t d)
{
    unsigned c = 0xff;
    for (int i = 0; i < 16; i++) c = crc8_step(c, (d >> i) & 1u);
    c = 0xff - c;
    unsigned r = 0;
    for (int i = 0; i < 8; i++) r |= ((c >> (7 - i)) & 1u) << i;
    return ((d >> 16) & 0xffu) == r && (d >> 24) == 0x4Eu;
}

// One frame from its decoded bits to its PDU slot.  bw: the frame's decoded bits, 32 per word, word k at
// bw[k * bstride] (bit t of the frame = bit t % 32 of word t / 32), nwords of them; s_crc: the CRC-32 byte table in LDS.
__device__ __forceinline__ void assemble_frame(const uint32_t *__restrict__ bw, size_t bstride, const gr80211_frame_info &fi,
                                               const int fr, const int nwords, const uint32_t *s_crc, uint8_t *__restrict__ out,
                                               int out_stride, int32_t *__restrict__ out_len, int32_t *__restrict__ n_pdu,
                                               uint8_t *__restrict__ bits_out, int bits_stride, int strict)
{
    const int T = fi.trellis;

    // ---- descrambler sequence (lib/decode_impl.cc:304-323).  With S[0..6] = the first seven
    // decoded bits and S[i] = S[i-7]^S[i-4], the data bits are decoded[i]^S[i] (zero for i<7).
    uint32_t W0 = 0, W1 = 0;
    {
        const uint32_t first = bw[0];
        uint32_t hist = 0;  // bit k = S[i-1-k]
        for (int i = 0; i < 64; i++) {
            uint32_t sbit;
            if (i < 7) sbit = (first >> i) & 1u;
            else sbit = ((hist >> 6) ^ (hist >> 3)) & 1u;
            hist = (hist << 1) | sbit;
            if (i < 32) W0 |= sbit << i;
            else W1 |= sbit << (i - 32);
        }
    }

    // ---- packet assembly (lib/decode_impl.cc:325-520).  PSDU dword q = data bits 16+32q .. 16+32q+31.
    uint8_t *o = out + (size_t)fr * out_stride;
    uint8_t *bo = bits_out ? bits_out + (size_t)fr * bits_stride : nullptr;
    const bool vht = fi.format == GR80211_F_VHT;
    const int T32 = (T + 31) / 32;
    int wpos = 0;      // bytes committed to the output slot
    int npdu = 0;
    // MPDU being collected
    int mlen = -1;     // -1: expecting a delimiter (VHT) ; >= 0 collecting
    int mgot = 0, mpad = 0;
    uint32_t crc = 0xFFFFFFFFu;
    int tl = 0;        // delimiter length, OR-accumulated as the reference does (lib/decode_impl.cc:336)
    int eof = 0;
    int procd = 16;    // bits consumed, as tmpBitsProcd
    bool done = false;
    // strict: de-aggregate by the standard's rules instead of the reference's (VHT and aggregated HT PSDUs): every
    // delimiter is checked (CRC-8, signature) and carries its own length, the search resumes on the next 4-byte
    // boundary after a bad one, zero-length delimiters are padding
    const bool walk = strict && (vht || fi.ampdu);
    const int plen = fi.len < (T - 16) / 8 ? fi.len : (T - 16) / 8;  // PSDU bytes the walk may use
    if (walk) {
        mlen = -1;
    } else if (!vht) {
        if (fi.ampdu) done = true;  // HT A-MPDU: nothing published (lib/decode_impl.cc:432-435)
        mlen = fi.len; mpad = fi.len;
    } else if (procd >= T) {
        done = true;
    }
    uint32_t Sprev2 = W0, Sprev1 = W1;  // scrambler words k-2, k-1 relative to the next one to make
    uint32_t Dcur = bw[0] ^ W0;         // descrambled word 0
    if (bo && bits_stride >= 4) *(uint32_t *)(bo) = Dcur;
    for (int k = 1; k <= T32 && k <= nwords; k++) {
        // descrambled word k
        uint32_t Sk;
        if (k == 1) Sk = W1;
        else { Sk = Sprev1 ^ ((Sprev2 >> 8) | (Sprev1 << 24)); Sprev2 = Sprev1; Sprev1 = Sk; }
        const uint32_t raw = k < nwords ? bw[(size_t)k * bstride] : 0u;
        const uint32_t Dnext = raw ^ Sk;
        if (bo && k < T32 && (k + 1) * 4 <= bits_stride) *(uint32_t *)(bo + (size_t)k * 4) = Dnext;
        const uint32_t psdu = (Dcur >> 16) | (Dnext << 16);  // PSDU dword q = k-1
        Dcur = Dnext;
        if (done) continue;
        if (walk && mlen < 0) {
            const int at = 4 * (k - 1);  // byte offset of this dword in the PSDU
            if (at + 4 > plen) { done = true; continue; }
            if (!delimiter_ok(psdu)) continue;
            const int l = (int)((psdu >> 4) & 0xfffu) | (vht ? (int)((psdu >> 2) & 3u) << 12 : 0);
            if (l == 0 || at + 4 + l > plen) continue;
            mlen = l; mpad = (l + 3) & ~3; mgot = 0; crc = 0xFFFFFFFFu;
            continue;
        }
        if (mlen < 0) {
            // A-MPDU delimiter: bit0 EOF, bits2-3 len[13:12], bits4-15 len[11:0] (lib/decode_impl.cc:339-358)
            procd += 32;
            if (procd > T) { done = true; continue; }
            eof = (int)(psdu & 1u);
            tl |= (int)((psdu >> 2) & 1u) << 12;
            tl |= (int)((psdu >> 3) & 1u) << 13;
            tl |= (int)((psdu >> 4) & 0xfffu);
            const int padded = (tl / 4 + ((tl % 4) != 0)) * 4;
            procd += padded * 8;
            if (procd > T) { done = true; continue; }
            mlen = tl; mpad = padded; mgot = 0; crc = 0xFFFFFFFFu;
            if (mpad == 0) {  // zero-length subframe: CRC of nothing never matches; move on
                mlen = -1;
                if (eof) done = true;
            }
            continue;
        }
        // payload dword
        const bool store = (mlen <= GR80211_MAX_PSDU) && (wpos + 3 + mlen + 1 <= out_stride);
#pragma unroll
        for (int b = 0; b < 4; b++) {
            if (mgot + b < mlen) {
                const uint32_t by = (psdu >> (8 * b)) & 0xffu;
                crc = crc_byte(crc, by, s_crc);
                if (store) o[wpos + 3 + mgot + b] = (uint8_t)by;
            }
        }
        mgot += 4;
        if (mgot >= mpad) {
            if (store && (crc ^ 0xFFFFFFFFu) == 558161692u) {  // lib/decode_impl.cc:374,453
                o[wpos] = (uint8_t)fi.format;
                o[wpos + 1] = (uint8_t)(mlen % 256);
                o[wpos + 2] = (uint8_t)(mlen / 256);
                o[wpos + 3 + mlen] = (uint8_t)fi.mcs;
                wpos += mlen + 4;
                npdu++;
            }
            if (walk) {
                mlen = -1;
            } else if (vht) {
                mlen = -1;
                if (eof) done = true;
            } else {
                done = true;
            }
        }
    }
    out_len[fr] = wpos;
    n_pdu[fr] = npdu;
}

}  // namespace gr80211
