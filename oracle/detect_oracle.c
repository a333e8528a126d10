/*
 * CPU oracle of the frame-detection front end (TEST INFRASTRUCTURE, never the product path): the blocks that
 * sit upstream of `signal` in the reference's receive flowgraph (examples/rx.grc:
 * presiso -> trigger -> sync -> signal).  Plain sequential restatements:
 *
 *   orc_presiso  examples/presiso.grc:68-137,218-229 -- a hier block of stock GNU Radio blocks: delay 16,
 *                multiply-conjugate, moving average 48 (complex), complex-to-mag, |x|^2, moving average 64,
 *                divide.  GNU Radio is a third-party dependency that is not in the reference tree
 *                (find_package(Gnuradio "3.10"), CMakeLists.txt); its moving average is a running sum that is
 *                re-seeded from the block's history at the start of every work call and handles at most
 *                max_iter (4000) items per call -- restated here with calls of exactly max_iter items.  The
 *                scheduler's real call sizes are not reproducible, so this stage is pinned to tolerance only.
 *   orc_trigger  lib/trigger_impl.cc:59-117
 *   orc_sync     lib/sync_impl.cc:61-196 (ltf_autoCorrelation 139-171, ltf_cfo 173-191), driven the way the
 *                scheduler drives it when input is plentiful: idle scans consume up to the trigger byte, the
 *                sync state waits for SYNC_MAX_BUF_LEN (240) items and consumes SYNC_MAX_RES_LEN (111).
 *
 * Pinned by: the reference's IQ fixtures (sync byte expected at L-LTF start + 16 = sample 1376, the position
 * the RX fixtures of this repository already use) and the Python-reference frames under tests/golden/.
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef float complex cf;

#define SYNC_MAX_BUF_LEN 240
#define SYNC_MAX_RES_LEN (SYNC_MAX_BUF_LEN - 128 - 1)
#define MAVG_MAX_ITER 4000

/* x[n] for n < 0 is the zero history GNU Radio blocks start with. */
static inline cf hist(const cf *x, long n) { return n < 0 ? 0.0f : x[n]; }

/* ac[n] = |sum_{k<48} x[n-k-16] conj(x[n-k])| / sum_{k<64} |x[n-k]|^2 ; conj[n] = the 48-term sum. */
void orc_presiso(const float *x_ri, long n, float *ac, float *conj_ri)
{
    const cf *x = (const cf *)x_ri;
    cf *cj = (cf *)conj_ri;
    for (long base = 0; base < n; base += MAVG_MAX_ITER) {
        long cnt = n - base < MAVG_MAX_ITER ? n - base : MAVG_MAX_ITER;
        /* seed: the length-1 items before the first output of this call */
        cf s48 = 0.0f;
        float s64 = 0.0f;
        for (long k = base - 47; k < base; k++) s48 += hist(x, k - 16) * conjf(hist(x, k));
        for (long k = base - 63; k < base; k++) {
            cf v = hist(x, k);
            s64 += crealf(v) * crealf(v) + cimagf(v) * cimagf(v);
        }
        for (long i = base; i < base + cnt; i++) {
            cf vi = hist(x, i);
            s48 += hist(x, i - 16) * conjf(vi);
            s64 += crealf(vi) * crealf(vi) + cimagf(vi) * cimagf(vi);
            cj[i] = s48;
            /* complex_to_mag: sqrt(re^2 + im^2) in float; divide: float */
            float mag = sqrtf(crealf(s48) * crealf(s48) + cimagf(s48) * cimagf(s48));
            ac[i] = mag / s64;
            s48 -= hist(x, i - 47 - 16) * conjf(hist(x, i - 47));
            cf vo = hist(x, i - 63);
            s64 -= crealf(vo) * crealf(vo) + cimagf(vo) * cimagf(vo);
        }
    }
}

/* lib/trigger_impl.cc:74-108: bit 0 = trigger, bit 1 = "autocorrelation at a new maximum of this plateau". */
void orc_trigger(const float *ac, long n, uint8_t *trig)
{
    int nPlateau = 0, fPlateau = 0, fPlateauEnd = 0, countDown = 0;
    float conjAc = 0.0f;
    for (long i = 0; i < n; i++) {
        trig[i] = 0;
        if (ac[i] > 0.3f) {
            nPlateau++;
            if (ac[i] > conjAc) {
                conjAc = ac[i];
                trig[i] |= 0x02;
            }
            if (nPlateau > 20 && (fPlateau + fPlateauEnd) == 0) {
                fPlateau = 1;
                fPlateauEnd = 1;
                countDown = 80;
            }
        } else {
            nPlateau = 0;
            fPlateauEnd = 0;
            conjAc = 0.0f;
        }
        if (fPlateau) {
            countDown--;
            if (countDown == 0) {
                fPlateau = 0;
                trig[i] |= 0x01;
            }
        }
    }
}

/* lib/sync_impl.cc:139-171 */
static void ltf_autocorr(const cf *sig, float *ac, float *pwr)
{
    cf ms = 0.0f;
    float s1 = 0.0f, s2 = 0.0f;
    for (int i = 0; i < 64; i++) {
        ms += sig[i] * conjf(sig[i + 64]);
        s1 += cabsf(sig[i]) * cabsf(sig[i]);
        s2 += cabsf(sig[i + 64]) * cabsf(sig[i + 64]);
    }
    for (int i = 0; i < SYNC_MAX_RES_LEN; i++) {
        ac[i] = cabsf(ms) / sqrtf(s1) / sqrtf(s2);
        pwr[i] = s1;
        ms -= sig[i] * conjf(sig[i + 64]);
        s1 -= cabsf(sig[i]) * cabsf(sig[i]);
        s2 -= cabsf(sig[i + 64]) * cabsf(sig[i + 64]);
        ms += sig[i + 64] * conjf(sig[i + 128]);
        s1 += cabsf(sig[i + 64]) * cabsf(sig[i + 64]);
        s2 += cabsf(sig[i + 128]) * cabsf(sig[i + 128]);
    }
}

/* lib/sync_impl.cc:173-191 */
static float ltf_cfo(const cf *sig, cf conj_avg)
{
    cf tmp[128];
    cf sum = 0.0f;
    float step_stf = atan2f(cimagf(conj_avg), crealf(conj_avg)) / 16.0f;
    for (int i = 0; i < 128; i++) tmp[i] = sig[i] * (cosf(i * step_stf) + I * sinf(i * step_stf));
    for (int i = 0; i < 64; i++) sum += tmp[i] * conjf(tmp[i + 64]);
    cf avg = sum / 64.0f;
    float step_ltf = atan2f(cimagf(avg), crealf(avg)) / 64.0f;
    return step_stf + step_ltf;
}

/*
 * lib/sync_impl.cc:61-137.  Writes one record per emitted sync byte: position (absolute item index), rad, snr,
 * rssi.  Also returns, per record, the trigger index it came from and the index whose conj value was used
 * (-1: none seen yet), for diagnosis.  Returns the number of records (at most cap are stored).
 */
long orc_sync(const uint8_t *trig, const float *conj_ri, const float *sig_ri, long n, long *pos, float *rad,
              float *snr, float *rssi, long *trig_idx, long *conj_idx, long cap)
{
    const cf *cj = (const cf *)conj_ri;
    const cf *sig = (const cf *)sig_ri;
    cf conj_avg = 0.0f;
    long conj_at = -1, count = 0, i = 0;
    float ac[SYNC_MAX_RES_LEN], pwr[SYNC_MAX_RES_LEN];
    while (i < n) {
        if (!(trig[i] & 0x01)) {
            if (trig[i] & 0x02) {
                conj_avg = cj[i];
                conj_at = i;
            }
            i++;
            continue;
        }
        /* sync state: needs 240 items from the trigger byte on */
        if (n - i < SYNC_MAX_BUF_LEN) break;
        ltf_autocorr(sig + i, ac, pwr);
        int mx = 0;
        for (int j = 1; j < SYNC_MAX_RES_LEN; j++)
            if (ac[j] > ac[mx]) mx = j; /* std::max_element: first maximum */
        if (ac[mx] > 0.5) {
            float lim = ac[mx] * 0.8; /* the reference stores the double product in a float */
            double mxd = (double)ac[mx];
            int l = mx, r = mx;
            for (int j = mx; j >= 0; j--)
                if (ac[j] < lim) { l = j; break; }
            for (int j = mx; j < SYNC_MAX_RES_LEN; j++)
                if (ac[j] < lim) { r = j; break; }
            int m = (l + r) / 2;
            if (count < cap) {
                pos[count] = i + m;
                rad[count] = ltf_cfo(sig + i + m, conj_avg);
                snr[count] = (float)(10.0 * log10(mxd / (1.0 - mxd)));
                rssi[count] = pwr[mx] / 64.0f;
                if (trig_idx) trig_idx[count] = i;
                if (conj_idx) conj_idx[count] = conj_at;
            }
            count++;
        }
        i += SYNC_MAX_RES_LEN;
    }
    return count;
}

/* presiso -> trigger -> sync over one capture. */
long orc_detect(const float *x_ri, long n, long *pos, float *rad, float *snr, float *rssi, long cap, float *ac_ws,
                float *conj_ws, uint8_t *trig_ws)
{
    orc_presiso(x_ri, n, ac_ws, conj_ws);
    orc_trigger(ac_ws, n, trig_ws);
    return orc_sync(trig_ws, conj_ws, x_ri, n, pos, rad, snr, rssi, 0, 0, cap);
}
