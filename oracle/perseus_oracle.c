/*
 * perseus_oracle.c -- see perseus_oracle.h.  TEST INFRASTRUCTURE ONLY.
 *
 * Build: make -C oracle   (gcc -O3 -march=native -fopenmp, NO -ffast-math:
 * the unpack divide must stay an IEEE divide; SURVEY.md 8c shows fast-math
 * changes no bit, but the oracle should not depend on that.)
 */
#include "perseus_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------ */
uint32_t orc_lcg_fill(uint8_t *dst, size_t nbytes, uint32_t seed)
{
    uint32_t s = seed;
    for (size_t i = 0; i < nbytes; i++) {
        s = s * 1664525u + 1013904223u;
        dst[i] = (uint8_t)(s >> 24);
    }
    return s;
}

/* ------------------------------------------------------------------------
 * A1 (perseustest.c:411-426): the three wire bytes of a component occupy
 * bytes 1..3 of a little-endian int32 whose byte 0 is zero, i.e. the value
 * is the sign-extended 24-bit sample times 256.
 */
static inline int32_t msb_align24(const uint8_t *p)
{
    uint32_t u = ((uint32_t)p[0] << 8) | ((uint32_t)p[1] << 16) |
                 ((uint32_t)p[2] << 24);
    return (int32_t)u;
}

/* A2 (perseustest.c:466-502): float = (float)int32 / (INT_MAX - 256); the
 * int divisor is converted to float by the usual arithmetic conversions
 * (2147483391 rounds to 2147483392.0f). */
void orc_unpack24_f32(const uint8_t *in, size_t nbytes, float *out_iq)
{
    const size_t ns = nbytes / 6;
    const float full_scale = (float)(INT_MAX - 256);
    for (size_t k = 0; k < ns; k++) {
        const uint8_t *p = in + 6 * k;
        out_iq[2 * k + 0] = (float)msb_align24(p) / full_scale;
        out_iq[2 * k + 1] = (float)msb_align24(p + 3) / full_scale;
    }
}

/* A3 (perseustest.c:432-460): MSB-aligned int32 pairs. */
void orc_unpack24_i32(const uint8_t *in, size_t nbytes, int32_t *out_iq)
{
    const size_t ns = nbytes / 6;
    for (size_t k = 0; k < ns; k++) {
        const uint8_t *p = in + 6 * k;
        out_iq[2 * k + 0] = msb_align24(p);
        out_iq[2 * k + 1] = msb_align24(p + 3);
    }
}

void orc_unpack24_f32_callback_style(const uint8_t *in, size_t nbytes,
                                     float *out_iq, size_t buf_bytes)
{
    size_t off = 0;
    while (off < nbytes) {
        size_t n = nbytes - off < buf_bytes ? nbytes - off : buf_bytes;
        orc_unpack24_f32(in + off, n, out_iq + 2 * (off / 6));
        off += n;
    }
}

/* N1 (authored): inverse of A2.  x*8388607 is evaluated in float (one
 * rounding), then rounded to the nearest integer, ties to even (rintf in the
 * default rounding mode), then saturated to the 24-bit range. */
static inline int32_t quant24(float x)
{
    float v = rintf(x * 8388607.0f);
    if (!(v >= -8388608.0f))
        v = -8388608.0f;          /* also catches NaN */
    if (v > 8388607.0f)
        v = 8388607.0f;
    return (int32_t)v;
}

void orc_pack24_f32(const float *in_iq, size_t ns, uint8_t *out)
{
    for (size_t k = 0; k < ns; k++) {
        const uint32_t i = (uint32_t)quant24(in_iq[2 * k]), q = (uint32_t)quant24(in_iq[2 * k + 1]);
        uint8_t *p = out + 6 * k;
        p[0] = (uint8_t)i; p[1] = (uint8_t)(i >> 8); p[2] = (uint8_t)(i >> 16);
        p[3] = (uint8_t)q; p[4] = (uint8_t)(q >> 8); p[5] = (uint8_t)(q >> 16);
    }
}

/* ------------------------------------------------------------------------
 * A6 (perseus-sdr.c:584): FREG = (uint32)(f / fclk * 2^32), double
 * arithmetic, truncation toward zero.
 */
uint32_t orc_nco_freg(double center_freq_hz, double adc_clk_hz)
{
    return (uint32_t)(center_freq_hz / adc_clk_hz * 4.294967296E9);
}

/* perseus-sdr.c:589-615 with the cut-off table of perseusfx2.h:70-93. */
int orc_presel_id(double f, int enable_presel)
{
    static const double fc[10] = { 1.7e6, 2.1e6, 3.0e6, 4.2e6, 6.0e6,
                                   8.4e6, 12.0e6, 17.0e6, 24.0e6, 32.0e6 };
    if (!enable_presel)
        return 10;
    for (int i = 0; i < 10; i++)
        if (f < fc[i])
            return i;
    return 10;
}

/* ------------------------------------------------------------------------
 * A7 (perseus-sdr.c:776-811): walk the ascending table; a request above an
 * entry moves on (or takes the last entry); otherwise compare with the
 * midpoint to the previous entry, midpoint itself going to the LOWER rate.
 */
int orc_rate_index(int sps, const int *table, int n)
{
    int prev = 0;
    for (int i = 0; i < n; i++) {
        if (sps > table[i]) {
            if (i < n - 1) {
                prev = table[i];
                continue;
            }
            return i;
        }
        int mid = (table[i] + prev) / 2;
        if (sps <= mid)
            return i == 0 ? 0 : i - 1;
        return i;
    }
    return -1;
}

/* ------------------------------------------------------------------------
 * A8 (authored).  Phase accumulator is exact integer arithmetic, so the LO
 * at absolute sample n depends only on n and freg.
 */
void orc_nco_mix_f64(const float *x_iq, size_t ns, uint64_t n0, uint32_t freg,
                     double *out_iq)
{
    const double two_pi_over_2p32 = 6.283185307179586476925286766559 / 4294967296.0;
    for (size_t n = 0; n < ns; n++) {
        uint32_t ph = (uint32_t)((n0 + n) * (uint64_t)freg);
        double a = two_pi_over_2p32 * (double)ph;
        double c = cos(a), s = -sin(a);   /* exp(-j a) = c + j s */
        double xr = x_iq[2 * n], xi = x_iq[2 * n + 1];
        out_iq[2 * n + 0] = xr * c - xi * s;
        out_iq[2 * n + 1] = xr * s + xi * c;
    }
}

/* The same NCO as a phase ACCUMULATOR that is retuned while it runs (authored; models the FPGA's
 * NCO, whose tuning word perseus_set_ddc_center_freq rewrites while streaming, perseus-sdr.c:584,
 * examples/fifo.c:43-49): sample n is mixed with exp(-j*2*pi*acc(n)/2^32), acc(0) = 0,
 * acc(n+1) = acc(n) + freg(n) mod 2^32, where freg(n) = word[i] for seg_start[i] <= n < seg_start[i+1].
 * A new word changes the increment, never the accumulated phase (phase-continuous retune).
 * seg_start[0] must be 0; n0 = absolute index of x_iq[0].                                          */
void orc_nco_mix_retuned_f64(const float *x_iq, size_t ns, uint64_t n0, const uint64_t *seg_start,
                             const uint32_t *word, int nseg, double *out_iq)
{
    const double two_pi_over_2p32 = 6.283185307179586476925286766559 / 4294967296.0;
    /* acc at the start of each segment */
    uint32_t acc0[64];
    if (nseg < 1 || nseg > 64 || seg_start[0] != 0)
        return;
    acc0[0] = 0;
    for (int i = 1; i < nseg; i++)
        acc0[i] = acc0[i - 1] + (uint32_t)((seg_start[i] - seg_start[i - 1]) * (uint64_t)word[i - 1]);
    int sg = 0;
    for (size_t n = 0; n < ns; n++) {
        const uint64_t na = n0 + n;
        while (sg + 1 < nseg && na >= seg_start[sg + 1])
            sg++;
        while (sg > 0 && na < seg_start[sg])
            sg--;
        const uint32_t ph = acc0[sg] + (uint32_t)((na - seg_start[sg]) * (uint64_t)word[sg]);
        const double a = two_pi_over_2p32 * (double)ph;
        const double c = cos(a), s = -sin(a);
        const double xr = x_iq[2 * n], xi = x_iq[2 * n + 1];
        out_iq[2 * n + 0] = xr * c - xi * s;
        out_iq[2 * n + 1] = xr * s + xi * c;
    }
}

size_t orc_fir_decim_f64(const double *x, size_t ns, const float *taps,
                         int ntaps, int D, double *y)
{
    if (D <= 0 || ntaps <= 0)
        return 0;
    const size_t nout = (ns + (size_t)D - 1) / (size_t)D;
#pragma omp parallel for schedule(static)
    for (long long m = 0; m < (long long)nout; m++) {
        const long long top = m * D;
        double ar = 0.0, ai = 0.0;
        int kmax = ntaps - 1;
        if ((long long)kmax > top)
            kmax = (int)top;
        for (int k = 0; k <= kmax; k++) {
            const double h = (double)taps[k];
            ar += h * x[2 * (top - k)];
            ai += h * x[2 * (top - k) + 1];
        }
        y[2 * m] = ar;
        y[2 * m + 1] = ai;
    }
    return nout;
}

size_t orc_resample_f64(const double *x, size_t ns, const float *taps, int ntaps, int L, int M, double *y)
{
    if (L <= 0 || M <= 0 || ntaps <= 0)
        return 0;
    const size_t nout = (ns * (size_t)L + (size_t)M - 1) / (size_t)M;
#pragma omp parallel for schedule(static)
    for (long long m = 0; m < (long long)nout; m++) {
        const unsigned long long t = (unsigned long long)m * (unsigned long long)M;
        const long long n = (long long)(t / (unsigned long long)L);
        const int ph = (int)(t % (unsigned long long)L);
        double ar = 0.0, ai = 0.0;
        long long j = 0;
        for (int k = ph; k < ntaps && n - j >= 0; k += L, j++) {
            const double h = (double)taps[k];
            ar += h * x[2 * (n - j)];
            ai += h * x[2 * (n - j) + 1];
        }
        y[2 * m] = ar;
        y[2 * m + 1] = ai;
    }
    return nout;
}

size_t orc_ddc_chain(const uint8_t *packed, size_t ns, uint32_t freg,
                     int mix_enable, int nstages, const int *D,
                     const int *ntaps, const float *const *taps, const int *interp,
                     float *out_iq, size_t out_capacity)
{
    if (nstages < 0 || nstages > 8)
        return (size_t)-1;
    float *xf = (float *)malloc(sizeof(float) * 2 * (ns ? ns : 1));
    double *cur = (double *)malloc(sizeof(double) * 2 * (ns ? ns : 1));
    if (!xf || !cur) {
        free(xf);
        free(cur);
        return (size_t)-1;
    }
    orc_unpack24_f32(packed, ns * 6, xf);
    if (mix_enable) {
        orc_nco_mix_f64(xf, ns, 0, freg, cur);
    } else {
        for (size_t i = 0; i < 2 * ns; i++)
            cur[i] = (double)xf[i];
    }
    free(xf);
    size_t n = ns;
    for (int s = 0; s < nstages; s++) {
        const int L = (interp && interp[s] > 1) ? interp[s] : 1;
        size_t nout = (n * (size_t)L + (size_t)D[s] - 1) / (size_t)D[s];
        double *nxt = (double *)malloc(sizeof(double) * 2 * (nout ? nout : 1));
        if (!nxt) {
            free(cur);
            return (size_t)-1;
        }
        if (L > 1)
            orc_resample_f64(cur, n, taps[s], ntaps[s], L, D[s], nxt);
        else
            orc_fir_decim_f64(cur, n, taps[s], ntaps[s], D[s], nxt);
        free(cur);
        cur = nxt;
        n = nout;
    }
    if (n > out_capacity) {
        free(cur);
        return (size_t)-1;
    }
    for (size_t i = 0; i < 2 * n; i++)
        out_iq[i] = (float)cur[i];
    free(cur);
    return n;
}

/* ------------------------------------------------------------------------
 * EVERY output of one batch of a cascade against the double oracle (authored; what the full-size GPU tests and
 * bench.py's `verified` use).  The stream is the packed buffer of ns_buf samples repeated for ever from absolute ADC
 * sample 0 -- the full-size tests and the bench feed the same batch again and again -- and zero before sample 0
 * (the oracle's zero initial history: every stage's samples with a negative index are zero).  Compared are the
 * last stage's outputs M with first_in <= M*Dtot < first_in + n_in: the ones the batch that starts at absolute
 * sample first_in produces, got_iq[0] being the first of them.  The definition is orc_ddc_chain's (unpack to float as
 * A2, NCO in double with the exact 32-bit phase of the absolute index, y[m] = sum h[k] x[m*D - k] in double); the work
 * is cut into chunks of outputs, each computed from its own input span plus the cascade's halo, OpenMP over chunks.
 * Plain decimators only.  Returns the number of outputs compared, or -1 on bad arguments.
 */
long long orc_chain_check(const uint8_t *packed, size_t ns_buf, uint64_t first_in, size_t n_in, uint32_t freg,
                          int mix_enable, int nstages, const int *D, const int *ntaps, const float *const *taps,
                          const float *got_iq, size_t n_got, double tol, orc_check_stats *st)
{
    if (!packed || !got_iq || !st || ns_buf == 0 || nstages < 1 || nstages > 8)
        return -1;
    long long dtot = 1;
    for (int s = 0; s < nstages; s++) {
        if (D[s] < 1 || ntaps[s] < 1)
            return -1;
        dtot *= D[s];
    }
    const long long m_first = (long long)((first_in + (uint64_t)dtot - 1) / (uint64_t)dtot);
    const long long m_end = (long long)((first_in + n_in + (uint64_t)dtot - 1) / (uint64_t)dtot);
    const long long n_cmp = m_end - m_first;
    memset(st, 0, sizeof *st);
    st->first_bad = -1;
    if (n_cmp <= 0 || (size_t)n_cmp > n_got)
        return n_cmp <= 0 ? 0 : -1;
    long long CH = (1LL << 18) / dtot;            /* about 2^18 ADC samples per chunk */
    if (CH < 64)
        CH = 64;
    const long long nchunks = (n_cmp + CH - 1) / CH;
    /* spans per level for a chunk of CH outputs: level nstages = the outputs, level 0 = ADC samples */
    long long span[9];
    span[nstages] = CH;
    for (int s = nstages - 1; s >= 0; s--)
        span[s] = (span[s + 1] - 1) * D[s] + ntaps[s];
    const double two_pi_over_2p32 = 6.283185307179586476925286766559 / 4294967296.0;
    const float full_scale = (float)(INT_MAX - 256);
    double g_err = 0.0, g_ref = 0.0, g_ratio = 0.0;
    long long g_bad = 0, g_first = -1;
    int failed = 0;
#pragma omp parallel
    {
        double *buf[2];
        buf[0] = (double *)malloc(sizeof(double) * 2 * (size_t)span[0]);
        buf[1] = (double *)malloc(sizeof(double) * 2 * (size_t)span[1]);
        double t_err = 0.0, t_ref = 0.0, t_ratio = 0.0;
        long long t_bad = 0, t_first = -1;
        if (!buf[0] || !buf[1]) {
#pragma omp atomic write
            failed = 1;
        } else {
#pragma omp for schedule(dynamic, 1)
            for (long long c = 0; c < nchunks; c++) {
                long long lo[9], hi[9];            /* inclusive index ranges per level */
                lo[nstages] = m_first + c * CH;
                hi[nstages] = lo[nstages] + CH - 1 < m_end - 1 ? lo[nstages] + CH - 1 : m_end - 1;
                for (int s = nstages - 1; s >= 0; s--) {
                    lo[s] = lo[s + 1] * D[s] - (ntaps[s] - 1);
                    hi[s] = hi[s + 1] * D[s];
                }
                /* level 0: unpack (A2) and mix */
                double *x = buf[0];
                const long long n0 = hi[0] - lo[0] + 1;
                for (long long i = 0; i < n0; i++) {
                    const long long n = lo[0] + i;
                    double xr = 0.0, xi = 0.0;
                    if (n >= 0) {
                        const uint8_t *q = packed + 6 * (size_t)((uint64_t)n % (uint64_t)ns_buf);
                        xr = (double)((float)msb_align24(q) / full_scale);
                        xi = (double)((float)msb_align24(q + 3) / full_scale);
                        if (mix_enable) {
                            const uint32_t ph = (uint32_t)((uint64_t)n * (uint64_t)freg);
                            const double a = two_pi_over_2p32 * (double)ph;
                            const double cc = cos(a), ss = -sin(a);
                            const double r = xr * cc - xi * ss, im = xr * ss + xi * cc;
                            xr = r;
                            xi = im;
                        }
                    }
                    x[2 * i] = xr;
                    x[2 * i + 1] = xi;
                }
                /* the stages: y[m] = sum_k h[k] x[m*D - k]; indices below zero hold zeros at every level */
                int cur = 0;
                for (int s = 0; s < nstages; s++) {
                    const double *in = buf[cur];
                    double *out = buf[cur ^ 1];
                    const float *h = taps[s];
                    const int nt = ntaps[s], d = D[s];
                    for (long long m = lo[s + 1]; m <= hi[s + 1]; m++) {
                        double ar = 0.0, ai = 0.0;
                        if (m >= 0) {
                            const double *top = in + 2 * (m * d - lo[s]);
                            for (int k = 0; k < nt; k++) {
                                ar += (double)h[k] * top[-2 * k];
                                ai += (double)h[k] * top[-2 * k + 1];
                            }
                        }
                        out[2 * (m - lo[s + 1])] = ar;
                        out[2 * (m - lo[s + 1]) + 1] = ai;
                    }
                    cur ^= 1;
                }
                const double *ref = buf[cur];
                const long long nn = hi[nstages] - lo[nstages] + 1;
                const float *g = got_iq + 2 * (lo[nstages] - m_first);
                double c_err = 0.0, c_ref = 0.0;
                for (long long i = 0; i < 2 * nn; i++) {
                    const double r = fabs(ref[i]);
                    if (r > c_ref)
                        c_ref = r;
                }
                for (long long i = 0; i < 2 * nn; i++) {
                    const double e = fabs((double)g[i] - ref[i]);
                    if (!(e <= tol * c_ref)) {               /* also NaN */
                        t_bad++;
                        const long long idx = lo[nstages] - m_first + i / 2;
                        if (t_first < 0 || idx < t_first)
                            t_first = idx;
                    }
                    if (e > c_err || e != e)
                        c_err = e != e ? INFINITY : e;
                }
                if (c_err > t_err)
                    t_err = c_err;
                if (c_ref > t_ref)
                    t_ref = c_ref;
                const double ratio = c_ref > 0.0 ? c_err / c_ref : c_err;
                if (ratio > t_ratio)
                    t_ratio = ratio;
            }
        }
#pragma omp critical
        {
            if (t_err > g_err)
                g_err = t_err;
            if (t_ref > g_ref)
                g_ref = t_ref;
            if (t_ratio > g_ratio)
                g_ratio = t_ratio;
            g_bad += t_bad;
            if (t_first >= 0 && (g_first < 0 || t_first < g_first))
                g_first = t_first;
        }
        free(buf[0]);
        free(buf[1]);
    }
    if (failed)
        return -1;
    st->max_err = g_err;
    st->max_ref = g_ref;
    st->worst_chunk_ratio = g_ratio;
    st->n_compared = n_cmp;
    st->n_bad = g_bad;
    st->first_bad = g_first;
    st->chunk_outputs = CH;
    return n_cmp;
}

/* ------------------------------------------------------------------------
 * CPU baseline: unpack + one decimating FIR, float accumulate.  Chunks of
 * outputs are independent (the input is read with its halo), so OpenMP
 * splits the output range.  Each thread unpacks its input span into a
 * private planar float buffer first (so the FIR inner loop vectorises).
 */
/* four packed samples -> planar floats, the bytes placed as msb_align24 places them (A1), by two-source byte shuffles
 * (GCC vector extensions: pshufb / vpermt2b on x86); reads 32 bytes from p */
typedef uint8_t orc_v16u8 __attribute__((vector_size(16)));
typedef int32_t orc_v4i32 __attribute__((vector_size(16)));
typedef float orc_v4f32 __attribute__((vector_size(16)));
static inline void unpack4_planar(const uint8_t *p, float *xi, float *xq, float full_scale)
{
    orc_v16u8 v0, v1;
    memcpy(&v0, p, 16);
    memcpy(&v1, p + 8, 16);
    /* I of samples 0..3 at byte offsets 0, 6, 12, 18; Q at 3, 9, 15, 21 (offset b >= 16 is byte b - 8 of v1: index 16 + b - 8);
     * byte 0 of every lane is masked to zero afterwards */
    const orc_v16u8 mi = { 0, 0, 1, 2, 0, 6, 7, 8, 0, 12, 13, 14, 0, 26, 27, 28 };
    const orc_v16u8 mq = { 0, 3, 4, 5, 0, 9, 10, 11, 0, 15, 24, 25, 0, 29, 30, 31 };
    const orc_v4i32 keep = { (int32_t)0xffffff00, (int32_t)0xffffff00, (int32_t)0xffffff00, (int32_t)0xffffff00 };
    const orc_v4i32 vi = (orc_v4i32)__builtin_shuffle(v0, v1, mi) & keep;
    const orc_v4i32 vq = (orc_v4i32)__builtin_shuffle(v0, v1, mq) & keep;
    const orc_v4f32 fs = { full_scale, full_scale, full_scale, full_scale };
    const orc_v4f32 fi = __builtin_convertvector(vi, orc_v4f32) / fs;
    const orc_v4f32 fq = __builtin_convertvector(vq, orc_v4f32) / fs;
    memcpy(xi, &fi, 16);
    memcpy(xq, &fq, 16);
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

size_t orc_stage1_f32(const uint8_t *packed, size_t ns, const float *taps,
                      int ntaps, int D, float *out_iq, int threads)
{
    if (D <= 0 || ntaps <= 0)
        return 0;
    const size_t nout = (ns + (size_t)D - 1) / (size_t)D;
    const size_t CH = 4096;              /* outputs per work item */
    const size_t nchunks = (nout + CH - 1) / CH;
#ifdef _OPENMP
    if (threads > 0)
        omp_set_num_threads(threads);
#else
    (void)threads;
#endif
    /* reversed taps so the inner loop walks the input forwards */
    float *hr = (float *)malloc(sizeof(float) * (size_t)ntaps);
    for (int k = 0; k < ntaps; k++)
        hr[k] = taps[ntaps - 1 - k];
#pragma omp parallel
    {
        const size_t span = CH * (size_t)D + (size_t)ntaps;
        float *xi = (float *)malloc(sizeof(float) * span);
        float *xq = (float *)malloc(sizeof(float) * span);
        const float full_scale = (float)(INT_MAX - 256);
#pragma omp for schedule(dynamic, 1)
        for (long long c = 0; c < (long long)nchunks; c++) {
            const size_t m0 = (size_t)c * CH;
            const size_t m1 = m0 + CH < nout ? m0 + CH : nout;
            /* inputs needed: [m0*D-(ntaps-1), (m1-1)*D] */
            const long long first = (long long)(m0 * D) - (ntaps - 1);
            const long long last = (long long)((m1 - 1) * D);
            long long i = first;
            for (; i <= last && i < 0; i++) {
                xi[i - first] = 0.0f;
                xq[i - first] = 0.0f;
            }
            /* four samples (24 bytes) a step through byte shuffles; the last samples of the buffer one by one (the
             * 16-byte loads must stay inside it) */
            const long long vec_end = (long long)ns - 6 < last + 1 ? (long long)ns - 6 : last + 1;
            for (; i + 4 <= vec_end; i += 4)
                unpack4_planar(packed + 6 * (size_t)i, xi + (i - first), xq + (i - first), full_scale);
            for (; i <= last; i++) {
                size_t j = (size_t)(i - first);
                if ((size_t)i >= ns) {
                    xi[j] = 0.0f;
                    xq[j] = 0.0f;
                } else {
                    const uint8_t *p = packed + 6 * (size_t)i;
                    xi[j] = (float)msb_align24(p) / full_scale;
                    xq[j] = (float)msb_align24(p + 3) / full_scale;
                }
            }
            for (size_t m = m0; m < m1; m++) {
                const float *pi = xi + (m - m0) * (size_t)D;
                const float *pq = xq + (m - m0) * (size_t)D;
                float ar = 0.0f, ai = 0.0f;
#pragma omp simd reduction(+ : ar, ai)
                for (int k = 0; k < ntaps; k++) {
                    ar += hr[k] * pi[k];
                    ai += hr[k] * pq[k];
                }
                out_iq[2 * m] = ar;
                out_iq[2 * m + 1] = ai;
            }
        }
        free(xi);
        free(xq);
    }
    free(hr);
    return nout;
}

/* ------------------------------------------------------------------------
 * CPU baseline, the way the reference runs: ONE thread, the stream delivered in callbacks of
 * buf_bytes (6144: perseus-in.c:206-207 -> examples/perseustest.c:466-502), each callback unpacks its buffer as
 * user_data_callback_c_f does (one sample at a time, float = int / (INT_MAX - 256)), mixes it (authored: a double phasor
 * stepped per sample, re-seeded from the exact 32-bit phase at every callback) and pushes it through the streaming FIR
 * chain (authored: each stage keeps its last ntaps - 1 samples and its decimation phase), float accumulation, planar
 * lines so that the tap loop vectorises.  Returns the number of outputs written to out_iq (capacity out_cap complex
 * samples), (size_t)-1 on bad arguments.  Same definition as orc_ddc_chain, so the result agrees with it to float
 * rounding (tests/test_oracle.py).
 */
typedef struct {
    float *li, *lq;      /* the line: ntaps - 1 history samples, then the callback's new ones */
    float *hr;           /* taps reversed */
    int    nt, d;
    long long next;      /* index, relative to the first new sample, of the sample the next output's window ends on */
} orc_stream_stage;

size_t orc_stream_f32_callback_style(const uint8_t *packed, size_t nbytes, size_t buf_bytes, uint32_t freg, int mix_enable,
                                     int nstages, const int *D, const int *ntaps, const float *const *taps, float *out_iq,
                                     size_t out_cap)
{
    if (nstages < 1 || nstages > 8 || buf_bytes < 6)
        return (size_t)-1;
    const size_t bs = buf_bytes / 6;                     /* samples a callback */
    orc_stream_stage st[8];
    memset(st, 0, sizeof st);
    size_t cap = bs;
    int bad = 0;
    for (int s = 0; s < nstages; s++) {
        st[s].nt = ntaps[s];
        st[s].d = D[s];
        st[s].li = (float *)calloc((size_t)ntaps[s] + cap, sizeof(float));
        st[s].lq = (float *)calloc((size_t)ntaps[s] + cap, sizeof(float));
        st[s].hr = (float *)malloc(sizeof(float) * (size_t)ntaps[s]);
        if (!st[s].li || !st[s].lq || !st[s].hr) {
            bad = 1;
            break;
        }
        for (int k = 0; k < ntaps[s]; k++)
            st[s].hr[k] = taps[s][ntaps[s] - 1 - k];
        cap = cap / (size_t)D[s] + 1;
    }
    float *yi = (float *)malloc(sizeof(float) * (bs + 1)), *yq = (float *)malloc(sizeof(float) * (bs + 1));
    size_t n_out = 0;
    if (bad || !yi || !yq)
        goto done;
    {
        const float full_scale = (float)(INT_MAX - 256);
        const double two_pi_over_2p32 = 6.283185307179586476925286766559 / 4294967296.0;
        const double astep = two_pi_over_2p32 * (double)freg;
        const double sc = cos(astep), ss = -sin(astep);                    /* exp(-j astep) */
        uint64_t n_abs = 0;
        for (size_t off = 0; off + 6 <= nbytes; off += buf_bytes) {
            const size_t nb = (nbytes - off < buf_bytes ? nbytes - off : buf_bytes) / 6;
            /* the callback: unpack (A2) [+ mix] into stage 0's line */
            float *xi = st[0].li + (st[0].nt - 1), *xq = st[0].lq + (st[0].nt - 1);
            const uint8_t *p = packed + off;
            if (mix_enable) {
                const double a0 = two_pi_over_2p32 * (double)(uint32_t)(n_abs * (uint64_t)freg);
                double c = cos(a0), sn = -sin(a0);                       /* (a double phasor: 1024 float steps drift to 2e-5) */
                for (size_t k = 0; k < nb; k++) {
                    const float xr = (float)msb_align24(p + 6 * k) / full_scale, xim = (float)msb_align24(p + 6 * k + 3) / full_scale;
                    xi[k] = (float)(xr * c - xim * sn);
                    xq[k] = (float)(xr * sn + xim * c);
                    const double c2 = c * sc - sn * ss;
                    sn = c * ss + sn * sc;
                    c = c2;
                }
            } else {
                for (size_t k = 0; k < nb; k++) {
                    xi[k] = (float)msb_align24(p + 6 * k) / full_scale;
                    xq[k] = (float)msb_align24(p + 6 * k + 3) / full_scale;
                }
            }
            n_abs += nb;
            size_t n_in = nb;
            for (int s = 0; s < nstages; s++) {
                orc_stream_stage *g = &st[s];
                const int nt = g->nt;
                size_t m = 0;
                long long pos = g->next;
                for (; pos < (long long)n_in; pos += g->d, m++) {
                    const float *wi = g->li + pos, *wq = g->lq + pos;      /* window [pos - (nt-1), pos] of the new samples */
                    float ar = 0.0f, ai = 0.0f;
#pragma omp simd reduction(+ : ar, ai)
                    for (int k = 0; k < nt; k++) {
                        ar += g->hr[k] * wi[k];
                        ai += g->hr[k] * wq[k];
                    }
                    yi[m] = ar;
                    yq[m] = ai;
                }
                g->next = pos - (long long)n_in;
                memmove(g->li, g->li + n_in, sizeof(float) * (size_t)(nt - 1));
                memmove(g->lq, g->lq + n_in, sizeof(float) * (size_t)(nt - 1));
                if (s + 1 < nstages) {
                    memcpy(st[s + 1].li + (st[s + 1].nt - 1), yi, sizeof(float) * m);
                    memcpy(st[s + 1].lq + (st[s + 1].nt - 1), yq, sizeof(float) * m);
                } else {
                    for (size_t k = 0; k < m; k++) {
                        if (n_out >= out_cap) {
                            n_out = (size_t)-1;
                            goto done;
                        }
                        out_iq[2 * n_out] = yi[k];
                        out_iq[2 * n_out + 1] = yq[k];
                        n_out++;
                    }
                }
                n_in = m;
            }
        }
    }
done:
    for (int s = 0; s < nstages; s++) {
        free(st[s].li);
        free(st[s].lq);
        free(st[s].hr);
    }
    free(yi);
    free(yq);
    return bad ? (size_t)-1 : n_out;
}
