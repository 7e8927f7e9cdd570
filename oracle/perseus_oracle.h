/*
 * perseus_oracle.h -- CPU restatement of the libperseus-sdr I/Q ingest +
 * decimation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - 24-bit unpack (orc_unpack24_f32 / orc_unpack24_i32): restates
 *     examples/perseustest.c:411-502 of the reference.  PINNED against the
 *     reference itself run on this machine: oracle/_ref/perseustest_ref is the
 *     reference's example client compiled from /root/reference/examples as it
 *     lies (oracle/Makefile, target `ref`) against this repository's drop-in
 *     header and library -- the reference's libusb dependency is exactly what
 *     that library replaces, so no stand-in header or library is involved.
 *     tests/test_reference_client.py runs it: the bytes its callbacks write
 *     (float and int32) equal this oracle's for the LCG stream and for the
 *     exhaustive 2^24 vector (SHA-256 7e5c094b...2884, also the value SURVEY.md
 *     8c recorded).  The library core itself (perseus-sdr.c etc.) still cannot
 *     be built here (needs <libusb-1.0/libusb.h>), which is fine: it holds no
 *     sample arithmetic.
 *   - NCO tuning word / nearest-rate / preselector id: restate the one-line
 *     formulas at perseus-sdr.c:584, :776-811, :589-615; pinned by the KATs
 *     in SURVEY.md 4.
 *   - NCO mix and FIR decimation: PARITY UNPINNED.  The reference holds no
 *     software model of them (they live in opaque FPGA bitstreams), so these
 *     functions are authored definitions: 32-bit phase accumulator
 *     phase(n) = (n*freg) mod 2^32, LO = exp(-j*2*pi*phase/2^32), and
 *     y[m] = sum_k h[k] * x[m*D - k] with zero history, all in double.
 */
#ifndef PERSEUS_ORACLE_H
#define PERSEUS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- synthetic input (LCG bytes) ---------------- */
/* s = s*1664525 + 1013904223 (mod 2^32); byte = s >> 24; returns new state.
 * The first byte produced comes from the state AFTER one step from `seed`. */
uint32_t orc_lcg_fill(uint8_t *dst, size_t nbytes, uint32_t seed);

/* ---- A1/A2/A3: 24-bit packed I/Q unpack (examples/perseustest.c) -------- */
/* nbytes/6 samples are converted; trailing bytes ignored (perseustest.c:477) */
void orc_unpack24_f32(const uint8_t *in, size_t nbytes, float *out_iq);
void orc_unpack24_i32(const uint8_t *in, size_t nbytes, int32_t *out_iq);

/* ---- N1 (authored): float -> 24-bit packed, the inverse of A2 --------------
 * code = clamp(rint(x * 8388607), -8388608, 8388607), round half to even;
 * pack(unpack(c)) == c for every 24-bit code c (tested exhaustively).         */
void orc_pack24_f32(const float *in_iq, size_t nsamples, uint8_t *out);

/* ---- A6: NCO tuning word (perseus-sdr.c:584) --------------------------- */
uint32_t orc_nco_freg(double center_freq_hz, double adc_clk_hz);
/* preselector filter id chosen by perseus-sdr.c:589-615 (10 = wide band) */
int orc_presel_id(double center_freq_hz, int enable_presel);

/* ---- A7: nearest sampling-rate selection (perseus-sdr.c:776-811) ------- */
/* table must be sorted ascending; returns index or -1 */
int orc_rate_index(int sps, const int *table, int n);

/* ---- A8 (authored): NCO mix and polyphase FIR decimation --------------- */
/* out[n] = x[n] * exp(-j*2*pi*((n0+n)*freg mod 2^32)/2^32), complex double */
void orc_nco_mix_f64(const float *x_iq, size_t nsamples, uint64_t n0,
                     uint32_t freg, double *out_iq);

/* The NCO as a phase accumulator retuned while it runs (authored): acc(0)=0, acc(n+1)=acc(n)+
 * freg(n), freg(n) = word[i] for seg_start[i] <= n < seg_start[i+1]; seg_start[0] == 0, nseg <= 64.
 * Phase-continuous at every retune, like a hardware NCO.  n0 = absolute index of x_iq[0].     */
void orc_nco_mix_retuned_f64(const float *x_iq, size_t nsamples, uint64_t n0, const uint64_t *seg_start,
                             const uint32_t *word, int nseg, double *out_iq);

/* y[m] = sum_{k<ntaps} h[k]*x[m*D-k], x[i<0]=0, m = 0..ceil(n/D)-1.
 * returns number of outputs written. */
size_t orc_fir_decim_f64(const double *x_iq, size_t nsamples,
                         const float *taps, int ntaps, int D, double *y_iq);

/* rational L/M resampler (authored): upsample by L (zero
 * stuffing), filter with h, keep every M-th sample:
 *   y[m] = sum_j h[j*L + (m*M mod L)] * x[floor(m*M/L) - j],  m = 0..ceil(n*L/M)-1 */
size_t orc_resample_f64(const double *x_iq, size_t nsamples, const float *taps, int ntaps,
                        int L, int M, double *y_iq);

/* whole chain: unpack -> (mix if mix_enable) -> nstages FIR decimators.
 * interp may be NULL (all plain decimators); interp[s] > 1 makes stage s a
 * rational interp[s]/D[s] resampler.
 * All intermediates double; result cast to float.  Returns outputs written
 * (capacity in complex samples), or (size_t)-1 on bad arguments. */
size_t orc_ddc_chain(const uint8_t *packed, size_t nsamples,
                     uint32_t freg, int mix_enable,
                     int nstages, const int *D, const int *ntaps,
                     const float *const *taps, const int *interp,
                     float *out_iq, size_t out_capacity);

/* ---- every output of one batch against the double oracle (authored) ----
 * The stream = `packed` (ns_buf samples) repeated for ever from absolute sample 0, zeros before it; compared are the
 * last stage's outputs M with first_in <= M*Dtot < first_in + n_in (what the batch starting at first_in produces),
 * got_iq[0] the first of them.  Same definition as orc_ddc_chain, computed in chunks (OpenMP).  An output is BAD when
 * |got - ref| > tol * (max |ref| of its chunk of `chunk_outputs` outputs) or NaN.  Plain decimators only.
 * Returns the number of outputs compared (-1: bad arguments / n_got too small).                                  */
typedef struct {
    double    max_err;            /* max |got - ref| over all outputs (I and Q separately)          */
    double    max_ref;            /* max |ref|                                                      */
    double    worst_chunk_ratio;  /* max over chunks of (chunk max err / chunk max |ref|)           */
    long long n_compared, n_bad;
    long long first_bad;          /* batch-relative index of the first bad output, -1: none          */
    long long chunk_outputs;
} orc_check_stats;
long long orc_chain_check(const uint8_t *packed, size_t ns_buf, uint64_t first_in, size_t n_in, uint32_t freg,
                          int mix_enable, int nstages, const int *D, const int *ntaps, const float *const *taps,
                          const float *got_iq, size_t n_got, double tol, orc_check_stats *st);

/* ---- CPU baseline fast path (float accumulate, OpenMP over chunks) ------
 * unpack + single-stage decimate-by-D, the work bench.py times beside the
 * GPU kernel.  Same definition as the chain above with one stage and no mix,
 * but accumulated in float.  threads<=0 -> omp default. Returns outputs. */
size_t orc_stage1_f32(const uint8_t *packed, size_t nsamples,
                      const float *taps, int ntaps, int D,
                      float *out_iq, int threads);
/* reference-style single-thread callback loop: 6144-byte buffers through
 * orc_unpack24_f32 (perseustest.c:466-502 without the fwrite). */
void orc_unpack24_f32_callback_style(const uint8_t *in, size_t nbytes,
                                     float *out_iq, size_t buf_bytes);
int orc_max_threads(void);
/* the whole chain the way the reference would run it on a CPU: ONE thread, callbacks of buf_bytes (6144), each unpacked
 * as user_data_callback_c_f does, [mixed,] pushed through streaming float FIR stages (plain decimators).  Outputs
 * written (capacity out_cap complex samples) or (size_t)-1. */
size_t orc_stream_f32_callback_style(const uint8_t *packed, size_t nbytes, size_t buf_bytes, uint32_t freg, int mix_enable,
                                     int nstages, const int *D, const int *ntaps, const float *const *taps, float *out_iq,
                                     size_t out_cap);

#ifdef __cplusplus
}
#endif
#endif
