/*
 * perseus_api.c -- the perseus_* callback API (include/perseus-sdr.h) on top of
 * a sample source and the GPU DDC pipeline (include/perseus_ddc.h).
 *
 * What is mirrored from the reference (semantics, not text):
 *   - the state machine and its precondition ladder: NULL descr -> not open ->
 *     firmware -> FPGA configured -> already started   (perseus-sdr.c:563-573,
 *     :647-660), with the same error codes and errorset()/errornone() protocol
 *   - perseus_init() returns the receiver COUNT and starts ONE library thread
 *     that delivers every callback, serialized and in order (perseus-sdr.c:166-188,
 *     :736-774; perseus-in.c:187-264)
 *   - a ring of 8 transfer buffers in one contiguous allocation; the callback
 *     buffer is library-owned and valid only during the call (perseus-in.c:63-96)
 *   - the transfer dispatcher (perseus-in.c:187-264): a completed transfer reaches the
 *     client only if it is the expected slot AND full length; short and out-of-sequence
 *     ones are dropped and logged at level 0, a timeout is tolerated (level 1) and the
 *     transfer goes round again, error / stall / no-device / overflow kill that one
 *     transfer and the queue completes when all 8 are dead; the expected slot moves to
 *     idx+1 after every live completion.  perseus_stop_async_input() returns only when no
 *     further callback can occur and prints the rate line (perseus-sdr.c:709-722)
 *   - up to 8 receivers, each with its own queue (perseus-sdr.c:43-47, perseus-in.h:87):
 *     in DDC mode receiver i runs on GPU i % ngpu and the delivery thread first SUBMITS a
 *     batch for every receiver, then collects them, so the GPUs work at the same time
 *   - NCO word, preselector choice, attenuator encoding, nearest-rate selection
 *     (perseus-sdr.c:584, :589-615, :496-517, :776-811)
 * What is replaced: libusb/FX2/FPGA-bitstream plumbing -> a sample source
 * (LCG / zero / raw file) and, in DDC mode, the GPU pipeline doing the FPGA's
 * NCO mix + decimating FIR chain.  Known quirks of the reference that are NOT
 * reproduced are listed in DESIGN.md.
 */
#define _GNU_SOURCE
#include "../../include/perseus-amd-ext.h"
#include "../../include/perseus_ddc.h"

#include <errno.h>
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <string.h>
#include <sys/time.h>
#include <unistd.h>

/* ---- the three exported globals (reference perseuserr.c:31-33) -------------- */
int  perseus_dbg_level = 0;
char perseus_error_str[1024] = "";
int  perseus_error = 0;

char *perseus_errorstr(void)
{
    static char none[] = "no error";
    return perseus_error == 0 ? none : perseus_error_str;
}

#define MAX_DESCR   8           /* reference perseus-sdr.c:43 */
#define QUEUE_SIZE  8           /* reference perseus-sdr.c:683 */
#define FLT_WB      10
#define FLT_UNDEF   255
#define SIO_FIFOEN   0x01
#define SIO_DITHER   0x02
#define SIO_GAINHIGH 0x04

/* rates for which the reference ships an FPGA image (generate_fpga_code.sh) */
static const int k_rates[] = { 48000, 95000, 96000, 125000, 192000, 250000,
                               500000, 1000000, 1600000, 2000000 };
#define N_RATES ((int)(sizeof(k_rates) / sizeof(k_rates[0])))

typedef struct {
    int   nstages;
    int   interp[4];            /* L of a rational L/decim stage, 1 otherwise */
    int   decim[4];
    int   ntaps[4];
    float *taps[4];
} ddc_plan;

/* transfer status, the cases of the reference's dispatcher (perseus-in.c:199-257) */
enum { XFER_COMPLETED = 0, XFER_TIMED_OUT, XFER_ERROR, XFER_STALL, XFER_NO_DEVICE, XFER_OVERFLOW };

/* fault injection: what the virtual USB side does wrong, and when */
enum { FAULT_NONE = 0, FAULT_SHORT, FAULT_TIMEOUT, FAULT_OOS, FAULT_ERROR, FAULT_STALL, FAULT_NODEV, FAULT_OVERFLOW,
       FAULT_EOF };
typedef struct {
    int kind;
    uint64_t at;                /* transfer number (1 = first), or 0             */
    uint64_t every;             /* every k-th transfer, or 0                     */
} fault_rule;
#define MAX_FAULTS 32
#define MAX_RETUNES 256

#include "out_segments.h"      /* out_seg, out_segs: where the output lies between the GPU and the callbacks */

struct perseus_descr_ds {
    int index;
    int present;                /* enumerated by perseus_init                   */
    int is_open;
    int is_preserie;
    int firmware_downloaded;
    int fpga_configured;
    eeprom_prodid product_id;
    uint8_t frontendctl;        /* atten_id<<4 | presel_id                       */
    uint8_t presel_flt_id;
    uint8_t sio_ctl;
    atomic_uint freg;           /* written by client threads, read by the delivery thread */
    double adc_clk_freq;
    int sample_rate;            /* selected table entry, 0 = none                */
    perseus_amd_config cfg;
    char file_path[1024];
    /* streaming state */
    /* flags shared between client threads and the delivery thread are atomics;
     * everything else below is touched either before `streaming` is published
     * or under pump_lock (ThreadSanitizer-clean, unlike the reference's
     * volatile flags)                                              */
    atomic_int streaming;       /* transfer queue exists                         */
    atomic_int cancelling;
    atomic_int source_done;
    perseus_input_callback cb;
    void *cb_extra;
    uint32_t buffersize;
    uint8_t *ring;              /* QUEUE_SIZE * buffersize                       */
    int next_slot;              /* slot the virtual USB side fills next          */
    int idx_expected;           /* slot the dispatcher expects (perseus-in.c:55,260) */
    int slot_dead[QUEUE_SIZE];  /* transfer killed by a fatal status, never resubmitted */
    int n_dead;
    uint64_t seq;               /* transfers completed by the source             */
    atomic_ullong delivered, dropped, timeouts;
    fault_rule faults[MAX_FAULTS];
    int n_faults;
    char fault_script[256];
    unsigned long bytes_received;
    struct timeval t_start, t_stop;
    uint32_t lcg_state;
    FILE *fp;
    /* DDC mode */
    ddc_plan plan;
    pddc_pipeline *pipe;
    /* two pinned input batch buffers: while the GPU works on one batch (H2D, kernels, D2H
     * on three streams) the source fills the next (pddc_pipeline_push_host_async)       */
    uint8_t *batch_in[2];       /* batch_samples * 6                             */
    float *batch_out[2];        /* ring mode only (zc == 0): pipeline output of one batch */
    size_t out_cap;             /* a batch's output at most, in complex samples  */
    int cur;                    /* buffer pair the next batch goes into          */
    out_segs os;                /* the batches' outputs, oldest first; the last os.n_pend still on the GPU (out_segments.h) */
    int input_done;             /* the source has nothing more to give           */
    int gpu_source;             /* the LCG stream is generated on the device     */
    int batch_auto;             /* batch_samples was not chosen by the client (PERSEUS_AMD_BATCH, set_config, set_batch): the
                                   library picks it for the kind of source when the stream starts (effective_batch)     */
    uint32_t batch_eff;         /* the batch size of THIS stream (set by start; cfg.batch_samples stays the client's / the
                                   default value, so the next stream of this descriptor decides anew)                   */
    uint64_t ganged_batches;    /* batches that shared their launches with other receivers of the GPU */
    uint64_t n_in_place, n_gathered;   /* transfers delivered from the output buffer itself / copied into their slot */
    int gpu_dev;                /* HIP device of the current / last stream, -1: none */
    /* The decimated output never moves on the host: `fifo` is ONE pinned buffer, every batch reserves its place in it
     * when it is submitted (the last kernel or the D2H copy writes there), and a callback gets a pointer INTO it
     * wherever its buffer's worth of bytes lies in one piece; only a buffer that straddles two batches (or the wrap) is
     * gathered into the transfer's ring slot first.  (Until round 4 every byte was copied twice, batch buffer -> byte
     * ring -> ring slot: 5-6 GB/s of payload on the one delivery thread, which bounded the unpaced stream.) */
    /* That needs batches that are large against the transfers (every batch at least two buffers' worth of output: then
     * eight segments always hold a whole buffer).  A stream of small batches -- a low output rate, or a client that asks
     * for 8192-sample batches -- keeps the byte ring of rounds 1-3 (zc == 0): batch buffer -> ring -> slot; there the
     * copies are a few hundred bytes a batch. */
    int zc;
    uint8_t *fifo;              /* zc: the pinned output buffer (os.cap bytes, os.ready of them deliverable); else the byte ring */
    size_t fifo_len, fifo_cap;  /* ring mode: bytes in the ring; its size        */
    size_t fifo_rd;             /* ring mode: read position                      */
    uint64_t adc_samples;       /* ADC-rate samples handed to the GPU so far     */
    uint64_t batches;
    uint32_t freg_applied;      /* word the last submitted batch was mixed with  */
    uint64_t retune_at[MAX_RETUNES];   /* first ADC sample of every tuning-word segment */
    uint32_t retune_word[MAX_RETUNES];
    int n_retunes;
    pthread_mutex_t pump_lock;  /* held by the delivery thread while it works on this descriptor */
};

static perseus_descr g_list[MAX_DESCR];
static int g_entries = 0;
static pthread_t g_thread;
static int g_thread_on = 0;
static atomic_int g_thread_stop;
static atomic_int g_peak_inflight;     /* most receivers that had a GPU batch in flight at the same time */

/* ------------------------------------------------------------------------- */
static double now_s(void)
{
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return tv.tv_sec + 1e-6 * tv.tv_usec;
}

static int rate_index(int sps)
{
    /* nearest entry, midpoint to the LOWER rate, above the top -> last
     * (semantics of reference perseus-sdr.c:776-811) */
    int prev = 0;
    for (int i = 0; i < N_RATES; i++) {
        if (sps > k_rates[i]) {
            if (i < N_RATES - 1) {
                prev = k_rates[i];
                continue;
            }
            return i;
        }
        int mid = (k_rates[i] + prev) / 2;
        if (sps <= mid)
            return i == 0 ? 0 : i - 1;
        return i;
    }
    return -1;
}

/* ---- Kaiser-window low-pass designer for the default DDC plans -------------
 * (authored: the reference has no tap values, they live in the FPGA images)   */
static double bessel_i0(double x)
{
    double s = 1.0, t = 1.0;
    for (int k = 1; k < 64; k++) {
        t *= (x / (2.0 * k)) * (x / (2.0 * k));
        s += t;
        if (t < 1e-18 * s)
            break;
    }
    return s;
}

static void kaiser_lowpass(float *h, int n, double fc /* cycles/sample */, double atten_db)
{
    const double beta = atten_db > 50 ? 0.1102 * (atten_db - 8.7)
                        : atten_db > 21 ? 0.5842 * pow(atten_db - 21, 0.4) + 0.07886 * (atten_db - 21) : 0.0;
    const double m = (n - 1) / 2.0, i0b = bessel_i0(beta);
    double sum = 0.0;
    double *t = (double *)malloc(sizeof(double) * (size_t)n);
    for (int k = 0; k < n; k++) {
        const double x = k - m;
        const double sinc = fabs(x) < 1e-12 ? 2.0 * fc : sin(2.0 * M_PI * fc * x) / (M_PI * x);
        const double r = x / (m > 0 ? m : 1.0);
        const double w = bessel_i0(beta * sqrt(r * r < 1.0 ? 1.0 - r * r : 0.0)) / i0b;
        t[k] = sinc * w;
        sum += t[k];
    }
    for (int k = 0; k < n; k++)
        h[k] = (float)(t[k] / sum);
    free(t);
}

static void plan_free(ddc_plan *p)
{
    for (int i = 0; i < 4; i++) {
        free(p->taps[i]);
        p->taps[i] = NULL;
    }
    p->nstages = 0;
}

/* the plain decimator stages' taps: shortest equiripple designs for the specification below, made offline
 * (tools/design_plans.py -> plan_taps.inc); the window designer above remains for the rational stages and as the fallback */
#include "plan_taps.inc"

static const float *plan_table_taps(int rate, int stage, int decim, int *ntaps)
{
    for (size_t k = 0; k < sizeof(kPlanTapTable) / sizeof(kPlanTapTable[0]); k++)
        if (kPlanTapTable[k].rate == rate && kPlanTapTable[k].stage == stage && kPlanTapTable[k].decim == decim) {
            *ntaps = kPlanTapTable[k].ntaps;
            return kPlanTapTable[k].taps;
        }
    return NULL;
}

/* Plans from the 80 MS/s ADC rate to the ten rates of the reference's FPGA
 * images.  Integer ratios are decimator cascades; the four non-integer rates
 * end in a rational L/M polyphase resampler fed at 250 kS/s. */
static int plan_build(ddc_plan *p, int rate)
{
    static const struct { int rate, n, d[4], l[4]; } tab[] = {
        /* 2 MS/s as 10 * 4, not 8 * 5: the tuned decimate-by-10 first stage runs on the matrix cores like the decimate-by-8
         * one (k_fir_i8x<.., D = 10>), writes 0.8 instead of 1 byte per ADC sample between the stages and leaves the second
         * stage a fifth fewer samples -- 2^28 samples, buffers placed: 0.389 -> 0.369 ms (690 -> 727 GS/s) */
        { 2000000, 2, { 10, 4, 0, 0 },  { 1, 1, 0, 0 } },  { 1600000, 2, { 10, 5, 0, 0 },  { 1, 1, 0, 0 } },
        { 1000000, 2, { 10, 8, 0, 0 },  { 1, 1, 0, 0 } },  /* (likewise: 10 * 8 instead of 8 * 10) */
        { 500000, 3, { 8, 8, 5, 0 },    { 1, 1, 2, 0 } },   /* (the fused pair to 1.25 MS/s, then x2/5: 8 * 4 * 5 cannot fuse) */
        { 250000, 3, { 8, 8, 5, 0 },    { 1, 1, 1, 0 } },  { 125000, 3, { 8, 8, 10, 0 },   { 1, 1, 1, 0 } },
        /* the non-integer ratios, all from 250 kS/s -- behind the SAME two decimate-by-8 stages as the 250 / 125 kS/s plans,
         * which run as one fused kernel (6.125 B per ADC sample instead of 8.3 for 8 * 10 * 5 or 8 * 5 * 5: round 3
         * 480-600 GS/s at 2^28 samples, now 690-815) -- x96/125, x48/125, x24/125 and x19/50 */
        { 192000, 4, { 8, 8, 5, 125 },  { 1, 1, 1, 96 } }, { 96000, 4, { 8, 8, 5, 125 },   { 1, 1, 1, 48 } },
        { 48000, 4, { 8, 8, 5, 125 },   { 1, 1, 1, 24 } }, { 95000, 4, { 8, 8, 5, 50 },    { 1, 1, 1, 19 } },
    };
    plan_free(p);
    for (size_t t = 0; t < sizeof(tab) / sizeof(tab[0]); t++) {
        if (tab[t].rate != rate)
            continue;
        double fs = PERSEUS_ADC_CLK_FREQ;
        const double fpass = 0.4 * rate;
        for (int i = 0; i < tab[t].n; i++) {
            const int D = tab[t].d[i], L = tab[t].l[i];
            const double fs_out = fs * L / D;
            const double fproto = fs * L;              /* rate the prototype filter runs at */
            const int last = (i == tab[t].n - 1);
            const double atten = L > 1 ? 80.0 : 90.0;
            /* protect +-fpass of the FINAL band: stop-band starts where aliases
             * would fold onto it; the last stage may alias into its own transition band */
            double fstop = last ? 0.6 * rate : fs_out - fpass;
            double dw = 2.0 * M_PI * (fstop - fpass) / fproto;
            int n = (int)ceil((atten - 8.0) / (2.285 * dw)) + 1;
            if (n < 8)
                n = 8;
            if (L > 1)
                n = (n + L - 1) / L * L;            /* whole polyphase branches */
            if (i == 0 && D == 8 && L == 1) {       /* fused kernel: whole tap blocks of 8 */
                n = (n + 7) / 8 * 8;
                if (n > PDDC_FAST_MAX_TAPS)
                    n = PDDC_FAST_MAX_TAPS;
            }
            if (n > (L > 1 ? PDDC_MAX_TAPS : PDDC_MAX_TAPS_DECIM))
                n = (L > 1 ? PDDC_MAX_TAPS : PDDC_MAX_TAPS_DECIM) / L * L;
            /* the same specification met by the shortest equiripple filter (offline design): e.g. 250 kS/s 32 / 41 / 117
             * taps where the window method needs 48 / 56 / 144 */
            int nt = 0;
            const float *tt = L == 1 ? plan_table_taps(rate, i, D, &nt) : NULL;
            if (tt && nt <= PDDC_MAX_TAPS_DECIM)
                n = nt;
            p->decim[i] = D;
            p->interp[i] = L;
            p->ntaps[i] = n;
            p->taps[i] = (float *)malloc(sizeof(float) * (size_t)n);
            if (!p->taps[i]) {
                plan_free(p);
                return 0;
            }
            if (tt && nt == n)
                memcpy(p->taps[i], tt, sizeof(float) * (size_t)n);
            else
                kaiser_lowpass(p->taps[i], n, 0.5 * (fpass + fstop) / fproto, atten);
            if (L > 1)                               /* zero stuffing divides the gain by L */
                for (int k = 0; k < n; k++)
                    p->taps[i][k] *= (float)L;
            fs = fs_out;
        }
        p->nstages = tab[t].n;
        return p->nstages;
    }
    return 0;
}

/* ---- source ----------------------------------------------------------------- */
static size_t source_fill(perseus_descr *d, uint8_t *dst, size_t nbytes)
{
    switch (d->cfg.source) {
    case PERSEUS_AMD_SRC_ZERO:
        memset(dst, 0, nbytes);
        return nbytes;
    case PERSEUS_AMD_SRC_FILE: {
        if (!d->fp)
            return 0;
        size_t got = fread(dst, 1, nbytes, d->fp);
        return got;
    }
    default: {
        uint32_t s = d->lcg_state;
        for (size_t i = 0; i < nbytes; i++) {
            s = s * 1664525u + 1013904223u;
            dst[i] = (uint8_t)(s >> 24);
        }
        d->lcg_state = s;
        return nbytes;
    }
    }
}

static void pace_until(perseus_descr *d, double samples_done, double rate)
{
    if (!d->cfg.pace || rate <= 0)
        return;
    const double due = (d->t_start.tv_sec + 1e-6 * d->t_start.tv_usec) + samples_done / rate;
    for (;;) {
        double dt = due - now_s();
        if (dt <= 0 || d->cancelling || g_thread_stop)
            return;
        if (dt > 0.01)
            dt = 0.01;
        usleep((useconds_t)(dt * 1e6));
    }
}

/* ---- fault injection script ------------------------------------------------------
 * "kind@n" = at transfer number n (1 = first), "kind%k" = every k-th transfer, comma separated;
 * kinds: short timeout oos error stall nodev overflow eof.  Example: "short%7,timeout@9,error@20" */
static int parse_faults(perseus_descr *d, const char *script)
{
    static const struct { const char *name; int kind; } names[] = {
        { "short", FAULT_SHORT }, { "timeout", FAULT_TIMEOUT }, { "oos", FAULT_OOS }, { "error", FAULT_ERROR },
        { "stall", FAULT_STALL }, { "nodev", FAULT_NODEV }, { "overflow", FAULT_OVERFLOW }, { "eof", FAULT_EOF },
    };
    d->n_faults = 0;
    if (d->cfg.drop_every > 0) {
        d->faults[d->n_faults++] = (fault_rule){ FAULT_SHORT, 0, (uint64_t)d->cfg.drop_every };
    }
    if (!script)
        return 0;
    const char *p = script;
    while (*p) {
        while (*p == ',' || *p == ' ')
            p++;
        if (!*p)
            break;
        int kind = FAULT_NONE;
        size_t len = 0;
        for (size_t i = 0; i < sizeof(names) / sizeof(names[0]); i++) {
            len = strlen(names[i].name);
            if (strncmp(p, names[i].name, len) == 0 && (p[len] == '@' || p[len] == '%')) {
                kind = names[i].kind;
                break;
            }
        }
        if (kind == FAULT_NONE || d->n_faults >= MAX_FAULTS)
            return -1;
        const char mode = p[len];
        char *endp = NULL;
        const unsigned long long v = strtoull(p + len + 1, &endp, 10);
        if (endp == p + len + 1 || v == 0)
            return -1;
        d->faults[d->n_faults++] = (fault_rule){ kind, mode == '@' ? v : 0, mode == '%' ? v : 0 };
        p = endp;
    }
    return 0;
}

static int fault_for(const perseus_descr *d, uint64_t n)     /* n = 1-based transfer number */
{
    for (int i = 0; i < d->n_faults; i++)                     /* a rule for exactly this transfer wins ... */
        if (d->faults[i].at && d->faults[i].at == n)
            return d->faults[i].kind;
    for (int i = 0; i < d->n_faults; i++)                     /* ... over the periodic ones */
        if (d->faults[i].every && n % d->faults[i].every == 0)
            return d->faults[i].kind;
    return FAULT_NONE;
}

/* ---- the dispatcher: what the reference does with one completed transfer
 * (perseus-in.c:187-264), status by status -------------------------------------- */
static void dispatch(perseus_descr *d, int idx, int status, uint32_t actual, const uint8_t *data)
{
    if (d->cancelling)
        return;
    switch (status) {
    case XFER_COMPLETED:
        d->bytes_received += actual;
        if (idx == d->idx_expected) {
            if (actual == d->buffersize) {
                perseus_input_callback cb = d->cb;
                if (cb) {
                    /* `data`: the transfer's payload -- its ring slot, or (DDC modes) the bytes where the GPU put them.
                     * Like the reference's buffers it is the library's and valid until the callback returns
                     * (perseus-in.c:206-207 resubmits the transfer right after it). */
                    cb((void *)(data ? data : d->ring + (size_t)idx * d->buffersize), (int)d->buffersize, d->cb_extra);
                    d->delivered++;
                }
            } else {
                d->dropped++;
                dbgprintf(0, "transfer %d shorter than requested (%u of %u bytes): dropped", idx, actual,
                          d->buffersize);
            }
        } else {
            d->dropped++;
            dbgprintf(0, "transfer %d out of sequence (slot %d was expected): dropped", idx, d->idx_expected);
        }
        break;
    case XFER_TIMED_OUT:
        d->timeouts++;
        dbgprintf(1, "transfer %d timed out (%u bytes so far): resubmitted", idx, actual);
        break;
    default: {
        static const char *what[] = { "", "", "failed", "stalled", "lost its device", "overflowed" };
        dbgprintf(0, "transfer %d %s: not resubmitted", idx, status <= XFER_OVERFLOW ? what[status] : "failed");
        if (!d->slot_dead[idx]) {
            d->slot_dead[idx] = 1;
            d->n_dead++;
        }
        if (d->n_dead == QUEUE_SIZE) {                /* perseus_input_queue_check_completion */
            dbgprintf(0, "all %d transfers are dead: the input queue has completed", QUEUE_SIZE);
            d->source_done = 1;
        }
        return;                                       /* the expected slot does not move */
    }
    }
    d->idx_expected = (idx + 1) % QUEUE_SIZE;
}

static int next_live_slot(perseus_descr *d)
{
    for (int k = 0; k < QUEUE_SIZE; k++) {
        const int sidx = (d->next_slot + k) % QUEUE_SIZE;
        if (!d->slot_dead[sidx]) {
            d->next_slot = (sidx + 1) % QUEUE_SIZE;
            return sidx;
        }
    }
    return -1;
}

/* One turn of the virtual USB side: the next live transfer completes with whatever the fault
 * script says.  `fill` brings buffersize bytes of payload into a slot (wire source or the
 * decimated FIFO) and returns the bytes it had; `avail` says how many whole buffers `fill`
 * can still provide without blocking (an out-of-sequence pair needs two).
 * Returns 0 when nothing could be done (no payload, queue dead).                          */
typedef const uint8_t *(*fill_fn)(perseus_descr *d, uint8_t *slot, size_t *got);

static int turn(perseus_descr *d, fill_fn fill, size_t avail)
{
    if (d->n_dead == QUEUE_SIZE) {
        d->source_done = 1;
        return 0;
    }
    const int fault = fault_for(d, d->seq + 1);
    if (fault == FAULT_EOF) {
        d->source_done = 1;
        d->input_done = 1;
        return 0;
    }
    if (fault == FAULT_TIMEOUT || (fault >= FAULT_ERROR && fault <= FAULT_OVERFLOW)) {
        /* nothing arrived: no payload is consumed */
        const int sidx = next_live_slot(d);
        d->seq++;
        dispatch(d, sidx, fault == FAULT_TIMEOUT ? XFER_TIMED_OUT : XFER_ERROR + (fault - FAULT_ERROR), 0, NULL);
        return 1;
    }
    if (avail == 0)
        return 0;
    if (fault == FAULT_OOS && QUEUE_SIZE - d->n_dead >= 2) {
        /* two neighbouring transfers complete in the wrong order (with a single live transfer left
         * there is no neighbour: the rule is ignored) */
        if (avail < 2)
            return 0;                                 /* wait until two buffers of payload exist */
        const int a = next_live_slot(d), b = next_live_slot(d);
        size_t ga = 0, gb = 0;
        const uint8_t *pa = fill(d, d->ring + (size_t)a * d->buffersize, &ga);
        const uint8_t *pb = fill(d, d->ring + (size_t)b * d->buffersize, &gb);
        d->seq += 2;
        dispatch(d, b, XFER_COMPLETED, (uint32_t)gb, pb);
        dispatch(d, a, XFER_COMPLETED, (uint32_t)ga, pa);
        return 1;
    }
    const int sidx = next_live_slot(d);
    size_t got = 0;
    const uint8_t *data = fill(d, d->ring + (size_t)sidx * d->buffersize, &got);
    d->seq++;
    if (got == 0) {                                   /* the source had nothing at all */
        d->source_done = 1;
        return 0;
    }
    if (fault == FAULT_SHORT && got == d->buffersize)
        got = d->buffersize / 2;                      /* the payload was consumed, half of it "arrived" */
    dispatch(d, sidx, XFER_COMPLETED, (uint32_t)got, data);
    return 1;
}

/* ---- wire mode: the source plays the receiver ----------------------------------------- */
static const uint8_t *fill_wire(perseus_descr *d, uint8_t *slot, size_t *got)
{
    *got = source_fill(d, slot, d->buffersize);
    if (*got < d->buffersize)
        d->input_done = 1;                            /* bounded source (file) ended */
    return slot;
}

static int pump_wire(perseus_descr *d)
{
    if (d->input_done) {
        d->source_done = 1;
        return 0;
    }
    pace_until(d, (double)(d->seq + 1) * (d->buffersize / 6), (double)d->sample_rate);
    return turn(d, fill_wire, 2);
}

/* ---- DDC mode: ADC-rate batches -> GPU -> FIFO of decimated bytes -> transfers ---------- */
/* ring mode (zc == 0): a byte ring between the batch buffers and the transfers */
static size_t ring_room(const perseus_descr *d) { return d->fifo_cap - d->fifo_len; }

static void ring_put(perseus_descr *d, const uint8_t *src, size_t n)
{
    size_t wr = (d->fifo_rd + d->fifo_len) % d->fifo_cap;
    const size_t first = n < d->fifo_cap - wr ? n : d->fifo_cap - wr;
    memcpy(d->fifo + wr, src, first);
    memcpy(d->fifo, src + first, n - first);
    d->fifo_len += n;
}

static const uint8_t *fill_ring(perseus_descr *d, uint8_t *slot, size_t *got)
{
    const size_t n = d->buffersize;
    *got = 0;
    if (d->fifo_len < n)
        return slot;
    const size_t first = n < d->fifo_cap - d->fifo_rd ? n : d->fifo_cap - d->fifo_rd;
    memcpy(slot, d->fifo + d->fifo_rd, first);
    memcpy(slot + first, d->fifo, n - first);
    d->fifo_rd = (d->fifo_rd + n) % d->fifo_cap;
    d->fifo_len -= n;
    d->n_gathered++;
    *got = n;
    return slot;
}

/* the next buffersize deliverable bytes: where they lie, or gathered into `slot` when they are in two pieces */
static size_t ready_bytes(const perseus_descr *d) { return d->zc ? d->os.ready : d->fifo_len; }

static const uint8_t *fill_fifo(perseus_descr *d, uint8_t *slot, size_t *got)
{
    if (!d->zc)
        return fill_ring(d, slot, got);
    *got = 0;
    if (d->os.ready < d->buffersize)
        return slot;
    int in_place = 0;
    const uint8_t *p = oseg_take(&d->os, d->fifo, d->buffersize, slot, &in_place);
    if (in_place)
        d->n_in_place++;
    else
        d->n_gathered++;
    *got = d->buffersize;
    return p;
}

static size_t out_bytes_per_sample(const perseus_descr *d)
{
    return d->cfg.mode == PERSEUS_AMD_MODE_DDC_WIRE ? 6 : 8;
}

/* the receiver's own part of a submission: the next batch buffer filled from the source, the pace kept, the tuning
 * word latched.  Returns the samples of the batch (0: nothing to submit) */
static size_t batch_prepare(perseus_descr *d)
{
    const int k = d->cur;
    size_t ns = d->batch_eff;
    if (!d->gpu_source) {
        size_t got = source_fill(d, d->batch_in[k], ns * 6);
        got -= got % 48;                         /* whole groups of 8 samples */
        if (got < ns * 6)
            d->input_done = 1;                   /* bounded source (file) ended */
        ns = got / 6;
        if (ns == 0)
            return 0;
    }
    pace_until(d, (double)(d->adc_samples + ns), d->adc_clk_freq);
    /* A retune takes effect at the batch boundary (reference clients retune while streaming,
     * examples/fifo.c:43-49): sample-accurate there and phase-continuous, and logged so that
     * a test can build the same piecewise stream (perseus_amd_get_retune_log).              */
    const uint32_t word = d->freg;
    if (d->batches == 0 || word != d->freg_applied) {
        if (d->n_retunes < MAX_RETUNES) {
            d->retune_at[d->n_retunes] = d->adc_samples;
            d->retune_word[d->n_retunes] = word;
            d->n_retunes++;
        }
        d->freg_applied = word;
        dbgprintf(3, "NCO word %u from ADC sample %llu", word, (unsigned long long)d->adc_samples);
    }
    pddc_pipeline_set_freg(d->pipe, word);
    return ns;
}

/* where the next batch's output goes: its place in the output buffer, or (ring mode) the batch buffer of the pair;
 * `off` is what batch_pushed() records for it */
static void *out_place(perseus_descr *d, size_t *off)
{
    if (!d->zc) {
        *off = (size_t)d->cur;
        return d->batch_out[d->cur];
    }
    if (!oseg_reserve(&d->os, d->out_cap * out_bytes_per_sample(d), off))
        return NULL;
    return d->fifo + *off;
}

static void batch_pushed(perseus_descr *d, int rc, size_t ns, size_t n_out, int ticket, size_t off)
{
    if (rc != PDDC_OK) {
        dbgprintf(0, "GPU pipeline failed (%d: %s); stream stopped", rc, pddc_last_error());
        d->input_done = 1;
        return;
    }
    d->adc_samples += ns;
    d->batches++;
    oseg_push(&d->os, ticket, off, n_out * out_bytes_per_sample(d));
    d->cur ^= 1;
}

/* fill the next batch buffer from the source and hand it to the GPU; returns at once */
static void submit_batch(perseus_descr *d)
{
    const int k = d->cur;
    const size_t ns = batch_prepare(d);
    if (ns == 0)
        return;
    size_t n_out = 0, off = 0;
    int ticket = -1;
    int rc;
    void *dst = out_place(d, &off);
    if (!dst)
        return;                                  /* (can_submit said there was room) */
    if (d->gpu_source)
        rc = pddc_pipeline_push_synth_async(d->pipe, d->cfg.lcg_seed, d->adc_samples * 6, ns, dst, d->out_cap, &n_out,
                                            &ticket);
    else
        rc = pddc_pipeline_push_host_async(d->pipe, d->batch_in[k], ns, dst, d->out_cap, &n_out, &ticket);
    batch_pushed(d, rc, ns, n_out, ticket, off);
}

/* pass 1: keep the GPU fed.  A free-running source has two batches in flight (source, PCIe and
 * kernels overlap); a paced, real-time one is not read ahead, that would only add latency.   */
static int can_submit(const perseus_descr *d)
{
    const int depth = d->cfg.pace ? 1 : 2;
    const size_t worst = d->out_cap * out_bytes_per_sample(d);
    return d->os.n_pend < depth && !d->input_done && !d->source_done &&
           /* (no further ahead of the callbacks than the ring was: at most three batches' output waiting, so a retune still
            * takes effect within a few batches of an unpaced stream) */
           (d->zc ? d->os.n - d->os.n_pend <= 2 && oseg_reserve(&d->os, worst, NULL)
                  : ring_room(d) >= worst * (size_t)(d->os.n_pend + 1)) &&
           !(d->cfg.max_buffers && d->seq >= d->cfg.max_buffers);
}

static int ddc_submit(perseus_descr *d)
{
    int did = 0;
    while (can_submit(d)) {
        submit_batch(d);
        did = 1;
        if (d->os.n_pend == 0)
            break;
    }
    return did;
}

/* pass 2: take the oldest batch's output into the FIFO (waits for it: every receiver's batch was
 * submitted in pass 1, so the GPUs are all busy while this thread waits for the first of them) */
static int ddc_collect(perseus_descr *d)
{
    if (d->os.n_pend == 0)
        return 0;
    const out_seg b = oseg_oldest_pending(&d->os);
    if (pddc_pipeline_wait_ticket(d->pipe, b.ticket) != PDDC_OK) {
        d->os.n_pend--;
        dbgprintf(0, "GPU pipeline failed (%s); stream stopped", pddc_last_error());
        d->input_done = 1;
        d->source_done = 1;
        return 0;
    }
    if (d->zc) {
        oseg_ready(&d->os);                      /* its bytes are where they will be delivered from */
    } else {
        ring_put(d, (const uint8_t *)d->batch_out[b.off], b.len);
        oseg_pop_pending(&d->os);                /* (ring mode lists only what is still on the GPU) */
    }
    return 1;
}

/* pass 3: frame the FIFO into transfers and dispatch them */
static int ddc_deliver(perseus_descr *d, int budget)
{
    int did = 0;
    while (budget-- > 0 && !d->source_done && !d->cancelling) {
        if (d->cfg.max_buffers && d->seq >= d->cfg.max_buffers) {
            d->source_done = 1;
            break;
        }
        if (!turn(d, fill_fifo, ready_bytes(d) / d->buffersize))
            break;
        did = 1;
    }
    if (d->input_done && d->os.n_pend == 0 && ready_bytes(d) < d->buffersize)
        d->source_done = 1;
    return did;
}

/* ---- submit helpers ---------------------------------------------------------------------------
 * Pass 0 of the delivery thread hands every streaming receiver's next batch to its GPU: a dozen HIP calls per
 * receiver (generator or H2D copy, kernels, D2H copy, events), ~65 us of host time each -- with eight receivers on
 * one thread that serial half millisecond per round, not the GPUs, set the pace (eight receivers took 11x the time of
 * one).  The calls of different receivers touch different pipelines and streams, so they are farmed out: the
 * delivery thread posts the round's receivers as jobs, the helpers and the delivery thread itself take them one by
 * one, and the round goes on when all are done.  Callbacks are NOT farmed out: passes 1 and 2 (collect, deliver)
 * stay on the one library thread, serialized and in order, as in the reference (perseus-sdr.c:749-770).        */
#define MAX_HELPERS (MAX_DESCR - 1)
static pthread_t g_helper[MAX_HELPERS];
static int g_nhelpers = 0;
static pthread_mutex_t g_job_lock = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_job_cv = PTHREAD_COND_INITIALIZER, g_done_cv = PTHREAD_COND_INITIALIZER;
static int g_job_list[MAX_DESCR], g_job_n = 0, g_job_next = 0, g_job_done = 0, g_job_busy = 0, g_job_inflight = 0;
static int g_helpers_stop = 0;

static void submit_one(int i, int *busy, int *inflight)
{
    perseus_descr *d = &g_list[i];
    pthread_mutex_lock(&d->pump_lock);
    if (d->streaming && !d->cancelling && !d->source_done && d->cfg.mode != PERSEUS_AMD_MODE_WIRE) {
        *busy |= ddc_submit(d);
        *inflight += d->os.n_pend > 0;
    }
    pthread_mutex_unlock(&d->pump_lock);
}

/* take jobs of the posted round until none is left; returns with g_job_lock held */
static void run_jobs_locked(void)
{
    while (g_job_next < g_job_n) {
        const int i = g_job_list[g_job_next++];
        pthread_mutex_unlock(&g_job_lock);
        int busy = 0, inflight = 0;
        submit_one(i, &busy, &inflight);
        pthread_mutex_lock(&g_job_lock);
        g_job_busy |= busy;
        g_job_inflight += inflight;
        if (++g_job_done == g_job_n)
            pthread_cond_broadcast(&g_done_cv);
    }
}

static void *helper_fn(void *arg)
{
    (void)arg;
    pthread_mutex_lock(&g_job_lock);
    while (!g_helpers_stop) {
        if (g_job_next < g_job_n)
            run_jobs_locked();
        else
            pthread_cond_wait(&g_job_cv, &g_job_lock);
    }
    pthread_mutex_unlock(&g_job_lock);
    return NULL;
}

/* ---- gang submission ------------------------------------------------------------------------------
 * Receivers that share a GPU and stream from the on-device source hand their batches to the GPU TOGETHER
 * (pddc_gang_push_async): one generator launch, one first-stage launch, one decimator launch for all of them, the
 * receiver being the grid's second dimension, and one event -- instead of a chain of launches per receiver, each
 * shorter than the launch gap in front of it.  Outputs are bit-identical to the per-receiver path.  The reference's
 * shape is the same: eight descriptors, one poll thread that serves them all (perseus-sdr.c:43-47, 736-774).
 * PERSEUS_AMD_GANG=0 turns it off (every receiver a chain of its own, farmed out to the submit helpers).        */
#define MAX_GANG_GPUS 16
static pddc_gang *g_gang[MAX_GANG_GPUS];
static int g_gang_off = 0;

/* (only free-running sources: a gang round holds every member's lock while it prepares the batches, and a PACED source
 * sleeps there up to a batch period -- its receiver keeps the helper path, which sleeps under its own lock alone and
 * delivers at its own due time) */
static int gang_candidate(const perseus_descr *d)
{
    return d->gpu_source && d->gpu_dev >= 0 && d->gpu_dev < MAX_GANG_GPUS && d->cfg.mode == PERSEUS_AMD_MODE_DDC &&
           !d->cfg.pace;
}

/* the receivers sub[0..m) (ascending, all on GPU `dev`): up to `depth` rounds of one batch each */
static void gang_submit(int dev, const int *sub, int m, int *busy, int *inflight)
{
    if (!g_gang[dev] && pddc_gang_create(&g_gang[dev], dev) != PDDC_OK) {
        dbgprintf(0, "no gang on GPU %d (%s): receivers submit one by one", dev, pddc_last_error());
        g_gang_off = 1;
        return;
    }
    for (int k = 0; k < m; k++)
        pthread_mutex_lock(&g_list[sub[k]].pump_lock);         /* ascending order everywhere: no deadlock */
    for (int round = 0; round < 2; round++) {
        perseus_descr *mem[MAX_DESCR];
        int nm = 0;
        size_t ns = 0;
        for (int k = 0; k < m; k++) {
            perseus_descr *d = &g_list[sub[k]];
            if (!d->streaming || d->cancelling || d->source_done || !gang_candidate(d) || !can_submit(d))
                continue;
            if (nm == 0)
                ns = d->batch_eff;
            if (d->batch_eff != ns) {                  /* another batch length: a chain of its own */
                submit_batch(d);
                *busy = 1;
                continue;
            }
            mem[nm++] = d;
        }
        if (nm == 0)
            break;
        *busy = 1;
        /* (a round of one still goes through the gang's stream: changing between the gang's stream and the pipeline's
         * own costs a wait for everything in flight, and the next round is likely to be a full one again) */
        pddc_gang_item it[MAX_DESCR];
        int ni = 0;
        perseus_descr *in[MAX_DESCR];
        size_t off[MAX_DESCR];
        for (int k = 0; k < nm; k++) {
            perseus_descr *d = mem[k];
            if (batch_prepare(d) != ns)
                continue;
            memset(&it[ni], 0, sizeof(it[ni]));
            it[ni].pipe = d->pipe;
            it[ni].seed = d->cfg.lcg_seed;
            it[ni].byte_offset = d->adc_samples * 6;
            it[ni].h_out = out_place(d, &off[ni]);
            if (!it[ni].h_out)
                continue;                                      /* (can_submit said there was room) */
            it[ni].out_capacity = d->out_cap;
            in[ni++] = d;
        }
        if (ni == 0)
            continue;
        int shared = 0;
        const int rc = pddc_gang_push_async(g_gang[dev], it, ni, ns, &shared);
        for (int k = 0; k < ni; k++) {
            batch_pushed(in[k], rc, ns, it[k].n_out, it[k].ticket, off[k]);
            if (rc == PDDC_OK && shared > 1)
                in[k]->ganged_batches++;
        }
    }
    for (int k = m - 1; k >= 0; k--) {
        *inflight += g_list[sub[k]].os.n_pend > 0;
        pthread_mutex_unlock(&g_list[sub[k]].pump_lock);
    }
}

/* pass 0 for the receivers in list[0..n): in parallel when there are several and helpers exist */
static void submit_round(const int *list_in, int n_in, int *busy, int *inflight)
{
    int list[MAX_DESCR], n = 0;
    if (!g_gang_off && n_in > 1) {
        /* receivers that share a GPU go together; what is left goes one by one below */
        char taken[MAX_DESCR] = { 0 };
        for (int a = 0; a < n_in; a++) {
            const perseus_descr *da = &g_list[list_in[a]];
            if (taken[a] || !gang_candidate(da))
                continue;
            int sub[MAX_DESCR], m = 0;
            for (int b = a; b < n_in; b++) {
                const perseus_descr *db = &g_list[list_in[b]];
                if (!taken[b] && gang_candidate(db) && db->gpu_dev == da->gpu_dev)
                    sub[m++] = list_in[b];
            }
            if (m < 2)
                continue;
            for (int b = a; b < n_in; b++)
                for (int k = 0; k < m; k++)
                    if (list_in[b] == sub[k])
                        taken[b] = 1;
            gang_submit(da->gpu_dev, sub, m, busy, inflight);
        }
        for (int a = 0; a < n_in; a++)
            if (!taken[a] || g_gang_off)
                list[n++] = list_in[a];
    } else {
        memcpy(list, list_in, sizeof(int) * (size_t)n_in);
        n = n_in;
    }
    if (n <= 1 || g_nhelpers == 0) {
        for (int k = 0; k < n; k++)
            submit_one(list[k], busy, inflight);
        return;
    }
    pthread_mutex_lock(&g_job_lock);
    memcpy(g_job_list, list, sizeof(int) * (size_t)n);
    g_job_n = n;
    g_job_next = 0;
    g_job_done = 0;
    g_job_busy = 0;
    g_job_inflight = 0;
    pthread_cond_broadcast(&g_job_cv);
    run_jobs_locked();                           /* the delivery thread takes its share */
    while (g_job_done < g_job_n)
        pthread_cond_wait(&g_done_cv, &g_job_lock);
    *busy |= g_job_busy;
    *inflight += g_job_inflight;
    g_job_n = g_job_next = 0;
    pthread_mutex_unlock(&g_job_lock);
}

static void helpers_start(int n)
{
    if (n > MAX_HELPERS)
        n = MAX_HELPERS;
    const char *e = getenv("PERSEUS_AMD_SUBMIT_THREADS");        /* 0: everything on the delivery thread */
    if (e)
        n = atoi(e) < n ? atoi(e) : n;
    pthread_mutex_lock(&g_job_lock);
    g_helpers_stop = 0;
    g_job_n = g_job_next = g_job_done = 0;
    pthread_mutex_unlock(&g_job_lock);
    g_nhelpers = 0;
    for (int k = 0; k < n; k++)
        if (pthread_create(&g_helper[g_nhelpers], NULL, helper_fn, NULL) == 0)
            g_nhelpers++;
}

static void helpers_stop(void)
{
    pthread_mutex_lock(&g_job_lock);
    g_helpers_stop = 1;
    pthread_cond_broadcast(&g_job_cv);
    pthread_mutex_unlock(&g_job_lock);
    for (int k = 0; k < g_nhelpers; k++)
        pthread_join(g_helper[k], NULL);
    g_nhelpers = 0;
}

/* PERSEUS_AMD_TIMING=1: where the delivery thread's time goes (submit / wait+collect / callbacks), printed at perseus_exit */
static int g_timing = 0;
static double g_t_pass[3];
static unsigned long long g_rounds;

static void *worker_fn(void *arg)
{
    (void)arg;
    while (!g_thread_stop) {
        int busy = 0;
        int inflight = 0;
        {
            /* pass 0: every streaming DDC receiver's next batch goes to its GPU -- all of them before any is waited for */
            int list[MAX_DESCR], n = 0;
            for (int i = 0; i < g_entries; i++) {
                perseus_descr *d = &g_list[i];
                if (d->streaming && !d->cancelling && !d->source_done && d->cfg.mode != PERSEUS_AMD_MODE_WIRE)
                    list[n++] = i;
            }
            const double t0 = g_timing ? now_s() : 0.0;
            submit_round(list, n, &busy, &inflight);
            if (g_timing && n > 0) {
                g_t_pass[0] += now_s() - t0;
                g_rounds++;
            }
        }
        for (int pass = 1; pass < 3; pass++) {
            const double t0 = g_timing ? now_s() : 0.0;
            /* after pass 0 every receiver's batch has been submitted and none has been waited for yet */
            if (pass == 1 && inflight > g_peak_inflight)
                g_peak_inflight = inflight;
            for (int i = 0; i < g_entries; i++) {
                perseus_descr *d = &g_list[i];
                if (!d->streaming || d->cancelling || d->source_done)
                    continue;
                if (pass < 2 && d->cfg.mode == PERSEUS_AMD_MODE_WIRE)
                    continue;
                pthread_mutex_lock(&d->pump_lock);
                if (d->streaming && !d->cancelling && !d->source_done) {
                    if (d->cfg.mode == PERSEUS_AMD_MODE_WIRE) {
                        if (d->cfg.max_buffers && d->seq >= d->cfg.max_buffers)
                            d->source_done = 1;
                        else
                            busy |= pump_wire(d);
                    } else if (pass == 1) {
                        busy |= ddc_collect(d);
                    } else {
                        busy |= ddc_deliver(d, 64);
                    }
                }
                pthread_mutex_unlock(&d->pump_lock);
            }
            if (g_timing)
                g_t_pass[pass] += now_s() - t0;
        }
        if (!busy)
            usleep(1000);
    }
    return NULL;
}

/* ---- API ---------------------------------------------------------------------- */
void perseus_set_debug(int level) { perseus_dbg_level = level; }

static uint32_t effective_batch(const perseus_descr *d);

static void default_config(perseus_descr *d)
{
    const char *e;
    memset(&d->cfg, 0, sizeof(d->cfg));
    d->cfg.mode = PERSEUS_AMD_MODE_WIRE;
    d->cfg.source = PERSEUS_AMD_SRC_LCG;
    d->cfg.lcg_seed = 12345u + (uint32_t)d->index;
    d->cfg.pace = 1;
    d->cfg.gpu_device = -1;
    d->cfg.batch_samples = 0;                   /* 0: the library picks per stream (effective_batch) */
    d->batch_auto = 1;
    if ((e = getenv("PERSEUS_AMD_MODE"))) {
        if (strcmp(e, "ddc") == 0)
            d->cfg.mode = PERSEUS_AMD_MODE_DDC;
        else if (strcmp(e, "ddc-wire") == 0)
            d->cfg.mode = PERSEUS_AMD_MODE_DDC_WIRE;
    }
    if ((e = getenv("PERSEUS_AMD_SOURCE"))) {
        if (strncmp(e, "lcg", 3) == 0) {
            d->cfg.source = PERSEUS_AMD_SRC_LCG;
            if (e[3] == ':')
                d->cfg.lcg_seed = (uint32_t)strtoul(e + 4, NULL, 0) + (uint32_t)d->index;
        } else if (strcmp(e, "zero") == 0) {
            d->cfg.source = PERSEUS_AMD_SRC_ZERO;
        } else if (strncmp(e, "file:", 5) == 0) {
            d->cfg.source = PERSEUS_AMD_SRC_FILE;
            snprintf(d->file_path, sizeof(d->file_path), "%s", e + 5);
            d->cfg.file_path = d->file_path;
        }
    }
    if ((e = getenv("PERSEUS_AMD_PACE")))
        d->cfg.pace = atoi(e) != 0;
    if ((e = getenv("PERSEUS_AMD_BATCH"))) {     /* a client's choice: 8 .. PERSEUS_AMD_BATCH_MAX, rounded down to a multiple of 8 */
        const long long v = atoll(e) / 8 * 8;
        if (v >= 8 && v <= (long long)PERSEUS_AMD_BATCH_MAX) {
            d->cfg.batch_samples = (uint32_t)v;
            d->batch_auto = 0;
        } else {
            dbgprintf(1, "PERSEUS_AMD_BATCH=%s ignored: not in 8 .. %u", e, PERSEUS_AMD_BATCH_MAX);
        }
    }
    if ((e = getenv("PERSEUS_AMD_DROP")))
        d->cfg.drop_every = atoi(e);
    if ((e = getenv("PERSEUS_AMD_MAX_BUFFERS")))
        d->cfg.max_buffers = strtoull(e, NULL, 0);
    d->cfg.ep_packet_size = 512;
    if ((e = getenv("PERSEUS_AMD_EP_PACKET")))
        d->cfg.ep_packet_size = atoi(e);
    d->cfg.cpu_source = 0;
    if ((e = getenv("PERSEUS_AMD_CPU_SOURCE")))
        d->cfg.cpu_source = atoi(e) != 0;
    d->fault_script[0] = 0;
    d->cfg.fault_script = NULL;
    if ((e = getenv("PERSEUS_AMD_FAULTS"))) {
        snprintf(d->fault_script, sizeof(d->fault_script), "%s", e);
        d->cfg.fault_script = d->fault_script;
    }
}

int perseus_init(void)
{
    dbgprintf(3, "perseus_init()");
    if (g_thread_on)
        perseus_exit();
    memset(g_list, 0, sizeof(g_list));
    int n = 1;
    const char *e = getenv("PERSEUS_AMD_DEVICES");
    if (e)
        n = atoi(e);
    if (n < 0)
        n = 0;
    if (n > MAX_DESCR)
        n = MAX_DESCR;
    for (int i = 0; i < n; i++) {
        g_list[i].index = i;
        g_list[i].present = 1;
        pthread_mutex_init(&g_list[i].pump_lock, NULL);
        default_config(&g_list[i]);
        dbgprintf(2, "Found virtual receiver %d (source %d, mode %d)", i, g_list[i].cfg.source,
                  g_list[i].cfg.mode);
    }
    g_entries = n;
    g_peak_inflight = 0;
    {
        const char *e = getenv("PERSEUS_AMD_GANG");
        g_gang_off = e && atoi(e) == 0;
        e = getenv("PERSEUS_AMD_TIMING");
        g_timing = e && atoi(e) != 0;
        g_t_pass[0] = g_t_pass[1] = g_t_pass[2] = 0.0;
        g_rounds = 0;
    }
    if (g_entries > 0) {
        g_thread_stop = 0;
        helpers_start(g_entries - 1);              /* submit helpers: one per further receiver */
        if (pthread_create(&g_thread, NULL, worker_fn, NULL) != 0) {
            helpers_stop();
            return errorset(PERSEUS_CANTCREAT, "can't create the sample delivery thread");
        }
        g_thread_on = 1;
    }
    return errornone(g_entries);
}

int perseus_exit(void)
{
    dbgprintf(3, "perseus_exit(): thread=%d", g_thread_on);
    for (int i = 0; i < g_entries; i++)
        if (g_list[i].streaming)
            perseus_stop_async_input(&g_list[i]);
    if (g_thread_on) {
        g_thread_stop = 1;
        pthread_join(g_thread, NULL);
        g_thread_on = 0;
        helpers_stop();
    }
    for (int i = 0; i < g_entries; i++) {
        perseus_close(&g_list[i]);
        plan_free(&g_list[i].plan);
    }
    if (g_timing && g_rounds)
        fprintf(stderr, "perseus: delivery thread, %llu rounds: submit %.1f us, wait + collect %.1f us, deliver %.1f us per round\n",
                g_rounds, 1e6 * g_t_pass[0] / g_rounds, 1e6 * g_t_pass[1] / g_rounds, 1e6 * g_t_pass[2] / g_rounds);
    for (int i = 0; i < MAX_GANG_GPUS; i++) {
        pddc_gang_destroy(g_gang[i]);
        g_gang[i] = NULL;
    }
    g_entries = 0;
    g_thread_stop = 0;
    return errornone(0);
}

perseus_descr *perseus_open(int nDev)
{
    dbgprintf(3, "perseus_open(%d)", nDev);
    if (nDev < 0 || nDev >= g_entries) {
        errorset(PERSEUS_INVALIDDEV, "invalid device id %d", nDev);
        return NULL;
    }
    perseus_descr *d = &g_list[nDev];
    if (d->is_open) {
        errorset(PERSEUS_ALREADYOPEN, "device %d already open", nDev);
        return NULL;
    }
    d->is_open = 1;
    d->is_preserie = 0;
    d->firmware_downloaded = 1;      /* a virtual receiver always answers (perseus-sdr.c:288-297) */
    d->fpga_configured = 0;
    d->adc_clk_freq = PERSEUS_ADC_CLK_FREQ;
    d->presel_flt_id = FLT_UNDEF;
    d->frontendctl = 0;
    d->sio_ctl = 0;
    d->freg = 0;
    d->sample_rate = 0;
    return errornone(d);
}

int perseus_close(perseus_descr *d)
{
    dbgprintf(3, "perseus_close(%p)", (void *)d);
    if (d == NULL)
        return errorset(PERSEUS_NULLDESCR, "null descriptor");
    if (!d->is_open)
        return errornone(0);
    if (d->streaming)
        perseus_stop_async_input(d);
    d->is_open = 0;
    return errornone(0);
}

int perseus_firmware_download(perseus_descr *d, char *fname)
{
    dbgprintf(3, "perseus_firmware_download(%p,%s)", (void *)d, fname ? fname : "Null");
    if (d == NULL)
        return errorset(PERSEUS_NULLDESCR, "null descriptor");
    if (!d->is_open)
        return errorset(PERSEUS_DEVNOTOPEN, "device not open");
    if (fname != NULL)
        return errorset(PERSEUS_FNNOTAVAIL, "Firmware download from files not implemented");
    /* "firmware found": fill in the product id the EEPROM would hold */
    d->presel_flt_id = FLT_UNDEF;
    d->product_id.sn = (uint16_t)(1000 + d->index);
    d->product_id.prodcode = PERSEUS_PRODCODE;
    d->product_id.hwrel = 3;
    d->product_id.hwver = 1;
    static const uint8_t sig[6] = { 'M', 'I', '3', '5', '5', 'X' };
    memcpy(d->product_id.signature, sig, 6);
    d->firmware_downloaded = 1;
    return errornone(0);
}

int perseus_get_product_id(perseus_descr *d, eeprom_prodid *prodid)
{
    dbgprintf(3, "perseus_get_product_id(%p,...)", (void *)d);
    if (d == NULL)
        return errorset(PERSEUS_NULLDESCR, "null descriptor");
    if (prodid == NULL)
        return errorset(PERSEUS_NULLDESCR, "null eeprom_prodid pointer");
    if (!d->firmware_downloaded)
        return errorset(PERSEUS_FWNOTLOADED, "firmware not loaded");
    memcpy(prodid, &d->product_id, sizeof(*prodid));
    return errornone(0);
}

#define CHECK_OPEN_FW(d)                                                   \
    do {                                                                   \
        if ((d) == NULL)                                                   \
            return errorset(PERSEUS_NULLDESCR, "null descriptor");         \
        if (!(d)->is_open)                                                 \
            return errorset(PERSEUS_DEVNOTOPEN, "device not open");        \
        if (!(d)->firmware_downloaded)                                     \
            return errorset(PERSEUS_FWNOTLOADED, "firmware not loaded");   \
    } while (0)

#define CHECK_FPGA(d)                                                      \
    do {                                                                   \
        if (!(d)->fpga_configured)                                         \
            return errorset(PERSEUS_FPGANOTCFGD, "FPGA not configured");   \
    } while (0)

int perseus_set_attenuator(perseus_descr *d, uint8_t atten_id)
{
    dbgprintf(3, "perseus_set_attenuator(%p,%d)", (void *)d, atten_id);
    CHECK_OPEN_FW(d);
    if (d->is_preserie)
        atten_id ^= PERSEUS_ATT_30DB;
    d->frontendctl = (uint8_t)((atten_id << 4) | (d->frontendctl & 0x0F));
    return errornone(0);
}

static const int k_att_db[4] = { 0, 10, 20, 30 };

int perseus_set_attenuator_in_db(perseus_descr *d, int db)
{
    dbgprintf(3, "perseus_set_attenuator_in_db(%p,%d)", (void *)d, db);
    CHECK_OPEN_FW(d);
    CHECK_FPGA(d);
    for (int i = 0; i < 4; i++)
        if (k_att_db[i] == db)
            return perseus_set_attenuator(d, (uint8_t)i);
    return errorset(PERSEUS_ATTERROR, "set attenuator error, bad value: %d", db);
}

int perseus_get_attenuator_values(perseus_descr *d, int *buf, unsigned int size)
{
    (void)d;
    if (size == 0)
        return errorset(PERSEUS_ERRPARAM, "Zero length buffer");
    for (unsigned i = 0; i < size; i++)
        buf[i] = -1;
    for (unsigned i = 0; i < 4; i++) {
        if (i >= size)
            return errorset(PERSEUS_BUFFERSIZE, "Insufficient buffer size");
        buf[i] = k_att_db[i];
    }
    return errornone(0);
}

int perseus_set_attenuator_n(perseus_descr *d, int nlo)
{
    dbgprintf(3, "perseus_set_attenuator_n(%p,%d)", (void *)d, nlo);
    CHECK_OPEN_FW(d);
    CHECK_FPGA(d);
    if (nlo < 0 || nlo >= 4)
        return errorset(PERSEUS_ERRPARAM, "Invalid index in vector");
    if (perseus_set_attenuator(d, (uint8_t)nlo) < 0)
        return errorset(PERSEUS_ATTERROR, "set attenuator error");
    return errornone(0);
}

int perseus_set_adc(perseus_descr *d, int dither, int preamp)
{
    dbgprintf(3, "perseus_set_adc(%p,%d,%d)", (void *)d, dither, preamp);
    CHECK_OPEN_FW(d);
    CHECK_FPGA(d);
    d->sio_ctl = (uint8_t)(dither ? d->sio_ctl | SIO_DITHER : d->sio_ctl & ~SIO_DITHER);
    d->sio_ctl = (uint8_t)(preamp ? d->sio_ctl | SIO_GAINHIGH : d->sio_ctl & ~SIO_GAINHIGH);
    return errornone(0);
}

int perseus_set_ddc_center_freq(perseus_descr *d, double hz, int enablePresel)
{
    static const double fc[10] = { 1.7e6, 2.1e6, 3.0e6, 4.2e6, 6.0e6, 8.4e6, 12e6, 17e6, 24e6, 32e6 };
    dbgprintf(3, "perseus_set_ddc_center_freq(%p,%.3f,%d)", (void *)d, hz, enablePresel);
    CHECK_OPEN_FW(d);
    CHECK_FPGA(d);
    if (hz < PERSEUS_DDC_FREQ_MIN || hz > PERSEUS_DDC_FREQ_MAX)
        return errorset(PERSEUS_ERRPARAM, "center_freq not in the range [%d..%d]", PERSEUS_DDC_FREQ_MIN,
                        PERSEUS_DDC_FREQ_MAX);
    /* FREG = Flo/Fclk * 2^32, truncated (reference perseus-sdr.c:584); read by the
     * delivery thread at the next batch boundary */
    d->freg = pddc_nco_freg(hz, d->adc_clk_freq);
    uint8_t flt = FLT_WB;
    if (enablePresel)
        for (int i = 0; i < 10; i++)
            if (hz < fc[i]) {
                flt = (uint8_t)i;
                break;
            }
    if (flt != d->presel_flt_id) {
        d->frontendctl = (uint8_t)((d->frontendctl & 0xF0) | (flt & 0x0F));
        d->presel_flt_id = flt;
    }
    return errornone(0);
}

int perseus_get_sampling_rates(perseus_descr *d, int *buf, unsigned int size)
{
    (void)d;                    /* may be NULL (reference perseustest.c:60) */
    if (size == 0)
        return errorset(PERSEUS_ERRPARAM, "Zero length buffer");
    for (unsigned i = 0; i < size; i++)
        buf[i] = 0;
    for (unsigned i = 0; i < (unsigned)N_RATES; i++) {
        if (i >= size)
            return errorset(PERSEUS_BUFFERSIZE, "Insufficient buffer size");
        buf[i] = k_rates[i];
    }
    return errornone(0);
}

int perseus_set_sampling_rate(perseus_descr *d, int sps)
{
    dbgprintf(3, "perseus_set_sampling_rate(%p,%d)", (void *)d, sps);
    CHECK_OPEN_FW(d);
    const int idx = rate_index(sps);
    if (idx < 0)
        return errorset(PERSEUS_FPGANOTCFGD, "FPGA not configured: sampling rate not found");
    if (d->streaming)
        return errorset(PERSEUS_ASYNCSTARTED, "cannot change the sampling rate while streaming");
    d->sample_rate = k_rates[idx];
    plan_build(&d->plan, d->sample_rate);       /* 0 stages for non-integer ratios */
    d->fpga_configured = 1;
    dbgprintf(3, "rate %d S/s selected (%d-stage GPU plan)", d->sample_rate, d->plan.nstages);
    return errornone(0);
}

int perseus_set_sampling_rate_n(perseus_descr *d, unsigned int n)
{
    dbgprintf(3, "perseus_set_sampling_rate_n(%p,%u)", (void *)d, n);
    CHECK_OPEN_FW(d);
    if (n >= (unsigned)N_RATES)
        return errorset(PERSEUS_ERRPARAM, "Invalid index in vector");
    return perseus_set_sampling_rate(d, k_rates[n]);
}

int perseus_is_preserie(perseus_descr *d, int *flag)
{
    if (d == NULL)
        return errorset(PERSEUS_NULLDESCR, "null descriptor");
    if (!d->is_open)
        return errorset(PERSEUS_DEVNOTOPEN, "device not open");
    if (flag)
        *flag = d->is_preserie;
    if (d->is_preserie)
        return errorset(PERSEUS_SNNOTAVAILABLE, "preserie unit");
    return errornone(0);
}

static void free_stream(perseus_descr *d)
{
    free(d->ring);
    d->ring = NULL;
    if (d->pipe)
        pddc_pipeline_wait(d->pipe);         /* nothing may still be copying into the buffers */
    for (int k = 0; k < 2; k++) {
        pddc_host_free(d->batch_in[k]);
        d->batch_in[k] = NULL;
        pddc_host_free(d->batch_out[k]);
        d->batch_out[k] = NULL;
    }
    if (d->zc)
        pddc_host_free(d->fifo);
    else
        free(d->fifo);
    d->fifo = NULL;
    d->zc = 0;
    d->fifo_len = d->fifo_cap = d->fifo_rd = 0;
    memset(&d->os, 0, sizeof(d->os));
    if (d->pipe) {
        pddc_pipeline_destroy(d->pipe);
        d->pipe = NULL;
    }
    if (d->fp) {
        fclose(d->fp);
        d->fp = NULL;
    }
}

static int start_locked(perseus_descr *d, uint32_t buffersize, perseus_input_callback cb, void *extra);

int perseus_start_async_input(perseus_descr *d, uint32_t buffersize, perseus_input_callback cb, void *extra)
{
    dbgprintf(3, "perseus_start_async_input(%p,%u,...)", (void *)d, buffersize);
    CHECK_OPEN_FW(d);
    CHECK_FPGA(d);
    if (d->streaming)
        return errorset(PERSEUS_ASYNCSTARTED, "async input already started");
    if (buffersize > 16320)
        return errorset(PERSEUS_ERRPARAM, "max bulk buffer size is 16320 bytes");
    /* the endpoint's max packet size decides the granule (reference perseus-sdr.c:664-680); the
     * virtual receiver reports 512 unless configured as a 510-byte (or a broken) endpoint */
    switch (d->cfg.ep_packet_size) {
    case 512:
        if (buffersize == 0 || (buffersize % 6144) != 0)
            return errorset(PERSEUS_BUFFERSIZE,
                            "buffer size should be an integer multiple of 6144 bytes (1024 I/Q samples)");
        break;
    case 510:
        if (buffersize == 0 || (buffersize % 510) != 0)
            return errorset(PERSEUS_BUFFERSIZE,
                            "buffer size should be an integer multiple of 510 bytes (85 IQ samples)");
        break;
    default:
        return errorset(PERSEUS_ERRPARAM, "Unexpected max packet size: %d", d->cfg.ep_packet_size);
    }

    pthread_mutex_lock(&d->pump_lock);      /* the delivery thread reads these fields under the same lock */
    const int rc_start = start_locked(d, buffersize, cb, extra);
    pthread_mutex_unlock(&d->pump_lock);
    return rc_start;
}

static int start_locked(perseus_descr *d, uint32_t buffersize, perseus_input_callback cb, void *extra)
{
    d->ring = (uint8_t *)malloc((size_t)QUEUE_SIZE * buffersize);
    if (!d->ring)
        return errorset(PERSEUS_NOMEM, "can't allocate the transfer buffers");
    d->buffersize = buffersize;
    d->gpu_dev = -1;
    d->gpu_source = 0;
    d->lcg_state = d->cfg.lcg_seed;
    if (parse_faults(d, d->cfg.fault_script) < 0) {
        free_stream(d);
        return errorset(PERSEUS_ERRPARAM, "bad fault injection script \"%s\"", d->cfg.fault_script);
    }
    if (d->cfg.source == PERSEUS_AMD_SRC_FILE) {
        d->fp = fopen(d->cfg.file_path ? d->cfg.file_path : "", "rb");
        if (!d->fp) {
            free_stream(d);
            return errorset(PERSEUS_FILENOTFOUND, "can't open source file %s",
                            d->cfg.file_path ? d->cfg.file_path : "(null)");
        }
    }
    if (d->cfg.mode != PERSEUS_AMD_MODE_WIRE) {
        if (d->plan.nstages == 0) {
            free_stream(d);
            return errorset(PERSEUS_FPGANOTCFGD,
                            "no decimation plan from 80 MS/s to %d S/s",
                            d->sample_rate);
        }
        pddc_stage_desc sd[4];
        for (int i = 0; i < d->plan.nstages; i++) {
            sd[i].decim = d->plan.decim[i];
            sd[i].ntaps = d->plan.ntaps[i];
            sd[i].taps = d->plan.taps[i];
            sd[i].interp = d->plan.interp[i];
        }
        int ndev = pddc_device_count();
        if (ndev <= 0) {
            free_stream(d);
            return errorset(PERSEUS_DEVNOTFOUND, "DDC mode needs a GPU and none is visible (no CPU fallback)%s%s",
                            ndev < 0 ? ": " : "", ndev < 0 ? pddc_last_error() : "");
        }
        int dev = d->cfg.gpu_device >= 0 ? d->cfg.gpu_device : d->index % ndev;
        d->gpu_dev = dev;
        int rc = pddc_pipeline_create(&d->pipe, dev, sd, d->plan.nstages,
                                      PDDC_F_MIX | (d->cfg.mode == PERSEUS_AMD_MODE_DDC_WIRE ? PDDC_F_OUT_PACKED24 : 0));
        if (rc != PDDC_OK) {
            free_stream(d);
            return errorset(PERSEUS_DEVCONF, "GPU pipeline creation failed (%d): %s", rc, pddc_last_error());
        }
        pddc_pipeline_set_freg(d->pipe, d->freg);
        d->batch_eff = effective_batch(d);                     /* (perseus_amd_effective_batch reports it while the stream runs) */
        d->out_cap = pddc_pipeline_max_output(d->pipe, d->batch_eff) + 8;
        /* the synthetic stream is generated on the GPU (bit-identical to the host loop,
         * pddc_synth_lcg) unless the configuration insists on the CPU generator */
        d->gpu_source = d->cfg.source == PERSEUS_AMD_SRC_LCG && !d->cfg.cpu_source;
        int hrc = 0;
        for (int k = 0; k < 2; k++) {
            if (!d->gpu_source)
                hrc |= pddc_host_alloc((void **)&d->batch_in[k], (size_t)d->batch_eff * 6);
        }
        /* every batch at least two buffers' worth of output (out_cap is the most a batch gives, + 8; the least is within
         * two samples of that): the callbacks read the output where the GPU puts it */
        d->zc = d->out_cap > 10 && (d->out_cap - 10) * out_bytes_per_sample(d) >= 2 * (size_t)buffersize;
        if (d->zc) {
            /* room for two batches on the GPU, one being delivered, and the piece at the end that a batch does not fit into */
            d->os.cap = 6 * oseg_align(d->out_cap * 8) + oseg_align(2 * (size_t)buffersize);
            hrc |= pddc_host_alloc((void **)&d->fifo, d->os.cap);
        } else {
            for (int k = 0; k < 2; k++)
                hrc |= pddc_host_alloc((void **)&d->batch_out[k], d->out_cap * 8);
            d->fifo_cap = 3 * d->out_cap * 8 + 2 * (size_t)buffersize;
            d->fifo = (uint8_t *)malloc(d->fifo_cap);
        }
        if (hrc || !d->fifo) {
            free_stream(d);
            return errorset(PERSEUS_NOMEM, "can't allocate the batch buffers");
        }
    }
    d->cb = cb;
    d->cb_extra = extra;
    d->next_slot = 0;
    d->idx_expected = 0;
    memset(d->slot_dead, 0, sizeof(d->slot_dead));
    d->n_dead = 0;
    d->seq = 0;
    d->delivered = d->dropped = d->timeouts = 0;
    d->bytes_received = 0;
    d->adc_samples = 0;
    d->batches = 0;
    d->ganged_batches = 0;
    d->n_in_place = d->n_gathered = 0;
    d->n_retunes = 0;
    d->cur = 0;
    d->os.n_pend = d->os.n = d->os.head = 0;
    d->os.ready = 0;
    d->input_done = 0;
    d->fifo_len = d->fifo_rd = 0;
    d->source_done = 0;
    d->cancelling = 0;
    gettimeofday(&d->t_start, NULL);
    d->sio_ctl |= SIO_FIFOEN;
    d->streaming = 1;
    return errornone(0);
}

int perseus_stop_async_input(perseus_descr *d)
{
    dbgprintf(3, "perseus_stop_async_input(%p)", (void *)d);
    if (d == NULL)
        return errorset(PERSEUS_NULLDESCR, "null descriptor");
    if (!d->streaming)
        return errorset(PERSEUS_ASYNCSTARTED, "async input not started");
    d->cancelling = 1;
    gettimeofday(&d->t_stop, NULL);
    /* the delivery thread holds pump_lock while it fills a buffer or runs this
     * descriptor's callback: once we own it no callback is in flight, and with
     * `cancelling` set none can start (reference perseus-sdr.c:709-716) */
    const int on_worker = g_thread_on && pthread_equal(pthread_self(), g_thread);
    if (!on_worker)
        pthread_mutex_lock(&d->pump_lock);
    d->cb = NULL;
    d->streaming = 0;
    const double elapsed = 1e-6 * (d->t_stop.tv_usec - d->t_start.tv_usec) + (d->t_stop.tv_sec - d->t_start.tv_sec);
    /* same line as the reference (perseus-sdr.c:719-722); a sample is 6 bytes on
     * the wire, 8 bytes in float32 DDC mode */
    const double per_ksample = (d->cfg.mode == PERSEUS_AMD_MODE_DDC ? 8.0 : 6.0) * 1000.0;
    dbgprintf(3, "Elapsed time: %f s - kSamples read: %ld - Rate: %.1f kS/s\n", elapsed,
              (long)(d->bytes_received / per_ksample), elapsed > 0 ? d->bytes_received / elapsed / per_ksample : 0.0);
    free_stream(d);
    d->sio_ctl &= (uint8_t)~SIO_FIFOEN;
    d->cancelling = 0;
    if (!on_worker)
        pthread_mutex_unlock(&d->pump_lock);
    return errornone(0);
}

/* ---- extension API ---------------------------------------------------------- */
int perseus_amd_get_config(perseus_descr *d, perseus_amd_config *cfg)
{
    if (d == NULL || cfg == NULL)
        return errorset(PERSEUS_NULLDESCR, "null descriptor");
    *cfg = d->cfg;
    return errornone(0);
}

int perseus_amd_set_config(perseus_descr *d, const perseus_amd_config *cfg)
{
    if (d == NULL || cfg == NULL)
        return errorset(PERSEUS_NULLDESCR, "null descriptor");
    if (d->streaming)
        return errorset(PERSEUS_ASYNCSTARTED, "cannot reconfigure while streaming");
    if (cfg->mode < PERSEUS_AMD_MODE_WIRE || cfg->mode > PERSEUS_AMD_MODE_DDC_WIRE)
        return errorset(PERSEUS_ERRPARAM, "bad mode %d", cfg->mode);
    if (cfg->source < PERSEUS_AMD_SRC_LCG || cfg->source > PERSEUS_AMD_SRC_FILE)
        return errorset(PERSEUS_ERRPARAM, "bad source %d", cfg->source);
    if (cfg->batch_samples % 8 || cfg->batch_samples > PERSEUS_AMD_BATCH_MAX)
        return errorset(PERSEUS_ERRPARAM, "batch_samples must be 0 (the library picks) or a multiple of 8 up to %u", PERSEUS_AMD_BATCH_MAX);
    if (cfg->ep_packet_size < 0 || cfg->ep_packet_size > 1024)
        return errorset(PERSEUS_ERRPARAM, "bad endpoint packet size %d", cfg->ep_packet_size);
    /* the strings are copied into the descriptor -- unless they already ARE the descriptor's
     * copies (get_config -> modify -> set_config hands them back) */
    const char *fp = cfg->file_path, *fs = cfg->fault_script;
    /* batch_samples: 0 = the library picks per stream, anything else is the client's choice (get_config reports 0 while the
     * choice is the library's, so get_config -> change another field -> set_config leaves it there) */
    d->batch_auto = cfg->batch_samples == 0;
    d->cfg = *cfg;
    if (d->cfg.ep_packet_size == 0)
        d->cfg.ep_packet_size = 512;
    if (fp) {
        if (fp != d->file_path)
            snprintf(d->file_path, sizeof(d->file_path), "%s", fp);
        d->cfg.file_path = d->file_path;
    }
    if (fs) {
        if (fs != d->fault_script)
            snprintf(d->fault_script, sizeof(d->fault_script), "%s", fs);
        d->cfg.fault_script = d->fault_script;
        if (parse_faults(d, d->fault_script) < 0)
            return errorset(PERSEUS_ERRPARAM, "bad fault injection script \"%s\"", d->fault_script);
    }
    return errornone(0);
}

/* GPU batch size of the next stream.  The client's choice if it made one; otherwise 2^22 samples (52 ms of signal: a paced,
 * real-time source must not wait longer for its output; a host-fed one is bound by filling the batch) -- except for a
 * free-running on-device source, where the batch only has to amortise the launch chain's fixed cost: 2^24.  Measured
 * through the C client, 250 kS/s plan, one receiver (bench.py --workload api250k): 2^22 80 GS/s of ADC-rate input, 2^24
 * 255, 2^26 200, 2^28 214 -- beyond 2^24 the delivery thread's two copies of every output byte (GPU batch -> ring ->
 * transfer buffer, 5 GB/s of callback payload) bound the stream, not the GPU.                                          */
static uint32_t effective_batch(const perseus_descr *d)
{
    const int gpu_source = d->cfg.source == PERSEUS_AMD_SRC_LCG && !d->cfg.cpu_source;
    if (!d->batch_auto && d->cfg.batch_samples)
        return d->cfg.batch_samples;
    if (!d->cfg.pace && gpu_source && d->cfg.mode != PERSEUS_AMD_MODE_WIRE)
        return PERSEUS_AMD_BATCH_UNPACED_DEVICE;
    return PERSEUS_AMD_BATCH_DEFAULT;
}

uint32_t perseus_amd_effective_batch(perseus_descr *d) { return d ? (d->streaming ? d->batch_eff : effective_batch(d)) : 0; }

int perseus_amd_set_batch(perseus_descr *d, uint32_t batch_samples)
{
    if (d == NULL)
        return errorset(PERSEUS_NULLDESCR, "null descriptor");
    if (d->streaming)
        return errorset(PERSEUS_ASYNCSTARTED, "cannot reconfigure while streaming");
    if (batch_samples % 8 || batch_samples > PERSEUS_AMD_BATCH_MAX)
        return errorset(PERSEUS_ERRPARAM, "batch_samples must be a multiple of 8 up to %u (0: the library picks)", PERSEUS_AMD_BATCH_MAX);
    if (batch_samples == 0) {
        d->batch_auto = 1;
        d->cfg.batch_samples = 0;
    } else {
        d->batch_auto = 0;
        d->cfg.batch_samples = batch_samples;
    }
    return errornone(0);
}

uint32_t perseus_amd_get_freg(perseus_descr *d) { return d ? d->freg : 0; }
int perseus_amd_get_sampling_rate(perseus_descr *d) { return d ? d->sample_rate : 0; }
int perseus_amd_get_frontendctl(perseus_descr *d) { return d ? d->frontendctl : -1; }
int perseus_amd_get_sioctl(perseus_descr *d) { return d ? d->sio_ctl : -1; }
uint64_t perseus_amd_buffers_delivered(perseus_descr *d) { return d ? d->delivered : 0; }
uint64_t perseus_amd_buffers_dropped(perseus_descr *d) { return d ? d->dropped : 0; }

int perseus_amd_source_running(perseus_descr *d)
{
    return d && d->streaming && !d->source_done;
}

int perseus_amd_get_stats(perseus_descr *d, perseus_amd_stats *st)
{
    if (d == NULL || st == NULL)
        return errorset(PERSEUS_NULLDESCR, "null descriptor");
    /* the delivery thread updates these under the pump lock while streaming (a client callback runs
     * ON that thread with the lock held: no second lock then) */
    const int on_worker = g_thread_on && pthread_equal(pthread_self(), g_thread);
    if (!on_worker)
        pthread_mutex_lock(&d->pump_lock);
    st->delivered = d->delivered;
    st->dropped = d->dropped;
    st->timeouts = d->timeouts;
    st->dead_transfers = (uint64_t)d->n_dead;
    st->transfers = d->seq;
    st->bytes_received = d->bytes_received;
    st->adc_samples = d->adc_samples;
    st->batches = d->batches;
    st->gpu_device = d->gpu_dev;
    st->gpu_source = d->gpu_source;
    st->ganged_batches = d->ganged_batches;
    st->buffers_in_place = d->n_in_place;
    st->buffers_gathered = d->n_gathered;
    st->peak_receivers_in_flight = g_peak_inflight;
    if (!on_worker)
        pthread_mutex_unlock(&d->pump_lock);
    return errornone(0);
}

int perseus_amd_get_retune_log(perseus_descr *d, uint64_t *first_sample, uint32_t *word, int capacity)
{
    if (d == NULL)
        return errorset(PERSEUS_NULLDESCR, "null descriptor");
    /* read under the pump lock: the delivery thread appends while streaming */
    const int on_worker = g_thread_on && pthread_equal(pthread_self(), g_thread);
    if (!on_worker)
        pthread_mutex_lock(&d->pump_lock);
    const int n = d->n_retunes;
    for (int i = 0; i < n && i < capacity; i++) {
        if (first_sample)
            first_sample[i] = d->retune_at[i];
        if (word)
            word[i] = d->retune_word[i];
    }
    if (!on_worker)
        pthread_mutex_unlock(&d->pump_lock);
    return errornone(n);
}

int perseus_amd_get_plan_interp(perseus_descr *d, int interp[4])
{
    if (d == NULL || interp == NULL)
        return errorset(PERSEUS_NULLDESCR, "null descriptor");
    for (int i = 0; i < d->plan.nstages; i++)
        interp[i] = d->plan.interp[i];
    return errornone(d->plan.nstages);
}

/* The plan perseus_set_sampling_rate(sps) would select -- nearest table rate, the midpoint to the lower one, as
 * perseus-sdr.c:776-811 -- WITHOUT a descriptor: no perseus_init / perseus_exit, no global state touched, so a host
 * that has receivers open can ask (bench.py and the tools used to run init/open/.../exit for this, which tore down
 * whatever the process had open).  Same outputs as perseus_amd_get_plan + perseus_amd_get_plan_interp; *rate gets
 * the table rate.  Returns the number of stages or PERSEUS_FPGANOTCFGD. */
int perseus_amd_plan_for_rate(int sps, int *rate, int decim[4], int ntaps[4], int interp[4], float *taps[4])
{
    const int idx = rate_index(sps);
    if (idx < 0)
        return PERSEUS_FPGANOTCFGD;
    ddc_plan pl;
    memset(&pl, 0, sizeof(pl));
    if (plan_build(&pl, k_rates[idx]) <= 0) {       /* (0: no plan for the rate, or no memory for its taps) */
        plan_free(&pl);
        return PERSEUS_FPGANOTCFGD;
    }
    if (rate)
        *rate = k_rates[idx];
    for (int i = 0; i < pl.nstages; i++) {
        if (decim)
            decim[i] = pl.decim[i];
        if (ntaps)
            ntaps[i] = pl.ntaps[i];
        if (interp)
            interp[i] = pl.interp[i];
        if (taps && taps[i])
            memcpy(taps[i], pl.taps[i], sizeof(float) * (size_t)pl.ntaps[i]);
    }
    const int n = pl.nstages;
    plan_free(&pl);
    return n;
}

int perseus_amd_get_plan(perseus_descr *d, int decim[4], int ntaps[4], float *taps[4])
{
    if (d == NULL)
        return errorset(PERSEUS_NULLDESCR, "null descriptor");
    for (int i = 0; i < d->plan.nstages; i++) {
        if (decim)
            decim[i] = d->plan.decim[i];
        if (ntaps)
            ntaps[i] = d->plan.ntaps[i];
        if (taps && taps[i])
            memcpy(taps[i], d->plan.taps[i], sizeof(float) * (size_t)d->plan.ntaps[i]);
    }
    return errornone(d->plan.nstages);
}
