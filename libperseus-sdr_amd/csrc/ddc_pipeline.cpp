/*
 * ddc_pipeline.cpp -- host side of the thin C ABI (include/perseus_ddc.h).
 *
 * Owns the per-stream state the reference never needed because its DDC lives
 * in the FPGA (SURVEY.md 5 "checkpoint/resume"): FIR history per stage,
 * decimation phase per stage, and the 64-bit sample counter that makes the
 * NCO phase a pure function of the absolute sample index
 * (phase(n) = n * freg mod 2^32, freg per perseus-sdr.c:584).
 *
 * There is no CPU implementation behind these entry points: if HIP reports no
 * device they fail with PDDC_ENODEV.
 */
#include "../../include/perseus_ddc.h"
#include "ddc_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

using namespace pddc;

static thread_local char g_err[512] = "no error";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

/* the same for the library's other translation units (ddc_multi.cpp); not exported */
extern "C" __attribute__((visibility("hidden"))) int pddc_set_error_(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return fail(e__ == hipErrorOutOfMemory ? PDDC_ENOMEM                                   \
                        : (e__ == hipErrorNoDevice || e__ == hipErrorInvalidDevice) ? PDDC_ENODEV  \
                                                                                    : PDDC_EHIP,   \
                        "%s: %s", #expr, hipGetErrorString(e__));                                  \
    } while (0)

/* ------------------------------------------------------------------------ */
struct Stage {
    int decim = 1;
    int interp = 1;               /* L of a rational L/decim stage (1 = plain decimator) */
    int ntaps = 0;
    std::vector<float> taps;      /* host copy (after optional fp16 rounding)  */
    float *d_taps = nullptr;      /* h[k] linear: points INTO d_taps_base (zero-padded on both sides) */
    float *d_taps_base = nullptr; /* the allocation                             */
    float *d_taps_dup = nullptr;  /* the same table with every tap twice, (h[k], h[k]): what k_fir_generic reads;
                                     points INTO d_taps_dup_base                 */
    float *d_taps_dup_base = nullptr;
    float *d_taps_blk = nullptr;  /* [ntb][8] block-reversed (fused kernel)     */
    float *d_taps_poly = nullptr; /* rational stage: [L][Kp] polyphase rows g[ph][j] = h[j*L + ph] */
    float *d_taps_seg = nullptr;  /* stage 2 as the fused cascade's third stage: h[k] zero padded to spl*seglen */
    float *d_taps_firp = nullptr; /* k_firp (plain decimators by 4, 5, 8, 10): (h[k], h[k]) zero padded to firp_taps_len */
    void *d_taps_f16 = nullptr;   /* PDDC_F_TAPS_FP16, stage 0 without NCO: the taps as binary16 values (1 KB with padding) -- the
                                   * only form of them k_fir_i8x reads then; its matrix waves quantise them into their operand
                                   * registers themselves (FirI8xArgs::taps16) */
    float i8_two_e = 0.0f;        /* ... scaled by this power of two first (2^E of fir_i8x_build_tables) */
    bool i8x_ok = false;          /* stage 0: k_fir_i8x can hold these taps (not all zero, finite) */
    int poly_k = 0, poly_kp = 0;
    int ntb = 0;                  /* tap blocks if fused-capable, else 0        */
    int hist = 0;                 /* history length in samples (mult. of 8)     */
    int hist_elem = 8;            /* bytes per history sample: 6 packed, 8 float2 */
    /* the `hist` input samples that precede the next batch.  Two buffers,
     * alternated every call: a launch reads d_hist[cur] and leaves the new
     * history in d_hist[cur^1], so no block can see a half-updated history
     * and no separate update kernel sits between two launches.               */
    void *d_hist[2] = { nullptr, nullptr };
    int cur = 0;
    /* input data of stages >= 1 (and the unpacked floats of a generic stage 0) */
    float *d_buf = nullptr;
    size_t buf_cap = 0;           /* capacity in samples                        */
    bool buf_in_ws = false;       /* d_buf lies in the caller's workspace (pddc_pipeline_set_workspace): not ours to free */
    float *d_buf_alt = nullptr;   /* overlap mode, stage 2 only: the second half of the double buffer the fused pair writes */
    size_t buf_alt_cap = 0;
    unsigned long long consumed = 0;   /* inputs consumed since reset           */
};

struct pddc_pipeline {
    int device = 0;
    uint32_t flags = 0;
    int nstages = 0;
    Stage st[PDDC_MAX_STAGES];
    uint32_t freg = 0;
    /* phase(n) = n*freg + phase_off (mod 2^32).  A retune at sample n changes freg and moves
     * phase_off by n*(freg_old - freg_new), so the phase is continuous there -- the FPGA's NCO
     * is a phase accumulator, a new tuning word changes its increment, never its value.
     * freg_applied: the word the last processed batch was mixed with (the fused kernel re-mixes
     * its raw packed history and needs it for the first batch after a retune).               */
    uint32_t phase_off = 0;
    uint32_t freg_applied = 0;
    bool fresh = true;            /* nothing processed since create / reset / seek */
    /* Tuning-word segments that still reach into stage 0's history window [n0 - H, n0): {first
     * sample, word, offset}; the last one is the word in force.  Stage 0 keeps its history as raw
     * packed samples and mixes them when it reads them, with ONE word (freg_applied): that is exact
     * as long as the whole window was mixed with one word.  Batches shorter than the history with
     * retunes between them break that; such a batch takes the float route (process(): "mixed
     * history"), where each stretch of the history is mixed with its own word.                  */
    struct WordSeg {
        long long n_begin;
        uint32_t freg, off;
    };
    std::vector<WordSeg> segs;
    float *d_hist_f32 = nullptr;  /* stage 0's history as mixed float2, for that route */
    int fail_at_stage = -1;       /* test hook: the next process() fails when it reaches this stage */
    /* measurement hook: HIP events around the stage-0 (or fused-pair) kernel of every process() */
    bool time_stage0 = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    float lo_c[8], lo_s[8];
    float lo_c_applied[8], lo_s_applied[8];   /* step phasors of freg_applied */
    unsigned long long n0 = 0;    /* absolute sample counter (stage 0 input)    */
    int R = 4;                    /* outputs per lane of the fused kernel (4: 16 waves/CU) */
    int NT = 256;                 /* threads per block of the unfused stage-0 kernel (128: four blocks per CU) */
    /* staging for push_host / push_host_async: two slots, so that the H2D copy of batch k+1,
     * the kernels of batch k and the D2H copy of batch k-1 run at the same time (three streams,
     * PCIe is full duplex); a slot is reused only after its previous D2H has finished        */
    struct HostSlot {
        uint8_t *d_in = nullptr;
        size_t in_cap = 0;            /* samples */
        float *d_out = nullptr;
        size_t out_cap = 0;           /* samples */
        hipEvent_t ev_in = nullptr, ev_comp = nullptr, ev_out = nullptr;
        hipEvent_t ev_wait = nullptr; /* what the slot's ticket waits for: ev_out, or the event of the gang round it went out with */
        void *h_out_seen = nullptr;   /* gang rounds: the last host output buffer asked about, and its address as the   */
        float *h_out_dev = nullptr;   /* device sees it (pinned memory, pddc_host_alloc) or NULL: copy out as usual     */
        bool used = false;
    } slot[2];
    int next_slot = 0;
    hipStream_t s_in = nullptr, s_out = nullptr;   /* copy streams; own_stream computes */
    float *d_fout = nullptr;      /* float output of the last stage when the caller wants packed */
    size_t d_fout_cap = 0;
    hipStream_t own_stream = nullptr;
    unsigned *d_sched = nullptr;  /* k_fir8's tile scheduler words (zero between launches) + its sticky error word */
    /* fused cascade (stages 0+1+2 in one kernel, launch_fir8_fused3): geometry of the third stage, the seam slots
     * through which a chunk of tiles hands its last outputs to the chunk behind it, one flag per chunk            */
    /* overlap mode (pddc_pipeline_set_overlap): the stage behind the fused pair is not launched with its batch; the NEXT
     * batch's pair carries it along as extra thread blocks (Fir8Args::tail), pddc_pipeline_fence launches what is
     * left.  The pair writes two alternating workspace halves.                                                  */
    bool overlap = false;
    int ov_parity = 0;
    bool carry_pending = false;   /* carry_tail has not been launched yet */
    GenTail carry_tail;
    bool s3_ok = false;
    Fir8Stage3 s3;
    void *d_seam = nullptr;
    unsigned *d_flags = nullptr;
    /* gang submission (pddc_gang_push_async): several pipelines of one GPU share one launch chain.  gang_rec != NULL
     * turns pddc_pipeline_process into "record, do not launch": the first-stage launch and the tail behind it are
     * written there, the gang launches them for all its members at once.                                          */
    struct GangRec *gang_rec = nullptr;
    struct pddc_gang *gang = nullptr;        /* the gang whose stream the last push used (NULL: the pipeline's own)   */
    /* kernel selection (pddc_pipeline_set_option; defaults = what the measurements chose; the environment is looked at
     * once, when the pipeline is created, never on the data path) */
    struct Opts {
        int no_i8 = 0;            /* 1: the vector kernels only (k_fir8)                                              */
        int i8x = 1;              /* tuned first stages (PDDC_F_MIX) on k_fir_i8x (0: k_fir8)                         */
        int i8x_pair = 1;         /* ... and the cascade's first two stages as its fused pair (0: unfused, or k_fir8)  */
        int i8x_plain = 1;        /* untuned first stages on k_fir_i8x too (0: k_fir8): at 2^28 samples 32/48 taps 0.319 against
                                     k_fir8's 0.339 ms; at 2^22 5.9 against 12.9 us (profiles/r04).  Round 3's k_fir_i8, which
                                     this option used to fall back to, is gone: one kernel family since round 5           */
        int i8x_blocks = 0;       /* persistent grid override of k_fir_i8x (0: one block per CU)                       */
        int i8x_chunk = 0;        /* tiles per chunk of its walk (0: 1; fused pair 4, from 2^23 samples on 8)              */
        int i8x_layout = -1;      /* which waves finish a tile (ddc_fir_i8.hip "Who does what"): 0 the matrix waves, 1 the
                                     loaders, 2 two matrix + two finishing waves; -1: by form (0 without the NCO and for
                                     tuned stages up to 128 taps, 1 for 129..256 tuned taps and for the fused pair)      */
        int i8x_pair_max_log2 = 25;   /* the fused pair on the matrix cores up to 2^25-sample batches: 3.6x k_fir8's pair at 2^22, 1.6x at
                                       * 2^24, 1.14x at 2^25; at 2^26 k_fir8's pair -- bursts, porch, round 6 -- is 8-10 % ahead (83 against
                                       * 92 us, profiles/r06/f_pair_crossover.txt), at 2^28 7 % and carries the tail in its launch */
        int no_fuse2 = 0, fuse3 = 0;
    } opt;
    /* k_fir_i8x's operands follow the tuning word: they are rebuilt on the host when the word, the taps or the form
     * (mix / fused pair) change and travel to the device in stream order -- a retune is one 37..50 KB asynchronous copy.
     * Four slots (pinned host staging + device copy) so that a launch queued earlier keeps reading its own table; a slot is
     * reused only after everything that was queued when it was left has finished (its event).                         */
    struct I8xSlot {
        void *d = nullptr, *h = nullptr;
        hipEvent_t left = nullptr;
        bool left_valid = false;
    };
    struct I8x {
        I8xSlot slot[4];
        int cur = -1;
        uint32_t freg = 0;
        bool mix = false, fuse2 = false;
        unsigned taps_ver = 0;
        hipStream_t stream = nullptr;
        float scale = 0.0f, ct[2] = { 0.0f, 0.0f };
        /* the decimate-by-10 form: one table set per delay 0 .. 7 of the taps (a batch's decimation phase), built when first
         * needed for the present word and taps and kept -- the phases of a stream's batches come round again.  Two
         * buffers per delay with a `left` event each, as the slots above: a retune rebuilds a delay's set into the OTHER
         * buffer and waits for nothing but that buffer's last readers of two retunes ago (the advisor, round 4: this was a
         * hipDeviceSynchronize per delay -- up to eight device-wide stalls behind one retune) */
        struct D10 {
            I8xSlot buf[2];
            int cur = 0;
            bool valid = false;
            uint32_t freg = 0;
            unsigned taps_ver = 0;
            hipStream_t stream = nullptr;
            float scale = 0.0f, ct[2] = { 0.0f, 0.0f };
        } d10[8];
    } i8x;
    unsigned taps_ver = 1;        /* bumped whenever a stage's taps change */
};
static constexpr size_t kI8xSlotBytes = 64 * 1024;     /* tables (<= 48 KB) + the second stage's taps at kI8xTaps2Off */
static constexpr size_t kI8xTaps2Off = 60 * 1024;

/* what one member's process() leaves for the gang: kind 0 = nothing recorded */
struct GangRec {
    int kind = 0;                 /* 1: the packed /8 first stage alone (launch_fir8), 2: the fused pair; 3: k_fir_i8x
                                     (the tuned first stage on the matrix cores, alone or as its fused pair: `fuse2`)  */
    int ntb = 0, R = 4;
    bool mix = false;
    Fir8Args a;
    FirI8xArgs ax;                /* kind 3 */
    int hist = 0, chunk = 0, layout = -1, blocks = 0;
    bool fuse2 = false;
    GenTail tail;                 /* nblocks == 0: the plan ends with the first-stage kernel                       */
};

struct pddc_gang {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t s_gen = nullptr;             /* the on-device source of the NEXT round runs beside this round's kernels */
    hipEvent_t ev_gen[4] = { nullptr, nullptr, nullptr, nullptr };
    hipEvent_t ev[4] = { nullptr, nullptr, nullptr, nullptr };
    int next_ev = 0;
    std::mutex lock;
    std::vector<pddc_pipeline *> members;    /* pipelines whose last push went through this gang's stream            */
};

static bool stage0_fused(const pddc_pipeline *p);
static bool stage0_packed_generic(const pddc_pipeline *p);
static int stage0_i8_kind(const pddc_pipeline *p, size_t nsamples);
static int setup_stage3(pddc_pipeline *p);
static bool stages01_i8x(const pddc_pipeline *p, size_t nsamples);
static int leave_gang(pddc_pipeline *p);
static float *direct_out(pddc_pipeline::HostSlot &sl, void *h_out);

static float round_to_half(float v)
{
    /* round-to-nearest-even to IEEE binary16, returned widened to float */
    _Float16 h = (_Float16)v;
    return (float)h;
}

static void lo_steps(uint32_t freg, float *c, float *s)
{
    const double k = 6.283185307179586476925286766559 / 4294967296.0;
    for (int e = 0; e < 8; ++e) {
        const uint32_t ph = (uint32_t)((uint64_t)e * freg);
        c[e] = (float)std::cos(k * (double)ph);
        s[e] = (float)(-std::sin(k * (double)ph));
    }
}

static void compute_lo_steps(pddc_pipeline *p)
{
    lo_steps(p->freg, p->lo_c, p->lo_s);
    lo_steps(p->freg_applied, p->lo_c_applied, p->lo_s_applied);
}

static bool stage_fused_capable(const Stage &s)
{
    if (s.decim != 8 || s.interp != 1 || s.ntaps > PDDC_FAST_MAX_TAPS)
        return false;
    return true;
}

static int pick_ntb(int ntaps)
{
    const int need = (ntaps + 7) / 8;
    const int opts[4] = { 4, 8, 16, 32 };
    for (int o : opts)
        if (need <= o)
            return o;
    return 0;
}

static int upload_taps(pddc_pipeline *p, int si)
{
    Stage &s = p->st[si];
    p->taps_ver++;
    if (s.d_taps_base) {
        hipFree(s.d_taps_base);
        s.d_taps_base = nullptr;
        s.d_taps = nullptr;
    }
    if (s.d_taps_blk) {
        hipFree(s.d_taps_blk);
        s.d_taps_blk = nullptr;
    }
    /* zero-padded by 3*D + 8 on both sides: k_fir_generic reads tap p*D + j for every j of
     * a thread's P <= 4 merged windows (in aligned steps of 8) without bounds checks */
    {
        const size_t Z = 3 * (size_t)s.decim + 8;
        std::vector<float> padded(Z + (size_t)s.ntaps + Z + 8, 0.0f);
        std::copy(s.taps.begin(), s.taps.begin() + s.ntaps, padded.begin() + (long)Z);
        HIP_TRY(hipMalloc(&s.d_taps_base, sizeof(float) * padded.size()));
        HIP_TRY(hipMemcpy(s.d_taps_base, padded.data(), sizeof(float) * padded.size(), hipMemcpyHostToDevice));
        s.d_taps = s.d_taps_base + Z;
        std::vector<float> dup(2 * padded.size());
        for (size_t k = 0; k < padded.size(); ++k)
            dup[2 * k] = dup[2 * k + 1] = padded[k];
        if (s.d_taps_dup_base)
            hipFree(s.d_taps_dup_base);
        s.d_taps_dup_base = nullptr;
        HIP_TRY(hipMalloc(&s.d_taps_dup_base, sizeof(float) * dup.size()));
        HIP_TRY(hipMemcpy(s.d_taps_dup_base, dup.data(), sizeof(float) * dup.size(), hipMemcpyHostToDevice));
        s.d_taps_dup = s.d_taps_dup_base + 2 * Z;
    }
    if (s.d_taps_poly) {
        hipFree(s.d_taps_poly);
        s.d_taps_poly = nullptr;
    }
    if (s.d_taps_firp) {
        hipFree(s.d_taps_firp);
        s.d_taps_firp = nullptr;
    }
    if (s.d_taps_f16) {
        hipFree(s.d_taps_f16);
        s.d_taps_f16 = nullptr;
    }
    /* (the history length is fixed at create time -- 8 * tap blocks -- and not known yet when this runs for the first time) */
    const int i8_hist = s.hist ? s.hist : 8 * pick_ntb(s.ntaps);
    {
        double hmax = 0.0;
        for (int k = 0; k < s.ntaps; ++k)
            hmax = std::fmax(hmax, std::fabs((double)s.taps[k]));
        s.i8x_ok = si == 0 && hmax > 0.0 && std::isfinite(hmax);
    }
    if (si == 0 && s.i8x_ok && stage_fused_capable(s) && s.ntaps <= i8_hist && (p->flags & PDDC_F_TAPS_FP16) &&
        !(p->flags & (PDDC_F_NO_FAST | PDDC_F_MIX)) && fir_i8x_supported(i8_hist, false, false)) {
        /* binary16 tap storage on the matrix cores (BASELINE config 5): s.taps hold binary16 values already; the device gets
         * them as such, 2 bytes a tap, and no operand table (the host's serves for scale and offset constant only) */
        std::vector<int8_t> tab(fir_i8x_table_bytes(i8_hist, false));
        float sc = 0.0f, ct[2];
        int e2 = 0;
        if (fir_i8x_build_tables(s.taps.data(), s.ntaps, i8_hist, false, 0u, tab.data(), &sc, ct, &e2)) {
            std::vector<uint16_t> h16((size_t)kFirI8Taps16Len);
            fir_i8_taps16(s.taps.data(), s.ntaps, i8_hist, h16.data());
            HIP_TRY(hipMalloc(&s.d_taps_f16, h16.size() * sizeof(uint16_t)));
            HIP_TRY(hipMemcpy(s.d_taps_f16, h16.data(), h16.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            s.i8_two_e = std::ldexp(1.0f, e2);
        }
    }
    if (s.interp == 1 && !(p->flags & PDDC_F_NO_FAST) && firp_supported(s.decim, s.ntaps)) {
        /* stage 0 runs k_firp on the packed samples, which leaves the unpack scale to the taps (like k_fir8) */
        const float tap_scale = si == 0 ? kFir8PackedTapScale : 1.0f;
        std::vector<float> t2(2 * (size_t)firp_taps_len(s.decim, s.ntaps), 0.0f);
        for (int k = 0; k < s.ntaps; ++k)
            t2[2 * (size_t)k] = t2[2 * (size_t)k + 1] = s.taps[k] * tap_scale;
        HIP_TRY(hipMalloc(&s.d_taps_firp, sizeof(float) * t2.size()));
        HIP_TRY(hipMemcpy(s.d_taps_firp, t2.data(), sizeof(float) * t2.size(), hipMemcpyHostToDevice));
    }
    if (s.interp > 1 && resample_lds_supported(s.interp, s.decim, s.ntaps)) {
        s.poly_k = (s.ntaps + s.interp - 1) / s.interp;
        s.poly_kp = (s.poly_k + 3) / 4 * 4;
        std::vector<float> g((size_t)s.interp * s.poly_kp, 0.0f);
        for (int ph = 0; ph < s.interp; ++ph)
            for (int j = 0; j < s.poly_k; ++j) {
                const int k = j * s.interp + ph;
                g[(size_t)ph * s.poly_kp + j] = k < s.ntaps ? s.taps[k] : 0.0f;
            }
        HIP_TRY(hipMalloc(&s.d_taps_poly, sizeof(float) * g.size()));
        HIP_TRY(hipMemcpy(s.d_taps_poly, g.data(), sizeof(float) * g.size(), hipMemcpyHostToDevice));
    }
    s.ntb = stage_fused_capable(s) ? pick_ntb(s.ntaps) : 0;
    /* a second stage that can be fused behind stage 0 always uses 8 tap blocks
     * (64-sample history), fused or not, so both paths share one state format */
    if (si == 1 && s.ntb != 0 && s.ntaps <= 64)
        s.ntb = 8;
    if (s.ntb) {
        /* hb[j][e] = h[8j + 7 - e], zero beyond ntaps */
        /* stage 0's block table is only ever used by the packed-input kernel, which leaves
         * the unpack scale to the taps */
        const float tap_scale = si == 0 ? kFir8PackedTapScale : 1.0f;
        std::vector<float> blk((size_t)s.ntb * 8, 0.0f);
        for (int j = 0; j < s.ntb; ++j)
            for (int e = 0; e < 8; ++e) {
                const int k = 8 * j + 7 - e;
                blk[(size_t)j * 8 + e] = k < s.ntaps ? s.taps[k] * tap_scale : 0.0f;
            }
        HIP_TRY(hipMalloc(&s.d_taps_blk, sizeof(float) * blk.size()));
        HIP_TRY(hipMemcpy(s.d_taps_blk, blk.data(), sizeof(float) * blk.size(), hipMemcpyHostToDevice));
    }
    return PDDC_OK;
}

/* ------------------------------------------------------------------------ */
extern "C" {

int pddc_version(void) { return 100; }

const char *pddc_last_error(void) { return g_err; }

int pddc_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e == hipErrorNoDevice)
        return 0;
    if (e != hipSuccess)
        return fail(PDDC_EHIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

uint32_t pddc_nco_freg(double center_freq_hz, double adc_clk_hz)
{
    /* perseus-sdr.c:584 -- double arithmetic, truncation toward zero */
    return (uint32_t)(center_freq_hz / adc_clk_hz * 4.294967296E9);
}

/* host arithmetic only (no device needed): the plain form's tap operand, so that its digits can be checked on any machine */
int pddc_fir_i8_table(const float *taps, int ntaps, int hist, int8_t *table, size_t table_bytes, float *scale, float *cterm)
{
    if (!taps || !table || !scale || !cterm)
        return fail(PDDC_EINVAL, "null argument");
    if (!fir_i8x_supported(hist, false, false))
        return fail(PDDC_EINVAL, "history %d: 32, 64, 128 or 256", hist);
    const size_t need = fir_i8x_table_bytes(hist, false);
    if (table_bytes < need)
        return fail(PDDC_ECAPACITY, "table needs %zu bytes, buffer has %zu", need, table_bytes);
    float ct[2] = { 0.0f, 0.0f };
    if (!fir_i8x_build_tables(taps, ntaps, hist, false, 0u, table, scale, ct))
        return fail(PDDC_EINVAL, "no int8 form for these taps (1..%d taps, not all zero, finite)", hist);
    *cterm = ct[0];
    return PDDC_OK;
}

/* ... and k_fir_i8x's operands with the NCO */
int pddc_fir_i8x_tables(const float *taps, int ntaps, int hist, int mix, uint32_t freg, int8_t *tables, size_t tables_bytes,
                        float *scale, float *ct)
{
    if (!taps || !tables || !scale || !ct)
        return fail(PDDC_EINVAL, "null argument");
    if (!fir_i8x_supported(hist, mix != 0, false))
        return fail(PDDC_EINVAL, "hist must be 32, 64, 128 or 256");
    if (tables_bytes < fir_i8x_table_bytes(hist, mix != 0))
        return fail(PDDC_EINVAL, "tables need %zu bytes", fir_i8x_table_bytes(hist, mix != 0));
    if (!fir_i8x_build_tables(taps, ntaps, hist, mix != 0, freg, tables, scale, ct))
        return fail(PDDC_EINVAL, "taps cannot be quantised (all zero, not finite, or more than hist)");
    return fir_i8x_mode(hist, mix != 0) == 0 ? 1 : 2;
}

int pddc_fir_i8x_d10_tables(const float *taps, int ntaps, int delay, uint32_t freg, int8_t *tables, size_t tables_bytes,
                            float *scale, float *ct)
{
    if (!taps || !tables || !scale || !ct)
        return fail(PDDC_EINVAL, "null argument");
    if (tables_bytes < fir_i8x_d10_table_bytes())
        return fail(PDDC_EINVAL, "tables need %zu bytes", fir_i8x_d10_table_bytes());
    if (!fir_i8x_d10_build_tables(taps, ntaps, delay, freg, tables, scale, ct))
        return fail(PDDC_EINVAL, "taps cannot be quantised (all zero, not finite, or ntaps + delay > %d)", kFirI8xD10Hist);
    return 2;
}

int pddc_fir_i8x_taps2(const float *taps2, int ntaps2, int mix, uint32_t freg, float *out, size_t out_len)
{
    if (!taps2 || !out || ntaps2 < 1 || ntaps2 > 64 || out_len < (size_t)kFirI8xTaps2Len)
        return fail(PDDC_EINVAL, "bad argument (1..64 taps, %d floats out)", kFirI8xTaps2Len);
    fir_i8x_taps2(taps2, ntaps2, mix != 0, freg, out);
    return PDDC_OK;
}

/* host arithmetic only: the binary16 tap array k_fir_i8x reads under PDDC_F_TAPS_FP16 (no NCO), and the 2^E it scales them with */
int pddc_fir_i8_taps16(const float *taps, int ntaps, int hist, uint16_t *out, size_t out_len, double *two_e)
{
    if (!taps || !out || !two_e)
        return fail(PDDC_EINVAL, "null argument");
    if (!fir_i8x_supported(hist, false, false))
        return fail(PDDC_EINVAL, "history %d: 32, 64, 128 or 256", hist);
    if (out_len < (size_t)kFirI8Taps16Len)
        return fail(PDDC_ECAPACITY, "the array has %d entries, buffer has %zu", kFirI8Taps16Len, out_len);
    std::vector<int8_t> tab(fir_i8x_table_bytes(hist, false));
    std::vector<float> h16(taps, taps + (ntaps > 0 ? ntaps : 0));
    for (float &v : h16)
        v = round_to_half(v);
    float sc = 0.0f, ct[2] = { 0.0f, 0.0f };
    int e2 = 0;
    if (ntaps < 1 || !fir_i8x_build_tables(h16.data(), ntaps, hist, false, 0u, tab.data(), &sc, ct, &e2))
        return fail(PDDC_EINVAL, "no int8 form for these taps (1..%d taps, not all zero, finite)", hist);
    fir_i8_taps16(h16.data(), ntaps, hist, out);
    *two_e = std::ldexp(1.0, e2);
    return PDDC_OK;
}

static int require_device(void)
{
    int n = pddc_device_count();
    if (n < 0)
        return n;
    if (n == 0)
        return fail(PDDC_ENODEV, "no HIP device visible (this library has no CPU fallback)");
    return PDDC_OK;
}

int pddc_set_device(int device)
{
    int rc = require_device();
    if (rc)
        return rc;
    HIP_TRY(hipSetDevice(device));
    return PDDC_OK;
}

int pddc_malloc(void **d_ptr, size_t nbytes)
{
    if (!d_ptr)
        return fail(PDDC_EINVAL, "null pointer");
    int rc = require_device();
    if (rc)
        return rc;
    HIP_TRY(hipMalloc(d_ptr, nbytes ? nbytes : 16));
    return PDDC_OK;
}

/* HBM is laid out in a few classes of large extents (tens of GiB; profiles/r02/i_placement_map.txt): a kernel that
 * streams reads from one buffer and writes to another runs ~8 % faster when the two lie in extents of different
 * classes -- streams that share a class get in each other's way.  Two allocations made one after the other usually
 * share an extent.  This walks: allocate a candidate, time a read+write probe stream between the partner and it, put
 * an 8 GiB spacer behind it, try again further on, until both speeds have been seen (or max_candidates); the fastest
 * candidate is returned, everything else freed.                                                                  */
/* min_bytes: buffers smaller than this are not walked.  A candidate smaller than the probe's 1 GiB is ALLOCATED at
 * 1 GiB (the caller uses its first nbytes): the probe must write past the 256 MB last-level cache to see the HBM, and
 * a 33 MB buffer written once per launch matters as much as a large one -- the fused pair of the x320 cascade, which
 * writes 1/48 of what it reads, runs at 0.292 or 0.330 ms depending on it (tools/placement_probe.py --mode small).           */
static int malloc_apart_impl(void **d_ptr, size_t nbytes, const void *d_partner, size_t partner_bytes, int max_candidates,
                             float *ms_best, float *ms_worst, size_t min_bytes, hipStream_t stream = nullptr)
{
    if (!d_ptr || nbytes == 0)
        return fail(PDDC_EINVAL, "bad argument");
    *d_ptr = nullptr;
    int rc = require_device();
    if (rc)
        return rc;
    if (ms_best)
        *ms_best = 0.0f;
    if (ms_worst)
        *ms_worst = 0.0f;
    if (!d_partner || partner_bytes < (64u << 20) || nbytes < min_bytes || max_candidates <= 1 ||
        ((uintptr_t)d_partner & 15)) {
        HIP_TRY(hipMalloc(d_ptr, nbytes));             /* too small to matter (or nothing to stay away from) */
        return PDDC_OK;
    }
    size_t spacer_bytes = (size_t)8 << 30;
    const size_t total = (size_t)1 << 30;               /* 1 GiB read + 1 GiB written per probe launch: beyond the L3 */
    const size_t asked = nbytes;
    if (nbytes < total)
        nbytes = total;
    {
        /* candidates and spacers together never take more than half of what is free: other allocations on this GPU
         * (another pipeline, torch's allocator) must not hit out-of-memory because of a search */
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess)
            free_b = 0;
        (void)hipGetLastError();
        const size_t budget = free_b / 2;
        if (budget < 2 * nbytes + spacer_bytes) {
            HIP_TRY(hipMalloc(d_ptr, asked));
            return PDDC_OK;
        }
        const size_t per = nbytes + spacer_bytes;
        if ((size_t)max_candidates > budget / per)
            max_candidates = (int)(budget / per);
        if (max_candidates < 2)
            max_candidates = 2;
    }
    std::vector<void *> cands, spacers;
    std::vector<float> ms;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    auto cleanup = [&](void *keep) {
        for (void *c : cands)
            if (c != keep)
                hipFree(c);
        for (void *sp : spacers)
            hipFree(sp);
        hipEventDestroy(e0);
        hipEventDestroy(e1);
    };
    auto both_seen = [&]() {
        float lo = ms[0];
        for (float t : ms)
            lo = t < lo ? t : lo;
        for (float t : ms)
            if (t > 1.04f * lo && t < 1.25f * lo)
                return true;
        return false;
    };
    while ((int)cands.size() < max_candidates) {
        void *c = nullptr;
        if (hipMalloc(&c, nbytes) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        cands.push_back(c);
        hipError_t e = hipSuccess;
        for (int k = 0; k < 3 && e == hipSuccess; ++k)
            e = launch_stream_probe(d_partner, partner_bytes, c, nbytes & ~(size_t)15, total, stream);
        if (e == hipSuccess)
            e = hipEventRecord(e0, stream);
        for (int k = 0; k < 5 && e == hipSuccess; ++k)
            e = launch_stream_probe(d_partner, partner_bytes, c, nbytes & ~(size_t)15, total, stream);
        if (e == hipSuccess)
            e = hipEventRecord(e1, stream);
        if (e == hipSuccess)
            e = hipEventSynchronize(e1);
        float t = 0.0f;
        if (e == hipSuccess)
            e = hipEventElapsedTime(&t, e0, e1);
        if (e != hipSuccess) {                          /* a probe that fails costs the search, not the buffer */
            (void)hipGetLastError();
            cleanup(nullptr);
            HIP_TRY(hipMalloc(d_ptr, asked));
            return PDDC_OK;
        }
        ms.push_back(t / 5.0f);
        if (ms.size() >= 2 && both_seen())
            break;
        void *sp = nullptr;
        if (hipMalloc(&sp, spacer_bytes) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        spacers.push_back(sp);
    }
    if (cands.empty()) {
        cleanup(nullptr);
        return fail(PDDC_ENOMEM, "hipMalloc(%zu) failed", nbytes);
    }
    size_t best = 0, worst = 0;
    for (size_t k = 1; k < ms.size(); ++k) {
        if (ms[k] < ms[best])
            best = k;
        if (ms[k] > ms[worst])
            worst = k;
    }
    if (ms_best)
        *ms_best = ms[best];
    if (ms_worst)
        *ms_worst = ms[worst];
    *d_ptr = cands[best];
    cleanup(cands[best]);
    return PDDC_OK;
}

int pddc_malloc_apart(void **d_ptr, size_t nbytes, const void *d_partner, size_t partner_bytes, int max_candidates,
                      float *ms_best, float *ms_worst)
{
    return malloc_apart_impl(d_ptr, nbytes, d_partner, partner_bytes, max_candidates, ms_best, ms_worst, (size_t)1 << 20);
}

int pddc_free(void *d_ptr)
{
    if (d_ptr)
        HIP_TRY(hipFree(d_ptr));
    return PDDC_OK;
}

int pddc_memcpy_h2d(void *d_dst, const void *h_src, size_t nbytes, void *stream)
{
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, nbytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return PDDC_OK;
}

int pddc_memcpy_d2h(void *h_dst, const void *d_src, size_t nbytes, void *stream)
{
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, nbytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return PDDC_OK;
}

int pddc_stream_sync(void *stream)
{
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return PDDC_OK;
}

static int check_unpack_args(const void *d_in, const void *d_out, size_t ns)
{
    if (ns == 0)
        return PDDC_OK;
    if (!d_in || !d_out)
        return fail(PDDC_EINVAL, "null device pointer");
    if (((uintptr_t)d_in & 15) || ((uintptr_t)d_out & 15))
        return fail(PDDC_EINVAL, "device pointers must be 16-byte aligned");
    return PDDC_OK;
}

int pddc_unpack24_f32(const void *d_packed, size_t nsamples, void *d_out, void *stream)
{
    int rc = require_device();
    if (rc)
        return rc;
    if ((rc = check_unpack_args(d_packed, d_out, nsamples)))
        return rc;
    HIP_TRY(launch_unpack24(d_packed, (long long)nsamples, d_out, false, false, 0, 0, 0, nullptr, nullptr,
                            (hipStream_t)stream));
    return PDDC_OK;
}

int pddc_unpack24_i32(const void *d_packed, size_t nsamples, void *d_out, void *stream)
{
    int rc = require_device();
    if (rc)
        return rc;
    if ((rc = check_unpack_args(d_packed, d_out, nsamples)))
        return rc;
    HIP_TRY(launch_unpack24(d_packed, (long long)nsamples, d_out, true, false, 0, 0, 0, nullptr, nullptr,
                            (hipStream_t)stream));
    return PDDC_OK;
}

int pddc_pack24_f32(const void *d_in, size_t nsamples, void *d_out, void *stream)
{
    int rc = require_device();
    if (rc)
        return rc;
    if ((rc = check_unpack_args(d_in, d_out, nsamples)))
        return rc;
    HIP_TRY(launch_pack24(static_cast<const float *>(d_in), (long long)nsamples, d_out, (hipStream_t)stream));
    return PDDC_OK;
}

int pddc_synth_lcg(void *d_dst, size_t nbytes, uint32_t seed, uint64_t byte_offset, void *stream)
{
    int rc = require_device();
    if (rc)
        return rc;
    if (nbytes && (!d_dst || ((uintptr_t)d_dst & 15)))
        return fail(PDDC_EINVAL, "destination must be a 16-byte aligned device pointer");
    HIP_TRY(launch_synth_lcg(d_dst, nbytes, seed, byte_offset, (hipStream_t)stream));
    return PDDC_OK;
}

/* ---------------------------------------------------------------- pipeline */
int pddc_pipeline_create(pddc_pipeline **out, int device, const pddc_stage_desc *stages, int nstages,
                         uint32_t flags)
{
    if (!out)
        return fail(PDDC_EINVAL, "null out pointer");
    *out = nullptr;
    if (!stages || nstages < 1 || nstages > PDDC_MAX_STAGES)
        return fail(PDDC_EINVAL, "nstages must be 1..%d", PDDC_MAX_STAGES);
    for (int i = 0; i < nstages; ++i) {
        if (stages[i].decim < 1 || stages[i].decim > 4096)
            return fail(PDDC_EINVAL, "stage %d: bad decimation %d", i, stages[i].decim);
        if (stages[i].interp < 0 || stages[i].interp > 256)
            return fail(PDDC_EINVAL, "stage %d: bad interpolation %d", i, stages[i].interp);
        const int max_taps = stages[i].interp > 1 ? PDDC_MAX_TAPS : PDDC_MAX_TAPS_DECIM;
        if (stages[i].ntaps < 1 || stages[i].ntaps > max_taps || !stages[i].taps)
            return fail(PDDC_EINVAL, "stage %d: ntaps must be 1..%d", i, max_taps);
        /* a plain decimator may always end up on the generic kernel (ragged batches, odd phases):
         * its smallest block shape must fit the LDS, or the first process() would fail half way */
        if (stages[i].interp <= 1 && !fir_generic_supported(stages[i].decim, stages[i].ntaps))
            return fail(PDDC_EINVAL,
                        "stage %d: decimate-by-%d with %d taps does not fit: (63*D + ntaps + 10) samples of 8 bytes "
                        "must fit the 160 KiB of LDS (D <= ~320 for short filters); split the stage",
                        i, stages[i].decim, stages[i].ntaps);
    }
    int rc = require_device();
    if (rc)
        return rc;
    int ndev = pddc_device_count();
    if (device < 0 || device >= ndev)
        return fail(PDDC_ENODEV, "device %d out of range (0..%d)", device, ndev - 1);
    HIP_TRY(hipSetDevice(device));

    pddc_pipeline *p = new (std::nothrow) pddc_pipeline();
    if (!p)
        return fail(PDDC_ENOMEM, "out of host memory");
    p->device = device;
    p->flags = flags;
    p->nstages = nstages;
    /* kernel selection: the environment is read HERE, once per pipeline, and never on the data path
     * (pddc_pipeline_set_option changes it afterwards) */
    {
        auto env_int = [](const char *name, int dflt) {
            const char *e = getenv(name);
            return e ? atoi(e) : dflt;
        };
        p->opt.no_i8 = getenv("PDDC_NO_I8") ? 1 : 0;
        p->opt.i8x = env_int("PDDC_I8X", 1);
        p->opt.i8x_pair = env_int("PDDC_I8X_PAIR", 1);
        p->opt.i8x_plain = env_int("PDDC_I8X_PLAIN", 1);
        p->opt.i8x_blocks = env_int("PDDC_I8X_BLOCKS", 0);
        p->opt.i8x_chunk = env_int("PDDC_I8X_CHUNK", 0);
        p->opt.i8x_layout = env_int("PDDC_I8X_LAYOUT", -1);
        p->opt.i8x_pair_max_log2 = env_int("PDDC_I8X_PAIR_MAX_LOG2", 25);
        p->opt.no_fuse2 = getenv("PDDC_NO_FUSE2") ? 1 : 0;
        p->opt.fuse3 = env_int("PDDC_FUSE3", 0);
    }
    /* tuning knobs (development): outputs per lane and persistent grid size */
    if (const char *e = getenv("PDDC_FIR8_R"))
        if (atoi(e) == 4 || atoi(e) == 8)
            p->R = atoi(e);
    {
        const char *e = getenv("PDDC_FIR8_BLOCKS");      /* (process-wide; back to the default when the variable is gone) */
        fir8_set_grid_blocks(e ? atoi(e) : 0);
    }
    for (int i = 0; i < nstages; ++i) {
        Stage &s = p->st[i];
        s.decim = stages[i].decim;
        s.interp = stages[i].interp > 1 ? stages[i].interp : 1;
        s.ntaps = stages[i].ntaps;
        s.taps.assign(stages[i].taps, stages[i].taps + s.ntaps);
        if (flags & PDDC_F_TAPS_FP16)
            for (float &v : s.taps)
                v = round_to_half(v);
        if ((rc = upload_taps(p, i))) {
            pddc_pipeline_destroy(p);
            return rc;
        }
        /* history: enough for ntaps-1, rounded to the 8-sample granule; the
         * fused kernel wants exactly 8*ntb */
        s.hist = s.ntb ? 8 * s.ntb : ((s.ntaps - 1 + 7) / 8) * 8;
        if (s.interp > 1)           /* at most ceil(ntaps/L) inputs per output */
            s.hist = (((s.ntaps + s.interp - 1) / s.interp) + 7) / 8 * 8;
        if (s.hist == 0)
            s.hist = 8;
    }
    /* outputs per lane of the fused kernel: 8 for filters of 65..256 taps (40 % fewer LDS
     * reads per output; with the tap-outer window loop no SGPR spills: 127 taps 0.374 ->
     * 0.348 ms, 255 taps 0.52 -> 0.48 ms), 4 for the short first stages of a cascade */
    if (!getenv("PDDC_FIR8_R"))
        p->R = p->st[0].ntb >= 16 ? 8 : 4;
    if (const char *e = getenv("PDDC_FIR8_NT"))
        if (fir8_nt_supported(p->st[0].ntb, p->R, atoi(e)))
            p->NT = atoi(e);
    compute_lo_steps(p);
    hipError_t e = hipSuccess;
    for (int i = 0; i < nstages && e == hipSuccess; ++i) {
        Stage &s = p->st[i];
        s.hist_elem = (i == 0 && (stage0_fused(p) || stage0_packed_generic(p))) ? PDDC_PACKED_BYTES : 8;
        for (int b = 0; b < 2 && e == hipSuccess; ++b)
            e = hipMalloc(&s.d_hist[b], (size_t)s.hist * (size_t)s.hist_elem + 64);
    }
    if (e != hipSuccess) {
        pddc_pipeline_destroy(p);
        return fail(PDDC_ENOMEM, "hipMalloc history: %s", hipGetErrorString(e));
    }
    e = hipMalloc((void **)&p->d_sched, 64);
    if (e != hipSuccess) {
        pddc_pipeline_destroy(p);
        return fail(PDDC_ENOMEM, "hipMalloc scheduler words: %s", hipGetErrorString(e));
    }
    e = hipStreamCreateWithFlags(&p->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        pddc_pipeline_destroy(p);
        return fail(PDDC_EHIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    if ((rc = setup_stage3(p))) {
        pddc_pipeline_destroy(p);
        return rc;
    }
    if ((rc = pddc_pipeline_reset(p))) {
        pddc_pipeline_destroy(p);
        return rc;
    }
    if (tunables().debug.load())
        fprintf(stderr, "[pddc] pipeline %p sched %p taps_blk %p %p\n", (void *)p, (void *)p->d_sched,
                (void *)p->st[0].d_taps_blk, (void *)p->st[1].d_taps_blk);
    *out = p;
    return PDDC_OK;
}

int pddc_pipeline_destroy(pddc_pipeline *p)
{
    if (!p)
        return PDDC_OK;
    hipSetDevice(p->device);
    hipDeviceSynchronize();
    (void)leave_gang(p);
    for (int i = 0; i < PDDC_MAX_STAGES; ++i) {
        if (p->st[i].d_taps_base)
            hipFree(p->st[i].d_taps_base);
        if (p->st[i].d_taps_blk)
            hipFree(p->st[i].d_taps_blk);
        if (p->st[i].d_taps_dup_base)
            hipFree(p->st[i].d_taps_dup_base);
        if (p->st[i].d_taps_poly)
            hipFree(p->st[i].d_taps_poly);
        if (p->st[i].d_taps_seg)
            hipFree(p->st[i].d_taps_seg);
        if (p->st[i].d_taps_firp)
            hipFree(p->st[i].d_taps_firp);
        if (p->st[i].d_taps_f16)
            hipFree(p->st[i].d_taps_f16);
        if (p->st[i].d_buf && !p->st[i].buf_in_ws)
            hipFree(p->st[i].d_buf);
        if (p->st[i].d_buf_alt && !p->st[i].buf_in_ws)
            hipFree(p->st[i].d_buf_alt);
        for (int b = 0; b < 2; ++b)
            if (p->st[i].d_hist[b])
                hipFree(p->st[i].d_hist[b]);
    }
    for (auto &sl : p->slot) {
        if (sl.d_in)
            hipFree(sl.d_in);
        if (sl.d_out)
            hipFree(sl.d_out);
        if (sl.ev_in)
            hipEventDestroy(sl.ev_in);
        if (sl.ev_comp)
            hipEventDestroy(sl.ev_comp);
        if (sl.ev_out)
            hipEventDestroy(sl.ev_out);
    }
    if (p->s_in)
        hipStreamDestroy(p->s_in);
    if (p->s_out)
        hipStreamDestroy(p->s_out);
    if (p->d_fout)
        hipFree(p->d_fout);
    if (p->d_hist_f32)
        hipFree(p->d_hist_f32);
    for (auto &e : p->ev_pool) {
        hipEventDestroy(e.first);
        hipEventDestroy(e.second);
    }
    for (auto &e : p->i8x.d10)
        for (auto &b : e.buf) {
            if (b.d)
                hipFree(b.d);
            if (b.h)
                hipHostFree(b.h);
            if (b.left)
                hipEventDestroy(b.left);
        }
    for (auto &sl : p->i8x.slot) {
        if (sl.d)
            hipFree(sl.d);
        if (sl.h)
            hipHostFree(sl.h);
        if (sl.left)
            hipEventDestroy(sl.left);
    }
    if (p->d_sched)
        hipFree(p->d_sched);
    if (p->d_seam)
        hipFree(p->d_seam);
    if (p->d_flags)
        hipFree(p->d_flags);
    if (p->own_stream)
        hipStreamDestroy(p->own_stream);
    delete p;
    return PDDC_OK;
}

int pddc_pipeline_reset(pddc_pipeline *p)
{
    if (!p)
        return fail(PDDC_EINVAL, "null pipeline");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipDeviceSynchronize());
    p->n0 = 0;
    p->phase_off = 0;
    p->freg_applied = p->freg;
    compute_lo_steps(p);
    p->fresh = true;
    p->carry_pending = false;                         /* a reset stream has no tail to finish */
    p->ov_parity = 0;
    p->segs.assign(1, pddc_pipeline::WordSeg{ 0, p->freg, 0u });      /* samples before the start are zeros */
    HIP_TRY(hipMemset(p->d_sched, 0, 64));
    if (p->d_flags)
        HIP_TRY(hipMemset(p->d_flags, 0, sizeof(unsigned) * (size_t)fir8_fused3_max_chunks()));
    for (int i = 0; i < p->nstages; ++i) {
        Stage &s = p->st[i];
        s.consumed = 0;
        s.cur = 0;
        for (int b = 0; b < 2; ++b)
            HIP_TRY(hipMemset(s.d_hist[b], 0, (size_t)s.hist * (size_t)s.hist_elem + 64));
    }
    return PDDC_OK;
}

int pddc_pipeline_set_freg(pddc_pipeline *p, uint32_t freg)
{
    if (!p)
        return fail(PDDC_EINVAL, "null pipeline");
    if (freg != p->freg) {
        if (p->fresh) {
            /* before the first batch: the stream starts with this word, offset 0 */
            p->freg_applied = freg;
            p->segs.assign(1, pddc_pipeline::WordSeg{ 0, freg, 0u });
        } else {
            /* takes effect at sample n0 (the next one to be processed), phase-continuous there */
            p->phase_off += (uint32_t)p->n0 * (p->freg - freg);
            if (!p->segs.empty() && p->segs.back().n_begin == (long long)p->n0)
                p->segs.back() = pddc_pipeline::WordSeg{ (long long)p->n0, freg, p->phase_off };
            else
                p->segs.push_back(pddc_pipeline::WordSeg{ (long long)p->n0, freg, p->phase_off });
        }
        p->freg = freg;
        compute_lo_steps(p);
    }
    return PDDC_OK;
}

uint32_t pddc_pipeline_get_phase_offset(const pddc_pipeline *p) { return p ? p->phase_off : 0; }

/* kernel selection by API state (not by environment): name -> field */
static int *option_field(pddc_pipeline *p, const char *name)
{
    if (!name)
        return nullptr;
    const struct {
        const char *n;
        int *f;
    } tab[] = { { "no_i8", &p->opt.no_i8 },       { "i8x", &p->opt.i8x },
                { "i8x_pair", &p->opt.i8x_pair }, { "i8x_plain", &p->opt.i8x_plain },   { "i8x_blocks", &p->opt.i8x_blocks },
                { "i8x_chunk", &p->opt.i8x_chunk },
                { "i8x_layout", &p->opt.i8x_layout },
                { "i8x_pair_max_log2", &p->opt.i8x_pair_max_log2 },
                { "no_fuse2", &p->opt.no_fuse2 }, { "fuse3", &p->opt.fuse3 } };
    for (const auto &t : tab)
        if (!strcmp(t.n, name))
            return t.f;
    return nullptr;
}

int pddc_pipeline_set_option(pddc_pipeline *p, const char *name, int value)
{
    if (!p)
        return fail(PDDC_EINVAL, "null pipeline");
    if (p->carry_pending)
        return fail(PDDC_ESTATE, "overlap mode holds a tail back: pddc_pipeline_fence(p, stream) first");
    int *f = option_field(p, name);
    if (!f)
        return fail(PDDC_EINVAL, "unknown option '%s'", name ? name : "(null)");
    *f = value;
    return PDDC_OK;
}

int pddc_set_tunable(const char *name, int value)
{
    if (!set_tunable(name, value))
        return fail(PDDC_EINVAL, "unknown tunable '%s'", name ? name : "(null)");
    return PDDC_OK;
}

int pddc_get_tunable(const char *name, int *value)
{
    if (!value || !get_tunable(name, value))
        return fail(PDDC_EINVAL, "unknown tunable '%s'", name ? name : "(null)");
    return PDDC_OK;
}

int pddc_pipeline_get_option(const pddc_pipeline *p, const char *name, int *value)
{
    if (!p || !value)
        return fail(PDDC_EINVAL, "null argument");
    int *f = option_field(const_cast<pddc_pipeline *>(p), name);
    if (!f)
        return fail(PDDC_EINVAL, "unknown option '%s'", name ? name : "(null)");
    *value = *f;
    return PDDC_OK;
}

int pddc_pipeline_set_center_freq(pddc_pipeline *p, double hz)
{
    if (!p)
        return fail(PDDC_EINVAL, "null pipeline");
    /* same range check as perseus-sdr.c:575 */
    if (hz < 0.0 || hz > PDDC_ADC_CLK_HZ / 2)
        return fail(PDDC_EINVAL, "center frequency %.3f not in [0, %.0f]", hz, PDDC_ADC_CLK_HZ / 2);
    return pddc_pipeline_set_freg(p, pddc_nco_freg(hz, PDDC_ADC_CLK_HZ));
}

uint32_t pddc_pipeline_get_freg(const pddc_pipeline *p) { return p ? p->freg : 0; }

int pddc_pipeline_seek(pddc_pipeline *p, uint64_t abs_sample)
{
    if (!p)
        return fail(PDDC_EINVAL, "null pipeline");
    /* every stage must land on an output boundary, and the rational stages on a phase-0 one */
    unsigned long long pos = abs_sample;
    unsigned long long at[PDDC_MAX_STAGES];
    for (int i = 0; i < p->nstages; ++i) {
        const Stage &s = p->st[i];
        at[i] = pos;
        if (pos % (unsigned long long)s.decim)
            return fail(PDDC_EINVAL, "position %llu is not on an output boundary of stage %d",
                        (unsigned long long)abs_sample, i);
        pos = pos / (unsigned long long)s.decim * (unsigned long long)(s.interp > 1 ? s.interp : 1);
    }
    if (abs_sample % PDDC_INPUT_GRANULE)
        return fail(PDDC_EINVAL, "position must be a multiple of %d", PDDC_INPUT_GRANULE);
    int rc = pddc_pipeline_reset(p);
    if (rc)
        return rc;
    p->n0 = abs_sample;
    for (int i = 0; i < p->nstages; ++i)
        p->st[i].consumed = at[i];
    return PDDC_OK;
}

int pddc_pipeline_total_decim(const pddc_pipeline *p)
{
    if (!p)
        return fail(PDDC_EINVAL, "null pipeline");
    /* rounded overall rate ratio in/out (exact for integer plans) */
    double d = 1.0;
    for (int i = 0; i < p->nstages; ++i)
        d *= (double)p->st[i].decim / (double)p->st[i].interp;
    return (int)(d + 0.5);
}

int pddc_pipeline_set_taps(pddc_pipeline *p, int stage, const float *taps, int ntaps)
{
    if (!p || !taps)
        return fail(PDDC_EINVAL, "null argument");
    if (stage < 0 || stage >= p->nstages)
        return fail(PDDC_EINVAL, "stage %d out of range", stage);
    if (p->carry_pending)
        return fail(PDDC_ESTATE, "overlap mode holds a tail back: pddc_pipeline_fence(p, stream) first");
    Stage &s = p->st[stage];
    /* the history length is fixed at create time; a new tap set must fit it */
    const int need_hist = s.interp > 1 ? (ntaps + s.interp - 1) / s.interp : ntaps - 1;
    if (ntaps < 1 || need_hist > s.hist || (s.ntb && ntaps > 8 * s.ntb))
        return fail(PDDC_EINVAL, "ntaps %d does not fit the stage geometry (history %d)", ntaps, s.hist);
    if (s.interp <= 1 && !fir_generic_supported(s.decim, ntaps))
        return fail(PDDC_EINVAL, "decimate-by-%d with %d taps does not fit the LDS", s.decim, ntaps);
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipDeviceSynchronize());
    const int keep_ntb = s.ntb;
    s.ntaps = ntaps;
    s.taps.assign(taps, taps + ntaps);
    if (p->flags & PDDC_F_TAPS_FP16)
        for (float &v : s.taps)
            v = round_to_half(v);
    int rc = upload_taps(p, stage);
    if (rc)
        return rc;
    if (keep_ntb && s.ntb != keep_ntb) {
        /* keep the tile geometry (history length) chosen at create time */
        s.ntb = keep_ntb;
        /* stage 0's block table is only ever used by the packed-input kernel, which leaves
         * the unpack scale to the taps */
        const float tap_scale = stage == 0 ? kFir8PackedTapScale : 1.0f;
        std::vector<float> blk((size_t)s.ntb * 8, 0.0f);
        for (int j = 0; j < s.ntb; ++j)
            for (int e = 0; e < 8; ++e) {
                const int k = 8 * j + 7 - e;
                blk[(size_t)j * 8 + e] = k < s.ntaps ? s.taps[k] * tap_scale : 0.0f;
            }
        hipFree(s.d_taps_blk);
        s.d_taps_blk = nullptr;
        HIP_TRY(hipMalloc(&s.d_taps_blk, sizeof(float) * blk.size()));
        HIP_TRY(hipMemcpy(s.d_taps_blk, blk.data(), sizeof(float) * blk.size(), hipMemcpyHostToDevice));
    }
    if (stage <= 2)
        return setup_stage3(p);
    return PDDC_OK;
}

/* outputs produced by a stage that has consumed `consumed` inputs and now
 * receives n more: outputs m with consumed <= m*D < consumed+n              */
static void stage_outputs(unsigned long long consumed, size_t n, int D, int L, size_t *first_off,
                          unsigned long long *m0, size_t *n_out)
{
    if (L > 1) {
        /* rational: output m needs input floor(m*D/L); outputs with consumed <= that < consumed+n */
        const unsigned long long a = (consumed * (unsigned long long)L + (unsigned long long)D - 1) / (unsigned long long)D;
        const unsigned long long b = ((consumed + n) * (unsigned long long)L + (unsigned long long)D - 1) / (unsigned long long)D;
        *first_off = 0;
        *m0 = a;
        *n_out = (size_t)(b - a);
        return;
    }
    const unsigned long long mm = (consumed + (unsigned long long)D - 1) / (unsigned long long)D;
    const unsigned long long off = mm * (unsigned long long)D - consumed;   /* 0..D-1 */
    *first_off = (size_t)off;
    *m0 = mm;
    *n_out = n > off ? (size_t)((n - off - 1) / (size_t)D + 1) : 0;
}

/* outputs the NEXT process() of nsamples will produce, from the stream position alone */
static size_t predict_outputs(const pddc_pipeline *p, size_t nsamples)
{
    size_t n = nsamples, off, nout;
    unsigned long long m0;
    for (int i = 0; i < p->nstages; ++i) {
        stage_outputs(p->st[i].consumed, n, p->st[i].decim, p->st[i].interp, &off, &m0, &nout);
        n = nout;
    }
    return n;
}

size_t pddc_pipeline_next_output(const pddc_pipeline *p, size_t nsamples_in)
{
    return p ? predict_outputs(p, nsamples_in) : 0;
}

size_t pddc_pipeline_max_output(const pddc_pipeline *p, size_t n)
{
    if (!p)
        return 0;
    for (int i = 0; i < p->nstages; ++i)
        n = (n * (size_t)p->st[i].interp + (size_t)p->st[i].decim - 1) / (size_t)p->st[i].decim + (p->st[i].interp > 1 ? 1 : 0);
    return n;
}

static bool stage0_fused(const pddc_pipeline *p)
{
    return p->st[0].ntb != 0 && !(p->flags & PDDC_F_NO_FAST) && fir8_supported(p->st[0].ntb, p->R);
}

int pddc_pipeline_uses_fused(const pddc_pipeline *p) { return p && stage0_fused(p) ? 1 : 0; }

/* how many tuning words do stage 0's history window [n0 - H, n0) and the batch behind it see?  1: one word, one offset */
static size_t words_in_window(const pddc_pipeline *p)
{
    const long long w0 = (long long)p->n0 - (long long)p->st[0].hist;
    size_t first = 0;
    while (first + 1 < p->segs.size() && p->segs[first + 1].n_begin <= w0)
        ++first;
    return p->segs.size() - first;
}

/* Which kernel runs the decimate-by-8 first stage of this batch: 0 the vector kernel (k_fir8), 2 k_fir_i8x (the wire bytes
 * on the int8 matrix cores; tuned: the NCO folded into the taps -- every first stage of 1..256 taps whose history window
 * was mixed with the word in force: the one batch behind a retune goes through k_fir8, which re-mixes its packed history
 * with the old word; the two share the stream state).  (1 was round 3's k_fir_i8, retired in round 5: its binary16-stored
 * taps are the plain form's second way to get its operand.)                                                        */
static int stage0_i8_kind_raw(const pddc_pipeline *p, size_t nsamples)
{
    const Stage &s0 = p->st[0];
    if (!stage0_fused(p) || p->opt.no_i8 || nsamples < (size_t)s0.hist || s0.ntaps > s0.hist)
        return 0;
    if (p->flags & PDDC_F_MIX)
        return p->opt.i8x && s0.i8x_ok && fir_i8x_supported(s0.hist, true, false) && words_in_window(p) == 1 ? 2 : 0;
    /* untuned: the plain form -- taps from the host's table, or (PDDC_F_TAPS_FP16) binary16 values the matrix waves
     * quantise themselves; a pipeline with that flag whose taps have no int8 form keeps the vector kernel */
    if (p->opt.i8x_plain && s0.i8x_ok && fir_i8x_supported(s0.hist, false, false) &&
        (!(p->flags & PDDC_F_TAPS_FP16) || s0.d_taps_f16 != nullptr))
        return 2;
    return 0;
}

/* can stages 0 and 1 run as k_fir_i8x's fused pair (given that stage 0 runs on it)?  stage 1 a plain decimate-by-8 of <= 64
 * taps, whole tiles of 8192 samples, batches up to 2^i8x_pair_max_log2 (beyond that k_fir8's pair streams better) */
static bool i8x_pair_ok(const pddc_pipeline *p, size_t nsamples)
{
    if (p->nstages < 2 || !p->opt.i8x_pair || p->opt.no_fuse2)
        return false;
    const Stage &s0 = p->st[0], &s1 = p->st[1];
    if (s1.decim != 8 || s1.interp != 1 || s1.ntb != 8 || s1.hist != 64 || s1.ntaps > 64)
        return false;
    if (!fir_i8x_supported(s0.hist, (p->flags & PDDC_F_MIX) != 0, true))
        return false;
    if (p->opt.i8x_pair_max_log2 < 40 && nsamples > ((size_t)1 << (p->opt.i8x_pair_max_log2 < 0 ? 0 : p->opt.i8x_pair_max_log2)))
        return false;
    return nsamples > 0 && nsamples % 8192 == 0 && s0.consumed % 8 == 0 && s1.consumed % 8 == 0;
}

static bool stages01_fusable(const pddc_pipeline *p, size_t nsamples);
/* ... with one more rule for cascades: where this batch cannot take k_fir_i8x's fused pair (not whole 8192-sample tiles, or
 * larger than the pair is good for) but CAN take k_fir8's (tiles of 4096), the vector pair wins over matrix-core stage 0 +
 * a second-stage kernel of its own */
static int stage0_i8_kind(const pddc_pipeline *p, size_t nsamples)
{
    const int k = stage0_i8_kind_raw(p, nsamples);
    if (k == 2 && p->nstages >= 2 && !i8x_pair_ok(p, nsamples) && stages01_fusable(p, nsamples))
        return 0;
    /* (Measured and not added: in overlap mode a two-stage plan's tail -- the /5 of 2 MS/s, the /10 of 1 MS/s -- could ride
     * in k_fir8's launch, which k_fir_i8x, one block of twelve waves per CU, cannot offer.  2^28 samples, same process,
     * each twice: 8 * 5 vector + carried tail 0.405 / 0.457 ms, k_fir_i8x + tail in line 0.425 / 0.408; 8 * 10 0.452 / 0.407
     * against 0.4255 / 0.4259: no winner, and the matrix-core path is the steadier one.  tools/plan_rates.py --overlap) */
    return k;
}

/* the tuned decimate-by-10 first stage (the 1.6 MS/s plan's) on k_fir_i8x's paired-rows form: up to 57 taps (the taps are
 * delayed by up to 7 samples to put a batch's first window on a multiple of 8), one tuning word in the history window */
static bool stage0_packed_generic(const pddc_pipeline *p);
static bool i8x_d10_ok(const pddc_pipeline *p)
{
    const Stage &s0 = p->st[0];
    return stage0_packed_generic(p) && (p->flags & PDDC_F_MIX) && p->opt.i8x && !p->opt.no_i8 && s0.decim == 10 && s0.interp == 1 &&
           s0.ntaps + 7 <= kFirI8xD10Hist && s0.hist >= 8 && s0.hist % 8 == 0 && s0.hist <= kFirI8xD10Hist && s0.i8x_ok &&
           words_in_window(p) == 1;
}

int pddc_pipeline_stage0_on_i8(const pddc_pipeline *p, size_t nsamples)
{
    return !p ? 0 : i8x_d10_ok(p) && nsamples >= 8 ? 2 : stage0_i8_kind(p, nsamples);
}

/* a first stage that is a plain decimator but not the fused decimate-by-8 (e.g. the /10 of the
 * 1.6 MS/s plan): the generic kernel reads the packed samples itself (unpack and mix while it
 * stages), so no float2 intermediate is written.  Its history is then kept packed too.        */
static bool stage0_packed_generic(const pddc_pipeline *p)
{
    return p->st[0].interp == 1 && !stage0_fused(p) && !(p->flags & PDDC_F_NO_FAST);
}

int pddc_pipeline_stage0_reads_packed(const pddc_pipeline *p)
{
    return p && (stage0_fused(p) || stage0_packed_generic(p)) ? 1 : 0;
}

/* stages 0+1 run as one kernel when: stage 0 is the fused decimate-by-8, stage 1
 * is a plain decimate-by-8 with <= 64 taps, the batch is whole tiles, and both
 * stages sit on an 8-sample phase boundary */
static bool stages01_fusable(const pddc_pipeline *p, size_t nsamples)
{
    if (p->nstages < 2 || !stage0_fused(p) || p->opt.no_fuse2)
        return false;
    const Stage &s0 = p->st[0], &s1 = p->st[1];
    if (s1.decim != 8 || s1.interp != 1 || s1.ntb != 8 || s1.hist != 64)
        return false;
    if (!fir8_fused2_supported(s0.ntb, s1.ntb, p->R))
        return false;
    return nsamples > 0 && nsamples % (size_t)fir8_tile_inputs(p->R) == 0 && s0.consumed % 8 == 0 &&
           s1.consumed % 8 == 0;
}

/* the cascade's first two stages as k_fir_i8x's fused pair: stage 0 on the matrix cores, stage 1 -- a plain decimate-by-8
 * of <= 64 taps -- on its values while they are still in LDS; whole tiles of 8192 samples */
static bool stages01_i8x(const pddc_pipeline *p, size_t nsamples)
{
    return stage0_i8_kind(p, nsamples) == 2 && i8x_pair_ok(p, nsamples);
}

/* k_fir_i8x's operands for the word in force (rebuilt and uploaded in stream order when word, taps, form or stream changed)
 * and the constant part of its argument record */
static int i8x_prepare(pddc_pipeline *p, bool mix, bool fuse2, hipStream_t s, FirI8xArgs &q)
{
    pddc_pipeline::I8x &x = p->i8x;
    const Stage &s0 = p->st[0];
    const uint32_t word = mix ? p->freg : 0u;
    if (!(x.cur >= 0 && x.freg == word && x.mix == mix && x.fuse2 == fuse2 && x.taps_ver == p->taps_ver && x.stream == s)) {
        if (x.cur >= 0) {
            if (x.stream == s) {
                HIP_TRY(hipEventRecord(x.slot[x.cur].left, s));
                x.slot[x.cur].left_valid = true;
            } else {
                /* another stream (joining or leaving a gang): whatever reads the old tables there has to be through */
                HIP_TRY(hipDeviceSynchronize());
                for (auto &sl : x.slot)
                    sl.left_valid = false;
            }
        }
        const int nx = (x.cur + 1) & 3;
        pddc_pipeline::I8xSlot &sl = x.slot[nx];
        if (!sl.d) {
            HIP_TRY(hipMalloc(&sl.d, kI8xSlotBytes));
            HIP_TRY(hipHostMalloc(&sl.h, kI8xSlotBytes, hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&sl.left, hipEventDisableTiming));
        }
        if (sl.left_valid) {
            HIP_TRY(hipEventSynchronize(sl.left));
            sl.left_valid = false;
        }
        if (fir_i8x_table_bytes(s0.hist, mix) > kI8xTaps2Off)
            return fail(PDDC_EINVAL, "k_fir_i8x: tap tables do not fit their slot");
        float sc = 0.0f, ct[2] = { 0.0f, 0.0f };
        if (!fir_i8x_build_tables(s0.taps.data(), s0.ntaps, s0.hist, mix, word, static_cast<int8_t *>(sl.h), &sc, ct))
            return fail(PDDC_EINVAL, "k_fir_i8x: the taps cannot be quantised (all zero, or not finite)");
        float *t2 = reinterpret_cast<float *>(static_cast<uint8_t *>(sl.h) + kI8xTaps2Off);
        if (fuse2)
            fir_i8x_taps2(p->st[1].taps.data(), p->st[1].ntaps, mix, word, t2);
        /* (binary16-stored taps, no NCO: the matrix waves quantise the device's binary16 array themselves -- no table goes
         * to the device; the host's serves for scale and offset constant only) */
        if (!(s0.d_taps_f16 && !mix) || fuse2)
            HIP_TRY(hipMemcpyAsync(sl.d, sl.h, kI8xSlotBytes, hipMemcpyHostToDevice, s));
        x.cur = nx;
        x.freg = word;
        x.mix = mix;
        x.fuse2 = fuse2;
        x.taps_ver = p->taps_ver;
        x.stream = s;
        x.scale = sc;
        x.ct[0] = ct[0];
        x.ct[1] = ct[1];
    }
    q.atab = x.slot[x.cur].d;
    if (s0.d_taps_f16 && !mix) {
        q.atab = nullptr;
        q.taps16 = s0.d_taps_f16;
        q.two_e = s0.i8_two_e;
    }
    q.taps2 = fuse2 ? reinterpret_cast<const float *>(static_cast<const uint8_t *>(x.slot[x.cur].d) + kI8xTaps2Off) : nullptr;
    q.scale = x.scale;
    q.ct[0] = x.ct[0];
    q.ct[1] = x.ct[1];
    q.n0 = p->n0;
    q.freg = word;
    q.phase_off = mix ? p->phase_off : 0u;
    return PDDC_OK;
}

static int i8x_d10_prepare(pddc_pipeline *p, int delay, hipStream_t s, FirI8xArgs &q)
{
    pddc_pipeline::I8x::D10 &e = p->i8x.d10[delay & 7];
    const Stage &s0 = p->st[0];
    if (!(e.valid && e.freg == p->freg && e.taps_ver == p->taps_ver && e.stream == s)) {
        const size_t nb = fir_i8x_d10_table_bytes();
        if (e.valid) {
            /* a retune or new taps: the set in use stays where it is for the launches already queued -- an event behind them --
             * and the new one goes into the other buffer */
            if (e.stream == s) {
                HIP_TRY(hipEventRecord(e.buf[e.cur].left, s));
                e.buf[e.cur].left_valid = true;
            } else {
                /* another stream (joining or leaving a gang): whatever reads the old tables there has to be through */
                HIP_TRY(hipDeviceSynchronize());
                for (auto &b : e.buf)
                    b.left_valid = false;
            }
            e.cur ^= 1;
        }
        pddc_pipeline::I8xSlot &b = e.buf[e.cur];
        if (!b.d) {
            HIP_TRY(hipMalloc(&b.d, nb));
            HIP_TRY(hipHostMalloc(&b.h, nb, hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&b.left, hipEventDisableTiming));
        }
        if (b.left_valid) {
            HIP_TRY(hipEventSynchronize(b.left));
            b.left_valid = false;
        }
        e.valid = false;
        if (!fir_i8x_d10_build_tables(s0.taps.data(), s0.ntaps, delay, p->freg, static_cast<int8_t *>(b.h), &e.scale, e.ct))
            return fail(PDDC_EINVAL, "k_fir_i8x: the taps cannot be quantised (all zero, or not finite)");
        HIP_TRY(hipMemcpyAsync(b.d, b.h, nb, hipMemcpyHostToDevice, s));
        e.freg = p->freg;
        e.taps_ver = p->taps_ver;
        e.stream = s;
        e.valid = true;
    }
    q.atab = e.buf[e.cur].d;
    q.taps2 = nullptr;
    q.scale = e.scale;
    q.ct[0] = e.ct[0];
    q.ct[1] = e.ct[1];
    q.n0 = p->n0;
    q.freg = p->freg;
    q.phase_off = p->phase_off;
    return PDDC_OK;
}

extern "C" int pddc_pipeline_uses_fused_pair(const pddc_pipeline *p, size_t nsamples)
{
    return !p ? 0 : stages01_i8x(p, nsamples) ? 2 : stages01_fusable(p, nsamples) ? 1 : 0;
}

/* The whole cascade in one kernel: behind the fused pair, stage 2 -- a plain decimator at 1/64 of the input rate --
 * runs on the pair's outputs while they are still in LDS (k_fir8<.., FUSE3>): the 1/8 B per input sample that the
 * pair wrote and the tail kernel read back, the tail's launch and the two launch gaps all go (x320: one streaming
 * pass of 6 + 8/320 bytes per input sample).  Geometry and seam buffers are prepared at create time.            */
static int setup_stage3(pddc_pipeline *p)
{
    p->s3_ok = false;
    if (p->nstages < 3 || !stage0_fused(p))
        return PDDC_OK;
    const Stage &s0 = p->st[0], &s1 = p->st[1];
    Stage &s2 = p->st[2];
    if (s1.decim != 8 || s1.interp != 1 || s1.ntb != 8 || s1.hist != 64 || s2.interp != 1 || s2.hist_elem != 8)
        return PDDC_OK;
    Fir8Stage3 q;
    q.d = s2.decim;
    q.ntaps = s2.ntaps;
    q.h = s2.hist;
    if (!fir8_fused3_geometry(s0.ntb, s1.ntb, p->R, &q))
        return PDDC_OK;
    std::vector<float> seg((size_t)q.spl * q.seglen, 0.0f);
    std::copy(s2.taps.begin(), s2.taps.begin() + s2.ntaps, seg.begin());
    if (s2.d_taps_seg)
        HIP_TRY(hipFree(s2.d_taps_seg));
    s2.d_taps_seg = nullptr;
    HIP_TRY(hipMalloc(&s2.d_taps_seg, sizeof(float) * seg.size()));
    HIP_TRY(hipMemcpy(s2.d_taps_seg, seg.data(), sizeof(float) * seg.size(), hipMemcpyHostToDevice));
    const size_t nch = (size_t)fir8_fused3_max_chunks();
    if (p->d_seam)
        HIP_TRY(hipFree(p->d_seam));
    p->d_seam = nullptr;
    HIP_TRY(hipMalloc(&p->d_seam, nch * (size_t)q.seam_stride));
    if (!p->d_flags) {
        HIP_TRY(hipMalloc((void **)&p->d_flags, sizeof(unsigned) * nch));
        HIP_TRY(hipMemset(p->d_flags, 0, sizeof(unsigned) * nch));
    }
    p->s3 = q;
    p->s3_ok = true;
    return PDDC_OK;
}

/* Opt-in (PDDC_FUSE3=1): measured on MI355X the one-kernel cascade is SLOWER than the fused pair followed by the tail
 * kernel -- x320 at 2^28 samples 0.323 against 0.287 + 0.027 ms, no gain at 2^20..2^22 either (profiles/r03/
 * c_fused_cascade_*.txt) -- because k_fir8 is bound by each block's own dependency chain, and work added to the same
 * waves lengthens it one for one; the tail costs less on OTHER waves (pddc_pipeline_set_overlap).                  */
static bool stages012_fusable(const pddc_pipeline *p, size_t nsamples)
{
    return p->s3_ok && p->nstages >= 3 && p->opt.fuse3 == 1 && stages01_fusable(p, nsamples);
}

extern "C" int pddc_pipeline_uses_fused_cascade(const pddc_pipeline *p, size_t nsamples)
{
    return p && stages012_fusable(p, nsamples) ? 1 : 0;
}

/* Waits for everything queued on `stream` and reports a kernel-side failure (the fused cascade's bounded wait for a
 * neighbouring chunk gave up: PDDC_EHIP, sticky until reset).  Tests and bench.py call it after their runs.        */
extern "C" int pddc_pipeline_check(pddc_pipeline *p, void *stream)
{
    if (!p)
        return fail(PDDC_EINVAL, "null pipeline");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    unsigned w[4] = { 0, 0, 0, 0 };
    HIP_TRY(hipMemcpy(w, p->d_sched, sizeof(w), hipMemcpyDeviceToHost));
    if (w[2] != 0)
        return fail(PDDC_EHIP, "fused cascade: a block gave up waiting for the chunk in front of it (code %u)", w[2]);
    return PDDC_OK;
}

/* the stage's input buffer, grown when a batch is larger than any seen before (synchronising).  Plain allocations: WHERE
 * a buffer lies in HBM matters to the kernel that writes it (NOTEBOOK.md rounds 1-3 5 (o)-(r)), but a search for a good place is the
 * host's decision and never happens inside process(): pddc_pipeline_place_buffers, or pddc_pipeline_set_workspace.  */
static int ensure_buf(Stage &s, size_t need, const void * = nullptr, size_t = 0)
{
    if (s.d_buf && s.buf_cap >= need)
        return PDDC_OK;
    if (s.buf_in_ws)
        return fail(PDDC_ECAPACITY, "batch needs %zu samples of stage buffer, the workspace was sized for %zu", need,
                    s.buf_cap);
    const size_t cap = need + need / 4 + 64;
    HIP_TRY(hipDeviceSynchronize());
    if (s.d_buf)
        HIP_TRY(hipFree(s.d_buf));
    s.d_buf = nullptr;
    s.buf_cap = 0;
    HIP_TRY(hipMalloc(&s.d_buf, sizeof(float) * 2 * cap));
    s.buf_cap = cap;
    if (tunables().debug.load())
        fprintf(stderr, "[pddc] stage buffer %p (%zu samples) hist %p %p\n", (void *)s.d_buf, cap, s.d_hist[0],
                s.d_hist[1]);
    return PDDC_OK;
}

/* samples of stage i's input buffer (i >= 1) for batches of up to max_nsamples: every stage may emit one output more
 * than the ratio says, depending on where the batch begins in its decimation phase                                  */
static size_t ws_stage_samples(const pddc_pipeline *p, int i, size_t max_nsamples)
{
    size_t n = max_nsamples;
    for (int k = 0; k < i; ++k)
        n = (n * (size_t)p->st[k].interp) / (size_t)p->st[k].decim + 2;
    return n + 8 + 64;
}

extern "C" size_t pddc_pipeline_workspace_size(const pddc_pipeline *p, size_t max_nsamples)
{
    if (!p)
        return 0;
    size_t total = 0;
    for (int i = 1; i < p->nstages; ++i)
        total += ((ws_stage_samples(p, i, max_nsamples) * 8 + 255) & ~(size_t)255) *
                 ((p->overlap && i == p->nstages - 1) ? 2 : 1);
    return total;
}

extern "C" int pddc_pipeline_set_workspace(pddc_pipeline *p, void *d_ws, size_t nbytes, size_t max_nsamples)
{
    if (p && p->carry_pending)
        return fail(PDDC_ESTATE, "overlap mode holds a tail back that reads the present buffers: pddc_pipeline_fence(p, stream) first");
    if (!p)
        return fail(PDDC_EINVAL, "null pipeline");
    if (d_ws && (((uintptr_t)d_ws & 255) || nbytes < pddc_pipeline_workspace_size(p, max_nsamples)))
        return fail(PDDC_EINVAL, "workspace must be 256-byte aligned and hold pddc_pipeline_workspace_size() = %zu bytes",
                    pddc_pipeline_workspace_size(p, max_nsamples));
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipDeviceSynchronize());                 /* nothing in flight may still use the old buffers */
    uint8_t *at = static_cast<uint8_t *>(d_ws);
    for (int i = 1; i < p->nstages; ++i) {
        Stage &s = p->st[i];
        if (s.d_buf && !s.buf_in_ws)
            HIP_TRY(hipFree(s.d_buf));
        if (s.d_buf_alt && !s.buf_in_ws)
            HIP_TRY(hipFree(s.d_buf_alt));
        s.d_buf = s.d_buf_alt = nullptr;
        s.buf_cap = s.buf_alt_cap = 0;
        s.buf_in_ws = false;
        if (d_ws) {
            const size_t cap = ws_stage_samples(p, i, max_nsamples);
            s.d_buf = reinterpret_cast<float *>(at);
            s.buf_cap = cap;
            s.buf_in_ws = true;
            at += (cap * 8 + 255) & ~(size_t)255;
            if (p->overlap && i == p->nstages - 1) {
                s.d_buf_alt = reinterpret_cast<float *>(at);
                s.buf_alt_cap = cap;
                at += (cap * 8 + 255) & ~(size_t)255;
            }
        }
    }
    return PDDC_OK;
}

/* Explicit placement of the pipeline's own inter-stage buffers for batches of up to `max_nsamples` samples that will
 * be read from `d_packed`: every buffer a kernel streams at least 32 MiB per batch into is allocated through the
 * candidate walk of pddc_malloc_apart against the buffer that kernel reads (the packed batch for the first one).  Probes
 * run on `stream`; candidates and spacers take at most half of the free memory and are freed again; a probe that fails
 * leaves a plain allocation.  Takes about a second per buffer, once.  (Until round 3 process() did this by itself when a
 * large batch first arrived -- a search inside the asynchronous hot-path call, on the NULL stream, with up to 216 GiB of
 * transient allocations; now it only happens where the host asks for it.)                                       */
int pddc_pipeline_place_buffers(pddc_pipeline *p, const void *d_packed, size_t max_nsamples, void *stream)
{
    if (!p || !d_packed || max_nsamples == 0)
        return fail(PDDC_EINVAL, "bad argument");
    if (p->carry_pending)
        return fail(PDDC_ESTATE, "overlap mode holds a tail back: pddc_pipeline_fence(p, stream) first");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipDeviceSynchronize());
    const void *src = d_packed;
    size_t src_bytes = max_nsamples * 6;
    for (int i = 1; i < p->nstages; ++i) {
        Stage &s = p->st[i];
        if (s.buf_in_ws)
            return fail(PDDC_ESTATE, "the inter-stage buffers come from the caller's workspace (pddc_pipeline_set_workspace)");
        const size_t cap = ws_stage_samples(p, i, max_nsamples);
        /* with the fused pair stage 1's buffer is never written: the pair writes stage 2's, reading the packed batch */
        const bool skipped = i == 1 && p->nstages >= 2 && stages01_fusable(p, max_nsamples - max_nsamples % (size_t)fir8_tile_inputs(p->R));
        if (skipped)
            continue;
        if (cap * 8 >= ((size_t)32 << 20) && src_bytes >= ((size_t)64 << 20)) {
            void *ptr = nullptr;
            float fast = 0.0f, slow = 0.0f;
            int rc = malloc_apart_impl(&ptr, cap * 8, src, src_bytes, 8, &fast, &slow, (size_t)1 << 20, (hipStream_t)stream);
            if (rc)
                return rc;
            if (s.d_buf)
                HIP_TRY(hipFree(s.d_buf));
            s.d_buf = static_cast<float *>(ptr);
            s.buf_cap = cap;
            if (tunables().debug.load())
                fprintf(stderr, "[pddc] stage %d buffer placed: probe %.3f ms (slowest candidate %.3f)\n", i, fast, slow);
        }
        src = s.d_buf;
        src_bytes = cap * 8;
        if (!src)
            break;
    }
    return PDDC_OK;
}

/* Overlap mode.  A cascade behind the fused pair is kernels in a row, each waiting for the one before: the pair (0.29 ms
 * at 2^28 samples), a launch gap, the tail (23 us at a fifth of the HBM rate: it has 1/64 of the samples and a tenth
 * of the arithmetic), another gap -- 11 % of the x320 step (profiles/r02/k_trace_c320.txt).  What was tried first
 * (profiles/r03/): the tail INSIDE the pair's kernel (stages012_fusable above: slower, that kernel is bound by every
 * block's own dependency chain); the tail on a side stream behind an event (every cross-stream event put 18 us between
 * two pairs on the main stream, and the two grids, ready at the same instant, were dealt out interleaved: the pair took
 * 0.338 instead of 0.287 ms).  What works needs neither stream nor event: the tail of batch k becomes part of the
 * LAUNCH of batch k+1 -- extra thread blocks behind the pair's persistent ones in the same grid (Fir8Args::tail), on
 * waves the pair leaves idle -- and pddc_pipeline_fence launches the one tail that is left at the end.          */
int pddc_pipeline_set_overlap(pddc_pipeline *p, int enable)
{
    if (!p)
        return fail(PDDC_EINVAL, "null pipeline");
    if (p->carry_pending)
        return fail(PDDC_ESTATE, "a tail is still to be launched: pddc_pipeline_fence first");
    p->overlap = enable != 0;
    p->ov_parity = 0;
    return PDDC_OK;
}

/* launches, on `stream`, the tail that overlap mode still holds back (nothing to do otherwise): behind it every output
 * of every process() so far is complete in stream order                                                        */
int pddc_pipeline_fence(pddc_pipeline *p, void *stream)
{
    if (!p)
        return fail(PDDC_EINVAL, "null pipeline");
    if (p->carry_pending) {
        HIP_TRY(hipSetDevice(p->device));
        HIP_TRY(launch_gen_tail(p->carry_tail, (hipStream_t)stream));
        p->carry_pending = false;
    }
    return PDDC_OK;
}

/* stage-0 timing: a pair of events per process(), reused from a pool */
static int stage0_event(pddc_pipeline *p, hipStream_t s, bool start)
{
    if (!p->time_stage0)
        return PDDC_OK;
    if (start) {
        if (p->ev_used == p->ev_pool.size()) {
            if (p->ev_pool.size() >= 8192)
                return PDDC_OK;                       /* enough samples: stop recording */
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            p->ev_pool.emplace_back(a, b);
        }
        HIP_TRY(hipEventRecord(p->ev_pool[p->ev_used].first, s));
    } else if (p->ev_used < p->ev_pool.size()) {
        HIP_TRY(hipEventRecord(p->ev_pool[p->ev_used].second, s));
        p->ev_used++;
    }
    return PDDC_OK;
}

static void fill_fir8_args(const pddc_pipeline *p, Fir8Args &a)
{
    a.n0 = p->n0;
    a.freg = p->freg;
    a.phase_off = p->phase_off;
    a.freg_hist = p->freg_applied;
    a.sched = p->d_sched;
    for (int e = 0; e < 8; ++e) {
        a.lo_c[e] = p->lo_c[e];
        a.lo_s[e] = p->lo_s[e];
        a.lo_c_hist[e] = p->lo_c_applied[e];
        a.lo_s_hist[e] = p->lo_s_applied[e];
    }
}

static void fill_stage3_args(const pddc_pipeline *p, Fir8Args &a, float *dst, size_t off, size_t n_out, bool advance)
{
    const Stage &s2 = p->st[2];
    a.s3 = p->s3;
    a.s3.taps = s2.d_taps_seg;
    a.s3.hist = s2.d_hist[s2.cur];
    a.s3.hist_out = advance ? s2.d_hist[s2.cur ^ 1] : nullptr;
    a.s3.out = dst;
    a.s3.seam = p->d_seam;
    a.s3.flags = p->d_flags;
    a.s3.n_out = (long long)n_out;
    a.s3.off = (int)off;
}

int pddc_pipeline_process(pddc_pipeline *p, const void *d_packed, size_t nsamples, void *d_out,
                          size_t out_capacity, size_t *n_out_ret, void *stream_v)
{
    if (!p)
        return fail(PDDC_EINVAL, "null pipeline");
    if (n_out_ret)
        *n_out_ret = 0;
    if (nsamples == 0)
        return PDDC_OK;
    if (!d_packed || !d_out)
        return fail(PDDC_EINVAL, "null device pointer");
    if (nsamples % PDDC_INPUT_GRANULE)
        return fail(PDDC_EINVAL, "nsamples (%zu) must be a multiple of %d", nsamples, PDDC_INPUT_GRANULE);
    if (((uintptr_t)d_packed & 15) || ((uintptr_t)d_out & 15))
        return fail(PDDC_EINVAL, "device pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream_v;
    HIP_TRY(hipSetDevice(p->device));

    /* plan: outputs per stage */
    size_t n_in[PDDC_MAX_STAGES + 1], off[PDDC_MAX_STAGES];
    unsigned long long m0[PDDC_MAX_STAGES];
    n_in[0] = nsamples;
    for (int i = 0; i < p->nstages; ++i)
        stage_outputs(p->st[i].consumed, n_in[i], p->st[i].decim, p->st[i].interp, &off[i], &m0[i], &n_in[i + 1]);
    const size_t n_final = n_in[p->nstages];
    if (n_final > out_capacity)
        return fail(PDDC_ECAPACITY, "output capacity %zu < %zu", out_capacity, n_final);

    const bool mix = (p->flags & PDDC_F_MIX) != 0;
    int rc;
    /* destination of stage i: the next stage's input buffer, or the caller's */
    /* (reads_packed: the kernel writing it streams the packed batch -- stage 0, or the fused pair for stage 1) */
    auto stage_dst = [&](int i, float **dst, bool reads_packed) -> int {
        if (i + 1 < p->nstages) {
            (void)reads_packed;
            /* growing the buffer frees it: a tail that overlap mode still holds back may read it (its input half or its
             * output side) -- it goes out first */
            if (p->carry_pending && !p->st[i + 1].buf_in_ws && !(p->st[i + 1].d_buf && p->st[i + 1].buf_cap >= n_in[i + 1] + 8)) {
                int rf = pddc_pipeline_fence(p, s);
                if (rf)
                    return rf;
            }
            int r = ensure_buf(p->st[i + 1], n_in[i + 1] + 8);
            if (r)
                return r;
            *dst = p->st[i + 1].d_buf;
        } else if (p->flags & PDDC_F_OUT_PACKED24) {
            if (p->d_fout_cap < n_final + 8) {
                HIP_TRY(hipDeviceSynchronize());
                if (p->d_fout)
                    HIP_TRY(hipFree(p->d_fout));
                p->d_fout = nullptr;
                p->d_fout_cap = 0;
                HIP_TRY(hipMalloc(&p->d_fout, (n_final + n_final / 4 + 64) * 8));
                p->d_fout_cap = n_final + n_final / 4 + 64;
            }
            *dst = p->d_fout;
        } else {
            *dst = static_cast<float *>(d_out);
        }
        return PDDC_OK;
    };

    /* Overlap mode: the plan's LAST stage `ti` (a plain decimator behind one first-stage kernel: the fused pair, or the
     * unfused fused-/8 stage 0) is held back and rides along with the next batch's first-stage launch.  Prepares its
     * record: shape, the half of the double-buffered input the first stage writes this time, history, output.   */
    auto carry_setup = [&](int ti, float **dst_first, GenTail *mine, bool use_alt, size_t lds_cap) -> int {
        Stage &sl = p->st[ti];
        mine->H = sl.hist;
        mine->D = sl.decim;
        mine->ntaps = sl.ntaps;
        mine->n_out = (long long)n_in[ti + 1];
        if (!gen_tail_shape(mine, lds_cap, sl.d_taps_firp != nullptr))
            return 1;                                    /* does not fit: in line */
        if (use_alt) {
            if (sl.buf_in_ws) {
                if (sl.d_buf_alt == nullptr || sl.buf_alt_cap < n_in[ti] + 8)
                    return fail(PDDC_ECAPACITY, "overlap mode: the workspace was set before pddc_pipeline_set_overlap, "
                                                "or for smaller batches (it needs two halves for the last stage)");
            } else if (sl.buf_alt_cap < n_in[ti] + 8) {
                if (p->carry_pending) {                 /* (the held-back tail may read the half that is about to be freed) */
                    int rf = pddc_pipeline_fence(p, s);
                    if (rf)
                        return rf;
                }
                HIP_TRY(hipDeviceSynchronize());
                if (sl.d_buf_alt)
                    HIP_TRY(hipFree(sl.d_buf_alt));
                sl.d_buf_alt = nullptr;
                sl.buf_alt_cap = 0;
                HIP_TRY(hipMalloc(&sl.d_buf_alt, sizeof(float) * 2 * sl.buf_cap));
                sl.buf_alt_cap = sl.buf_cap;
            }
            *dst_first = sl.d_buf_alt;
        }
        float *dst_last;
        int r = stage_dst(ti, &dst_last, false);
        if (r)
            return r;
        mine->in = *dst_first;
        mine->hist = static_cast<const float *>(sl.d_hist[sl.cur]);
        mine->hist_out = static_cast<float *>(sl.d_hist[sl.cur ^ 1]);
        mine->out = dst_last;
        mine->taps = sl.d_taps_dup;
        mine->taps2 = sl.d_taps_firp;
        mine->first = (long long)off[ti];
        mine->n_batch = (long long)n_in[ti];
        return PDDC_OK;
    };
    auto carry_wanted = [&](int ti) {
        return p->overlap && p->nstages == ti + 1 && p->R == 4 && stage0_fused(p) && p->st[ti].interp == 1 &&
               !(p->flags & (PDDC_F_OUT_PACKED24 | PDDC_F_NO_FAST)) && n_in[ti + 1] > 0;
    };

    /* Stream state (which history buffer is current, inputs consumed per stage, the sample
     * counter) is committed only after every launch of the batch has been accepted: a failure
     * half way leaves the pipeline exactly where it was, and the batch can be retried.       */
    bool flip[PDDC_MAX_STAGES] = { false, false, false, false };
    int first = 0;
    int skip_from = PDDC_MAX_STAGES;        /* overlap mode: stages from here on are held back, not launched here   */
    /* which tuning words does stage 0's history window [n0 - H, n0) hold?  Drop the segments that
     * ended before it.  One word, or the old word all through with the new one starting exactly now:
     * the packed-history kernels handle it (freg_hist).  Anything else: the mixed-history route.  */
    bool mixed_hist = false;
    if (mix && (stage0_fused(p) || stage0_packed_generic(p))) {
        const long long w0 = (long long)p->n0 - (long long)p->st[0].hist;
        while (p->segs.size() >= 2 && p->segs[1].n_begin <= w0)
            p->segs.erase(p->segs.begin());
        const size_t k = p->segs.size();
        const bool ok = k == 1 || (k == 2 && p->segs[1].n_begin == (long long)p->n0);
        mixed_hist = !ok;
        const uint32_t hist_word = p->segs[0].freg;
        if (ok && hist_word != p->freg_applied) {      /* cannot happen while set_freg keeps both in step */
            p->freg_applied = hist_word;
            compute_lo_steps(p);
        }
    }
    /* Gang mode: this call only RECORDS the first-stage launch and the one plain decimator that may follow it; the gang
     * launches them for all its members together.  Anything else (other plan shapes, the rare routes, an empty tail)
     * answers 1 before anything has been touched, and the gang runs this member's batch as a chain of its own.   */
    GangRec *const g = p->gang_rec;
    if (g) {
        g->kind = 0;
        g->tail = GenTail{};
        const int i8k = mixed_hist ? 0 : stage0_i8_kind(p, nsamples);
        const int nfirst = (i8k == 2 ? stages01_i8x(p, nsamples) : stages01_fusable(p, nsamples)) ? 2 : 1;
        const int last = p->nstages - 1;
        const bool tail_ok = p->nstages == nfirst ||
                             (p->nstages == nfirst + 1 && p->st[last].interp == 1 && n_in[p->nstages] > 0);
        if (mixed_hist || !stage0_fused(p) || p->NT != 256 || p->overlap || p->carry_pending ||
            p->time_stage0 || p->fail_at_stage >= 0 || (p->flags & (PDDC_F_OUT_PACKED24 | PDDC_F_NO_FAST)) ||
            stages012_fusable(p, nsamples) || !tail_ok || n_in[1] == 0 || nsamples < (size_t)p->st[0].hist ||
            (i8k != 2 && !fir8_many_supported(nfirst, p->st[0].ntb, p->R)))
            return 1;
    }
    const int i8kind = mixed_hist ? 0 : stage0_i8_kind(p, nsamples);
    if (i8kind == 2 && stages01_i8x(p, nsamples) && !stages012_fusable(p, nsamples)) {
        /* stages 0 and 1 as k_fir_i8x's fused pair: stage 0 on the int8 matrix cores with the NCO in its taps, stage 1 on
         * its values while they are in LDS.  The stage behind the pair runs in line (no carried tail on this kernel). */
        Stage &s0 = p->st[0], &s1 = p->st[1];
        float *dst;
        if ((rc = pddc_pipeline_fence(p, s)))
            return rc;
        if ((rc = stage_dst(1, &dst, true)))
            return rc;
        GenTail mine;
        const bool gtail = g && p->nstages == 3;
        if (gtail && (rc = carry_setup(2, &dst, &mine, false, 160u * 1024u)))
            return rc;                                      /* (1: no shape for this tail -- nothing touched yet) */
        FirI8xArgs q;
        if ((rc = i8x_prepare(p, mix, true, s, q)))
            return rc;
        q.in = d_packed;
        q.hist = s0.d_hist[s0.cur];
        q.hist_out = s0.d_hist[s0.cur ^ 1];
        q.out = dst;
        q.n_in = (long long)nsamples;
        q.hist2 = s1.d_hist[s1.cur];
        q.hist2_out = s1.d_hist[s1.cur ^ 1];
        if ((rc = stage0_event(p, s, true)))
            return rc;
        if (g) {
            g->kind = 3;
            g->mix = mix;
            g->ax = q;
            g->hist = s0.hist;
            g->fuse2 = true;
            g->chunk = p->opt.i8x_chunk;
            g->layout = p->opt.i8x_layout;
            g->blocks = p->opt.i8x_blocks;
            if (gtail) {
                g->tail = mine;
                flip[2] = true;
            }
        } else {
            HIP_TRY(launch_fir_i8x(q, s0.hist, mix, true, s, p->opt.i8x_blocks, p->opt.i8x_chunk, p->opt.i8x_layout));
        }
        if ((rc = stage0_event(p, s, false)))
            return rc;
        flip[0] = flip[1] = true;
        first = gtail ? 3 : 2;
    } else if (!mixed_hist && stages012_fusable(p, nsamples)) {
        /* stages 0, 1 and 2 in ONE kernel: neither intermediate touches HBM */
        Stage &s0 = p->st[0], &s1 = p->st[1];
        float *dst;
        if ((rc = pddc_pipeline_fence(p, s)))
            return rc;
        if ((rc = stage_dst(2, &dst, true)))
            return rc;
        Fir8Args a;
        a.in = d_packed;
        a.hist = s0.d_hist[s0.cur];
        a.hist_out = s0.d_hist[s0.cur ^ 1];
        a.out = nullptr;
        a.taps_blk = s0.d_taps_blk;
        a.taps2_blk = s1.d_taps_blk;
        a.hist2 = s1.d_hist[s1.cur];
        a.hist2_out = s1.d_hist[s1.cur ^ 1];
        a.n_in = (long long)nsamples;
        fill_fir8_args(p, a);
        fill_stage3_args(p, a, dst, off[2], n_in[3], true);
        if ((rc = stage0_event(p, s, true)))
            return rc;
        HIP_TRY(launch_fir8_fused3(s0.ntb, p->R, mix, a, s));
        if ((rc = stage0_event(p, s, false)))
            return rc;
        flip[0] = flip[1] = flip[2] = true;
        first = 3;
    } else if (!mixed_hist && stages01_fusable(p, nsamples)) {
        /* stages 0 and 1 in ONE kernel: the 8 B/sample-at-1/8-rate intermediate
         * (1 B written + 1 B read per input sample) never touches HBM */
        Stage &s0 = p->st[0], &s1 = p->st[1];
        float *dst;
        if ((rc = stage_dst(1, &dst, true)))
            return rc;
        /* overlap mode: the one stage behind the pair is held back and rides along with the NEXT batch's pair */
        GenTail mine;
        bool ov = !g && carry_wanted(2);
        if (ov) {
            rc = carry_setup(2, &dst, &mine, p->ov_parity == 1, kCarryLdsCap);
            if (rc < 0)
                return rc;
            ov = rc == PDDC_OK;
        }
        const bool gtail = g && p->nstages == 3;
        if (gtail && (rc = carry_setup(2, &dst, &mine, false, 160u * 1024u)))
            return rc;                                      /* (1: no shape for this tail -- nothing touched yet) */
        if (!ov && (rc = pddc_pipeline_fence(p, s)))        /* in line: what is held back goes first */
            return rc;
        Fir8Args a;
        a.in = d_packed;
        a.hist = s0.d_hist[s0.cur];
        a.hist_out = s0.d_hist[s0.cur ^ 1];
        a.out = dst;
        a.taps_blk = s0.d_taps_blk;
        a.taps2_blk = s1.d_taps_blk;
        a.hist2 = s1.d_hist[s1.cur];
        a.hist2_out = s1.d_hist[s1.cur ^ 1];
        a.n_in = (long long)nsamples;
        fill_fir8_args(p, a);
        if (ov && p->carry_pending)
            a.tail = p->carry_tail;
        if ((rc = stage0_event(p, s, true)))
            return rc;
        if (g) {
            g->kind = 2;
            g->ntb = s0.ntb;
            g->R = p->R;
            g->mix = mix;
            g->a = a;
            if (gtail) {
                g->tail = mine;
                flip[2] = true;
            }
        } else {
            HIP_TRY(launch_fir8_fused2(s0.ntb, p->R, mix, a, s));
        }
        if (ov)
            p->carry_pending = false;       /* (failure atomicity: only once the launch was accepted) */
        if ((rc = stage0_event(p, s, false)))
            return rc;
        flip[0] = flip[1] = true;
        first = gtail ? 3 : 2;
        if (ov) {                           /* the previous tail went out with this launch; this batch's is held back */
            p->carry_tail = mine;
            p->carry_pending = true;
            p->ov_parity ^= 1;
            flip[2] = true;
            first = 3;
        }
    } else if (!(!mixed_hist && carry_wanted(1)) && (rc = pddc_pipeline_fence(p, s))) {
        return rc;                                          /* any other route runs the later stages in line */
    }
    for (int i = first; i < p->nstages && i < skip_from; ++i) {
        Stage &st = p->st[i];
        if (p->fail_at_stage >= 0 && p->fail_at_stage <= i) {
            p->fail_at_stage = -1;
            return fail(PDDC_EHIP, "injected failure at stage %d (pddc_pipeline_inject_failure)", i);
        }
        float *dst;
        if ((rc = stage_dst(i, &dst, i == 0)))
            return rc;
        void *h_in = st.d_hist[st.cur], *h_out = st.d_hist[st.cur ^ 1];
        const void *x = d_packed;                       /* this stage's input batch */
        bool hist_done = false;
        if (i == 0 && mixed_hist) {
            /* rare: batches shorter than the history with retunes between them.  The packed history
             * is unpacked and mixed stretch by stretch, each with the word and offset that applied to
             * it, the batch likewise with the current word, and the generic decimator runs on floats;
             * the packed history for the next call is carried as usual.                            */
            const int H = st.hist;
            if ((rc = ensure_buf(st, nsamples + 8)))
                return rc;
            if (!p->d_hist_f32)
                HIP_TRY(hipMalloc(&p->d_hist_f32, (size_t)PDDC_MAX_TAPS * 8 + 256));
            const long long w0 = (long long)p->n0 - H;
            for (size_t k = 0; k < p->segs.size(); ++k) {
                const long long a0 = std::max(w0, k == 0 ? w0 : p->segs[k].n_begin);
                const long long a1 = std::min((long long)p->n0, k + 1 < p->segs.size() ? p->segs[k + 1].n_begin
                                                                                       : (long long)p->n0);
                if (a1 <= a0)
                    continue;
                float lc[8], ls[8];
                lo_steps(p->segs[k].freg, lc, ls);
                HIP_TRY(launch_unpack24(static_cast<const uint8_t *>(h_in) + (a0 - w0) * PDDC_PACKED_BYTES, a1 - a0,
                                        p->d_hist_f32 + 2 * (a0 - w0), false, true, (unsigned long long)a0,
                                        p->segs[k].freg, p->segs[k].off, lc, ls, s));
            }
            HIP_TRY(launch_unpack24(d_packed, (long long)nsamples, st.d_buf, false, true, p->n0, p->freg, p->phase_off,
                                    p->lo_c, p->lo_s, s));
            if (n_in[1] > 0)
                HIP_TRY(launch_fir_generic(st.d_buf, p->d_hist_f32, H, (long long)off[0], (long long)n_in[1], st.decim,
                                           st.d_taps_dup, st.ntaps, dst, nullptr, (long long)nsamples, s));
            /* hist_done stays false: the packed history moves on below (x == d_packed) */
        } else if (i == 0 && stage0_fused(p)) {
            /* overlap mode, two-stage plan: stage 1 is held back and rides along with the next batch's stage 0 */
            GenTail mine;
            bool ov = !g && carry_wanted(1) && p->NT == 256 && i8kind != 2;   /* (k_fir_i8x carries no tail: in line) */
            if (ov) {
                rc = carry_setup(1, &dst, &mine, p->ov_parity == 1, kCarryLdsCap);
                if (rc < 0)
                    return rc;
                ov = rc == PDDC_OK;
            }
            const bool gtail = g && p->nstages == 2;
            if (gtail && (rc = carry_setup(1, &dst, &mine, false, 160u * 1024u)))
                return rc;
            if (!ov && (rc = pddc_pipeline_fence(p, s)))
                return rc;
            Fir8Args a;
            a.in = d_packed;
            a.hist = h_in;
            a.hist_out = nsamples >= (size_t)st.hist ? h_out : nullptr;
            a.out = dst;
            a.taps_blk = st.d_taps_blk;
            a.n_in = (long long)nsamples;
            fill_fir8_args(p, a);
            if (ov && p->carry_pending)
                a.tail = p->carry_tail;
            if ((rc = stage0_event(p, s, true)))
                return rc;
            if (g && i8kind == 2) {
                FirI8xArgs q;
                if ((rc = i8x_prepare(p, mix, false, s, q)))
                    return rc;
                q.in = d_packed;
                q.hist = h_in;
                q.hist_out = a.hist_out;
                q.out = dst;
                q.n_in = (long long)nsamples;
                g->kind = 3;
                g->mix = mix;
                g->ax = q;
                g->hist = st.hist;
                g->fuse2 = false;
                g->chunk = p->opt.i8x_chunk;
                g->layout = p->opt.i8x_layout;
                g->blocks = p->opt.i8x_blocks;
                if (gtail) {
                    g->tail = mine;
                    flip[1] = true;
                    skip_from = 1;
                }
            } else if (g) {
                g->kind = 1;
                g->ntb = st.ntb;
                g->R = p->R;
                g->mix = mix;
                g->a = a;
                if (gtail) {
                    g->tail = mine;
                    flip[1] = true;
                    skip_from = 1;
                }
            } else if (i8kind == 2 && !ov) {
                /* the NCO in the taps, the wire bytes on the int8 matrix cores (k_fir_i8x; same history as k_fir8) */
                FirI8xArgs q;
                if ((rc = i8x_prepare(p, mix, false, s, q)))
                    return rc;
                q.in = d_packed;
                q.hist = h_in;
                q.hist_out = a.hist_out;
                q.out = dst;
                q.n_in = (long long)nsamples;
                HIP_TRY(launch_fir_i8x(q, st.hist, mix, false, s, p->opt.i8x_blocks, p->opt.i8x_chunk, p->opt.i8x_layout));
            } else {
                HIP_TRY(launch_fir8(st.ntb, p->R, IN_PACKED24, mix, a, s, p->NT));
            }
            if ((rc = stage0_event(p, s, false)))
                return rc;
            hist_done = a.hist_out != nullptr;
            if (ov) {
                p->carry_tail = mine;
                p->carry_pending = true;
                p->ov_parity ^= 1;
                flip[1] = true;
                skip_from = 1;
            }
        } else if (i == 0 && stage0_packed_generic(p)) {
            if (n_in[1] > 0 && i8x_d10_ok(p)) {
                /* the matrix cores: the decimation phase of this batch (its first output's newest sample, 0 .. 9) goes into
                 * the taps as a delay, so that the windows end on a multiple of 8 samples */
                const int first = (int)off[0], delay = (8 - (first & 7)) & 7;
                FirI8xArgs q;
                if ((rc = i8x_d10_prepare(p, delay, s, q)))
                    return rc;
                q.in = d_packed;
                q.hist = h_in;
                q.hist_out = nsamples >= (size_t)st.hist ? h_out : nullptr;
                q.out = static_cast<float *>(dst);
                q.n_in = (long long)nsamples;
                q.n_out = (long long)n_in[1];
                q.in_off = first + delay;
                q.hist_len = st.hist;
                q.n0 = p->n0 + (unsigned long long)q.in_off;
                if ((rc = stage0_event(p, s, true)))
                    return rc;
                HIP_TRY(launch_fir_i8x_d10(q, s, p->opt.i8x_blocks, p->opt.i8x_chunk, p->opt.i8x_layout));
                if ((rc = stage0_event(p, s, false)))
                    return rc;
                x = d_packed;                 /* (a batch shorter than the history: the packed history moves on below) */
                hist_done = q.hist_out != nullptr;
            } else if (n_in[1] > 0) {
                if (st.d_taps_firp)           /* register-blocked kernel for /4 /5 /8 /10 */
                    HIP_TRY(launch_firp_packed(d_packed, h_in, st.hist, (long long)off[0], (long long)n_in[1], st.decim,
                                               st.d_taps_firp, st.ntaps, dst, h_out, (long long)nsamples, mix, p->n0,
                                               p->freg, p->phase_off, p->freg_applied, p->lo_c, p->lo_s, p->lo_c_applied,
                                               p->lo_s_applied, s));
                else
                    HIP_TRY(launch_fir_generic_packed(d_packed, h_in, st.hist, (long long)off[0], (long long)n_in[1],
                                                      st.decim, st.d_taps_dup, st.ntaps, dst, h_out, (long long)nsamples,
                                                      mix, p->n0, p->freg, p->phase_off, p->freg_applied, p->lo_c, p->lo_s,
                                                      p->lo_c_applied, p->lo_s_applied, s));
                hist_done = true;             /* block 0 of the kernel wrote the new (packed) history */
            }
        } else {
            if (i == 0) {
                /* generic first stage: unpack(+mix) to float2, then the generic FIR */
                if ((rc = ensure_buf(st, nsamples + 8)))
                    return rc;
                HIP_TRY(launch_unpack24(d_packed, (long long)nsamples, st.d_buf, false, mix, p->n0, p->freg,
                                        p->phase_off, p->lo_c, p->lo_s, s));
            }
            x = st.d_buf;
            const bool fast = i > 0 && st.ntb != 0 && !(p->flags & PDDC_F_NO_FAST) &&
                              fir8_supported(st.ntb, p->R) && (n_in[i] % 8 == 0) && off[i] == 0 &&
                              (st.consumed % 8 == 0) && n_in[i] > 0;
            if (fast) {
                Fir8Args a;
                a.in = x;
                a.hist = h_in;
                a.hist_out = n_in[i] >= (size_t)st.hist ? h_out : nullptr;
                a.out = dst;
                a.taps_blk = st.d_taps_blk;
                a.n_in = (long long)n_in[i];
                fill_fir8_args(p, a);
                HIP_TRY(launch_fir8(st.ntb, p->R, IN_F32C, false, a, s));
                hist_done = a.hist_out != nullptr;
            } else if (st.interp > 1) {
                if (n_in[i + 1] > 0 && st.d_taps_poly && !(p->flags & PDDC_F_NO_FAST)) {
                    HIP_TRY(launch_resample_lds(static_cast<const float *>(x), static_cast<const float *>(h_in), st.hist,
                                                st.consumed, m0[i], (long long)n_in[i + 1], st.interp, st.decim,
                                                st.d_taps_poly, st.poly_k, st.poly_kp, dst, static_cast<float *>(h_out),
                                                (long long)n_in[i], s));
                    hist_done = true;
                } else if (n_in[i + 1] > 0) {
                    HIP_TRY(launch_resample(static_cast<const float *>(x), static_cast<const float *>(h_in), st.hist,
                                            st.consumed, m0[i], (long long)n_in[i + 1], st.interp, st.decim,
                                            st.d_taps, st.ntaps, dst, s));
                }
            } else if (n_in[i + 1] > 0) {
                if (st.d_taps_firp)
                    HIP_TRY(launch_firp(IN_F32C, false, x, h_in, st.hist, (long long)off[i], (long long)n_in[i + 1], st.decim,
                                        st.d_taps_firp, st.ntaps, dst, h_out, (long long)n_in[i], nullptr, s));
                else
                    HIP_TRY(launch_fir_generic(static_cast<const float *>(x), static_cast<const float *>(h_in),
                                               st.hist, (long long)off[i], (long long)n_in[i + 1], st.decim,
                                               st.d_taps_dup, st.ntaps, dst, static_cast<float *>(h_out),
                                               (long long)n_in[i], s));
                hist_done = true;             /* block 0 of the kernel wrote the new history */
            }
        }
        if (n_in[i] > 0) {
            if (!hist_done)
                HIP_TRY(launch_hist_update(h_out, h_in, st.hist, x, (long long)n_in[i], st.hist_elem, s));
            flip[i] = true;
        }
    }
    if (p->flags & PDDC_F_OUT_PACKED24)
        HIP_TRY(launch_pack24(p->d_fout, (long long)n_final, d_out, s));
    for (int i = 0; i < p->nstages; ++i) {          /* commit */
        if (flip[i])
            p->st[i].cur ^= 1;
        p->st[i].consumed += n_in[i];
    }
    p->n0 += nsamples;
    if (p->freg_applied != p->freg) {
        p->freg_applied = p->freg;
        compute_lo_steps(p);
    }
    p->fresh = false;
    if (n_out_ret)
        *n_out_ret = n_final;
    return PDDC_OK;
}

int pddc_host_alloc(void **h_ptr, size_t nbytes)
{
    if (!h_ptr || nbytes == 0)
        return fail(PDDC_EINVAL, "bad argument");
    int rc = require_device();
    if (rc)
        return rc;
    hipError_t e = hipHostMalloc(h_ptr, nbytes, hipHostMallocDefault);
    if (e != hipSuccess)
        return fail(PDDC_ENOMEM, "hipHostMalloc(%zu): %s", nbytes, hipGetErrorString(e));
    return PDDC_OK;
}

int pddc_host_free(void *h_ptr)
{
    if (h_ptr)
        HIP_TRY(hipHostFree(h_ptr));
    return PDDC_OK;
}

/* a staging slot ready for a batch of nsamples: events, device buffers for the packed input and the final output */
static int prep_slot(pddc_pipeline *p, pddc_pipeline::HostSlot &sl, size_t nsamples)
{
    if (!sl.ev_in) {
        HIP_TRY(hipEventCreateWithFlags(&sl.ev_in, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&sl.ev_comp, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&sl.ev_out, hipEventDisableTiming));
    }
    const size_t max_out = pddc_pipeline_max_output(p, nsamples) + 1;
    if (sl.in_cap < nsamples || sl.out_cap < max_out) {
        if (sl.used)                          /* growing a slot: its last batch must be out first */
            if (sl.ev_wait)                     /* (nullptr: known complete) */
                HIP_TRY(hipEventSynchronize(sl.ev_wait));
        if (sl.in_cap < nsamples) {
            if (sl.d_in)
                HIP_TRY(hipFree(sl.d_in));
            sl.d_in = nullptr;
            sl.in_cap = 0;
            HIP_TRY(hipMalloc(&sl.d_in, nsamples * 6 + 64));
            sl.in_cap = nsamples;
        }
        if (sl.out_cap < max_out) {
            if (sl.d_out)
                HIP_TRY(hipFree(sl.d_out));
            sl.d_out = nullptr;
            sl.out_cap = 0;
            HIP_TRY(hipMalloc(&sl.d_out, max_out * 8 + 64));
            sl.out_cap = max_out;
        }
    }
    return PDDC_OK;
}

/* A pipeline's pushes are ordered by the stream they go through: its own, or a gang's.  Changing from one to the other
 * (the number of receivers that stream together changed) waits for what is still in flight on the old one.        */
/* A slot's ticket may wait for an event of the gang round it went out with.  Once the gang's stream has been synchronised
 * that batch is complete; the event itself goes on being re-recorded by rounds this pipeline is not in (or is destroyed with
 * the gang), so the slot forgets it: ev_wait == nullptr means "complete".                                            */
static void forget_gang_events(pddc_pipeline *p, const pddc_gang *g)
{
    for (auto &sl : p->slot)
        for (int i = 0; i < 4; ++i)
            if (sl.ev_wait && sl.ev_wait == g->ev[i])
                sl.ev_wait = nullptr;
}

static int leave_gang(pddc_pipeline *p)
{
    pddc_gang *g = p->gang;
    if (!g)
        return PDDC_OK;
    std::lock_guard<std::mutex> lk(g->lock);
    HIP_TRY(hipStreamSynchronize(g->stream));
    g->members.erase(std::remove(g->members.begin(), g->members.end(), p), g->members.end());
    p->gang = nullptr;
    forget_gang_events(p, g);
    return PDDC_OK;
}

/* one batch through a staging slot: input either copied from the host (h_packed) or generated
 * on the device (synthetic LCG source: no host -> device traffic at all), kernels, output D2H */
static int push_async(pddc_pipeline *p, const void *h_packed, bool synth, uint32_t seed, uint64_t byte_offset,
                      size_t nsamples, void *h_out, size_t out_capacity, size_t *n_out_ret, int *ticket)
{
    if (!p)
        return fail(PDDC_EINVAL, "null pipeline");
    if (n_out_ret)
        *n_out_ret = 0;
    if (ticket)
        *ticket = -1;
    if (nsamples == 0)
        return PDDC_OK;
    if ((!synth && !h_packed) || !h_out)
        return fail(PDDC_EINVAL, "null host pointer");
    if (nsamples % PDDC_INPUT_GRANULE)
        return fail(PDDC_EINVAL, "nsamples (%zu) must be a multiple of %d", nsamples, PDDC_INPUT_GRANULE);
    /* the caller's buffer is checked BEFORE anything is launched: a batch that does not fit is
     * refused with the stream position untouched, and can be pushed again with a larger buffer */
    {
        const size_t need = predict_outputs(p, nsamples);
        if (need > out_capacity)
            return fail(PDDC_ECAPACITY, "output capacity %zu < %zu", out_capacity, need);
    }
    HIP_TRY(hipSetDevice(p->device));
    int rc0 = leave_gang(p);                  /* the batch before may have gone out with a gang, on the gang's stream */
    if (rc0)
        return rc0;
    if (!p->s_in) {
        HIP_TRY(hipStreamCreateWithFlags(&p->s_in, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&p->s_out, hipStreamNonBlocking));
    }
    const int si = p->next_slot;
    pddc_pipeline::HostSlot &sl = p->slot[si];
    if ((rc0 = prep_slot(p, sl, nsamples)))
        return rc0;
    /* The on-device source has no host-to-device copy to overlap, and its decimated output is small: generator, kernels
     * and the copy out then all go on ONE stream.  The three-stream form below costs two cross-stream event hops per
     * batch, ~18 us each on this runtime (profiles/r03/d_side_stream_overlap_delay.txt) -- more than the kernels of a
     * 2^22-sample batch take.  Two slots still alternate, so the host fills / reads one while the GPU works on the other. */
    const bool one_stream = synth && !tunables().push_three_streams.load();
    /* ... except that from 2^24 samples on the GENERATOR of this batch goes to the side stream: it writes 6 bytes per
     * sample -- as much as the whole cascade reads -- and with two batches in flight it then runs beside the kernels of the
     * batch before (one event hop of ~18 us against 24 us of generator at 2^24, 94 us at 2^26) */
    const bool gen_aside = one_stream && nsamples >= ((size_t)1 << 24);
    hipStream_t st_in = one_stream && !gen_aside ? p->own_stream : p->s_in;
    hipStream_t st_out = one_stream ? p->own_stream : p->s_out;
    /* H2D (or the generator): the slot's input buffer is free once the kernels of its previous batch are done */
    if (sl.used && (!one_stream || gen_aside))
        HIP_TRY(hipStreamWaitEvent(st_in, sl.ev_comp, 0));
    if (synth)
        HIP_TRY(launch_synth_lcg(sl.d_in, nsamples * 6, seed, byte_offset, st_in));
    else
        HIP_TRY(hipMemcpyAsync(sl.d_in, h_packed, nsamples * 6, hipMemcpyHostToDevice, st_in));
    if (!one_stream || gen_aside) {
        HIP_TRY(hipEventRecord(sl.ev_in, st_in));
        /* kernels: after this batch has arrived and the slot's previous output has left */
        HIP_TRY(hipStreamWaitEvent(p->own_stream, sl.ev_in, 0));
        if (sl.used && !one_stream)
            HIP_TRY(hipStreamWaitEvent(p->own_stream, sl.ev_out, 0));
    }
    size_t n_out = 0;
    /* (one stream: the last stage writes into the caller's buffer itself where that is pinned memory, see direct_out) */
    float *direct = one_stream && !(p->flags & PDDC_F_OUT_PACKED24) ? direct_out(sl, h_out) : nullptr;
    int rc = pddc_pipeline_process(p, sl.d_in, nsamples, direct ? (void *)direct : (void *)sl.d_out,
                                   direct ? out_capacity : sl.out_cap, &n_out, p->own_stream);
    if (rc)
        return rc;
    if ((rc = pddc_pipeline_fence(p, p->own_stream)))     /* overlap mode: the D2H copy needs the tail's output */
        return rc;
    HIP_TRY(hipEventRecord(sl.ev_comp, p->own_stream));
    sl.used = true;
    p->next_slot = si ^ 1;
    if (!one_stream)
        HIP_TRY(hipStreamWaitEvent(st_out, sl.ev_comp, 0));
    if (n_out && !direct)
        HIP_TRY(hipMemcpyAsync(h_out, sl.d_out, n_out * ((p->flags & PDDC_F_OUT_PACKED24) ? 6 : 8),
                               hipMemcpyDeviceToHost, st_out));
    HIP_TRY(hipEventRecord(sl.ev_out, st_out));
    sl.ev_wait = sl.ev_out;
    if (n_out_ret)
        *n_out_ret = n_out;
    if (ticket)
        *ticket = si;
    return PDDC_OK;
}

int pddc_pipeline_push_host_async(pddc_pipeline *p, const void *h_packed, size_t nsamples, void *h_out,
                                  size_t out_capacity, size_t *n_out_ret, int *ticket)
{
    return push_async(p, h_packed, false, 0, 0, nsamples, h_out, out_capacity, n_out_ret, ticket);
}

int pddc_pipeline_push_synth_async(pddc_pipeline *p, uint32_t seed, uint64_t byte_offset, size_t nsamples,
                                   void *h_out, size_t out_capacity, size_t *n_out_ret, int *ticket)
{
    return push_async(p, nullptr, true, seed, byte_offset, nsamples, h_out, out_capacity, n_out_ret, ticket);
}

int pddc_pipeline_ticket_done(pddc_pipeline *p, int ticket)
{
    if (!p || ticket < 0 || ticket > 1)
        return fail(PDDC_EINVAL, "bad ticket");
    if (!p->slot[ticket].used)
        return fail(PDDC_ESTATE, "nothing was pushed on ticket %d", ticket);
    HIP_TRY(hipSetDevice(p->device));
    if (!p->slot[ticket].ev_wait)
        return 1;                             /* complete (it went out with a gang this pipeline has left since) */
    hipError_t e = hipEventQuery(p->slot[ticket].ev_wait);
    if (e == hipSuccess)
        return 1;
    if (e == hipErrorNotReady)
        return 0;
    return fail(PDDC_EHIP, "hipEventQuery: %s", hipGetErrorString(e));
}

int pddc_pipeline_wait_ticket(pddc_pipeline *p, int ticket)
{
    if (!p || ticket < 0 || ticket > 1)
        return fail(PDDC_EINVAL, "bad ticket");
    if (!p->slot[ticket].used)
        return fail(PDDC_ESTATE, "nothing was pushed on ticket %d", ticket);
    HIP_TRY(hipSetDevice(p->device));
    if (p->slot[ticket].ev_wait)
        HIP_TRY(hipEventSynchronize(p->slot[ticket].ev_wait));
    return PDDC_OK;
}

int pddc_pipeline_wait(pddc_pipeline *p)
{
    if (!p)
        return fail(PDDC_EINVAL, "null pipeline");
    HIP_TRY(hipSetDevice(p->device));
    if (p->s_in)
        HIP_TRY(hipStreamSynchronize(p->s_in));
    HIP_TRY(hipStreamSynchronize(p->own_stream));
    if (p->s_out)
        HIP_TRY(hipStreamSynchronize(p->s_out));
    if (p->gang) {
        std::lock_guard<std::mutex> lk(p->gang->lock);
        HIP_TRY(hipStreamSynchronize(p->gang->stream));
    }
    return PDDC_OK;
}

/* The final stage of a gang round (and of a push from the on-device source) writes into the caller's output buffer
 * ITSELF where that is pinned host memory (what pddc_host_alloc hands out; the device sees it through the bus): the few
 * dozen kilobytes a receiver's batch decimates to need no copy engine, and a round has no eight copies -- 7 us of stream
 * time and 6 us of host time each -- behind its kernels (8 receivers: 145 -> 221 GS/s of ADC-rate input).  Pageable
 * memory (or PDDC_GANG_COPY_OUT=1) keeps the staging buffer and the copy.                                          */
static float *direct_out(pddc_pipeline::HostSlot &sl, void *h_out)
{
    if (sl.h_out_seen != h_out) {
        sl.h_out_seen = h_out;
        sl.h_out_dev = nullptr;
        hipPointerAttribute_t at;
        if (!tunables().gang_copy_out.load() && hipPointerGetAttributes(&at, h_out) == hipSuccess &&
            at.type == hipMemoryTypeHost && at.devicePointer && at.hostPointer) {
            /* (h_out may point INTO a pinned allocation -- perseus_api.c hands out places inside one output buffer --: the
             * device's view of it is the allocation's plus the same offset, whichever of the two the runtime reports) */
            uint8_t *dv = static_cast<uint8_t *>(at.devicePointer) +
                          (static_cast<uint8_t *>(h_out) - static_cast<uint8_t *>(at.hostPointer));
            if (((uintptr_t)dv & 15) == 0)
                sl.h_out_dev = reinterpret_cast<float *>(dv);
        } else {
            (void)hipGetLastError();
        }
    }
    return sl.h_out_dev;
}

/* ---- gang: the pipelines of one GPU that stream together, one launch chain for all -------------------------------- */
int pddc_gang_create(pddc_gang **out, int device)
{
    if (!out)
        return fail(PDDC_EINVAL, "null argument");
    *out = nullptr;
    int rc = require_device();
    if (rc)
        return rc;
    HIP_TRY(hipSetDevice(device));
    pddc_gang *g = new (std::nothrow) pddc_gang;
    if (!g)
        return fail(PDDC_ENOMEM, "out of memory");
    g->device = device;
    hipError_t e = hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking);
    if (e == hipSuccess)
        e = hipStreamCreateWithFlags(&g->s_gen, hipStreamNonBlocking);
    for (int i = 0; i < 4 && e == hipSuccess; ++i) {
        e = hipEventCreateWithFlags(&g->ev[i], hipEventDisableTiming);
        if (e == hipSuccess)
            e = hipEventCreateWithFlags(&g->ev_gen[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        pddc_gang_destroy(g);
        return fail(PDDC_EHIP, "gang stream/events: %s", hipGetErrorString(e));
    }
    *out = g;
    return PDDC_OK;
}

int pddc_gang_destroy(pddc_gang *g)
{
    if (!g)
        return PDDC_OK;
    (void)hipSetDevice(g->device);
    {
        std::lock_guard<std::mutex> lk(g->lock);
        if (g->s_gen)
            (void)hipStreamSynchronize(g->s_gen);
        if (g->stream)
            (void)hipStreamSynchronize(g->stream);
        for (pddc_pipeline *p : g->members) {
            p->gang = nullptr;                /* their next push goes through their own stream again */
            forget_gang_events(p, g);         /* (both streams are idle: what their tickets waited for is complete) */
        }
        g->members.clear();
    }
    for (int i = 0; i < 4; ++i) {
        if (g->ev[i])
            (void)hipEventDestroy(g->ev[i]);
        if (g->ev_gen[i])
            (void)hipEventDestroy(g->ev_gen[i]);
    }
    if (g->s_gen)
        (void)hipStreamDestroy(g->s_gen);
    if (g->stream)
        (void)hipStreamDestroy(g->stream);
    delete g;
    return PDDC_OK;
}

static_assert(PDDC_GANG_MAX <= kFir8ManyMax, "a gang round must fit one k_fir8_many launch");

int pddc_gang_push_async(pddc_gang *g, pddc_gang_item *items, int n, size_t nsamples, int *n_ganged)
{
    if (n_ganged)
        *n_ganged = 0;
    if (!g || !items || n < 1 || n > PDDC_GANG_MAX)
        return fail(PDDC_EINVAL, "gang push: 1..%d items", PDDC_GANG_MAX);
    if (nsamples == 0 || nsamples % PDDC_INPUT_GRANULE)
        return fail(PDDC_EINVAL, "nsamples (%zu) must be a positive multiple of %d", nsamples, PDDC_INPUT_GRANULE);
    /* everything that can refuse the round is checked before anything is queued or any stream position moves */
    bool all_synth = true;
    for (int i = 0; i < n; ++i) {
        pddc_gang_item &it = items[i];
        it.n_out = 0;
        it.ticket = -1;
        if (!it.pipe || !it.h_out)
            return fail(PDDC_EINVAL, "gang item %d: null pipeline or output buffer", i);
        if (it.pipe->device != g->device)
            return fail(PDDC_EINVAL, "gang item %d: pipeline on GPU %d, gang on GPU %d", i, it.pipe->device, g->device);
        for (int j = 0; j < i; ++j)
            if (items[j].pipe == it.pipe)
                return fail(PDDC_EINVAL, "gang items %d and %d: the same pipeline", j, i);
        if (it.pipe->carry_pending)
            return fail(PDDC_ESTATE, "gang item %d: overlap mode holds a tail back (pddc_pipeline_fence first)", i);
        const size_t need = predict_outputs(it.pipe, nsamples);
        if (need > it.out_capacity)
            return fail(PDDC_ECAPACITY, "gang item %d: output capacity %zu < %zu", i, it.out_capacity, need);
        all_synth = all_synth && it.h_packed == nullptr;
    }
    HIP_TRY(hipSetDevice(g->device));
    std::lock_guard<std::mutex> lk(g->lock);
    hipStream_t s = g->stream;
    int si[PDDC_GANG_MAX];
    for (int i = 0; i < n; ++i) {
        pddc_pipeline *p = items[i].pipe;
        if (p->gang != g) {                    /* joining: what it queued on its own streams (or another gang's) first */
            int rc = leave_gang(p);
            if (!rc)
                rc = pddc_pipeline_wait(p);
            if (rc)
                return rc;
            p->gang = g;
            g->members.push_back(p);
        }
        si[i] = p->next_slot;
        int rc = prep_slot(p, p->slot[si[i]], nsamples);
        if (rc)
            return rc;
    }
    /* the inputs: one generator launch for the on-device sources, one copy each for the host-fed ones */
    if (all_synth) {
        SynthMany sm;
        for (int i = 0; i < n; ++i) {
            sm.dst[i] = items[i].pipe->slot[si[i]].d_in;
            sm.byte_offset[i] = items[i].byte_offset;
            sm.seed[i] = items[i].seed;
        }
        /* The generator depends on nothing but its buffers being free (the slot's previous batch, two rounds back): it
         * goes to a stream of its own and runs BESIDE the kernels of the round before, which the main stream is still
         * working on when this round is queued (a free-running source keeps two rounds in flight).  The main stream
         * meets an event that is long complete.                                                                  */
        if (n > 1 && !tunables().gang_gen_inline.load()) {
            hipEvent_t seen[PDDC_GANG_MAX];
            int nseen = 0;
            for (int i = 0; i < n; ++i) {
                const pddc_pipeline::HostSlot &sl = items[i].pipe->slot[si[i]];
                if (!sl.used || !sl.ev_wait || std::find(seen, seen + nseen, sl.ev_wait) != seen + nseen)
                    continue;
                seen[nseen++] = sl.ev_wait;
                HIP_TRY(hipStreamWaitEvent(g->s_gen, sl.ev_wait, 0));
            }
            HIP_TRY(launch_synth_lcg_many(sm, n, nsamples * 6, g->s_gen));
            HIP_TRY(hipEventRecord(g->ev_gen[g->next_ev], g->s_gen));
            HIP_TRY(hipStreamWaitEvent(s, g->ev_gen[g->next_ev], 0));
        } else {
            HIP_TRY(launch_synth_lcg_many(sm, n, nsamples * 6, s));
        }
    } else {
        for (int i = 0; i < n; ++i) {
            pddc_pipeline::HostSlot &sl = items[i].pipe->slot[si[i]];
            if (items[i].h_packed)
                HIP_TRY(hipMemcpyAsync(sl.d_in, items[i].h_packed, nsamples * 6, hipMemcpyHostToDevice, s));
            else
                HIP_TRY(launch_synth_lcg(sl.d_in, nsamples * 6, items[i].seed, items[i].byte_offset, s));
        }
    }
    /* every member plans its batch: the ones whose plan is "first-stage kernel [+ one plain decimator]" leave a record,
     * the others run their own launches on the gang's stream right here */
    GangRec rec[PDDC_GANG_MAX];
    bool open[PDDC_GANG_MAX];
    float *direct[PDDC_GANG_MAX];
    for (int i = 0; i < n; ++i) {
        pddc_pipeline *p = items[i].pipe;
        pddc_pipeline::HostSlot &sl = p->slot[si[i]];
        direct[i] = (p->flags & PDDC_F_OUT_PACKED24) ? nullptr : direct_out(sl, items[i].h_out);
        void *dst = direct[i] ? direct[i] : sl.d_out;
        const size_t cap = direct[i] ? items[i].out_capacity : sl.out_cap;
        p->gang_rec = n > 1 && !tunables().gang_solo.load() ? &rec[i] : nullptr;
        int rc = p->gang_rec ? pddc_pipeline_process(p, sl.d_in, nsamples, dst, cap, &items[i].n_out, s) : 1;
        p->gang_rec = nullptr;
        open[i] = rc == PDDC_OK && rec[i].kind != 0;
        if (rc == 1) {
            rc = pddc_pipeline_process(p, sl.d_in, nsamples, dst, cap, &items[i].n_out, s);
            if (rc == PDDC_OK)
                rc = pddc_pipeline_fence(p, s);
        }
        if (rc)
            return rc;                         /* (members before this one have moved on: the stream is broken) */
    }
    /* one launch per group of members with the same kernels */
    for (int i = 0; i < n; ++i) {
        if (!open[i])
            continue;
        Fir8Many fm;
        FirI8xMany xm;
        GenTailMany tm;
        int k = 0;
        bool any_tail = false;
        const bool x = rec[i].kind == 3;
        for (int j = i; j < n; ++j) {
            if (!open[j] || rec[j].kind != rec[i].kind || rec[j].mix != rec[i].mix ||
                (rec[j].tail.nblocks > 0) != (rec[i].tail.nblocks > 0) || rec[j].tail.kind != rec[i].tail.kind ||
                rec[j].tail.D != rec[i].tail.D || rec[j].tail.ntaps != rec[i].tail.ntaps)
                continue;
            if (x ? (rec[j].hist != rec[i].hist || rec[j].fuse2 != rec[i].fuse2 || rec[j].ax.n_in != rec[i].ax.n_in)
                  : (rec[j].ntb != rec[i].ntb || rec[j].R != rec[i].R || rec[j].a.n_in != rec[i].a.n_in))
                continue;
            if (x)
                xm.a[k] = rec[j].ax;
            else
                fm.a[k] = rec[j].a;
            tm.t[k] = rec[j].tail;
            any_tail = any_tail || rec[j].tail.nblocks > 0;
            open[j] = false;
            ++k;
        }
        if (x)
            HIP_TRY(launch_fir_i8x_many(xm, k, rec[i].hist, rec[i].mix, rec[i].fuse2, s, rec[i].blocks, rec[i].chunk, rec[i].layout));
        else
            HIP_TRY(launch_fir8_many(rec[i].kind, rec[i].ntb, rec[i].R, rec[i].mix, fm, k, s));
        if (any_tail)
            HIP_TRY(launch_gen_tail_many(tm, k, s));
        if (n_ganged)
            *n_ganged += k;
    }
    /* the outputs leave, one event for the round: every member's ticket waits for it */
    hipEvent_t ev = g->ev[g->next_ev];
    g->next_ev = (g->next_ev + 1) & 3;
    for (int i = 0; i < n; ++i) {
        pddc_pipeline *p = items[i].pipe;
        pddc_pipeline::HostSlot &sl = p->slot[si[i]];
        if (items[i].n_out && !direct[i])
            HIP_TRY(hipMemcpyAsync(items[i].h_out, sl.d_out, items[i].n_out * ((p->flags & PDDC_F_OUT_PACKED24) ? 6 : 8),
                                   hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipEventRecord(ev, s));
    for (int i = 0; i < n; ++i) {
        pddc_pipeline *p = items[i].pipe;
        pddc_pipeline::HostSlot &sl = p->slot[si[i]];
        sl.ev_wait = ev;
        sl.used = true;
        p->next_slot = si[i] ^ 1;
        items[i].ticket = si[i];
    }
    return PDDC_OK;
}

int pddc_pipeline_push_host(pddc_pipeline *p, const void *h_packed, size_t nsamples, void *h_out,
                            size_t out_capacity, size_t *n_out_ret)
{
    int ticket = -1;
    int rc = pddc_pipeline_push_host_async(p, h_packed, nsamples, h_out, out_capacity, n_out_ret, &ticket);
    if (rc)
        return rc;
    if (ticket >= 0)
        return pddc_pipeline_wait_ticket(p, ticket);
    return PDDC_OK;
}

/* ---- checkpoint / resume of the stream state ------------------------------------------------
 * What the reference never had to keep (its FPGA did): per stage the FIR history and the decimation
 * phase, the 64-bit sample counter, the NCO word with its phase offset and the tuning-word segments
 * that still reach into stage 0's history.  Saved as one host blob; restored into a pipeline of the
 * same plan (same or another GPU): the stream then continues bit-identically.                       */
struct StateHeader {
    uint32_t magic, version, nstages, flags;
    int32_t R, NT;
    uint64_t n0;
    uint32_t freg, phase_off, freg_applied, fresh, nsegs, pad;
    struct {
        int32_t decim, interp, ntaps, hist, hist_elem, pad;
        uint64_t consumed;
    } st[PDDC_MAX_STAGES];
};
static const uint32_t kStateMagic = 0x53434450u;      /* "PDCS" */

static size_t state_bytes(const pddc_pipeline *p)
{
    size_t n = sizeof(StateHeader) + p->segs.size() * (sizeof(long long) + 2 * sizeof(uint32_t));
    for (int i = 0; i < p->nstages; ++i)
        n += (size_t)p->st[i].hist * (size_t)p->st[i].hist_elem;
    return n;
}

size_t pddc_pipeline_state_size(const pddc_pipeline *p) { return p ? state_bytes(p) : 0; }

int pddc_pipeline_save_state(pddc_pipeline *p, void *h_buf, size_t capacity, size_t *used)
{
    if (!p || !h_buf)
        return fail(PDDC_EINVAL, "null argument");
    const size_t need = state_bytes(p);
    if (used)
        *used = need;
    if (capacity < need)
        return fail(PDDC_ECAPACITY, "state needs %zu bytes, buffer has %zu", need, capacity);
    if (p->carry_pending)
        return fail(PDDC_ESTATE, "overlap mode holds a tail back: pddc_pipeline_fence(p, stream) first");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipDeviceSynchronize());                  /* every batch pushed so far is part of the state */
    StateHeader h = {};
    h.magic = kStateMagic;
    h.version = 1;
    h.nstages = (uint32_t)p->nstages;
    h.flags = p->flags;
    h.R = p->R;
    h.NT = p->NT;
    h.n0 = p->n0;
    h.freg = p->freg;
    h.phase_off = p->phase_off;
    h.freg_applied = p->freg_applied;
    h.fresh = p->fresh ? 1u : 0u;
    h.nsegs = (uint32_t)p->segs.size();
    for (int i = 0; i < p->nstages; ++i) {
        const Stage &s = p->st[i];
        h.st[i].decim = s.decim;
        h.st[i].interp = s.interp;
        h.st[i].ntaps = s.ntaps;
        h.st[i].hist = s.hist;
        h.st[i].hist_elem = s.hist_elem;
        h.st[i].consumed = s.consumed;
    }
    uint8_t *w = static_cast<uint8_t *>(h_buf);
    memcpy(w, &h, sizeof(h));
    w += sizeof(h);
    for (const auto &sg : p->segs) {
        memcpy(w, &sg.n_begin, sizeof(long long));
        memcpy(w + 8, &sg.freg, 4);
        memcpy(w + 12, &sg.off, 4);
        w += 16;
    }
    for (int i = 0; i < p->nstages; ++i) {
        const Stage &s = p->st[i];
        const size_t nb = (size_t)s.hist * (size_t)s.hist_elem;
        HIP_TRY(hipMemcpy(w, s.d_hist[s.cur], nb, hipMemcpyDeviceToHost));
        w += nb;
    }
    return PDDC_OK;
}

int pddc_pipeline_restore_state(pddc_pipeline *p, const void *h_buf, size_t nbytes)
{
    if (!p || !h_buf || nbytes < sizeof(StateHeader))
        return fail(PDDC_EINVAL, "bad state buffer");
    StateHeader h;
    memcpy(&h, h_buf, sizeof(h));
    if (h.magic != kStateMagic || h.version != 1)
        return fail(PDDC_EINVAL, "not a pipeline state (magic %08x version %u)", h.magic, h.version);
    if ((int)h.nstages != p->nstages || h.flags != p->flags || h.R != p->R || h.NT != p->NT)
        return fail(PDDC_ESTATE, "state was saved from a different plan (stages %u/%d, flags %x/%x, R %d/%d)", h.nstages,
                    p->nstages, h.flags, p->flags, h.R, p->R);
    size_t need = sizeof(StateHeader) + (size_t)h.nsegs * 16;
    for (int i = 0; i < p->nstages; ++i) {
        const Stage &s = p->st[i];
        if (h.st[i].decim != s.decim || h.st[i].interp != s.interp || h.st[i].ntaps != s.ntaps ||
            h.st[i].hist != s.hist || h.st[i].hist_elem != s.hist_elem)
            return fail(PDDC_ESTATE, "state was saved from a different plan (stage %d)", i);
        need += (size_t)s.hist * (size_t)s.hist_elem;
    }
    if (nbytes < need || h.nsegs < 1 || h.nsegs > 4096)
        return fail(PDDC_EINVAL, "state buffer truncated (%zu of %zu bytes)", nbytes, need);
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipDeviceSynchronize());
    const uint8_t *r = static_cast<const uint8_t *>(h_buf) + sizeof(StateHeader);
    p->segs.clear();
    for (uint32_t k = 0; k < h.nsegs; ++k) {
        pddc_pipeline::WordSeg sg;
        memcpy(&sg.n_begin, r, sizeof(long long));
        memcpy(&sg.freg, r + 8, 4);
        memcpy(&sg.off, r + 12, 4);
        p->segs.push_back(sg);
        r += 16;
    }
    for (int i = 0; i < p->nstages; ++i) {
        Stage &s = p->st[i];
        const size_t nb = (size_t)s.hist * (size_t)s.hist_elem;
        HIP_TRY(hipMemcpy(s.d_hist[s.cur], r, nb, hipMemcpyHostToDevice));
        s.consumed = h.st[i].consumed;
        r += nb;
    }
    p->n0 = h.n0;
    p->freg = h.freg;
    p->phase_off = h.phase_off;
    p->freg_applied = h.freg_applied;
    p->fresh = h.fresh != 0;
    compute_lo_steps(p);
    return PDDC_OK;
}

int pddc_pipeline_time_stage0_inline(pddc_pipeline *p, int enable)
{
    if (!p)
        return fail(PDDC_EINVAL, "null pipeline");
    p->time_stage0 = enable != 0;
    p->ev_used = 0;
    return PDDC_OK;
}

int pddc_pipeline_stage0_time(pddc_pipeline *p, float *avg_ms, int *nlaunches)
{
    if (!p || !avg_ms)
        return fail(PDDC_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(p->device));
    double sum = 0.0;
    for (size_t k = 0; k < p->ev_used; ++k) {
        HIP_TRY(hipEventSynchronize(p->ev_pool[k].second));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, p->ev_pool[k].first, p->ev_pool[k].second));
        sum += ms;
    }
    *avg_ms = p->ev_used ? (float)(sum / (double)p->ev_used) : 0.0f;
    if (nlaunches)
        *nlaunches = (int)p->ev_used;
    p->ev_used = 0;
    return PDDC_OK;
}

int pddc_pipeline_inject_failure(pddc_pipeline *p, int stage)
{
    if (!p || stage < 0 || stage >= p->nstages)
        return fail(PDDC_EINVAL, "bad stage");
    p->fail_at_stage = stage;
    return PDDC_OK;
}

int pddc_pipeline_schedule(const pddc_pipeline *p, size_t nsamples, int out[5])
{
    if (!p || !out)
        return fail(PDDC_EINVAL, "null argument");
    if (!stage0_fused(p))
        return fail(PDDC_ESTATE, "stage 0 does not run the fused kernel");
    const bool fuse2 = stages01_fusable(p, nsamples);
    const int nt = fuse2 ? 256 : p->NT;
    out[0] = fir8_tile_inputs(p->R, nt);
    fir8_schedule_query((long long)nsamples, p->R, fuse2, nt, &out[1], &out[2], &out[3], &out[4],
                        stages012_fusable(p, nsamples) ? p->s3.g : 0);
    return PDDC_OK;
}

int pddc_measure_copy(void *d_dst, const void *d_src, size_t nbytes, int iters, void *stream_v, float *avg_ms)
{
    if (!d_dst || !d_src || !avg_ms || iters < 1 || nbytes == 0)
        return fail(PDDC_EINVAL, "bad argument");
    int rc = require_device();
    if (rc)
        return rc;
    hipStream_t s = (hipStream_t)stream_v;
    if (((uintptr_t)d_dst | (uintptr_t)d_src | nbytes) & 15)
        return fail(PDDC_EINVAL, "copy measurement wants 16-byte aligned pointers and size");
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    for (int i = 0; i < iters; ++i)                             /* as many untimed ones first: sustained clocks */
        HIP_TRY(launch_stream_copy(d_src, d_dst, nbytes, s));
    HIP_TRY(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i)
        HIP_TRY(launch_stream_copy(d_src, d_dst, nbytes, s));
    HIP_TRY(hipEventRecord(e1, s));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *avg_ms = ms / (float)iters;
    return PDDC_OK;
}

int pddc_pipeline_time_stage0(pddc_pipeline *p, const void *d_packed, size_t nsamples, void *d_out,
                              int iters, void *stream_v, float *avg_ms)
{
    if (!p || !avg_ms || iters < 1)
        return fail(PDDC_EINVAL, "bad argument");
    if (!stage0_fused(p))
        return fail(PDDC_ESTATE, "stage 0 does not run the fused kernel");
    if (nsamples % PDDC_INPUT_GRANULE || ((uintptr_t)d_packed & 15) || ((uintptr_t)d_out & 15))
        return fail(PDDC_EINVAL, "bad size/alignment");
    hipStream_t s = (hipStream_t)stream_v;
    HIP_TRY(hipSetDevice(p->device));
    Fir8Args a;
    a.in = d_packed;
    a.hist = p->st[0].d_hist[p->st[0].cur];
    a.hist_out = nullptr;                  /* state is not advanced */
    a.out = static_cast<float *>(d_out);
    const bool fuse2 = stages01_fusable(p, nsamples);
    const bool fuse3 = stages012_fusable(p, nsamples);
    if (p->nstages > 1) {                  /* stage 0 (or the fused pair) of a cascade writes an internal buffer */
        int rc = ensure_buf(p->st[1], nsamples / (size_t)p->st[0].decim + 8);
        if (rc)
            return rc;
        a.out = p->st[1].d_buf;
    }
    if (fuse3) {                           /* the fused cascade: third-stage outputs into that buffer, state untouched */
        size_t off3, n3;
        unsigned long long m3;
        stage_outputs(p->st[2].consumed, nsamples / 64, p->st[2].decim, 1, &off3, &m3, &n3);
        fill_stage3_args(p, a, a.out, off3, n3, false);
    }
    if (fuse2) {
        a.taps2_blk = p->st[1].d_taps_blk;
        a.hist2 = p->st[1].d_hist[p->st[1].cur];
        a.hist2_out = nullptr;
    }
    a.taps_blk = p->st[0].d_taps_blk;
    a.n_in = (long long)nsamples;
    fill_fir8_args(p, a);
    const bool mix = (p->flags & PDDC_F_MIX) != 0;
    const int i8kind = stage0_i8_kind(p, nsamples);                    /* what process() would launch for this batch */
    const bool x2 = !fuse3 && i8kind == 2 && stages01_i8x(p, nsamples);
    const bool x1 = !fuse3 && !x2 && !fuse2 && i8kind == 2;
    FirI8xArgs qx;
    if (x1 || x2) {
        int rc = i8x_prepare(p, mix, x2, s, qx);
        if (rc)
            return rc;
        qx.in = d_packed;
        qx.hist = a.hist;
        qx.hist_out = nullptr;
        qx.out = a.out;
        qx.n_in = (long long)nsamples;
        qx.hist2 = p->st[1].d_hist[p->st[1].cur];
        qx.hist2_out = nullptr;
    }
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) {
        if (fuse3)
            HIP_TRY(launch_fir8_fused3(p->st[0].ntb, p->R, mix, a, s));
        else if (x1 || x2)
            HIP_TRY(launch_fir_i8x(qx, p->st[0].hist, mix, x2, s, p->opt.i8x_blocks, p->opt.i8x_chunk, p->opt.i8x_layout));
        else if (fuse2)
            HIP_TRY(launch_fir8_fused2(p->st[0].ntb, p->R, mix, a, s));
        else
            HIP_TRY(launch_fir8(p->st[0].ntb, p->R, IN_PACKED24, mix, a, s, p->NT));
    }
    HIP_TRY(hipEventRecord(e1, s));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *avg_ms = ms / (float)iters;
    return PDDC_OK;
}

/* Where in ONE large allocation of the caller's does this pipeline's write side go?  HBM is laid out in a few classes of
 * large extents, and a kernel that reads one stream while it writes another runs 1-8 % faster when the two lie in
 * different classes (NOTEBOOK.md rounds 1-3, 5 (o)-(u)).  The input is at the arena's start; the probe is the pipeline's
 * OWN first kernel (a read+write model stream ranked slots differently from the kernels it stood for -- round 4's review
 * -- and is gone).  One fused stage: the kernel writes its output at out_offset of the candidate slot.  A cascade: its
 * inter-stage workspace goes there (pddc_pipeline_set_workspace; the first kernel writes into it) and STAYS at the chosen
 * slot; the caller puts the output behind it.  The slot right behind the input first ("first come"), then +32 / +48 /
 * +64 GiB, every slot only if none of those gains 3 %.  Stream state is not advanced.                              */
int pddc_pipeline_arena_place(pddc_pipeline *p, void *d_arena, size_t arena_bytes, size_t slot_bytes, size_t nsamples,
                              size_t out_offset, size_t *out_slot, float *ms_first_come, float *ms_best, int *nprobes,
                              void *stream_v)
{
    if (!p || !d_arena || !out_slot || slot_bytes == 0 || ((uintptr_t)d_arena & 255) || (slot_bytes & 255) || (out_offset & 255))
        return fail(PDDC_EINVAL, "bad argument");
    if (!stage0_fused(p))
        return fail(PDDC_ESTATE, "stage 0 does not run a fused kernel");
    const bool cascade = p->nstages > 1;
    const size_t ws = cascade ? pddc_pipeline_workspace_size(p, nsamples) : 0;
    const size_t n_out = pddc_pipeline_max_output(p, nsamples) + 8;
    if (nsamples * 6 > out_offset || out_offset + ws + n_out * 8 > slot_bytes)
        return fail(PDDC_EINVAL, "the slot does not hold the batch and the write side at this offset");
    const size_t nslot = arena_bytes / slot_bytes;
    if (nslot < 2)
        return fail(PDDC_EINVAL, "the arena holds fewer than two slots");
    uint8_t *base = static_cast<uint8_t *>(d_arena);
    std::vector<float> ms(nslot, -1.0f);
    int n = 0, rc = PDDC_OK;
    float t = 0.0f;
    auto side = [&](size_t o) -> void * {
        uint8_t *dst = base + o * slot_bytes + out_offset;
        if (cascade && (rc = pddc_pipeline_set_workspace(p, dst, ws, nsamples)))
            return nullptr;
        return dst;
    };
    /* The probes compare times a few per cent apart, so they must all see the same clocks: after idle the chip boosts
     * for a few launches and then undershoots for some tens of milliseconds.  ~40 ms of untimed launches up front, then
     * 6 untimed + 12 timed launches per slot, queued back to back. */
    void *d0 = side(1);
    if (rc || (rc = pddc_pipeline_time_stage0(p, base, nsamples, d0, 100, stream_v, &t)))
        return rc;
    auto probe = [&](size_t o) {
        if (o >= nslot || ms[o] >= 0.0f || rc)
            return;
        void *dst = side(o);
        if (rc || (rc = pddc_pipeline_time_stage0(p, base, nsamples, dst, 6, stream_v, &t)))
            return;
        if ((rc = pddc_pipeline_time_stage0(p, base, nsamples, dst, 12, stream_v, &t)))
            return;
        ms[o] = t;
        ++n;
    };
    const size_t gib8 = ((size_t)8 << 30) / slot_bytes ? ((size_t)8 << 30) / slot_bytes : 1;
    probe(1);
    probe(4 * gib8);
    probe(6 * gib8);
    probe(8 * gib8);
    auto best_of = [&]() {
        size_t b = 1;
        for (size_t o = 1; o < nslot; ++o)
            if (ms[o] >= 0.0f && ms[o] < ms[b])
                b = o;
        return b;
    };
    if (!rc && ms[best_of()] > 0.97f * ms[1])
        for (size_t o = 2; o < nslot; ++o)
            probe(o);
    if (rc)
        return rc;
    const size_t b = best_of();
    (void)side(b);
    if (rc)
        return rc;
    *out_slot = b;
    if (ms_first_come)
        *ms_first_come = ms[1];
    if (ms_best)
        *ms_best = ms[b];
    if (nprobes)
        *nprobes = n;
    return PDDC_OK;
}

} /* extern "C" */
