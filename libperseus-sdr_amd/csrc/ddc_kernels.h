/*
 * ddc_kernels.h -- internal launch interface between the host-side pipeline
 * (ddc_pipeline.cpp) and the gfx950 kernels (ddc_kernels.hip).
 * Not part of the public ABI (that is include/perseus_ddc.h).
 */
#ifndef PDDC_DDC_KERNELS_H
#define PDDC_DDC_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace pddc {

enum InFmt { IN_PACKED24 = 0, IN_F32C = 1 };

/* Process-wide development knobs of the launchers and of the host plumbing.  The environment (PDDC_<NAME>) is looked at
 * ONCE, the first time tunables() is called; afterwards they change only through set_tunable (pddc_set_tunable: tests,
 * tools).  Nothing on the data path calls getenv.  -1 / 0 = the built-in choice.                                      */
struct Tunables {
    std::atomic<int> fir8_dyn_pct{ -1 };      /* share of a k_fir8 launch's tiles that go out in dynamic chunks (-1: default) */
    std::atomic<int> fir8_chunk{ 0 };         /* tiles per dynamic chunk (0: default)                                  */
    std::atomic<int> fir8_walk{ -1 };         /* 1: chunks handed ROUND the blocks (chunk j -> block j mod nblocks, no counter);
                                               * 0: static runs + dynamic tail; -1: the launcher's default for the form  */
    std::atomic<int> gen_shape_nt{ 0 }, gen_shape_p{ 0 };   /* k_fir_generic block shape override (PDDC_GEN_SHAPE=NT,P) */
    std::atomic<int> no_firp{ 0 };            /* 1: plain decimators by 4/5/10 on k_fir_generic instead of k_firp      */
    std::atomic<int> firp_packed_p{ 0 };
    std::atomic<int> unpack_blocks{ 512 };
    std::atomic<int> debug{ 0 };              /* allocation / placement messages on stderr                             */
    std::atomic<int> push_three_streams{ 0 }, gang_copy_out{ 0 }, gang_gen_inline{ 0 }, gang_solo{ 0 };
};
Tunables &tunables();
bool set_tunable(const char *name, int value);       /* false: no such knob */
bool get_tunable(const char *name, int *value);

/* A THIRD stage fused behind the decimate-by-8 pair (launch_fir8_fused3): a plain decimate-by-d FIR on the second
 * stage's outputs, which then never reach HBM either -- the whole cascade is ONE streaming pass (x320 = 8*8*5:
 * 6 B read + 8/320 B written per input sample).  The stage runs at 1/64 of the input rate, so it is written for
 * simplicity, not for issue slots: the block keeps the second stage's outputs of a GROUP of g tiles (g*TO2 samples, a
 * multiple of d, so every group has the same output phase) in an LDS ring behind the h samples of history; lane j
 * computes output j of the group and each of the four waves a quarter of the taps (kept in two VGPRs, one tap per
 * lane, and read with v_readlane; one packed FMA for both rails per tap; the four partial sums meet in LDS) -- sliced
 * over the g tiles of the next group, `sl` taps per wave and tile, their LDS reads hidden behind the first-stage FIR.
 * A chunk of tiles starts with an unknown history: the outputs of its first group are held back, and when the chunk
 * ends the block takes the last h samples of the chunk in front of it -- published by that chunk's block through
 * `seam` with write-through (sc1) stores and a flag, MI355X_MICROARCH.md "inter-workgroup visibility" -- and adds
 * what they contribute.  Waiting only ever goes back in tile order, to a chunk that was taken earlier by a block
 * that is running, so it cannot deadlock; the spin is bounded all the same (sched[2] != 0 afterwards: timed out). */
struct Fir8Stage3 {
    const float *taps = nullptr;      /* [spl*seglen] h[k], zero beyond ntaps: wave w's segment starts at w*seglen  */
    const void  *hist = nullptr;      /* the h second-stage outputs (float2) that precede this batch               */
    void        *hist_out = nullptr;  /* receives the batch's last h of them (or NULL)                             */
    float       *out = nullptr;       /* float2 outputs of the third stage                                         */
    void        *seam = nullptr;      /* [chunks][seam_stride bytes]: each chunk's last h second-stage outputs     */
    unsigned    *flags = nullptr;     /* one word per chunk, zero between launches (the last block out clears them) */
    long long    n_out = 0;           /* outputs this batch produces                                               */
    int d = 0, ntaps = 0, h = 0;      /* decimation, taps, history (multiple of 8, >= ntaps - 1)                   */
    int off = 0;                      /* batch-relative index of the second-stage output that completes this
                                         batch's first third-stage output, 0 .. d-1                               */
    int g = 0, ng = 0;                /* tiles per group; outputs per group (g*TO2/d, a power of two <= 64)        */
    int spl = 0, seglen = 0, sl = 0;  /* tap segments (4: one per wave), taps per segment = g*sl, sl = taps a wave
                                         takes per tile (its slice; one of the instantiated lengths)             */
    int padf = 0;                     /* zero samples in front of the ring's history: max(0, spl*seglen - h), even */
    int seam_stride = 0;              /* bytes, multiple of 16, >= 8*h                                             */
};

/* One launch of the generic decimator on float2 -- 256 threads a block, one output a thread (k_fir_generic<1>) -- as a
 * plain record.  A launch of the fused pair can carry one along: the TAIL of the batch before (overlap mode of the
 * pipeline): `nblocks` extra thread blocks behind the pair's persistent ones run it on waves that would idle.   */
struct GenTail {
    const float *in = nullptr;        /* float2 input batch (the workspace half the previous pair wrote)           */
    const float *hist = nullptr;      /* the H samples in front of it                                              */
    float       *hist_out = nullptr;  /* receives the last H samples of [hist | in]                                */
    float       *out = nullptr;       /* float2 outputs                                                            */
    const float *taps = nullptr;      /* duplicated table (h[k], h[k]), see launch_fir_generic                     */
    long long    first = 0, n_out = 0, n_batch = 0;
    int          H = 0, D = 0, ntaps = 0;
    int          span = 0, a = 31;    /* block shape (gen_tail_shape)                                              */
    int          nblocks = 0;         /* 0: nothing carried                                                        */
    unsigned     lds = 0;             /* bytes of LDS a block needs                                                */
    int          kind = 0;            /* 0: k_fir_generic<1>'s body; 1: k_firp's (decimation 4, 5, 10), which wants: */
    const float *taps2 = nullptr;     /*    its own table, (h, h) pairs zero padded to firp_taps_len(D, ntaps)      */
    int          nbq = 0;
};
/* fills kind / span / a / nbq / nblocks / lds for D, ntaps, n_out (kind 1 where firp_supported and `have_taps2`);
 * false if a block's span does not fit `lds_cap` bytes */
bool gen_tail_shape(GenTail *t, size_t lds_cap, bool have_taps2);
/* the record as a launch of its own (what pddc_pipeline_fence queues for the last batch) */
hipError_t launch_gen_tail(const GenTail &t, hipStream_t s);

/* arguments of the fused decimate-by-8 kernel (k_fir8) */
struct Fir8Args {
    const void *in;          /* batch start: packed bytes or float2            */
    const void *hist;        /* 8*ntb samples that precede the batch           */
    void       *hist_out;    /* receives the batch's last 8*ntb samples (or NULL;
                                must not alias `hist`; needs n_in >= 8*ntb)     */
    float      *out;         /* float2 outputs                                 */
    const float *taps_blk;   /* [ntb][8] block-reversed taps (device)          */
    /* fused second decimate-by-8 stage (launch_fir8_fused2 only) */
    const float *taps2_blk = nullptr;  /* [8][8] block-reversed taps of the second stage  */
    const void  *hist2 = nullptr;      /* its history: the 64 stage-1 outputs (float2) that
                                precede this batch                               */
    void        *hist2_out = nullptr;  /* receives the batch's last 64 stage-1 outputs    */
    unsigned    *sched = nullptr;      /* 3 zero-initialised words of device memory: the tile
                                scheduler's chunk counter and exit counter (the kernel
                                leaves them zero again) and a sticky error word (fused third
                                stage: a bounded wait gave up); one set per stream  */
    Fir8Stage3   s3;                   /* launch_fir8_fused3 only                          */
    GenTail      tail;                 /* launch_fir8_fused2 only: the previous batch's tail, carried along */
    long long   n_in;        /* samples in the batch, multiple of 8            */
    unsigned long long n0;   /* absolute index of batch sample 0 (NCO phase)   */
    uint32_t    freg;        /* NCO tuning word                                */
    uint32_t    phase_off = 0;   /* phase(n) = n*freg + phase_off (mod 2^32): keeps the phase continuous
                                across retunes (the FPGA's accumulator never jumps)          */
    uint32_t    freg_hist = 0;   /* tuning word the samples in `hist` were mixed with (== freg unless
                                this is the first batch after a retune)                       */
    float       lo_c[8];     /* cos/sin of step e*freg, e=0..7 (host, double)  */
    float       lo_s[8];
    float       lo_c_hist[8];   /* the same for freg_hist                        */
    float       lo_s_hist[8];
};

/* ---- several streams, one launch chain ("gang"): the drop-in API's virtual receivers that share a GPU --------------
 * Up to kFir8ManyMax streams with the SAME plan and batch length go through one launch of each kernel: blockIdx.y is
 * the stream, and every record below is that stream's own (buffers, histories, tuning word, phase, scheduler words).
 * The records travel as kernel arguments (8 x sizeof(Fir8Args) stays under the 4 KiB a launch may carry).          */
constexpr int kFir8ManyMax = 8;
struct Fir8Many {
    Fir8Args a[kFir8ManyMax];
};
static_assert(sizeof(Fir8Many) <= 4000, "Fir8Many must fit the kernel-argument segment");
struct GenTailMany {
    GenTail t[kFir8ManyMax];
};
struct SynthMany {
    void              *dst[kFir8ManyMax];
    unsigned long long byte_offset[kFir8ManyMax];
    uint32_t           seed[kFir8ManyMax];
};
/* first stage of n streams: kind 2 = the fused pair (launch_fir8_fused2's kernel), kind 1 = the packed /8 stage alone
 * (launch_fir8's, IN_PACKED24).  The (ntb, R) shapes the pipeline picks (fir8_many_supported), 256 threads; every
 * a[i].n_in equal, a[i].tail empty, a[i].sched distinct. */
bool fir8_many_supported(int kind, int ntb, int R);
hipError_t launch_fir8_many(int kind, int ntb, int R, bool mix, const Fir8Many &m, int n, hipStream_t s);
/* n tails (same D, ntaps and kind; n_out may differ by a block) */
hipError_t launch_gen_tail_many(const GenTailMany &m, int n, hipStream_t s);
/* n generator streams of nbytes each */
hipError_t launch_synth_lcg_many(const SynthMany &m, int n, size_t nbytes, hipStream_t s);

/* ---- the first stages on the INT8 matrix cores (ddc_fir_i8.hip) -----------------------------------------------------
 * A 24-bit sample is three int8 planes -- the wire bytes -- and taps quantised to 32-bit integers are four balanced
 * base-256 digits: v_mfma_i32_16x16x64_i8 forms the byte-plane products exactly, the planes are recombined once per
 * output.  History and results as k_fir8's: `hist` packed samples in front, out[m] = sum_k h[k] x[8m - k]. */
constexpr int kFirI8Taps16Len = 128 + 6 * 64;               /* 1 KB */
/* host: the binary16 array k_fir_i8x's plain form reads with FirI8xArgs::taps16 (values must be binary16-representable) */
void fir_i8_taps16(const float *taps, int ntaps, int hist, uint16_t *out /* kFirI8Taps16Len */);

/* ---- k_fir_i8x: that product, without NCO or with the NCO folded into the taps, and optionally the second decimate-by-8 stage of a
 * cascade fused behind it (ddc_fir_i8.hip).  hist = 32, 64, 128 or 256 packed samples of history (the stage's 8 * ntb);
 * stream state as k_fir8's: packed history, and for the fused pair the 64 first-stage outputs (float2, mixed)
 * in front of the batch.  phase(n) = n * freg + phase_off for EVERY sample the batch's outputs touch, i.e. the history
 * window must have been mixed with the same word (the pipeline routes the one batch behind a retune through k_fir8). */
struct FirI8xArgs {
    const void *in;          /* packed batch, 16-byte aligned                                   */
    const void *hist;        /* the `hist` packed samples in front of it                        */
    void       *hist_out;    /* receives the batch's last `hist` samples (or NULL; needs n_in >= hist) */
    float      *out;         /* float2 outputs: n_in / 8, or n_in / 64 with the fused second stage */
    const void *atab;        /* fir_i8x_build_tables: 1 table (no NCO) or 2 (NCO; hist > 128: c, s; else the paired [c ; s], [-s ; c]) */
    long long   n_in;        /* samples, multiple of 8 (of 8192 with the fused second stage)     */
    float       scale;       /* integer result -> float                                         */
    float       ct[2];       /* the planes' unsigned -> signed offset times the tap sums, per component of u */
    unsigned long long n0;   /* absolute index of batch sample 0                                */
    uint32_t    freg, phase_off;
    const float *taps2 = nullptr;   /* fused second stage: fir_i8x_taps2 (kFirI8xTaps2Len floats, device)  */
    const void  *hist2 = nullptr;   /* the 64 first-stage outputs (float2) in front of the batch           */
    void        *hist2_out = nullptr;
    /* decimate-by-10 (launch_fir_i8x_d10; zero otherwise): output m of the batch is sum_k g[k] x[in_off + 10 m - k] */
    long long    n_out = 0;         /* outputs to store (0: n_in / 8)                                       */
    int          in_off = 0;        /* batch sample the first output's window ends on: a multiple of 8      */
    int          hist_len = 0;      /* samples the history buffer holds (0: the geometry's `hist`)           */
    /* binary16 tap STORAGE (PDDC_F_TAPS_FP16 without NCO; BASELINE config 5): instead of the operand table the device holds
     * the taps as IEEE binary16 values -- kFirI8Taps16Len of them, G[128 + tt] = h[hist - tt] for tt = 1 .. hist, zeros
     * elsewhere (fir_i8_taps16) -- and every block's matrix waves quantise them into their operand registers themselves:
     * the same integers fir_i8x_build_tables puts into the table.  atab is not read then. */
    const void  *taps16 = nullptr;
    float        two_e = 0.0f;      /* 2^E of fir_i8x_build_tables (exp2)                                    */
};
constexpr int kFirI8xTaps2Len = 2 * 68;
/* n streams (same geometry and batch length), one launch: blockIdx.y is the stream (the gang) */
struct FirI8xMany {
    FirI8xArgs a[kFir8ManyMax];
};
static_assert(sizeof(FirI8xMany) <= 4000, "FirI8xMany must fit the kernel-argument segment");
int    fir_i8x_mode(int hist, bool mix);                 /* 0: no NCO, 1: NCO split over waves (129..256 taps), 2: NCO, both tap sets in one operand */
size_t fir_i8x_table_bytes(int hist, bool mix);
/* host: the tap operand(s) for `ntaps` <= hist taps, NCO word freg when mix; false if the taps are all zero */
bool fir_i8x_build_tables(const float *taps, int ntaps, int hist, bool mix, uint32_t freg, int8_t *tables, float *scale,
                          float ct[2], int *exp2 = nullptr);
/* The tuned decimate-by-10 first stage (the 1.6 MS/s plan's) on the same kernel: paired rows, columns of 8 outputs 80
 * samples apart, tiles of 10240 samples, hist = 64.  A batch's first output need not sit on a multiple of 8 samples (the
 * stream's decimation phase: `first` = 0 .. 9), the loaders' groups do: the taps are delayed by `delay` = (-first) mod 8
 * samples -- g[k] = h[k - delay] e^{+j theta k}, ntaps + delay <= 64 -- and the windows end on in_off = first + delay. */
constexpr int kFirI8xD10Hist = 64;
bool fir_i8x_d10_build_tables(const float *taps, int ntaps, int delay, uint32_t freg, int8_t *tables, float *scale, float ct[2]);
size_t fir_i8x_d10_table_bytes();
hipError_t launch_fir_i8x_d10(const FirI8xArgs &a, hipStream_t s, int max_blocks = 0, int chunk = 0, int layout = -1);
/* host: the second stage's taps (<= 64) as the kernel reads them: complex g2[k] = h2[k] e^{+j 8 theta k}, descending */
void fir_i8x_taps2(const float *taps2, int ntaps2, bool mix, uint32_t freg, float *out /* kFirI8xTaps2Len */);
bool fir_i8x_supported(int hist, bool mix, bool fuse2);
/* max_blocks: persistent grid (0 = one block per CU); chunk: tiles per chunk of the round-robin walk (0 = 1, or 4 for the
 * fused pair; large = contiguous ranges per block) */
hipError_t launch_fir_i8x_many(const FirI8xMany &m, int n, int hist, bool mix, bool fuse2, hipStream_t s, int max_blocks = 0,
                               int chunk = 0, int layout = -1);
/* layout: which waves do what (ddc_fir_i8.hip "Who does what"): 0 a matrix wave on every SIMD, 1 matrix and post waves on
 * two SIMDs, loaders on the other two */
hipError_t launch_fir_i8x(const FirI8xArgs &a, int hist, bool mix, bool fuse2, hipStream_t s, int max_blocks = 0, int chunk = 0,
                          int layout = -1 /* by form */);

/* k_fir8 with packed input does not scale the unpacked integers (value * 256): the taps of
 * that stage must be uploaded multiplied by this, RN(1/8388607) / 256 -- the factor that
 * k_unpack24 applies per sample (bit-exact with the reference there; here the FIR tolerance
 * of 1e-6 applies and one rounding moves from every sample to every tap)                   */
constexpr float kFir8PackedTapScale = 0x1.000002p-31f;

/* tile geometry of k_fir8<NTB,R>: inputs per block tile */
constexpr int fir8_tile_inputs(int R, int NT = 256) { return 4 * NT * R; }
size_t fir8_lds_bytes(int ntb, int R);
bool   fir8_supported(int ntb, int R);
/* the two-level tile schedule a launch over n_in samples would use (tests place their
 * comparison windows on its seams) */
void   fir8_schedule_query(long long n_in, int R, bool fused, int NT, int *ntiles, int *nblocks, int *S, int *K,
                           int group = 0 /* fused third stage: tiles per group; S and K are multiples of it */);
/* block sizes of k_fir8: 256 threads always; 128 (four independent blocks per CU) for R = 8 packed first stages */
bool   fir8_nt_supported(int ntb, int R, int NT);
void   fir8_set_grid_blocks(int nblocks);   /* persistent grid override (0 = resident blocks x CUs) */

/* packed -> [mix] -> /8 -> /8 in one kernel: `out` receives the SECOND stage's
 * outputs (n_in/64); n_in must be a multiple of the tile (1024*R samples) */
bool fir8_fused2_supported(int ntb, int ntb2, int R);
size_t fir8_fused2_lds_bytes(int ntb, int R);        /* dynamic LDS of a fused-pair block                              */
constexpr size_t kCarryLdsCap = 50u * 1024u;        /* a carried tail block may ask for this much: three blocks of that
                                                       size (two of the first stage's, one of the tail's) share a CU */
hipError_t launch_fir8_fused2(int ntb, int R, bool mix, const Fir8Args &a, hipStream_t s);

/* packed -> [mix] -> /8 -> /8 -> /d3 in one kernel (a.s3 filled in, a.out unused): `a.s3.out` receives the THIRD
 * stage's outputs.  fir8_fused3_geometry fills the derived fields of s3 (g, ng, spl, seglen, sl, padf, seam_stride)
 * from d, ntaps, h and R and says whether the kernel can run the stage at all; fir8_fused3_max_chunks: how many
 * seam slots / flag words a launch may use at most.                                                           */
bool fir8_fused3_geometry(int ntb, int ntb2, int R, Fir8Stage3 *s3);
int  fir8_fused3_max_chunks();
hipError_t launch_fir8_fused3(int ntb, int R, bool mix, const Fir8Args &a, hipStream_t s);

/* returns hipSuccess or the launch error */
hipError_t launch_fir8(int ntb, int R, InFmt fmt, bool mix, const Fir8Args &a, hipStream_t s, int NT = 256);

hipError_t launch_unpack24(const void *d_in, long long nsamples, void *d_out, bool to_i32,
                           bool mix, unsigned long long n0, uint32_t freg, uint32_t phase_off,
                           const float *lo_c, const float *lo_s, hipStream_t s);

/* generic decimating FIR on float2: out[q] = sum_k h[k]*x[first + q*D - k],
 * x indexed relative to `in`; x[-H..-1] come from `hist` (H >= ntaps-1).
 * `taps` is the DUPLICATED table: entry k = the pair (h[k], h[k]) (a naturally aligned SGPR pair for the packed
 * FMA); it must be readable, as zeros, over entries [-3*D - 8, ntaps + 3*D + 8).  hist_out (or
 * NULL) receives the last H samples of [hist | in(n_batch)]; must not alias hist. */
hipError_t launch_fir_generic(const float *in, const float *hist, int H, long long first, long long n_out,
                              int D, const float *taps, int ntaps, float *out, float *hist_out,
                              long long n_batch, hipStream_t s);

/* The same decimator fed with 24-bit PACKED samples (stage 0 of a plan whose first decimation is not
 * 8): the block unpacks and -- `mix` -- mixes while it stages its input span; batch, hist (H samples,
 * H % 8 == 0) and hist_out are packed, n_batch % 8 == 0.  phase(n) = n*freg + phase_off for the batch,
 * and the history is mixed with freg_hist (phase-continuous at n0).                              */
hipError_t launch_fir_generic_packed(const void *in_packed, const void *hist_packed, int H, long long first,
                                     long long n_out, int D, const float *taps, int ntaps, float *out,
                                     void *hist_out_packed, long long n_batch, bool mix, unsigned long long n0,
                                     uint32_t freg, uint32_t phase_off, uint32_t freg_hist, const float *lo_c,
                                     const float *lo_s, const float *lo_c_hist, const float *lo_s_hist, hipStream_t s);

/* k_firp: the register-blocked decimator for the decimations the rate plans use besides 8 (4, 5, 10), float2 or packed
 * (+ NCO) input: same contract as launch_fir_generic / launch_fir_generic_packed except for the tap table -- (h[k], h[k])
 * pairs zero padded to firp_taps_len(D, ntaps) taps, nothing in front -- and that `mx` may be NULL for float2 input.
 * The history a stage keeps (H >= ntaps - 1, any length) is the same, so a stage can go back and forth between the
 * two kernels from batch to batch.                                                                               */
struct GenMixArgs;
bool firp_supported(int D, int ntaps);
int  firp_taps_len(int D, int ntaps);
int  firp_nbq(int D, int ntaps);
size_t firp_lds_bytes(int D, int ntaps);
hipError_t launch_firp(int infmt, bool mix, const void *in, const void *hist, int H, long long first, long long n_out,
                       int D, const float *taps2, int ntaps, float *out, void *hist_out, long long n_batch,
                       const GenMixArgs *mx, hipStream_t s);
hipError_t launch_firp_packed(const void *in_packed, const void *hist_packed, int H, long long first, long long n_out,
                              int D, const float *taps2, int ntaps, float *out, void *hist_out_packed, long long n_batch,
                              bool mix, unsigned long long n0, uint32_t freg, uint32_t phase_off, uint32_t freg_hist,
                              const float *lo_c, const float *lo_s, const float *lo_c_hist, const float *lo_s_hist,
                              hipStream_t s);

/* false when even the smallest block shape of the generic kernel cannot stage its input span
 * ((63*D + ntaps + 10) samples) in the 160 KiB of LDS: such a stage is refused at create time */
bool fir_generic_supported(int D, int ntaps);

/* dst = last H elements of [hist(H) | batch(n)], elem_bytes each (H*elem_bytes <= 16 KiB);
 * dst may alias hist */
hipError_t launch_hist_update(void *dst, const void *hist, int H, const void *batch, long long n,
                              int elem_bytes, hipStream_t s);

/* rational resampler: outputs m0..m0+n_out-1 of y[m] = sum_j g[jL+ph] x[floor(mM/L)-j];
 * `consumed` = absolute index of in[0]; x[-H..-1] from hist */
hipError_t launch_resample(const float *in, const float *hist, int H, unsigned long long consumed,
                           unsigned long long m0, long long n_out, int L, int M, const float *taps, int ntaps,
                           float *out, hipStream_t s);

/* The same from LDS-staged input and polyphase-ordered taps gpoly[ph*Kp + j] = h[j*L + ph] (K = ceil(ntaps/L)
 * taps per phase, rows zero-padded to Kp, a multiple of 4); also writes the next history (hist_out, or NULL):
 * the last H samples of [hist | in(n_batch)].  resample_lds_supported: the block's input span fits the LDS. */
bool resample_lds_supported(int L, int M, int ntaps);
hipError_t launch_resample_lds(const float *in, const float *hist, int H, unsigned long long consumed,
                               unsigned long long m0, long long n_out, int L, int M, const float *gpoly, int K, int Kp,
                               float *out, float *hist_out, long long n_batch, hipStream_t s);

/* float32 I/Q -> 24-bit packed (6 B/sample); in and out 16-byte aligned */
hipError_t launch_pack24(const float *in, long long nsamples, void *out, hipStream_t s);

/* plain streaming copy of nbytes (multiple of 16, both pointers 16-byte aligned) */
hipError_t launch_stream_copy(const void *src, void *dst, size_t nbytes, hipStream_t s);

/* equal read and write streams over two buffers that both wrap (placement probe) */
hipError_t launch_stream_probe(const void *src, size_t src_bytes, void *dst, size_t dst_bytes, size_t total_bytes,
                               hipStream_t s);

hipError_t launch_synth_lcg(void *dst, size_t nbytes, uint32_t seed, uint64_t byte_offset,
                            hipStream_t s);



} // namespace pddc
#endif
