/*
 * ddc_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the I/Q ingest +
 * decimation hot path.  Written for wave64 / 160 KiB LDS / HBM3E directly;
 * there is no other target.
 *
 *   k_unpack24   24-bit packed I/Q -> float32 (or MSB-aligned int32), optional
 *                NCO mix.  Bit-exact restatement of the reference client
 *                callbacks (examples/perseustest.c:432-502): float =
 *                (float)(v24*256) / 2147483392.0f == (float)v24 * RN(1/8388607),
 *                one v_cvt + one v_mul (exhaustively equal).
 *   k_fir8       fused  unpack -> [NCO mix] -> polyphase decimate-by-8 FIR.
 *                The 8 B/sample float intermediate never touches HBM:
 *                algorithmic traffic 6 B in + 1 B out per input sample.
 *                (+ a fused second /8 stage, + the previous batch's last stage as extra
 *                blocks of the launch; k_fir8_many: up to eight streams, one launch;
 *                body in fir8_block.inc)
 *   k_fir_i8x             the same filter on the INT8 matrix cores, with or without the NCO,
 *                optionally with the second /8 stage fused: ddc_fir_i8.hip.  Those are what
 *                a decimate-by-8 first stage runs on unless the pipeline's options say
 *                otherwise; k_fir8 keeps the pair with the carried tail above 2^26 samples.
 *   k_firp       register-blocked decimators by 4, 5, 8, 10 (packed first stages, float2 tails).
 *   k_fir_generic  any-D decimating FIR on float2 (later cascade stages).
 *   k_resample     rational L/M polyphase resampler (non-integer rates).
 *   k_pack24       float32 -> 24-bit packed (inverse of the unpack).
 *   k_hist_update  carries the FIR history between batches (generic path).
 *   k_synth_lcg    device-side synthetic source.
 *
 * k_fir8 design (NOTEBOOK.md rounds 1-3 section 4):
 *   - persistent grid (2 blocks per CU); a block = 256 threads = 4 waves; tile = 1024*R
 *     input samples; two-level schedule: a static run of tiles per block, then
 *     dynamic chunks from an atomic counter (the two blocks of a CU run unevenly)
 *   - load phase: every thread pulls whole 48-byte groups (8 samples) with
 *     3x global_load_dwordx4 one tile ahead (registers), unpacks with
 *     v_perm_b32 / v_cvt_f32_i32 (the 1/8388607 scale lives in the taps),
 *     optionally mixes with the NCO in tile-relative form (per-thread constant
 *     phasors; the tile's phasor goes on once per output at the stores), and
 *     writes PLANAR I / Q floats to LDS, rotated by one sample so that FIR windows
 *     are 16-byte aligned; the last NTB groups of a tile stay in LDS as the next
 *     tile's history
 *   - FIR phase: waves 0,2 filter the I plane, waves 1,3 the Q plane; each
 *     lane owns R consecutive outputs (a register sliding window over
 *     R+NTB-1 aligned 8-sample LDS groups, 2x ds_read_b128 each, conflict
 *     free through a 4-float pad every 8 groups); taps are wave-uniform and
 *     come through the scalar cache into SGPR pairs (s_load), so a tap costs
 *     no VGPR and no LDS traffic; the FMAs are PACKED (v_pk_fma_f32: every
 *     VALU op costs ~4 clocks per wave64 on this chip and the packed form does
 *     two FMAs in that slot), each output's dot product split into its even
 *     and odd terms so both operands are natural adjacent pairs; at R=8 the tap
 *     block is the outer loop, so one block of taps is live instead of eight
 *   - store phase: results are transposed through LDS (XOR-swizzled 16-byte
 *     chunks) into interleaved float2 and leave as coalesced nontemporal
 *     dwordx4 stores, one tile late so they sit behind the next load wait
 *   - optional fused second decimate-by-8 stage on the tile's outputs (NTB2)
 *   No fp32 MFMA here: 9 flop/B, a banded single-filter FIR wastes a third to a half of a
 *   matrix op and fp32 MFMA has the vector unit's own peak.  The INT8 matrix cores are another
 *   matter for this data: ddc_fir_i8.hip.
 */
#include "ddc_kernels.h"
#include "ddc_dev.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

/* wave issue priority of k_fir8's phases: unpack 0, FIR 2 (+3 % at 255 taps, NOTEBOOK.md rounds 1-3, 5 (v)) */
static constexpr int PDDC_PRIO_U = 0, PDDC_PRIO_F = 2;

namespace pddc {

Tunables &tunables()
{
    static Tunables t;
    static std::once_flag once;
    std::call_once(once, [] {
        auto env = [](const char *name, std::atomic<int> &v) {
            if (const char *e = getenv(name))
                v.store(atoi(e));
        };
        auto flag = [](const char *name, std::atomic<int> &v) {
            if (getenv(name))
                v.store(1);
        };
        env("PDDC_FIR8_DYN_PCT", t.fir8_dyn_pct);
        env("PDDC_FIR8_CHUNK", t.fir8_chunk);
        env("PDDC_FIR8_WALK", t.fir8_walk);
        if (const char *e = getenv("PDDC_GEN_SHAPE")) {
            int a = 0, b = 0;
            if (sscanf(e, "%d,%d", &a, &b) == 2) {
                t.gen_shape_nt.store(a);
                t.gen_shape_p.store(b);
            }
        }
        flag("PDDC_NO_FIRP", t.no_firp);
        env("PDDC_FIRP_PACKED_P", t.firp_packed_p);
        env("PDDC_UNPACK_BLOCKS", t.unpack_blocks);
        flag("PDDC_DEBUG", t.debug);
        flag("PDDC_PUSH_THREE_STREAMS", t.push_three_streams);
        flag("PDDC_GANG_COPY_OUT", t.gang_copy_out);
        flag("PDDC_GANG_GEN_INLINE", t.gang_gen_inline);
        flag("PDDC_GANG_SOLO", t.gang_solo);
    });
    return t;
}

static std::atomic<int> *tunable_by_name(const char *name)
{
    Tunables &t = tunables();
    const struct {
        const char *n;
        std::atomic<int> *v;
    } tab[] = { { "fir8_dyn_pct", &t.fir8_dyn_pct },   { "fir8_chunk", &t.fir8_chunk },       { "fir8_walk", &t.fir8_walk },
                { "gen_shape_nt", &t.gen_shape_nt },
                { "gen_shape_p", &t.gen_shape_p },     { "no_firp", &t.no_firp },             { "firp_packed_p", &t.firp_packed_p },
                { "unpack_blocks", &t.unpack_blocks }, { "debug", &t.debug },                 { "push_three_streams", &t.push_three_streams },
                { "gang_copy_out", &t.gang_copy_out }, { "gang_gen_inline", &t.gang_gen_inline }, { "gang_solo", &t.gang_solo } };
    if (name)
        for (const auto &e : tab)
            if (!strcmp(e.n, name))
                return e.v;
    return nullptr;
}

bool set_tunable(const char *name, int value)
{
    std::atomic<int> *v = tunable_by_name(name);
    if (v)
        v->store(value);
    return v != nullptr;
}

bool get_tunable(const char *name, int *value)
{
    std::atomic<int> *v = tunable_by_name(name);
    if (v && value)
        *value = v->load();
    return v != nullptr;
}

static constexpr int kFused3MaxChunks = 2048;      /* seam slots / flag words of a fused-cascade launch */
static constexpr int kPend3 = 4;                   /* open seams a block of the fused cascade may carry along */

/* float = (float)(v24*256) * RN(1/2147483392): the int->float convert is exact
 * (24 significant bits) and the product is bit-identical to the reference's
 * (float)int32 / (float)(INT_MAX-256) for all 2^24 codes (tests). */
static constexpr float kUnpackScale = 0x1.000002p-31f;   /* RN(1/8388607) / 256 */


/* ------------------------------------------------------------------------ */
/* 12 dwords (48 bytes) = 8 packed samples -> MSB-aligned int32 (value * 256),
 * exactly the iq_sample union placement of examples/perseustest.c:411-426.
 * One v_perm_b32 per component: bytes {b2,b1,b0,0x00}.                      */
__device__ __forceinline__ void unpack8_msb(const uint32_t (&w)[12], int32_t (&I)[8], int32_t (&Q)[8])
{
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const uint32_t a = w[3 * h], b = w[3 * h + 1], c = w[3 * h + 2];
        I[2 * h]     = (int32_t)__builtin_amdgcn_perm(a, a, 0x0201000cu);   /* bytes 0..2  */
        Q[2 * h]     = (int32_t)__builtin_amdgcn_perm(b, a, 0x0504030cu);   /* bytes 3..5  */
        I[2 * h + 1] = (int32_t)__builtin_amdgcn_perm(c, b, 0x0403020cu);   /* bytes 6..8  */
        Q[2 * h + 1] = (int32_t)(c & 0xffffff00u);                          /* bytes 9..11 */
    }
}

/* mix 8 consecutive samples whose first one has the local oscillator value cb + j*sb */
template <typename P>
__device__ __forceinline__ void mix8_lo(float (&xi)[8], float (&xq)[8], float cb, float sb, const P &p)
{
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        /* LO(nabs+e) = LO(nabs) * step[e] */
        const float c = cb * p.lo_c[e] - sb * p.lo_s[e];
        const float s = cb * p.lo_s[e] + sb * p.lo_c[e];
        const float r = xi[e] * c - xq[e] * s;
        const float i = xi[e] * s + xq[e] * c;
        xi[e] = r;
        xq[e] = i;
    }
}

/* (a + j b) * (c + j s) */
__device__ __forceinline__ void cmul(float &a, float &b, float c, float s)
{
    const float r = a * c - b * s;
    const float i = a * s + b * c;
    a = r;
    b = i;
}

/* (x.x + j x.y) * (p.x + j p.y) in TWO packed instructions: (xi c, xi s), then (xq, xq) * (-s, c) on top -- the
 * half selects and the sign ride in the instruction's op_sel / neg_lo fields.  Written as scalar code the compiler
 * vectorises four of these per pair of samples into 14 packed operations and 7 moves (k_firp<10,2,packed,MIX>, 21 of a
 * pair's 41 vector instructions); from vector code it folds the broadcasts but builds (-s, c) with an xor and a move. */
__device__ __forceinline__ f32x2 cmul_pk(f32x2 x, f32x2 p)
{
    f32x2 t, y;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(x), "v"(p));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(y) : "v"(x), "v"(p), "v"(t));
    return y;
}
/* the same for a wave-uniform second factor (kernel arguments): from vector code the compiler folds the broadcasts into
 * op_sel itself and builds (-s, c) on the scalar unit.  (An "s" constraint on a pair assembled from two kernarg floats
 * sent the whole table through scratch memory.) */
__device__ __forceinline__ f32x2 cmul_pk_s(f32x2 x, float c, float s)
{
    const f32x2 t = __builtin_shufflevector(x, x, 0, 0) * f32x2{ c, s };
    return __builtin_elementwise_fma(__builtin_shufflevector(x, x, 1, 1), f32x2{ -s, c }, t);
}

/* both complex values of an interleaved (I0,Q0,I1,Q1) vector times (c + j s) */
__device__ __forceinline__ f32x4 cmul2(f32x4 v, float c, float s)
{
    return f32x4{ v.x * c - v.y * s, v.x * s + v.y * c, v.z * c - v.w * s, v.z * s + v.w * c };
}

/* mix 8 consecutive samples whose first one has index nabs: phase = nabs*freg + off (mod 2^32).
 * `off` is the pipeline's phase offset for absolute indices (it keeps the phase continuous across
 * retunes, like the FPGA's phase accumulator) and 0 for tile-relative ones                     */
template <typename P>
__device__ __forceinline__ void mix8(float (&xi)[8], float (&xq)[8], unsigned long long nabs, const P &p,
                                     uint32_t off = 0u)
{
    float cb, sb;
    nco_lo((uint32_t)nabs * p.freg + off, cb, sb);
    mix8_lo(xi, xq, cb, sb, p);
}

/* ======================================================================== */
/* k_unpack24                                                               */
/* ======================================================================== */
struct UnpackArgs {
    const uint8_t *in;
    void          *out;
    long long      ns;
    unsigned long long n0;
    uint32_t       freg;
    uint32_t       phase_off;
    float          lo_c[8];
    float          lo_s[8];
};

template <bool TO_I32, bool MIX>
__global__ __launch_bounds__(256) void k_unpack24(UnpackArgs p)
{
    __shared__ __attribute__((aligned(16))) uint32_t slabs[4][1024];
    const int lane = threadIdx.x & 63;
    uint32_t *slab = slabs[threadIdx.x >> 6];
    const long long ngroups = (((p.ns + 7) >> 3) + 63) & ~63LL;   /* whole waves iterate together */
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < ngroups;
         g += (long long)gridDim.x * 256) {
        const long long s0 = g << 3;
        uint32_t w[12];
        if (s0 >= p.ns) {
#pragma unroll
            for (int d = 0; d < 12; ++d)
                w[d] = 0;
        } else if (s0 + 8 <= p.ns) {
            const uint4 *src = reinterpret_cast<const uint4 *>(p.in + s0 * 6);
            const uint4 a = src[0], b = src[1], c = src[2];
            w[0] = a.x; w[1] = a.y; w[2] = a.z;  w[3] = a.w;
            w[4] = b.x; w[5] = b.y; w[6] = b.z;  w[7] = b.w;
            w[8] = c.x; w[9] = c.y; w[10] = c.z; w[11] = c.w;
        } else {            /* ragged tail: byte loads, zero fill */
            const long long nb = (p.ns - s0) * 6;
#pragma unroll
            for (int d = 0; d < 12; ++d) {
                uint32_t v = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (4 * d + b < nb)
                        v |= (uint32_t)p.in[s0 * 6 + 4 * d + b] << (8 * b);
                w[d] = v;
            }
        }
        int32_t I[8], Q[8];
        unpack8_msb(w, I, Q);
        uint32_t o[16];
        if (TO_I32) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[2 * e]     = (uint32_t)I[e];
                o[2 * e + 1] = (uint32_t)Q[e];
            }
        } else {
            float xi[8], xq[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xi[e] = (float)I[e] * kUnpackScale;
                xq[e] = (float)Q[e] * kUnpackScale;
            }
            if (MIX)
                mix8(xi, xq, p.n0 + (unsigned long long)s0, p, p.phase_off);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[2 * e]     = __float_as_uint(xi[e]);
                o[2 * e + 1] = __float_as_uint(xq[e]);
            }
        }
        uint32_t *dst = reinterpret_cast<uint32_t *>(p.out) + s0 * 2;
        /* A lane holds 64 contiguous output bytes; storing them directly would
         * make every store instruction touch 64 separate 64-byte segments.  The
         * wave transposes its 4 KiB through a private LDS slab (XOR-swizzled
         * 16-byte chunks, conflict-free both ways, no block barrier) so that each
         * global_store_dwordx4 writes 1 KiB contiguous, with the nt hint.        */
        const long long wave_g0 = g - lane;                      /* first group of this wave */
        const bool wave_full = (wave_g0 + 64) * 8 <= p.ns;       /* wave-uniform */
        if (wave_full) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = 4 * lane + j;
                *reinterpret_cast<uint4 *>(slab + 4 * (c ^ ((c >> 3) & 7))) =
                    make_uint4(o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]);
            }
            /* same wave wrote and reads: program order + lgkmcnt is enough */
            u32x4 *wdst = reinterpret_cast<u32x4 *>(reinterpret_cast<uint32_t *>(p.out) + wave_g0 * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = 64 * j + lane;
                const u32x4 v = *reinterpret_cast<const u32x4 *>(slab + 4 * (q ^ ((q >> 3) & 7)));
                __builtin_nontemporal_store(v, wdst + q);
            }
        } else if (s0 + 8 <= p.ns) {
            uint4 *d4 = reinterpret_cast<uint4 *>(dst);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                d4[k] = make_uint4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
        } else if (s0 < p.ns) {
            const int rem = (int)(p.ns - s0);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (e < rem) {
                    dst[2 * e]     = o[2 * e];
                    dst[2 * e + 1] = o[2 * e + 1];
                }
        }
    }
}

hipError_t launch_unpack24(const void *d_in, long long ns, void *d_out, bool to_i32, bool mix,
                           unsigned long long n0, uint32_t freg, uint32_t phase_off, const float *lo_c,
                           const float *lo_s, hipStream_t s)
{
    if (ns <= 0)
        return hipSuccess;
    UnpackArgs a;
    a.in = static_cast<const uint8_t *>(d_in);
    a.out = d_out;
    a.ns = ns;
    a.n0 = n0;
    a.freg = freg;
    a.phase_off = phase_off;
    for (int e = 0; e < 8; ++e) {
        a.lo_c[e] = lo_c ? lo_c[e] : 1.0f;
        a.lo_s[e] = lo_s ? lo_s[e] : 0.0f;
    }
    const long long ngroups = (ns + 7) >> 3;
    long long blocks = (ngroups + 255) / 256;
    const int cap = tunables().unpack_blocks.load();                       /* 512: 2 per CU measured best (0.66 vs 0.70 ms) */
    if (blocks > cap)
        blocks = cap;
    const dim3 grid((unsigned)blocks), blk(256);
    if (to_i32)
        hipLaunchKernelGGL((k_unpack24<true, false>), grid, blk, 0, s, a);
    else if (mix)
        hipLaunchKernelGGL((k_unpack24<false, true>), grid, blk, 0, s, a);
    else
        hipLaunchKernelGGL((k_unpack24<false, false>), grid, blk, 0, s, a);
    return hipGetLastError();
}

/* ======================================================================== */
/* k_fir_generic, the kernel body (its description and launchers: further down) */
/* ======================================================================== */
struct GenMixArgs {
    unsigned long long n0;      /* absolute index of batch sample 0                            */
    uint32_t freg, phase_off;   /* phase(n) = n*freg + phase_off                               */
    uint32_t freg_hist;         /* word the history samples were mixed with (first batch after a retune) */
    float lo_c[8], lo_s[8];     /* step phasors of freg                                        */
    float lo_c_hist[8], lo_s_hist[8];
    /* k_firp's packed staging: a thread's pairs of samples lie 512 apart; LO(512 u freg), u = 0..15, from the host in
     * double.  (A recurrence phasor *= LO(512) carried the rounding of that ONE step phasor u times: 7.8e-7 of full
     * scale in the worst of 1200 random cases, against the 1e-6 bar; from this table it is one product from an exact
     * phase whatever u is.) */
    float lo512_c[16], lo512_s[16];
};

/* the body of one block: `bid` its index, `sd` its LDS, NT its threads -- also run by the extra blocks k_fir8 carries
 * along for the PREVIOUS batch's tail (Fir8Args::tail)                                                          */
template <int P, bool PACKED, bool MIX>
__device__ __forceinline__ void fir_generic_body(const float2 *__restrict__ in, const float2 *__restrict__ hist,
                                                 int H, long long first, long long n_out, int D,
                                                 const float PDDC_CONSTANT *taps, int ntaps,
                                                 float2 *__restrict__ out, int span, int a,
                                                 float2 *__restrict__ hist_out, long long n_batch,
                                                 const GenMixArgs &mx, const int bid, float2 *sd, const int NT)
{
    /* layout: span samples | 8 zero samples (the first, aligned step of the tap loop may
     * look up to 7 samples past a thread's windows, with zero taps); sample i at i + (i >> a) */
    const int tid = threadIdx.x;
    const int S = P * D;
    const long long q0 = (long long)bid * (NT * P);
    /* inputs needed: x[first + q0*D - (ntaps-1)  ..  first + (q0+NT*P-1)*D] */
    const long long x0 = first + q0 * D - (ntaps - 1);
    const long long last_needed = first + (n_out - 1) * (long long)D;    /* last valid input index */
    if (tid < 8) {
        const int i = span + tid;
        sd[i + (i >> a)] = make_float2(0.0f, 0.0f);
    }
    if (PACKED) {
        /* whole groups of 8 samples (48 bytes, 16-byte aligned both in the batch and in the
         * history, whose length is a multiple of 8), four groups in flight per thread */
        const uint8_t *inb = reinterpret_cast<const uint8_t *>(in);
        const uint8_t *hb8 = reinterpret_cast<const uint8_t *>(hist);
        const long long xa = (x0 >= 0 ? x0 : x0 - 7) / 8 * 8;     /* floor to a multiple of 8 */
        const int shift = (int)(x0 - xa);
        const int ngroups = (span + shift + 7) >> 3;
        /* the word and offset the samples in front of the batch were mixed with: the phase is
         * continuous at n0, so off_old = phase_off + n0*(freg - freg_hist)                    */
        const uint32_t off_old = mx.phase_off + (uint32_t)mx.n0 * (mx.freg - mx.freg_hist);
        for (int g0 = tid; g0 < ngroups; g0 += 4 * NT) {
            u32x4 raw[4][3];
            long long s0[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int g = g0 + NT * u;
                s0[u] = xa + 8LL * g;
                const uint8_t *src = nullptr;
                if (g < ngroups) {
                    if (s0[u] < 0) {
                        if (s0[u] >= -(long long)H)
                            src = hb8 + (s0[u] + H) * 6;
                    } else if (s0[u] < n_batch) {
                        src = inb + s0[u] * 6;
                    }
                }
#pragma unroll
                for (int w = 0; w < 3; ++w)
                    raw[u][w] = src ? reinterpret_cast<const u32x4 *>(src)[w] : u32x4{ 0u, 0u, 0u, 0u };
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int g = g0 + NT * u;
                if (g >= ngroups)
                    continue;
                const uint32_t w[12] = { raw[u][0].x, raw[u][0].y, raw[u][0].z, raw[u][0].w, raw[u][1].x, raw[u][1].y,
                                         raw[u][1].z, raw[u][1].w, raw[u][2].x, raw[u][2].y, raw[u][2].z, raw[u][2].w };
                int32_t I[8], Q[8];
                unpack8_msb(w, I, Q);
                float xi[8], xq[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xi[e] = (float)I[e] * kUnpackScale;
                    xq[e] = (float)Q[e] * kUnpackScale;
                }
                if (MIX) {
                    const bool old = s0[u] < 0;
                    float cb, sb;
                    nco_lo((uint32_t)(mx.n0 + (unsigned long long)s0[u]) * (old ? mx.freg_hist : mx.freg) +
                               (old ? off_old : mx.phase_off), cb, sb);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float sc = old ? mx.lo_c_hist[e] : mx.lo_c[e];
                        const float ss = old ? mx.lo_s_hist[e] : mx.lo_s[e];
                        cmul(xi[e], xq[e], cb * sc - sb * ss, cb * ss + sb * sc);
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int i = 8 * g - shift + e;               /* LDS index of sample xa + 8g + e */
                    if (i >= 0 && i < span)
                        sd[i + (i >> a)] = make_float2(xi[e], xq[e]);
                }
            }
        }
    } else if (x0 >= 0 && x0 + span - 1 <= last_needed) {
        /* interior block: branch-free, so the loads of 8 rounds are in flight together
         * (with the guarded form below every round waits for its own load: 21 serial
         * round trips made this kernel 36 us for the x320 cascade's last stage)       */
        const float2 *src = in + x0;
        for (int i = tid; i < span; i += 8 * NT) {
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {          /* clamped, not branched: all 8 loads go out together */
                const int ii = i + NT * u;
                v[u] = src[ii < span ? ii : span - 1];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ii = i + NT * u;
                if (ii < span)
                    sd[ii + (ii >> a)] = v[u];
            }
        }
    } else {
        for (int i = tid; i < span; i += NT) {
            const long long xi = x0 + i;
            float2 v = make_float2(0.0f, 0.0f);
            if (xi < 0) {
                if (xi >= -(long long)H)          /* history: the H samples that precede the batch */
                    v = hist[xi + H];
            } else if (xi <= last_needed) {
                v = in[xi];
            }
            sd[i + (i >> a)] = v;
        }
    }
    if (hist_out != nullptr && bid == 0) {
        if (PACKED) {                                   /* 6 bytes per sample, moved as dwords */
            const int Hw = H * 6 / 4;
            const long long nw = n_batch * 6 / 4;
            const uint32_t *hw = reinterpret_cast<const uint32_t *>(hist), *bw = reinterpret_cast<const uint32_t *>(in);
            uint32_t *ow = reinterpret_cast<uint32_t *>(hist_out);
            for (int i = tid; i < Hw; i += NT) {
                const long long j = (long long)i + nw;
                ow[i] = j < Hw ? hw[j] : bw[j - Hw];
            }
        } else {
            for (int i = tid; i < H; i += NT) {
                const long long j = (long long)i + n_batch;
                hist_out[i] = j < H ? hist[j] : in[j - H];
            }
        }
    }
    __syncthreads();
    /* four partial sums per output (window sample index mod 4): the rounding error of a long
     * fp32 accumulation grows with the length of the chain, and the 1e-6 budget is shared by
     * all stages of a cascade                                                               */
    f32x2 acc[P][4];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int a4 = 0; a4 < 4; ++a4)
            acc[p][a4] = f32x2{ 0.0f, 0.0f };
    /* sample x[q*D - j], q = q0 + P*tid, has local index S*tid + r with r = ntaps-1-j (wave-
     * uniform) and sits at lane + r + (r >> a): S*tid is a multiple of 2^a, so the pad splits */
    const float2 *lane = sd + (S + (S >> a)) * tid;
    /* The taps arrive DUPLICATED, (h[k], h[k]) per entry, so that a tap is a naturally aligned SGPR pair and the
     * packed FMA takes it as it is: with single floats hipcc moved every odd tap into the low half of a pair first
     * (12 s_mov per step of 24 FMAs).  And the two addressing forms are two separate loops: as one loop with a branch
     * inside, hipcc copied all 6*P accumulator registers at the merge point on every step (18 v_mov per 24 FMAs at
     * P = 3 -- the kernel issued 2.5x the VALU instructions its FMAs account for, profiles/r02/j_pmc_generic_tail.txt). */
    const f32x2 PDDC_CONSTANT *taps2 = reinterpret_cast<const f32x2 PDDC_CONSTANT *>(taps);
    if (a >= 3) {                                  /* (r >> a) is constant over an aligned step: immediates */
        for (int r0 = ((ntaps - 1) + (P - 1) * D) | 7; r0 >= 0; r0 -= 8) {
            const int jb = (ntaps - 1) - r0;       /* tap of output q for the step's first sample */
            f32x2 hh[P][8];
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    hh[p][u] = taps2[p * D + jb + u];
            const float2 *x8 = lane + (r0 + (r0 >> a)) - 7;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float2 xv = x8[7 - u];
                const f32x2 x = { xv.x, xv.y };
#pragma unroll
                for (int p = 0; p < P; ++p)
                    acc[p][u & 3] = __builtin_elementwise_fma(hh[p][u], x, acc[p][u & 3]);
            }
        }
    } else {
        for (int r0 = ((ntaps - 1) + (P - 1) * D) | 7; r0 >= 0; r0 -= 8) {
            const int jb = (ntaps - 1) - r0;
            f32x2 hh[P][8];
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    hh[p][u] = taps2[p * D + jb + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 - u;
                const float2 xv = lane[r + (r >> a)];
                const f32x2 x = { xv.x, xv.y };
#pragma unroll
                for (int p = 0; p < P; ++p)
                    acc[p][u & 3] = __builtin_elementwise_fma(hh[p][u], x, acc[p][u & 3]);
            }
        }
    }
    const long long q = q0 + (long long)P * tid;
#pragma unroll
    for (int p = 0; p < P; ++p)
        if (q + p < n_out) {
            const f32x2 sum = (acc[p][0] + acc[p][1]) + (acc[p][2] + acc[p][3]);
            out[q + p] = make_float2(sum.x, sum.y);
        }
}

template <int P, bool PACKED, bool MIX>
__global__ __launch_bounds__(256) void k_fir_generic(const float2 *__restrict__ in, const float2 *__restrict__ hist,
                                                      int H, long long first, long long n_out, int D,
                                                      const float PDDC_CONSTANT *taps, int ntaps,
                                                      float2 *__restrict__ out, int span, int a,
                                                      float2 *__restrict__ hist_out, long long n_batch,
                                                      GenMixArgs mx)
{
    extern __shared__ __attribute__((aligned(16))) float2 sd[];
    fir_generic_body<P, PACKED, MIX>(in, hist, H, first, n_out, D, taps, ntaps, out, span, a, hist_out, n_batch, mx,
                                     (int)blockIdx.x, sd, (int)blockDim.x);
}

/* ======================================================================== */
/* k_firp : register-blocked decimator for any D (tails, first stages that are not /8) */
/* ======================================================================== */
/* out[q] = sum_k h[k] * x[first + q*D - k], like k_fir_generic, with k_fir8's arithmetic shape carried over to any D:
 * a lane owns P CONSECUTIVE outputs, and the TAP BLOCK j (taps jD .. jD+D-1) is the outer loop -- output m meets tap
 * block j on the sample block b = m - j (block b = inputs bD .. bD+D-1 of the tile), so at step j the lane's P outputs
 * need the P blocks m0-j .. m0-j+P-1: one NEW block of D LDS reads per step, kept with the P-1 before it in a ring of
 * registers whose slot numbers are static once the loop is unrolled by P, feeds P*D packed FMAs.  One block of D
 * taps is live at a time, wave-uniform, from the scalar cache as (h, h) pairs.  LDS reads per output: ntaps/P + D
 * (the generic kernel: ntaps, and the LDS, not the FMAs, bounded it: D = 10 / 287 taps 200 us at 2^25 samples where
 * the FMAs need 30).  Samples lie in LDS as interleaved (I, Q) pairs -- a packed FMA does both rails, so odd D needs
 * no parity tricks -- in SEGMENTS of P*D (one lane's own blocks) with one pad pair behind a segment of even length:
 * lane stride odd in pairs, conflict-free ds_read_b64.  The window walks backwards one segment per loop iteration,
 * so every address of the loop body is the iteration's base pointer plus a constant.
 * A block of 256 threads makes 256*P outputs from its own copy of the span (tile + NB*D samples of history: 2-4 %
 * re-read); the blocks resident on a CU overlap each other's load and filter phases.  (A persistent variant with the
 * next tile prefetched into registers measured SLOWER -- packed /10 first stage 0.63 against 0.55 ms at 2^28 samples:
 * the filter phase of a tile is too short to cover a load -- and was dropped, profiles/r03/e_firp_persistent.txt.)
 * INFMT = IN_PACKED24: the batch and its history are 24-bit packed (stage 0): unpack -- and MIX: NCO -- while staging,
 * exactly as k_fir_generic<.., PACKED, MIX> does.
 * The body is a device function: k_fir8 runs it in the extra blocks that carry the previous batch's tail.        */
struct FirpArgs {
    const void *in;          /* batch: float2 or packed                                                   */
    const void *hist;        /* the H samples in front of it, same format                                  */
    void       *hist_out;    /* receives the last H samples of [hist | batch] (or NULL)                    */
    float      *out;         /* float2 outputs                                                             */
    const float *taps2;      /* (h[k], h[k]) pairs, zero padded to nbq*P*D taps                            */
    long long   first, n_out, n_batch;
    int         H, nbq;      /* history samples; loop iterations = tap blocks / P                          */
    GenMixArgs  mx;
};

template <int D, int P>
struct FirpGeom {
    static constexpr int PD   = P * D;
    static constexpr int SEGW = PD + ((PD & 1) ? 0 : 1);      /* pairs per segment incl. pad */
    static constexpr int TO   = 256 * P;                      /* outputs per block           */
};

constexpr int firp_p_of(int D) { return D >= 8 ? 2 : 4; }     /* outputs per lane (LDS: 256 segments of P*D pairs) */

template <int D, int P, int INFMT, bool MIX>
__device__ __forceinline__ void firp_block(const FirpArgs &a, const int bid, float2 *sdp)
{
    using G = FirpGeom<D, P>;
    f32x2 *sd = reinterpret_cast<f32x2 *>(sdp) + G::SEGW;     /* one spare segment in front, one behind the span */
    const int tid = threadIdx.x;
    const int GH = a.nbq;                                     /* history segments in front of the tile          */
    const int span = (GH + 256) * G::PD;                      /* samples staged: NB*D of history + the tile     */
    const long long t0 = (long long)bid * G::TO;              /* first output of the block                      */
    /* tile input i (0 .. TI-1; history i < 0) is staged sample GH*PD + i; output m of the tile = sum_k h[k]*in[mD+D-1-k] */
    const long long s0 = a.first + t0 * D - (D - 1) - (long long)GH * G::PD;     /* batch index of staged sample 0 */
    auto slot_of = [&](int i) {                               /* staged sample i -> LDS pair index              */
        const int seg = i / G::PD;
        return seg * G::SEGW + (i - seg * G::PD);
    };
    if (INFMT == IN_PACKED24) {
        /* Staging is most of this kernel's work at the full input rate (the /10 first stage has 7 FMAs per sample).  A
         * lane takes PAIRS of samples -- 12 bytes, one global_load_dwordx3 -- and consecutive lanes consecutive pairs, so
         * its two LDS writes land 16 bytes from its neighbours' (2-way on ds_write_b64).  With a whole group of 8 samples
         * per lane, as k_fir_generic stages them, the lanes of a write are 64 bytes apart: an 8-way bank conflict on every
         * one, 0.3 ms of LDS-array time per 2^28 samples -- what held this kernel (and holds that one) at 3.3 TB/s.
         * The samples stay the integers the unpack yields (the host folds RN(1/8388607)/256 into this stage's taps, like
         * k_fir8); the NCO takes ONE sin/cos per thread -- a thread's pairs lie 512 samples apart, pair u's phasor is the
         * thread's times LO(512 u) from the host's table, and every rotation is two packed instructions (cmul_pk); only
         * a batch's two edge blocks leave this path (the first can meet samples of the previous batch, mixed with the
         * previous tuning word): there every pair takes its own sin/cos.                                           */
        const uint8_t *inb = static_cast<const uint8_t *>(a.in);
        const uint8_t *hb8 = static_cast<const uint8_t *>(a.hist);
        const GenMixArgs &mx = a.mx;
        const long long xa = s0 & ~1LL;                        /* pairs start at even sample indices (4-byte aligned) */
        const int shift = (int)(s0 - xa);                      /* 0 or 1 */
        const int npairs = (span + shift + 1) >> 1;
        constexpr int NPF = (G::PD * (256 + 16) / 2 + 1 + 255) / 256;       /* pairs per thread */
        struct W3 { uint32_t a, b, c; };
        if (xa >= 0 && xa + 2LL * 256 * NPF <= a.n_batch) {
            /* interior block (uniform): no history, nothing beyond the batch, and every staged index one of the
             * span's pairs can touch (-1 .. span) has a slot thanks to the spare segments: straight-line code but for
             * the span's last pairs (rounds of 256 pairs wholly behind the span are skipped, uniformly)          */
            const uint8_t *src0 = inb + xa * 6 + 12 * tid;
            W3 rw[NPF];
#pragma unroll
            for (int u = 0; u < NPF; ++u)
                if (256 * u < npairs)
                    rw[u] = *reinterpret_cast<const W3 *>(src0 + 12 * 256 * u);
            float g0c = 1.0f, g0s = 0.0f;
            if (MIX)
                nco_lo((uint32_t)(mx.n0 + (unsigned long long)(xa + 2LL * tid)) * mx.freg + mx.phase_off, g0c, g0s);
            const float stc = mx.lo_c[1], sts = mx.lo_s[1];
            const f32x2 g0 = { g0c, g0s };
            static_assert(NPF <= 16, "GenMixArgs::lo512 holds 16 steps");
#pragma unroll
            for (int u = 0; u < NPF; ++u) {
                if (tid + 256 * u >= npairs)
                    continue;
                f32x2 x0 = { (float)(int32_t)__builtin_amdgcn_perm(rw[u].a, rw[u].a, 0x0201000cu),
                             (float)(int32_t)__builtin_amdgcn_perm(rw[u].b, rw[u].a, 0x0504030cu) };
                f32x2 x1 = { (float)(int32_t)__builtin_amdgcn_perm(rw[u].c, rw[u].b, 0x0403020cu),
                             (float)(int32_t)(rw[u].c & 0xffffff00u) };
                if (MIX) {
                    const f32x2 g = u > 0 ? cmul_pk_s(g0, mx.lo512_c[u], mx.lo512_s[u]) : g0;
                    x1 = cmul_pk(x1, cmul_pk_s(g, stc, sts));
                    x0 = cmul_pk(x0, g);
                }
                const float x0i = x0.x, x0q = x0.y, x1i = x1.x, x1q = x1.y;
                /* staged index of the pair's first sample, plus one segment: -1 .. span+ lands in the spare segments */
                const unsigned i0 = (unsigned)(2 * (tid + 256 * u) - shift + G::PD);
                const unsigned seg = i0 / G::PD;
                const unsigned within = i0 - seg * G::PD;
                f32x2 *dstp = sd + (int)(seg * G::SEGW + within) - G::SEGW;
                dstp[0] = f32x2{ x0i, x0q };
                dstp[within == G::PD - 1 ? 1 + G::SEGW - G::PD : 1] = f32x2{ x1i, x1q };
            }
        } else {
        W3 raw[NPF];
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int j = tid + 256 * u;
            const long long sp = xa + 2LL * j;
            const uint8_t *src = nullptr;
            if (j < npairs) {
                if (sp < 0) {
                    if (sp >= -(long long)a.H)
                        src = hb8 + (sp + a.H) * 6;
                } else if (sp < a.n_batch) {
                    src = inb + sp * 6;
                }
            }
            raw[u] = src ? *reinterpret_cast<const W3 *>(src) : W3{ 0u, 0u, 0u };
        }
        const bool has_old = MIX && xa < 0;                    /* uniform: the batch's first block only */
        const uint32_t off_old = mx.phase_off + (uint32_t)mx.n0 * (mx.freg - mx.freg_hist);
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int j = tid + 256 * u;
            if (j >= npairs)
                continue;
            float gc = 1.0f, gs = 0.0f;                        /* a batch's two edge blocks: every pair from its own phase */
            if (MIX)
                nco_lo((uint32_t)(mx.n0 + (unsigned long long)(xa + 2LL * j)) * mx.freg + mx.phase_off, gc, gs);
            /* 3 dwords = 2 samples -> MSB-aligned integers (the iq_sample placement, cf. unpack8_msb) */
            float x0i = (float)(int32_t)__builtin_amdgcn_perm(raw[u].a, raw[u].a, 0x0201000cu);
            float x0q = (float)(int32_t)__builtin_amdgcn_perm(raw[u].b, raw[u].a, 0x0504030cu);
            float x1i = (float)(int32_t)__builtin_amdgcn_perm(raw[u].c, raw[u].b, 0x0403020cu);
            float x1q = (float)(int32_t)(raw[u].c & 0xffffff00u);
            if (MIX) {
                float c0 = gc, s0p = gs, c1, s1;
                float stc = mx.lo_c[1], sts = mx.lo_s[1];
                if (has_old) {                                 /* uniform branch */
                    const long long sp = xa + 2LL * j;
                    if (sp < 0) {
                        nco_lo((uint32_t)(mx.n0 + (unsigned long long)sp) * mx.freg_hist + off_old, c0, s0p);
                        stc = mx.lo_c_hist[1];
                        sts = mx.lo_s_hist[1];
                    }
                }
                c1 = c0 * stc - s0p * sts;
                s1 = c0 * sts + s0p * stc;
                cmul(x0i, x0q, c0, s0p);
                cmul(x1i, x1q, c1, s1);
            }
            const int i0 = 2 * j - shift;                      /* staged index of the pair's first sample */
            if (i0 >= 0 && i0 + 1 < span) {
                const int seg = i0 / G::PD;
                const int within = i0 - seg * G::PD;
                f32x2 *dstp = sd + seg * G::SEGW + within;
                dstp[0] = f32x2{ x0i, x0q };
                dstp[within == G::PD - 1 ? 1 + G::SEGW - G::PD : 1] = f32x2{ x1i, x1q };
            } else {
                if (i0 >= 0 && i0 < span)
                    sd[slot_of(i0)] = f32x2{ x0i, x0q };
                if (i0 + 1 >= 0 && i0 + 1 < span)
                    sd[slot_of(i0 + 1)] = f32x2{ x1i, x1q };
            }
        }
        }
    } else {
        const float2 *in = static_cast<const float2 *>(a.in);
        const float2 *hist = static_cast<const float2 *>(a.hist);
        if (s0 >= 0 && s0 + span + 1 <= a.n_batch) {
            /* interior block: the whole span in ONE round trip -- up to 12 loads of 16 bytes (two samples) per thread, all
             * in flight before the first LDS write.  (In batches of 8 eight-byte loads a tile took three dependent round
             * trips, 6 of its 13 us: a carried tail, one block per CU, then outlasted the first stage it rides with.)  */
            const float2 *src = in + s0;
            u32x4 v[12];
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                const int ii = 2 * (tid + 256 * u);
                v[u] = ii < span ? *reinterpret_cast<const u32x4 *>(src + ii) : u32x4{ 0u, 0u, 0u, 0u };
            }
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                const int ii = 2 * (tid + 256 * u);
                if (ii < span)
                    sd[slot_of(ii)] = f32x2{ __uint_as_float(v[u].x), __uint_as_float(v[u].y) };
                if (ii + 1 < span)
                    sd[slot_of(ii + 1)] = f32x2{ __uint_as_float(v[u].z), __uint_as_float(v[u].w) };
            }
        } else {
            for (int i = tid; i < span; i += 256) {
                const long long sx = s0 + i;
                float2 v = make_float2(0.0f, 0.0f);
                if (sx < 0) {
                    if (sx >= -(long long)a.H)
                        v = hist[sx + a.H];
                } else if (sx < a.n_batch) {
                    v = in[sx];
                }
                sd[slot_of(i)] = f32x2{ v.x, v.y };
            }
        }
    }
    if (a.hist_out != nullptr && bid == 0) {                   /* the next call's history */
        if (INFMT == IN_PACKED24) {
            const int Hw = a.H * 6 / 4;
            const long long nw = a.n_batch * 6 / 4;
            const uint32_t *hw = static_cast<const uint32_t *>(a.hist), *bw = static_cast<const uint32_t *>(a.in);
            uint32_t *ow = static_cast<uint32_t *>(a.hist_out);
            for (int i = tid; i < Hw; i += 256) {
                const long long j = (long long)i + nw;
                ow[i] = j < Hw ? hw[j] : bw[j - Hw];
            }
        } else {
            const float2 *in = static_cast<const float2 *>(a.in), *hist = static_cast<const float2 *>(a.hist);
            float2 *ho = static_cast<float2 *>(a.hist_out);
            for (int i = tid; i < a.H; i += 256) {
                const long long j = (long long)i + a.n_batch;
                ho[i] = j < a.H ? hist[j] : in[j - a.H];
            }
        }
    }
    __syncthreads();
    /* the lane's own segment: tile segment `tid`, i.e. staged segment GH + tid */
    const f32x2 *seg = sd + (GH + tid) * G::SEGW;
    f32x2 blk[P][D];
#pragma unroll
    for (int pb = 1; pb < P; ++pb)
#pragma unroll
        for (int e = 0; e < D; ++e)
            blk[pb][e] = seg[pb * D + e];
    f32x2 acc[P][2];
#pragma unroll
    for (int r = 0; r < P; ++r)
        acc[r][0] = acc[r][1] = f32x2{ 0.0f, 0.0f };
    const f32x2 PDDC_CONSTANT *tp = (const f32x2 PDDC_CONSTANT *)a.taps2;
    for (int q = 0; q < a.nbq; ++q) {
#pragma unroll
        for (int jj = 0; jj < P; ++jj) {
            const int slot = (P - jj) % P;
#pragma unroll
            for (int e = 0; e < D; ++e)
                blk[slot][e] = jj == 0 ? seg[e] : seg[-G::SEGW + (P - jj) * D + e];
#pragma unroll
            for (int e = 0; e < D; ++e) {
                const f32x2 h = tp[jj * D + e];
#pragma unroll
                for (int r = 0; r < P; ++r)
                    acc[r][e & 1] = __builtin_elementwise_fma(h, blk[(r - jj + P) % P][D - 1 - e], acc[r][e & 1]);
            }
        }
        seg -= G::SEGW;
        tp += P * D;
    }
    const long long m = t0 + (long long)P * tid;
    f32x2 *op = reinterpret_cast<f32x2 *>(a.out) + m;
#pragma unroll
    for (int r = 0; r < P; ++r)
        if (m + r < a.n_out)
            op[r] = acc[r][0] + acc[r][1];
}

/* one block of a carried tail (GenTail): the generic decimator's body, or k_firp's for decimations 4, 5, 10 */
__device__ __forceinline__ void run_tail_block(const GenTail &t, const int bid, float2 *sd_tail)
{
    if (t.kind == 1) {
        FirpArgs fa;
        fa.in = t.in;
        fa.hist = t.hist;
        fa.hist_out = t.hist_out;
        fa.out = t.out;
        fa.taps2 = t.taps2;
        fa.first = t.first;
        fa.n_out = t.n_out;
        fa.n_batch = t.n_batch;
        fa.H = t.H;
        fa.nbq = t.nbq;
        if (t.D == 4)
            firp_block<4, firp_p_of(4), IN_F32C, false>(fa, bid, sd_tail);
        else if (t.D == 5)
            firp_block<5, firp_p_of(5), IN_F32C, false>(fa, bid, sd_tail);
        else if (t.D == 8)
            firp_block<8, firp_p_of(8), IN_F32C, false>(fa, bid, sd_tail);
        else
            firp_block<10, firp_p_of(10), IN_F32C, false>(fa, bid, sd_tail);
    } else {
        const GenMixArgs nomix = {};
        fir_generic_body<1, false, false>(reinterpret_cast<const float2 *>(t.in), reinterpret_cast<const float2 *>(t.hist),
                                          t.H, t.first, t.n_out, t.D, (const float PDDC_CONSTANT *)t.taps, t.ntaps,
                                          reinterpret_cast<float2 *>(t.out), t.span, t.a,
                                          reinterpret_cast<float2 *>(t.hist_out), t.n_batch, nomix, bid, sd_tail, 256);
    }
}

/* ======================================================================== */
/* k_fir8 : fused unpack + mix + polyphase decimate-by-8                    */
/* ======================================================================== */
/* LDS plane layout: group G (8 samples) lives at float offset
 *   goff(G) = 8 + 8*G + 4*(G/8)      (G >= 0),   group -1 at offset 0.
 * Sample position p = i + 8*NTB - 1 (i = input index relative to the tile),
 * group = p >> 3, slot = p & 7: the one-sample rotation puts the window
 * x[8m-7 .. 8m] of every output m into ONE aligned group.
 * A 4-float pad every 8 groups (64 samples) makes the lane stride of the FIR
 * reads 68 floats = one 16-byte slot off the 256-byte bank row: conflict-free
 * ds_read_b128.  R=8: a lane owns one 64-sample segment.  R=4: a lane owns a
 * 32-sample half segment, and a wave takes the even (or the odd) halves so its
 * lanes still sit 68 floats apart and see the pad at the same window offset. */
template <int R>
__device__ __forceinline__ int goff(int G)
{
    return 8 + 8 * G + 4 * (G / 8);
}

template <int NTB, int R, int NT = 256>
struct Fir8Geom {
    static constexpr int TI      = 4 * NT * R;          /* inputs per tile: NT/2 lanes per plane x 8R */
    static constexpr int TO      = TI / 8;              /* outputs per tile         */
    static constexpr int GT      = TI / 8;              /* new groups per tile      */
    static constexpr int GPT     = GT / NT;             /* groups per thread / tile */
    static constexpr int NG      = GT + NTB;            /* groups incl. history     */
    static constexpr int PLANE   = 8 + 8 * NG + 4 * (NG / 8) + 8;   /* floats      */
    static constexpr int OT      = 2 * TO;              /* output staging, floats   */
    static constexpr int LDS_FLT = 2 * PLANE + OT;
};

/* second (fused) decimate-by-8 stage: its input is the tile's TO stage-1
 * outputs, kept in LDS in the same rotated / padded plane layout */
template <int NTB2, int R, int NTB = 4>
struct Fir8Geom2 {
    static constexpr int TO2   = 16 * R;                 /* stage-2 outputs per tile (TO/8)   */
    static constexpr int R2    = TO2 / 64;               /* per lane of waves 0 (I) and 1 (Q) */
    static constexpr int GT2   = 16 * R;                 /* new input groups per tile (TO/8)  */
    static constexpr int NG2   = GT2 + NTB2;
    /* the PORCH of a chunk's first tile: the 8*NTB2 first-stage outputs in front of it (the second stage's history) are
     * computed from the 8*NTB2 + NTB input groups in front of the tile, which for that one tile lie -- in the FIRST
     * stage's padded layout -- in the plane set the second stage is not using                                      */
    static constexpr int NPG   = 8 * NTB2 + NTB;
    static constexpr int PORCH = (8 + 8 * NPG + 4 * (NPG / 8) + 8 + 3) & ~3;
    /* plain layout (offset 8 + position): the second stage is ~3 % of the work, a
     * 2-way bank conflict on its reads is cheaper than the LDS a pad would cost  */
    static constexpr int PLANE0 = 8 + 8 * NG2 + 8;
    static constexpr int PLANE = NTB2 > 0 ? (PLANE0 > PORCH ? PLANE0 : PORCH) : 0;
    /* the second stage's outputs of BT tiles leave in ONE burst of BT*TO2*8 bytes (fir8_block.inc "store_tile2") */
    static constexpr int BT    = R == 4 ? (NTB <= 4 ? 16 : 12) : 8;
    static constexpr int BURST = NTB2 > 0 ? BT * 2 * TO2 : 0;
    /* two plane SETS, alternating tile by tile (the history of tile t+1 is carried into the other set while all four
     * waves may still be reading tile t's), two staging areas for the two halves of the tap sum, the burst buffer */
    static constexpr int LDS_FLT = NTB2 > 0 ? 4 * PLANE + 4 * TO2 + BURST : 0;
};

size_t fir8_lds_bytes(int ntb, int R)
{
    const int NG = 1024 * R / 8 + ntb;
    const int plane = 8 + 8 * NG + 4 * (NG / 8) + 8;
    return (size_t)(2 * plane + 2 * 128 * R) * sizeof(float);
}

/* one 8-sample group: global words -> (mixed) planar floats */
template <int INFMT, bool MIX, int NW>
__device__ __forceinline__ void group_to_float(const u32x4 (&raw)[NW], float (&xi)[8], float (&xq)[8],
                                               unsigned long long nabs, const Fir8Args &p)
{
    if (INFMT == IN_PACKED24) {
        const uint32_t w[12] = { raw[0].x, raw[0].y, raw[0].z, raw[0].w, raw[1].x, raw[1].y,
                                 raw[1].z, raw[1].w, raw[2].x, raw[2].y, raw[2].z, raw[2].w };
        int32_t I[8], Q[8];
        unpack8_msb(w, I, Q);
        /* no scaling here: the host folds RN(1/8388607)/256 into this stage's taps
         * (kFir8PackedTapScale), which saves 16 multiplies per group; the samples travel
         * through mix and LDS as the MSB-aligned integers, exact in fp32               */
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            xi[e] = (float)I[e];
            xq[e] = (float)Q[e];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32x4 f = raw[k < NW ? k : 0];
            xi[2 * k]     = __uint_as_float(f.x);
            xq[2 * k]     = __uint_as_float(f.y);
            xi[2 * k + 1] = __uint_as_float(f.z);
            xq[2 * k + 1] = __uint_as_float(f.w);
        }
    }
    if (MIX)     /* zero-filled groups stay zero; the index wraps correctly for negative offsets */
        mix8(xi, xq, nabs, p);
}

/* same, with the local oscillator value of the group's first sample supplied */
template <int INFMT, int NW>
__device__ __forceinline__ void group_to_float_lo(const u32x4 (&raw)[NW], float (&xi)[8], float (&xq)[8], float cb,
                                                  float sb, const Fir8Args &p)
{
    group_to_float<INFMT, false, NW>(raw, xi, xq, 0ull, p);
    mix8_lo(xi, xq, cb, sb, p);
}

/* rotated LDS write of group v: e=0 -> slot 7 of group v-1 ; e=1..7 -> slots 0..6 of group v */
template <int R>
__device__ __forceinline__ void group_to_lds(float *sI, float *sQ, int v, const float (&xi)[8],
                                             const float (&xq)[8])
{
    const int o_prev = (v == 0) ? 7 : goff<R>(v - 1) + 7;
    const int o_cur  = goff<R>(v);
    sI[o_prev] = xi[0];
    sQ[o_prev] = xq[0];
    *reinterpret_cast<float4 *>(sI + o_cur) = make_float4(xi[1], xi[2], xi[3], xi[4]);
    *reinterpret_cast<float4 *>(sQ + o_cur) = make_float4(xq[1], xq[2], xq[3], xq[4]);
    *reinterpret_cast<float2 *>(sI + o_cur + 4) = make_float2(xi[5], xi[6]);
    *reinterpret_cast<float2 *>(sQ + o_cur + 4) = make_float2(xq[5], xq[6]);
    sI[o_cur + 6] = xi[7];
    sQ[o_cur + 6] = xq[7];
}

/* The register sliding window of one lane: R outputs over R+NTB-1 aligned
 * 8-sample groups.  PAR (R=4 only) says whether the lane's half segment starts
 * 4 groups into a padded 8-group row, which moves the pad inside the window.  */
/* Partial sums per output: the packed FMA already splits a dot product into even and odd
 * terms; filters of 200+ taps split once more (pair index parity), which halves the length of
 * every fp32 accumulation chain again (255 taps: max error 5.7e-7 -> 3e-7 of full scale).   */
template <int NTB>
struct FirAcc {
    static constexpr int N = NTB >= 32 ? 2 : 1;
};

template <int NTB, int R, int PAR, bool PADDED = true>
__device__ __forceinline__ void fir_window(const float *base, const float PDDC_CONSTANT *hb,
                                           f32x2 (&acc)[R][FirAcc<NTB>::N])
{
    constexpr int NA = FirAcc<NTB>::N;
#pragma unroll
    for (int ub = 0; ub < R + NTB - 1; ++ub) {
        const int go = 8 * ub + (PADDED ? 4 * ((ub + PAR * 4) >> 3) : 0);
        const f32x4 d0 = *reinterpret_cast<const f32x4 *>(base + go);
        const f32x4 d1 = *reinterpret_cast<const f32x4 *>(base + go + 4);
        const f32x2 xs[4] = { { d0.x, d0.y }, { d0.z, d0.w }, { d1.x, d1.y }, { d1.z, d1.w } };
        /* pair index outer, output inner: neighbouring instructions touch
         * different accumulators (a dependent v_pk_fma pair costs an s_nop) */
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int j = r + NTB - 1 - ub;
                if (j >= 0 && j < NTB) {
                    const f32x2 PDDC_CONSTANT *h = reinterpret_cast<const f32x2 PDDC_CONSTANT *>(hb + 8 * j);
                    acc[r][i % NA] = __builtin_elementwise_fma(h[i], xs[i], acc[r][i % NA]);
                }
            }
        }
    }
}


/* The same sliding window with the loops exchanged: tap block j outermost, the R groups
 * that meet it (ub = r + NTB-1-j) held in VGPRs and shifted by one group per step.  Only one
 * tap block is live at a time (8 SGPRs + the next s_load) instead of R of them: the R=8
 * kernels otherwise keep 64 tap SGPRs live and spill (119 v_readlane per tile at 255 taps).  */
static constexpr bool kTapOuterR8 = true;
template <int NTB, int R, int PAR, bool PADDED = true>
__device__ __forceinline__ void fir_window_tap_outer(const float *base, const float PDDC_CONSTANT *hb,
                                                     f32x2 (&acc)[R][FirAcc<NTB>::N])
{
    constexpr int NA = FirAcc<NTB>::N;
    auto load_group = [&](int ub, f32x2 (&w)[4]) {
        const int go = 8 * ub + (PADDED ? 4 * ((ub + PAR * 4) >> 3) : 0);
        const f32x4 d0 = *reinterpret_cast<const f32x4 *>(base + go);
        const f32x4 d1 = *reinterpret_cast<const f32x4 *>(base + go + 4);
        w[0] = f32x2{ d0.x, d0.y };
        w[1] = f32x2{ d0.z, d0.w };
        w[2] = f32x2{ d1.x, d1.y };
        w[3] = f32x2{ d1.z, d1.w };
    };
    f32x2 W[R][4];
#pragma unroll
    for (int g = 0; g < R; ++g)
        load_group(g, W[g]);
#pragma unroll
    for (int j = NTB - 1; j >= 0; --j) {
        const f32x2 PDDC_CONSTANT *h = reinterpret_cast<const f32x2 PDDC_CONSTANT *>(hb + 8 * j);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < R; ++r)
                acc[r][i % NA] = __builtin_elementwise_fma(h[i], W[r][i], acc[r][i % NA]);
        if (j > 0) {
#pragma unroll
            for (int g = 0; g + 1 < R; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    W[g][i] = W[g + 1][i];
            load_group(NTB - j + R - 1, W[R - 1]);
        }
    }
}

/* Persistent grid with a two-level tile schedule.  Tiles are grouped into
 * CHUNKS of consecutive tiles; inside a chunk the FIR history is carried in LDS.
 *   static part : block b first owns the S tiles [b*S, (b+1)*S)
 *   dynamic part: the remaining tiles [nblk*S, ntiles) are cut into chunks of K
 *                 tiles that the blocks take from an atomic counter (p.sched[0])
 *                 as they run dry.  The two blocks resident on a CU do not run at
 *                 the same speed (the older wave wins the issue arbitration: with
 *                 equal static shares one block finished 15-25 % before its
 *                 neighbour, which then ran alone and badly overlapped); the
 *                 dynamic tail lets them finish together.
 *   round robin : (S < 0, round 6; the fused pair's default from 2^27 samples on) no
 *                 static runs: the batch is chunks of K tiles, chunk j goes to block
 *                 j mod nblk for the first JD = -S chunks and comes from the counter
 *                 after that -- all blocks read and write ONE window of the batch,
 *                 which is what the pair's small write stream wants (fir8_block.inc
 *                 "store_tile2", NOTEBOOK R6.1).
 * The first tile of a chunk takes its history from global memory (the previous
 * 8*NTB input samples, or p.hist for tile 0), prefetched with the tile itself --
 * fused pair: with them the PORCH, the 8*NTB2 sample groups whose first-stage
 * outputs are the second stage's history.
 * The counter is taken one tile ahead (thread 0, published through LDS) so its
 * latency is never waited for, and the last block to leave resets it.
 *
 * Per tile t:  U  unpack the prefetched registers into the LDS planes
 *              S  coalesced global stores of the previous tile (staged by its F)
 *              -- barrier A --
 *              P  issue the next tile's global loads (in flight during F)
 *              F  FIR from LDS, results to the output staging area
 *              -- barrier B --
 *              C  the last NTB groups become the next tile's history (copied by
 *                 the very threads that overwrite them in the next U, so no
 *                 third barrier is needed).                                */
/* (The ablation builds -- loads / FIR / stores removed --, the in-kernel clock probe and the 128-thread variant that the
 * measurements in NOTEBOOK.md rounds 1-3 5 come from are not in this file: tools/ubench/fir8_probe_and_ablations.patch.)          */
/* NT = threads per block.  256 (4 waves: two per plane) is the default; 128 (R = 8 only: one wave per
 * plane, half the tile, half the LDS) lets four independent blocks share a CU instead of two.       */
/* SL3 > 0 (FUSE3): a third stage (plain decimate-by-d3 FIR, struct Fir8Stage3) runs on the second stage's outputs in LDS
 * as well, SL3 taps per wave and tile (its work is sliced into the tile loop, see "third stage" below).          */
template <int NTB, int R, int INFMT, bool MIX, int NTB2, int NT = 256, int SL3 = 0>
__global__ __launch_bounds__(NT, 2) void k_fir8(Fir8Args p, int ntiles, int S, int K)
{
#include "fir8_block.inc"
}

/* one batch of EACH of up to kFir8ManyMax streams that share a GPU (the drop-in API's virtual receivers): blockIdx.y is
 * the stream, every stream has its own argument record (input, histories, tuning word and phase, output, scheduler words)
 * and the same batch length, so ntiles / S / K are common.  One launch instead of one per stream: a 2^22-sample batch
 * keeps a 512-block grid busy for a dozen microseconds, less than the launch gap in front of it.                   */
template <int NTB, int R, int INFMT, bool MIX, int NTB2>
__global__ __launch_bounds__(256, 2) void k_fir8_many(Fir8Many m, int ntiles, int S, int K)
{
    constexpr int NT = 256;
    constexpr int SL3 = 0;
    const Fir8Args &p = m.a[blockIdx.y];
#include "fir8_block.inc"
}

bool fir8_supported(int ntb, int R)
{
    return (R == 4 || R == 8) && (ntb == 4 || ntb == 8 || ntb == 16 || ntb == 32);
}

/* Persistent grid: 2 blocks per CU x 256 CUs.  R=4 could keep 4 blocks (16 waves)
 * resident per CU, but 512 concurrent streams measured 1-3 % faster than 1024
 * (grids that are not a multiple of 256 lose ~10 % to imbalance).            */
static constexpr int kFir8DefaultBlocks = 512;
static int g_fir8_blocks = 0;       /* override (development) */

/* two-level tile schedule of k_fir8: S static tiles per block, the rest in dynamic
 * chunks of K tiles.  Measured (profiles/r01/v7_schedule_sweep.txt): 127 taps R=4
 * 0.398 -> 0.385 ms with 15-25 % dynamic in 4-tile chunks, 255 taps R=8 0.538 ->
 * 0.511 ms with 25 % in 2-tile chunks; single-tile chunks lose (one atomic per
 * tile on one address).  A chunk of the fused pair starts with a PORCH (the second stage's history
 * computed from the 544 samples in front of the chunk: 0.37 of a tile's time; rounds 1-5 recomputed the whole
 * tile in front): 8 % in chunks of 8.  Large batches of the pair take the round-robin walk below instead.
 * Development overrides: PDDC_FIR8_DYN_PCT (share of the tiles handed out
 * dynamically), PDDC_FIR8_CHUNK (K).                                              */
struct Fir8Sched {
    int nblocks, S, K;
};

static Fir8Sched fir8_schedule(int ntiles, int R, bool fused, int NT = 256, int group = 0)
{
    /* (process-wide knobs, pddc_set_tunable: a test can switch schedules inside one process) */
    const int v = tunables().fir8_dyn_pct.load();
    const int dyn_pct = v < 0 ? -1 : (v > 100 ? 100 : v);
    const int chunk = tunables().fir8_chunk.load();
    Fir8Sched sc;
    /* 128-thread blocks: four per CU, tiles half as long (chunks of twice as many) */
    /* small batches (BASELINE config 5's low end): one block per tile up to one block per CU, then about three tiles
     * per block -- a block with a single tile overlaps nothing, its load, filter and store phases just follow each other
     * (bench.py --log2n 22 / 23 --steps 2000, 127 taps: 2^22 samples 19.0 -> 14.4 us with 256 instead of 512 blocks, 2^23 22.4 -> 20.2
     * with 340; x320 cascade 2^21 31.5 -> 25.4) -- and the full two blocks per CU from 1536 tiles on                */
    const int full = kFir8DefaultBlocks * (256 / NT);
    int want = g_fir8_blocks;
    if (want <= 0) {
        want = ntiles / 3;
        if (want < full / 2)
            want = full / 2;
        if (want > full)
            want = full;
    }
    sc.nblocks = ntiles < want ? ntiles : want;
    if (group > 0) {              /* fused third stage: every chunk but the batch's last is whole groups of tiles */
        const int cap = ntiles / group > 0 ? ntiles / group : 1;
        if (sc.nblocks > cap)
            sc.nblocks = cap;
    }
    sc.K = chunk > 0 ? chunk : (fused ? 8 : (R == 4 ? 4 : 2) * (256 / NT));
    if (group > 0)
        sc.K = chunk > 0 ? (chunk + group - 1) / group * group : 2 * group;
    /* fused pair: a dynamic chunk starts with a porch, so only a small share pays (round 2, same-box sweep under the
     * arena placement with whole warm-up tiles: static 0.2897 ms, 5-10 % in chunks of 8 0.2863-0.2867, 20 % 0.293) */
    const int pct = dyn_pct >= 0 ? dyn_pct : (fused ? 8 : 20);
    sc.S = (int)((long long)ntiles * (100 - pct) / 100 / sc.nblocks);
    /* The round-robin walk (S = -JD: chunk j -> block j mod nblocks for j < JD, the rest from the counter; fir8_block.inc).
     * Default for the fused pair from 64 tiles per block on (2^27 samples): chunks of 16 / 32 tiles, a porch each (0.37 of a
     * tile's time), 10 % of them dynamic.  Same box, 2^28 samples, the pair's kernel with its write stream in the read
     * stream's HBM extent class / in another (profiles/r06/b_ab_walks.txt): static runs + dynamic tail 0.299-0.312 /
     * 0.287-0.288 ms, round robin 0.295-0.298 / 0.288-0.291 -- it is the placement-insensitive one (3 % against 5-8 %), and
     * what a host gets without placing anything is the first figure.  Not with the fused third stage.               */
    const int walk = tunables().fir8_walk.load();
    if (group == 0 && (walk == 1 || (walk < 0 && fused && ntiles >= 64 * sc.nblocks))) {
        if (chunk <= 0 && fused)
            sc.K = ntiles >= 128 * sc.nblocks ? 32 : 16;
        const int rr_pct = dyn_pct >= 0 ? dyn_pct : (fused ? 10 : 0);
        const long long nd = ((long long)ntiles + sc.K - 1) / sc.K;
        long long jd = nd * (100 - rr_pct) / 100 / sc.nblocks * sc.nblocks;
        if (jd < sc.nblocks)
            jd = sc.nblocks < nd ? sc.nblocks : nd;          /* every block starts with a chunk of its own */
        if (jd > 0x3fffffff)
            jd = 0x3fffffff;
        sc.S = -(int)jd;
    }
    if (group > 0) {
        sc.S -= sc.S % group;
        if (sc.S == 0 && chunk <= 0)
            sc.K = group;         /* small batch: one group per block, all of them handed out dynamically */
        for (;;) {                /* a bounded number of chunks: each has a seam slot and a flag */
            const long long nd = ((long long)ntiles - (long long)sc.nblocks * sc.S + sc.K - 1) / sc.K;
            if ((sc.S > 0 ? sc.nblocks : 0) + nd <= kFused3MaxChunks)
                break;
            sc.K += group;
        }
    }
    return sc;
}

void fir8_schedule_query(long long n_in, int R, bool fused, int NT, int *ntiles, int *nblocks, int *S, int *K, int group)
{
    const long long TI = 4LL * NT * R;
    const int nt = (int)((n_in + TI - 1) / TI);
    const Fir8Sched sc = fir8_schedule(nt, R, fused, NT, group);
    *ntiles = nt;
    *nblocks = sc.nblocks;
    *S = sc.S;
    *K = sc.K;
}

template <int NTB, int R, int NT = 256>
static hipError_t launch_fir8_t(InFmt fmt, bool mix, const Fir8Args &a, hipStream_t s)
{
    using G = Fir8Geom<NTB, R, NT>;
    const size_t lds = (size_t)G::LDS_FLT * sizeof(float);
    const long long ntiles_ll = (a.n_in + G::TI - 1) / G::TI;
    if (ntiles_ll <= 0)
        return hipSuccess;
    if (ntiles_ll > 0x7fffffffLL)
        return hipErrorInvalidValue;
    const int ntiles = (int)ntiles_ll;
    if (a.sched == nullptr)
        return hipErrorInvalidValue;
    const Fir8Sched sc = fir8_schedule(ntiles, R, false, NT);
    /* a carried tail (packed R = 4 first stages only): its blocks behind the persistent ones, LDS for the larger of the two */
    const bool carry_ok = R == 4 && NT == 256 && fmt == IN_PACKED24;
    if (a.tail.nblocks < 0 || (a.tail.nblocks > 0 && (!carry_ok || a.tail.lds > kCarryLdsCap)))
        return hipErrorInvalidValue;
    const size_t lds_launch = a.tail.nblocks > 0 && a.tail.lds > lds ? a.tail.lds : lds;
    const size_t lds_attr = lds > kCarryLdsCap ? lds : kCarryLdsCap;
    const dim3 grid((unsigned)(sc.nblocks + a.tail.nblocks)), blk(NT);
#define PDDC_LAUNCH(FMT, MIXV)                                                                    \
    do {                                                                                          \
        static unsigned long long attr_done = 0;   /* one bit per device: the attribute is per device */ \
        int dev__ = 0;                                                                            \
        (void)hipGetDevice(&dev__);                                                               \
        if (!(attr_done >> (dev__ & 63) & 1ull)) {                                                \
            hipError_t e = hipFuncSetAttribute(                                                   \
                reinterpret_cast<const void *>(&k_fir8<NTB, R, FMT, MIXV, 0, NT>),                \
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_attr);                       \
            if (e != hipSuccess)                                                                  \
                return e;                                                                         \
            attr_done |= 1ull << (dev__ & 63);                                                    \
        }                                                                                         \
        hipLaunchKernelGGL((k_fir8<NTB, R, FMT, MIXV, 0, NT>), grid, blk, lds_launch, s, a, ntiles, sc.S, sc.K); \
    } while (0)
    if (fmt == IN_PACKED24) {
        if (mix)
            PDDC_LAUNCH(IN_PACKED24, true);
        else
            PDDC_LAUNCH(IN_PACKED24, false);
    } else if constexpr (NT == 256) {
        PDDC_LAUNCH(IN_F32C, false);
    } else {
        return hipErrorInvalidValue;         /* 128-thread blocks exist for the packed first stage only */
    }
#undef PDDC_LAUNCH
    return hipGetLastError();
}

/* fused pair: packed input -> [mix] -> /8 (NTB blocks) -> /8 (<= 64 taps) */
template <int NTB, int R>
static hipError_t launch_fir8_fused2_t(bool mix, const Fir8Args &a, hipStream_t s)
{
    using G = Fir8Geom<NTB, R>;
    using G2 = Fir8Geom2<8, R, NTB>;
    const size_t lds = (size_t)(2 * G::PLANE + G2::LDS_FLT) * sizeof(float);
    if (a.n_in <= 0 || a.n_in % G::TI)
        return hipErrorInvalidValue;             /* whole tiles only */
    const long long ntiles_ll = a.n_in / G::TI;
    if (ntiles_ll > 0x7fffffffLL)
        return hipErrorInvalidValue;
    const int ntiles = (int)ntiles_ll;
    if (a.sched == nullptr)
        return hipErrorInvalidValue;
    const Fir8Sched sc = fir8_schedule(ntiles, R, true);
    if (a.tail.nblocks < 0 || (a.tail.nblocks > 0 && (R != 4 || a.tail.lds > kCarryLdsCap)))
        return hipErrorInvalidValue;
    const size_t lds_launch = a.tail.nblocks > 0 && a.tail.lds > lds ? a.tail.lds : lds;
    const size_t lds_attr = lds > kCarryLdsCap ? lds : kCarryLdsCap;
    const dim3 grid((unsigned)(sc.nblocks + a.tail.nblocks)), blk(256);
#define PDDC_LAUNCH2(MIXV)                                                                        \
    do {                                                                                          \
        static unsigned long long attr_done = 0;                                                  \
        int dev__ = 0;                                                                            \
        (void)hipGetDevice(&dev__);                                                               \
        if (!(attr_done >> (dev__ & 63) & 1ull)) {                                                \
            hipError_t e = hipFuncSetAttribute(                                                   \
                reinterpret_cast<const void *>(&k_fir8<NTB, R, IN_PACKED24, MIXV, 8>),            \
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_attr);                       \
            if (e != hipSuccess)                                                                  \
                return e;                                                                         \
            attr_done |= 1ull << (dev__ & 63);                                                    \
        }                                                                                         \
        hipLaunchKernelGGL((k_fir8<NTB, R, IN_PACKED24, MIXV, 8>), grid, blk, lds_launch, s, a, ntiles, \
                           sc.S, sc.K);                                                           \
    } while (0)
    if (mix)
        PDDC_LAUNCH2(true);
    else
        PDDC_LAUNCH2(false);
#undef PDDC_LAUNCH2
    return hipGetLastError();
}

size_t fir8_fused2_lds_bytes(int ntb, int R)
{
    const int NG = 1024 * R / 8 + ntb;
    const int plane = 8 + 8 * NG + 4 * (NG / 8) + 8;
    const int npg = 64 + ntb, porch = (8 + 8 * npg + 4 * (npg / 8) + 8 + 3) & ~3;
    const int plane0 = 8 + 8 * (16 * R + 8) + 8, plane2 = plane0 > porch ? plane0 : porch;
    const int bt = R == 4 ? (ntb <= 4 ? 16 : 12) : 8;
    return (size_t)(2 * plane + 4 * plane2 + 4 * 16 * R + bt * 2 * 16 * R) * sizeof(float);
}

bool fir8_fused2_supported(int ntb, int ntb2, int R)
{
    return (R == 4 || R == 8) && (ntb == 4 || ntb == 8) && ntb2 > 0 && ntb2 <= 8;
}

/* ---- fused cascade: packed input -> [mix] -> /8 -> /8 -> /d3, one kernel ---------------------------------------- */
int fir8_fused3_max_chunks() { return kFused3MaxChunks; }

static size_t fir8_fused3_lds3(const Fir8Stage3 &q, int TO2)
{
    return (size_t)(4 * 64 + 3 * (q.padf + q.h + q.g * TO2) + kPend3 * 64) * sizeof(f32x2) + (4 * kPend3 + 4) * sizeof(int);
}

bool fir8_fused3_geometry(int ntb, int ntb2, int R, Fir8Stage3 *q)
{
    if (!q || R != 4 || ntb2 != 8 || !fir8_fused2_supported(ntb, ntb2, R))
        return false;                          /* instantiated for R = 4 (the fused pair's own default) */
    const int TO2 = 16 * R;
    if (q->d < 2 || q->ntaps < 1 || q->h < q->ntaps - 1 || (q->h & 7) || q->h > 512)
        return false;
    int a = TO2, b = q->d;
    while (b) {
        const int t = a % b;
        a = b;
        b = t;
    }
    q->g = q->d / a;                           /* g*TO2 is the smallest whole number of tiles that is a multiple of d */
    q->ng = TO2 / a;
    if (q->ng < 1 || q->ng > 64 || q->g * TO2 < q->h)        /* a one-group chunk must hold a whole history */
        return false;
    q->spl = 4;                                /* one tap segment per wave, one slice of it per tile of a group */
    {
        static const int slices[] = { 9, 16 };                    /* the instantiated slice lengths */
        const int need = ((q->ntaps + 3) / 4 + q->g - 1) / q->g;
        q->sl = 0;
        for (int v : slices)
            if (v >= need) {
                q->sl = v;
                break;
            }
        if (q->sl == 0)
            return false;
    }
    q->seglen = q->g * q->sl;
    if (q->seglen > 128 || (q->seglen > 64 && 64 % q->sl != 0))
        return false;                          /* a wave keeps its taps in two VGPRs, one tap per lane each; a slice
                                                  takes its taps from one of them                                  */
    q->padf = q->spl * q->seglen > q->h ? (q->spl * q->seglen - q->h + 1) & ~1 : 0;
    q->seam_stride = (8 * q->h + 255) & ~255;
    return fir8_fused3_lds3(*q, TO2) <= 40u * 1024u;
}

template <int NTB, int R, int SL3>
static hipError_t launch_fir8_fused3_t(bool mix, const Fir8Args &a, hipStream_t s)
{
    using G = Fir8Geom<NTB, R>;
    using G2 = Fir8Geom2<8, R, NTB>;
    const Fir8Stage3 &q = a.s3;
    const size_t lds = (size_t)(2 * G::PLANE + G2::LDS_FLT) * sizeof(float) + fir8_fused3_lds3(q, G2::TO2);
    constexpr size_t lds_cap = 96u * 1024u;
    if (a.n_in <= 0 || a.n_in % G::TI || lds > lds_cap)
        return hipErrorInvalidValue;             /* whole tiles only */
    const long long ntiles_ll = a.n_in / G::TI;
    if (ntiles_ll > 0x7fffffffLL)
        return hipErrorInvalidValue;
    const int ntiles = (int)ntiles_ll;
    if (a.sched == nullptr || q.taps == nullptr || q.hist == nullptr || q.out == nullptr || q.seam == nullptr ||
        q.flags == nullptr || q.g < 1 || q.off < 0 || q.off >= q.d || q.sl != SL3 || q.seglen != q.g * SL3)
        return hipErrorInvalidValue;
    const Fir8Sched sc = fir8_schedule(ntiles, R, true, 256, q.g);
    const dim3 grid((unsigned)sc.nblocks), blk(256);
#define PDDC_LAUNCH3(MIXV)                                                                        \
    do {                                                                                          \
        static unsigned long long attr_done = 0;                                                  \
        int dev__ = 0;                                                                            \
        (void)hipGetDevice(&dev__);                                                               \
        if (!(attr_done >> (dev__ & 63) & 1ull)) {                                                \
            hipError_t e = hipFuncSetAttribute(                                                   \
                reinterpret_cast<const void *>(&k_fir8<NTB, R, IN_PACKED24, MIXV, 8, 256, SL3>), \
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap);                        \
            if (e != hipSuccess)                                                                  \
                return e;                                                                         \
            attr_done |= 1ull << (dev__ & 63);                                                    \
        }                                                                                         \
        hipLaunchKernelGGL((k_fir8<NTB, R, IN_PACKED24, MIXV, 8, 256, SL3>), grid, blk, lds, s, a, ntiles, \
                           sc.S, sc.K);                                                           \
    } while (0)
    if (mix)
        PDDC_LAUNCH3(true);
    else
        PDDC_LAUNCH3(false);
#undef PDDC_LAUNCH3
    return hipGetLastError();
}

hipError_t launch_fir8_fused3(int ntb, int R, bool mix, const Fir8Args &a, hipStream_t s)
{
#define PDDC_CASE3(N, SL)                                                                          \
    if (ntb == N && R == 4 && a.s3.sl == SL)                                                       \
        return launch_fir8_fused3_t<N, 4, SL>(mix, a, s)
    PDDC_CASE3(4, 9);
    PDDC_CASE3(4, 16);
    PDDC_CASE3(8, 9);
    PDDC_CASE3(8, 16);
#undef PDDC_CASE3
    return hipErrorInvalidValue;
}

hipError_t launch_fir8_fused2(int ntb, int R, bool mix, const Fir8Args &a, hipStream_t s)
{
    if (ntb == 4 && R == 4) return launch_fir8_fused2_t<4, 4>(mix, a, s);
    if (ntb == 8 && R == 4) return launch_fir8_fused2_t<8, 4>(mix, a, s);
    if (ntb == 4 && R == 8) return launch_fir8_fused2_t<4, 8>(mix, a, s);
    if (ntb == 8 && R == 8) return launch_fir8_fused2_t<8, 8>(mix, a, s);
    return hipErrorInvalidValue;
}

/* ---- several streams in one launch ------------------------------------------------------------------------------- */
template <int NTB, int NTB2, int R>
static hipError_t launch_fir8_many_t(bool mix, const Fir8Many &m, int n, hipStream_t s)
{
    using G = Fir8Geom<NTB, R>;
    using G2 = Fir8Geom2<8, R, NTB>;
    const size_t lds = NTB2 ? (size_t)(2 * G::PLANE + G2::LDS_FLT) * sizeof(float) : (size_t)G::LDS_FLT * sizeof(float);
    const long long n_in = m.a[0].n_in;
    if (n_in <= 0 || (NTB2 && n_in % G::TI))
        return hipErrorInvalidValue;
    for (int i = 0; i < n; ++i) {
        if (m.a[i].n_in != n_in || m.a[i].sched == nullptr || m.a[i].tail.nblocks != 0)
            return hipErrorInvalidValue;
        for (int j = 0; j < i; ++j)
            if (m.a[j].sched == m.a[i].sched)
                return hipErrorInvalidValue;     /* every stream counts its own tiles */
    }
    const long long ntiles_ll = (n_in + G::TI - 1) / G::TI;
    if (ntiles_ll > 0x7fffffffLL)
        return hipErrorInvalidValue;
    const int ntiles = (int)ntiles_ll;
    /* Every stream gets the grid and the tile schedule a launch of its own would have: with the NCO on, the first tile of
     * a chunk mixes its history through another expression than a tile that inherits it in LDS, equal to a bit or two --
     * the same chunks mean the SAME BITS as the per-stream path.  The n grids queue behind each other on the chip; no
     * block of this kernel waits for another one.  (Tried: 512/n blocks per stream drawing the same chunks through the
     * chunk counter -- 8 streams 242-254 vs 246 GS/s through the API, no gain, not kept.)                          */
    const Fir8Sched sc = fir8_schedule(ntiles, R, NTB2 != 0);
    const dim3 grid((unsigned)sc.nblocks, (unsigned)n), blk(256);
#define PDDC_LAUNCHM(MIXV)                                                                        \
    do {                                                                                          \
        static unsigned long long attr_done = 0;                                                  \
        int dev__ = 0;                                                                            \
        (void)hipGetDevice(&dev__);                                                               \
        if (!(attr_done >> (dev__ & 63) & 1ull)) {                                                \
            hipError_t e = hipFuncSetAttribute(                                                   \
                reinterpret_cast<const void *>(&k_fir8_many<NTB, R, IN_PACKED24, MIXV, NTB2>),    \
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
            if (e != hipSuccess)                                                                  \
                return e;                                                                         \
            attr_done |= 1ull << (dev__ & 63);                                                    \
        }                                                                                         \
        hipLaunchKernelGGL((k_fir8_many<NTB, R, IN_PACKED24, MIXV, NTB2>), grid, blk, lds, s, m, ntiles, sc.S, sc.K); \
    } while (0)
    if (mix)
        PDDC_LAUNCHM(true);
    else
        PDDC_LAUNCHM(false);
#undef PDDC_LAUNCHM
    return hipGetLastError();
}

bool fir8_many_supported(int kind, int ntb, int R)
{
    /* the shapes the pipeline itself picks: R = 4 for first stages of up to 64 taps (alone or as the fused pair),
     * R = 8 for the long ones (128, 256 taps) */
    if (kind == 2)
        return R == 4 && (ntb == 4 || ntb == 8);
    if (kind == 1)
        return (R == 4 && (ntb == 4 || ntb == 8)) || (R == 8 && (ntb == 16 || ntb == 32));
    return false;
}

hipError_t launch_fir8_many(int kind, int ntb, int R, bool mix, const Fir8Many &m, int n, hipStream_t s)
{
    if (n < 1 || n > kFir8ManyMax || !fir8_many_supported(kind, ntb, R))
        return hipErrorInvalidValue;
    if (kind == 2)
        return ntb == 4 ? launch_fir8_many_t<4, 8, 4>(mix, m, n, s) : launch_fir8_many_t<8, 8, 4>(mix, m, n, s);
    switch (ntb) {
    case 4: return launch_fir8_many_t<4, 0, 4>(mix, m, n, s);
    case 8: return launch_fir8_many_t<8, 0, 4>(mix, m, n, s);
    case 16: return launch_fir8_many_t<16, 0, 8>(mix, m, n, s);
    default: return launch_fir8_many_t<32, 0, 8>(mix, m, n, s);
    }
}

void fir8_set_grid_blocks(int nblocks) { g_fir8_blocks = nblocks > 0 ? nblocks : 0; }

bool fir8_nt_supported(int ntb, int R, int NT)
{
    (void)ntb;
    (void)R;
    return NT == 256;
}

hipError_t launch_fir8(int ntb, int R, InFmt fmt, bool mix, const Fir8Args &a, hipStream_t s, int NT)
{
    if (fmt == IN_F32C && mix)
        return hipErrorInvalidValue;
    if (NT != 256)
        return hipErrorInvalidValue;
#define PDDC_CASE(N, RR)                                                                          \
    if (ntb == N && R == RR)                                                                      \
        return launch_fir8_t<N, RR>(fmt, mix, a, s)
    PDDC_CASE(4, 8);
    PDDC_CASE(8, 8);
    PDDC_CASE(16, 8);
    PDDC_CASE(32, 8);
    PDDC_CASE(4, 4);
    PDDC_CASE(8, 4);
    PDDC_CASE(16, 4);
    PDDC_CASE(32, 4);
#undef PDDC_CASE
    return hipErrorInvalidValue;
}

/* ======================================================================== */
/* k_fir_generic : any decimation, any tap count, float2 in / float2 out    */
/* ======================================================================== */
/* The low-rate path (stages 2.. of a cascade, and odd first stages).  A block of
 * NT threads makes NT*P outputs; its input span ((NT*P-1)*D + ntaps samples) is
 * staged as interleaved float2 in LDS.  A thread owns P CONSECUTIVE outputs
 * q..q+P-1 and walks the union of their windows once, 8 samples per step: the
 * sample x[q*D - j] serves output q+p with tap p*D + j, a wave-uniform index,
 * so the taps come through the scalar cache (s_load of 8 at a time from a table
 * zero-padded by 3*D+8 on both sides) and one ds_read_b64 feeds P packed FMAs.
 * The lane stride S = P*D must be odd in 8-byte units for conflict-free reads:
 * the launcher takes P odd for odd D; for even S sample i sits at i + (i >> a),
 * 2^a the largest power of two in S (stride S + S/2^a, odd).  The steps are
 * aligned to 8 samples of the block's local index, so for a >= 3 (and for no
 * pad at all, a = 31) the 8 reads of a step are one address plus immediates;
 * the scalar unit -- one per CU -- is the bottleneck of this kernel otherwise
 * (per-sample uniform index arithmetic made it 29 us instead of 13).
 * Block 0 also writes the next call's history (the last H samples of
 * [hist | batch]) when asked.                                                */
/* PACKED: the batch and its history are 24-bit packed samples (stage 0): the block unpacks --
 * and, MIX, mixes with the NCO -- while it stages its span, so the float2 intermediate of an
 * unpack kernel (8 B written + 8 B read per input sample) never exists: 6 + 8/D bytes per input
 * sample for ANY decimation, not only the fused decimate-by-8 (the 1.6 MS/s plan is 10*5).    */
struct GenShape {
    int NT = 0, P = 0, span = 0, a = 31;
    size_t lds = 0;
};

/* shape: P outputs per thread (fewer LDS reads per output; odd for odd D so that the lane
 * stride needs no pad), NT threads per block, picked with a small cost model: a round of
 * resident blocks costs the staging round trips (~2 us per 8 loads per thread) plus the
 * tap loop of the waves sharing a SIMD; a second, nearly empty round of blocks doubles a
 * kernel this short.  NT == 0: no shape fits the 160 KiB of LDS.                         */
static GenShape pick_generic_shape(long long n_out, int D, int ntaps, int ncu)
{
    static const int shapes[][2] = { { 256, 4 }, { 128, 4 }, { 64, 4 }, { 256, 3 }, { 128, 3 }, { 64, 3 },
                                     { 256, 2 }, { 128, 2 }, { 64, 2 }, { 256, 1 }, { 64, 1 } };
    GenShape g;
    double best = -1.0;
    const int force_nt = tunables().gen_shape_nt.load(), force_p = tunables().gen_shape_p.load();   /* development */
    for (const auto &sh : shapes) {
        const int nt = sh[0], pp = sh[1];
        if (force_nt ? (nt != force_nt || pp != force_p) : ((D & 1) ? (pp == 2 || pp == 4) : pp == 3))
            continue;
        const long long sp_ll = (long long)(nt * pp - 1) * D + ntaps;
        if (sp_ll > (1 << 20))
            continue;
        const int sp = (int)sp_ll;
        const int S = pp * D;
        const int aa = (S & 1) ? 31 : __builtin_ctz((unsigned)S);
        const size_t l = (size_t)(sp + 8 + ((sp + 8) >> aa) + 2) * sizeof(float2);
        if (l > (pp == 1 ? 160u : 64u) * 1024u)
            continue;
        const long long blocks = (n_out + nt * pp - 1) / (nt * pp);
        long long per_cu = (long long)(160 * 1024 / (l + 512));
        if (per_cu > 2048 / nt)
            per_cu = 2048 / nt;
        if (per_cu > 16)
            per_cu = 16;
        if (per_cu < 1)
            per_cu = 1;
        const long long rounds = (blocks + per_cu * ncu - 1) / (per_cu * ncu);
        long long resident = (blocks + ncu - 1) / ncu;          /* blocks sharing a CU in a round */
        if (resident > per_cu)
            resident = per_cu;
        const double waves_per_simd = (double)resident * nt / 256.0;
        const double t_stage = 2.0 * (double)((sp + 8 * nt - 1) / (8 * nt));
        const double per_sample = (aa >= 3 ? pp + 1.5 : pp + 4.0);       /* issue slots per window sample */
        const double t_fir = (double)((pp - 1) * D + ntaps) * per_sample * 4.0 / 2000.0 *
                             (waves_per_simd < 1.0 ? 1.0 : waves_per_simd);
        double t = (double)rounds * (t_stage + t_fir);
        /* measured (profiles/r02/f_generic_shape_sweep2.txt): with the tap loop as it is now, one output per
         * thread and 256 threads is the best or within a few % of the best shape for every stage of the rate
         * plans -- many waves hide the scalar-cache and LDS latencies of a step better than register reuse pays */
        if (nt == 256 && pp == 1 && !force_nt)
            t *= 0.25;
        if (best < 0.0 || t < best) {
            best = t;
            g.NT = nt, g.P = pp, g.span = sp, g.a = aa, g.lds = l;
        }
    }
    return g;
}

/* can launch_fir_generic stage a block of this decimator in LDS at all? */
bool fir_generic_supported(int D, int ntaps)
{
    return D >= 1 && ntaps >= 1 && pick_generic_shape(1 << 20, D, ntaps, 256).NT != 0;
}

/* `taps` is the DUPLICATED table -- entry k is the pair (h[k], h[k]), 8 bytes -- and must be readable
 * (zeros) over entries [-3*D - 8, ntaps + 3*D + 8): the pipeline uploads its tap tables that way.  hist_out (or NULL) receives the last H samples of
 * [hist(H) | in(n_batch)]; it must not alias hist.                                  */
static hipError_t launch_fir_generic_any(const void *in, const void *hist, int H, long long first, long long n_out,
                                         int D, const float *taps, int ntaps, float *out, void *hist_out,
                                         long long n_batch, int packed_mode /* 0 float2, 1 packed, 2 packed + mix */,
                                         const GenMixArgs &mx, hipStream_t s);

hipError_t launch_fir_generic(const float *in, const float *hist, int H, long long first, long long n_out,
                              int D, const float *taps, int ntaps, float *out, float *hist_out,
                              long long n_batch, hipStream_t s)
{
    GenMixArgs mx = {};
    return launch_fir_generic_any(in, hist, H, first, n_out, D, taps, ntaps, out, hist_out, n_batch, 0, mx, s);
}

hipError_t launch_fir_generic_packed(const void *in_packed, const void *hist_packed, int H, long long first,
                                     long long n_out, int D, const float *taps, int ntaps, float *out,
                                     void *hist_out_packed, long long n_batch, bool mix, unsigned long long n0,
                                     uint32_t freg, uint32_t phase_off, uint32_t freg_hist, const float *lo_c,
                                     const float *lo_s, const float *lo_c_hist, const float *lo_s_hist, hipStream_t s)
{
    if ((H & 7) || (n_batch & 7))
        return hipErrorInvalidValue;
    GenMixArgs mx = {};
    mx.n0 = n0;
    mx.freg = freg;
    mx.phase_off = phase_off;
    mx.freg_hist = freg_hist;
    for (int e = 0; e < 8; ++e) {
        mx.lo_c[e] = lo_c ? lo_c[e] : 1.0f;
        mx.lo_s[e] = lo_s ? lo_s[e] : 0.0f;
        mx.lo_c_hist[e] = lo_c_hist ? lo_c_hist[e] : 1.0f;
        mx.lo_s_hist[e] = lo_s_hist ? lo_s_hist[e] : 0.0f;
    }
    return launch_fir_generic_any(in_packed, hist_packed, H, first, n_out, D, taps, ntaps, out, hist_out_packed, n_batch,
                                  mix ? 2 : 1, mx, s);
}

static hipError_t launch_fir_generic_any(const void *in, const void *hist, int H, long long first, long long n_out,
                                         int D, const float *taps, int ntaps, float *out, void *hist_out,
                                         long long n_batch, int packed_mode, const GenMixArgs &mx, hipStream_t s)
{
    if (n_out <= 0)
        return hipSuccess;
    int dev = 0;
    (void)hipGetDevice(&dev);
    static int ncu_of[64] = { 0 };
    if (ncu_of[dev & 63] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        ncu_of[dev & 63] = n;
    }
    const int ncu = ncu_of[dev & 63];
    const GenShape g = pick_generic_shape(n_out, D, ntaps, ncu);
    const int NT = g.NT, P = g.P, span = g.span, a = g.a;
    const size_t lds = g.lds;
    if (NT == 0)
        return hipErrorInvalidValue;           /* span does not fit LDS even one output per thread */
    const dim3 grid((unsigned)((n_out + (long long)NT * P - 1) / ((long long)NT * P))), blk((unsigned)NT);
#define PDDC_GEN3(PP, PK, MX)                                                                      \
    do {                                                                                          \
        static int attr_lds[64] = { 0 };            /* per device */                             \
        if ((int)lds > attr_lds[dev & 63]) {                                                      \
            /* the whole LDS once (launch_gen_tail sets the same attribute of the same function) */   \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fir_generic<PP, PK, MX>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            if (e != hipSuccess)                                                                  \
                return e;                                                                         \
            attr_lds[dev & 63] = 160 * 1024;                                                      \
        }                                                                                         \
        hipLaunchKernelGGL((k_fir_generic<PP, PK, MX>), grid, blk, lds, s, reinterpret_cast<const float2 *>(in), \
                           reinterpret_cast<const float2 *>(hist), H, first, n_out, D,            \
                           (const float PDDC_CONSTANT *)taps, ntaps, reinterpret_cast<float2 *>(out), span, a, \
                           reinterpret_cast<float2 *>(hist_out), n_batch, mx);                    \
    } while (0)
#define PDDC_GEN(PP)                                                                               \
    do {                                                                                          \
        if (packed_mode == 2)                                                                     \
            PDDC_GEN3(PP, true, true);                                                            \
        else if (packed_mode == 1)                                                                \
            PDDC_GEN3(PP, true, false);                                                           \
        else                                                                                      \
            PDDC_GEN3(PP, false, false);                                                          \
    } while (0)
    if (P == 4)
        PDDC_GEN(4);
    else if (P == 3)
        PDDC_GEN(3);
    else if (P == 2)
        PDDC_GEN(2);
    else
        PDDC_GEN(1);
#undef PDDC_GEN
#undef PDDC_GEN3
    return hipGetLastError();
}

template <int D, int P, int INFMT, bool MIX>
__global__ __launch_bounds__(256) void k_firp(FirpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float2 sdp[];
    /* (Carrying the previous batch's tail in this grid, as k_fir8 does, was tried for the packed first stage -- tail
     * blocks behind the grid start too late to overlap anything, spread through it they cost the first stage more
     * than they save: 10*5 plan 0.507 -> 0.536 ms -- and is not done.)                                           */
    const int bid = (int)blockIdx.x;
    firp_block<D, P, INFMT, MIX>(a, bid, sdp);
}

/* which D are built: first stages /10 and /5, tails /4 /5 /8 /10 (the reference's rate plans) */
int firp_nbq(int D, int ntaps)
{
    const int P = firp_p_of(D);
    return ((ntaps + D - 1) / D + P - 1) / P;
}

size_t firp_lds_bytes(int D, int ntaps)
{
    const int P = firp_p_of(D), pd = P * D, segw = pd + ((pd & 1) ? 0 : 1);
    return (size_t)(firp_nbq(D, ntaps) + 256 + 2) * segw * sizeof(float2);
}

bool firp_supported(int D, int ntaps)
{
    if (tunables().no_firp.load())
        return false;
    if (!(D == 4 || D == 5 || D == 8 || D == 10) || ntaps < 1)
        return false;
    if ((long long)(firp_nbq(D, ntaps) + 256) * firp_p_of(D) * D > 6144)       /* a tile's span: 12 x 2 samples per thread */
        return false;
    return firp_lds_bytes(D, ntaps) <= 96u * 1024u;
}

int firp_taps_len(int D, int ntaps)            /* taps the duplicated table must hold (zero padded) */
{
    return firp_nbq(D, ntaps) * firp_p_of(D) * D;
}

template <int D, int P = firp_p_of(D)>
static hipError_t launch_firp_t(int infmt, bool mix, const FirpArgs &a, int ntaps, hipStream_t s)
{
    using G = FirpGeom<D, P>;
    const size_t lds = (size_t)(a.nbq + 256 + 2) * G::SEGW * sizeof(float2);
    const long long ntiles = (a.n_out + G::TO - 1) / G::TO;
    if (ntiles > 0x7fffffffLL)
        return hipErrorInvalidValue;
    const size_t lds_launch = lds;
    const dim3 grid((unsigned)ntiles), blk(256);
    int dev = 0;
    (void)hipGetDevice(&dev);
#define PDDC_FIRP(FMT, MX)                                                                         \
    do {                                                                                          \
        static bool attr_done[64] = { false };                                                    \
        if (!attr_done[dev & 63]) {                                                               \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_firp<D, P, FMT, MX>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
            if (e != hipSuccess)                                                                  \
                return e;                                                                         \
            attr_done[dev & 63] = true;                                                           \
        }                                                                                         \
        hipLaunchKernelGGL((k_firp<D, P, FMT, MX>), grid, blk, lds_launch, s, a);                 \
    } while (0)
    if (infmt == IN_PACKED24) {
        if (mix)
            PDDC_FIRP(IN_PACKED24, true);
        else
            PDDC_FIRP(IN_PACKED24, false);
    } else {
        PDDC_FIRP(IN_F32C, false);
    }
#undef PDDC_FIRP
    return hipGetLastError();
}

/* same contract as launch_fir_generic / launch_fir_generic_packed; `taps2` = (h, h) pairs zero padded to
 * firp_taps_len(D, ntaps) taps                                                                                 */
hipError_t launch_firp(int infmt, bool mix, const void *in, const void *hist, int H, long long first, long long n_out,
                       int D, const float *taps2, int ntaps, float *out, void *hist_out, long long n_batch,
                       const GenMixArgs *mx, hipStream_t s)
{
    if (n_out <= 0)
        return hipSuccess;
    if (!firp_supported(D, ntaps) || (infmt == IN_PACKED24 && ((H & 7) || (n_batch & 7))))
        return hipErrorInvalidValue;
    FirpArgs a;
    a.in = in;
    a.hist = hist;
    a.hist_out = hist_out;
    a.out = out;
    a.taps2 = taps2;
    a.first = first;
    a.n_out = n_out;
    a.n_batch = n_batch;
    a.H = H;
    a.nbq = firp_nbq(D, ntaps);
    a.mx = mx ? *mx : GenMixArgs{};
    /* development: PDDC_FIRP_PACKED_P=1 gives the packed /10 first stage tiles of 256 outputs (one per lane) */
    const int pp = tunables().firp_packed_p.load();
    if (D == 10 && infmt == IN_PACKED24 && pp == 1) {
        a.nbq = (ntaps + D - 1) / D;
        return launch_firp_t<10, 1>(infmt, mix, a, ntaps, s);
    }
    if (D == 4) return launch_firp_t<4>(infmt, mix, a, ntaps, s);
    if (D == 5) return launch_firp_t<5>(infmt, mix, a, ntaps, s);
    if (D == 8) return launch_firp_t<8>(infmt, mix, a, ntaps, s);      /* (a long decimate-by-8 SECOND stage: 1 MS/s = 10 * 8) */
    if (D == 10) return launch_firp_t<10>(infmt, mix, a, ntaps, s);
    return hipErrorInvalidValue;
}

hipError_t launch_firp_packed(const void *in_packed, const void *hist_packed, int H, long long first, long long n_out,
                              int D, const float *taps2, int ntaps, float *out, void *hist_out_packed, long long n_batch,
                              bool mix, unsigned long long n0, uint32_t freg, uint32_t phase_off, uint32_t freg_hist,
                              const float *lo_c, const float *lo_s, const float *lo_c_hist, const float *lo_s_hist,
                              hipStream_t s)
{
    GenMixArgs mx = {};
    mx.n0 = n0;
    mx.freg = freg;
    mx.phase_off = phase_off;
    mx.freg_hist = freg_hist;
    for (int e = 0; e < 8; ++e) {
        mx.lo_c[e] = lo_c ? lo_c[e] : 1.0f;
        mx.lo_s[e] = lo_s ? lo_s[e] : 0.0f;
        mx.lo_c_hist[e] = lo_c_hist ? lo_c_hist[e] : 1.0f;
        mx.lo_s_hist[e] = lo_s_hist ? lo_s_hist[e] : 0.0f;
    }
    for (int u = 0; u < 16; ++u) {                      /* exp(-j 2 pi (512 u freg mod 2^32) / 2^32), in double */
        const uint32_t ph = (uint32_t)(512ull * (unsigned long long)u * freg);
        const double th = 6.283185307179586476925 * (double)ph / 4294967296.0;
        mx.lo512_c[u] = (float)std::cos(th);
        mx.lo512_s[u] = (float)(-std::sin(th));
    }
    return launch_firp(IN_PACKED24, mix, in_packed, hist_packed, H, first, n_out, D, taps2, ntaps, out, hist_out_packed,
                       n_batch, &mx, s);
}

bool gen_tail_shape(GenTail *t, size_t lds_cap, bool have_taps2)
{
    if (!t || t->D < 1 || t->ntaps < 1 || t->n_out < 1)
        return false;
    if (have_taps2 && firp_supported(t->D, t->ntaps) && firp_lds_bytes(t->D, t->ntaps) <= lds_cap) {
        t->kind = 1;
        t->nbq = firp_nbq(t->D, t->ntaps);
        t->lds = (unsigned)firp_lds_bytes(t->D, t->ntaps);
        t->nblocks = (int)((t->n_out + 256 * firp_p_of(t->D) - 1) / (256 * firp_p_of(t->D)));
        return true;
    }
    t->kind = 0;
    const long long sp = 255LL * t->D + t->ntaps;                 /* 256 threads, one output each */
    if (sp > (1 << 20))
        return false;
    t->span = (int)sp;
    t->a = (t->D & 1) ? 31 : __builtin_ctz((unsigned)t->D);
    const size_t l = (size_t)(t->span + 8 + ((t->span + 8) >> t->a) + 2) * sizeof(float2);
    if (l > lds_cap)
        return false;
    t->lds = (unsigned)l;
    t->nblocks = (int)((t->n_out + 255) / 256);
    return true;
}

hipError_t launch_gen_tail(const GenTail &t, hipStream_t s)
{
    if (t.nblocks <= 0)
        return hipSuccess;
    if (t.kind == 1)
        return launch_firp(IN_F32C, false, t.in, t.hist, t.H, t.first, t.n_out, t.D, t.taps2, t.ntaps, t.out, t.hist_out,
                           t.n_batch, nullptr, s);
    int dev = 0;
    (void)hipGetDevice(&dev);
    static bool attr_done[64] = { false };
    if (!attr_done[dev & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fir_generic<1, false, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess)
            return e;
        attr_done[dev & 63] = true;
    }
    const GenMixArgs mx = {};
    hipLaunchKernelGGL((k_fir_generic<1, false, false>), dim3((unsigned)t.nblocks), dim3(256), t.lds, s,
                       reinterpret_cast<const float2 *>(t.in), reinterpret_cast<const float2 *>(t.hist), t.H, t.first,
                       t.n_out, t.D, (const float PDDC_CONSTANT *)t.taps, t.ntaps, reinterpret_cast<float2 *>(t.out),
                       t.span, t.a, reinterpret_cast<float2 *>(t.hist_out), t.n_batch, mx);
    return hipGetLastError();
}

/* the tails of several streams in one launch: blockIdx.y is the stream */
__global__ __launch_bounds__(256) void k_gen_tail_many(GenTailMany m)
{
    extern __shared__ __attribute__((aligned(16))) float2 sd_tail_many[];
    const GenTail &t = m.t[blockIdx.y];
    if ((int)blockIdx.x >= t.nblocks)
        return;
    run_tail_block(t, (int)blockIdx.x, sd_tail_many);
}

hipError_t launch_gen_tail_many(const GenTailMany &m, int n, hipStream_t s)
{
    if (n < 1 || n > kFir8ManyMax)
        return hipErrorInvalidValue;
    int nb = 0;
    unsigned lds = 0;
    for (int i = 0; i < n; ++i) {
        if (m.t[i].kind != m.t[0].kind || m.t[i].D != m.t[0].D || m.t[i].ntaps != m.t[0].ntaps)
            return hipErrorInvalidValue;
        nb = m.t[i].nblocks > nb ? m.t[i].nblocks : nb;
        lds = m.t[i].lds > lds ? m.t[i].lds : lds;
    }
    if (nb <= 0)
        return hipSuccess;
    int dev = 0;
    (void)hipGetDevice(&dev);
    static bool attr_done[64] = { false };
    if (!attr_done[dev & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gen_tail_many),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess)
            return e;
        attr_done[dev & 63] = true;
    }
    hipLaunchKernelGGL(k_gen_tail_many, dim3((unsigned)nb, (unsigned)n), dim3(256), lds, s, m);
    return hipGetLastError();
}

/* ======================================================================== */
/* k_resample : rational L/M polyphase resampler on float2 (low rate)       */
/* ======================================================================== */
/* y[m] = sum_j g[j*L + ph] * x[n - j],  n = floor(m*M/L), ph = (m*M) mod L:
 * upsample by L, filter with g (ntaps = K*L), keep every M-th sample.  Used
 * for the reference's non-integer rates (48k/95k/96k/192k from 1-2 MS/s), so
 * one output per thread with taps and samples straight from L2 is enough.    */
__global__ __launch_bounds__(256) void k_resample(const float *__restrict__ in, const float *__restrict__ hist,
                                                   int H, unsigned long long consumed, unsigned long long m0,
                                                   long long n_out, int L, int M, const float *__restrict__ taps,
                                                   int ntaps, float *__restrict__ out)
{
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= n_out)
        return;
    const unsigned long long t = (m0 + (unsigned long long)q) * (unsigned long long)M;
    const long long n = (long long)(t / (unsigned long long)L) - (long long)consumed;   /* index into the batch */
    const int ph = (int)(t % (unsigned long long)L);
    float ar0 = 0.0f, ai0 = 0.0f, ar1 = 0.0f, ai1 = 0.0f;
    int j = 0;
    for (int k = ph; k < ntaps; k += L, ++j) {
        const long long xi = n - j;
        float2 v = make_float2(0.0f, 0.0f);
        if (xi >= 0)
            v = *reinterpret_cast<const float2 *>(in + 2 * xi);
        else if (xi >= -(long long)H)
            v = *reinterpret_cast<const float2 *>(hist + 2 * (xi + H));
        const float h = taps[k];
        if (j & 1) {
            ar1 = fmaf(h, v.x, ar1);
            ai1 = fmaf(h, v.y, ai1);
        } else {
            ar0 = fmaf(h, v.x, ar0);
            ai0 = fmaf(h, v.y, ai0);
        }
    }
    *reinterpret_cast<float2 *>(out + 2 * q) = make_float2(ar0 + ar1, ai0 + ai1);
}

/* The same with the block's input span staged in LDS and the taps in polyphase order
 * g[ph][j] = h[j*L + ph] (rows of Kp floats, zero padded, Kp % 4 == 0): every load of the tap loop
 * is independent of the others (float4 rows from L1/L2, samples from LDS), where the direct form
 * above walks 50+ dependent L2 round trips per output (23 us for the 80 k outputs of a 2^26-sample
 * batch of the 96 kS/s plan, 15 % of that plan's step).  Block 0 also leaves the next call's
 * history, so no separate update kernel runs behind a rational stage.                          */
__global__ __launch_bounds__(256) void k_resample_lds(const float2 *__restrict__ in, const float2 *__restrict__ hist,
                                                       int H, unsigned long long consumed, unsigned long long m0,
                                                       long long n_out, int L, int M, const float *__restrict__ gpoly,
                                                       int K, int Kp, float2 *__restrict__ out, int span_max,
                                                       float2 *__restrict__ hist_out, long long n_batch)
{
    extern __shared__ __attribute__((aligned(16))) float2 xs[];
    const int tid = threadIdx.x, NT = blockDim.x;
    const long long q0 = (long long)blockIdx.x * NT;
    /* first input the block needs: n(q0) - (K - 1); n(m) = floor(m*M/L) - consumed (batch-relative) */
    const unsigned long long t0 = (m0 + (unsigned long long)q0) * (unsigned long long)M;
    const long long n_lo = (long long)(t0 / (unsigned long long)L) - (long long)consumed - (K - 1);
    for (int i = tid; i < span_max; i += NT) {
        const long long xi = n_lo + i;
        float2 v = make_float2(0.0f, 0.0f);
        if (xi < 0) {
            if (xi >= -(long long)H)
                v = hist[xi + H];
        } else if (xi < n_batch) {
            v = in[xi];
        }
        xs[i] = v;
    }
    if (hist_out != nullptr && blockIdx.x == 0) {
        for (int i = tid; i < H; i += NT) {
            const long long j = (long long)i + n_batch;
            hist_out[i] = j < H ? hist[j] : in[j - H];
        }
    }
    __syncthreads();
    const long long q = q0 + tid;
    if (q >= n_out)
        return;
    const unsigned long long t = (m0 + (unsigned long long)q) * (unsigned long long)M;
    const int nl = (int)((long long)(t / (unsigned long long)L) - (long long)consumed - n_lo);   /* LDS index of x[n] */
    const int ph = (int)(t % (unsigned long long)L);
    const f32x4 *g = reinterpret_cast<const f32x4 *>(gpoly + (size_t)ph * Kp);
    f32x2 acc[4] = { { 0.0f, 0.0f }, { 0.0f, 0.0f }, { 0.0f, 0.0f }, { 0.0f, 0.0f } };
    for (int j4 = 0; j4 < Kp; j4 += 4) {
        const f32x4 h4 = g[j4 >> 2];
        const float hh[4] = { h4.x, h4.y, h4.z, h4.w };
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = nl - j4 - u;                      /* taps beyond K are zero, so clamp the index only */
            const float2 xv = xs[idx >= 0 ? idx : 0];
            acc[u] = __builtin_elementwise_fma(f32x2{ hh[u], hh[u] }, f32x2{ xv.x, xv.y }, acc[u]);
        }
    }
    const f32x2 sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    out[q] = make_float2(sum.x, sum.y);
}

hipError_t launch_resample(const float *in, const float *hist, int H, unsigned long long consumed,
                           unsigned long long m0, long long n_out, int L, int M, const float *taps, int ntaps,
                           float *out, hipStream_t s)
{
    if (n_out <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(k_resample, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s, in, hist, H, consumed,
                       m0, n_out, L, M, taps, ntaps, out);
    return hipGetLastError();
}

bool resample_lds_supported(int L, int M, int ntaps)
{
    if (L < 1 || M < 1)
        return false;
    const int K = (ntaps + L - 1) / L;
    const long long span = (long long)(63) * M / L + K + 3;          /* at least 64 outputs per block must fit */
    return span * 8 <= 64 * 1024;
}

hipError_t launch_resample_lds(const float *in, const float *hist, int H, unsigned long long consumed,
                               unsigned long long m0, long long n_out, int L, int M, const float *gpoly, int K, int Kp,
                               float *out, float *hist_out, long long n_batch, hipStream_t s)
{
    if (n_out <= 0)
        return hipSuccess;
    int NT = 256;
    long long span = 0;
    for (; NT >= 64; NT >>= 1) {
        span = (long long)(NT - 1) * M / L + K + 3;
        if (span * 8 <= 64 * 1024)
            break;
    }
    if (NT < 64)
        return hipErrorInvalidValue;
    const size_t lds = (size_t)span * sizeof(float2);
    int dev = 0;
    (void)hipGetDevice(&dev);
    static int attr_lds[64] = { 0 };
    if ((int)lds > attr_lds[dev & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_resample_lds),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
        attr_lds[dev & 63] = (int)lds;
    }
    hipLaunchKernelGGL(k_resample_lds, dim3((unsigned)((n_out + NT - 1) / NT)), dim3((unsigned)NT), lds, s,
                       reinterpret_cast<const float2 *>(in), reinterpret_cast<const float2 *>(hist), H, consumed, m0, n_out,
                       L, M, gpoly, K, Kp, reinterpret_cast<float2 *>(out), (int)span,
                       reinterpret_cast<float2 *>(hist_out), n_batch);
    return hipGetLastError();
}

/* ======================================================================== */
/* k_pack24 : float32 I/Q -> 24-bit packed wire format (the inverse of A2)  */
/* ======================================================================== */
/* code = clamp(rint(x * 8388607), -2^23, 2^23-1), ties to even, NaN -> -2^23;
 * pack(unpack(c)) == c for all 2^24 codes.  Runs at the decimated rate, so a
 * plain 8-samples-per-thread layout is enough.                                 */
__device__ __forceinline__ uint32_t quant24(float x)
{
    float v = __builtin_rintf(x * 8388607.0f);
    v = fminf(fmaxf(v, -8388608.0f), 8388607.0f);       /* fmaxf(NaN, lo) = lo */
    return (uint32_t)(int32_t)v & 0xffffffu;
}

__global__ __launch_bounds__(256) void k_pack24(const float *__restrict__ in, uint8_t *__restrict__ out, long long ns)
{
    const long long ngroups = (ns + 7) >> 3;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < ngroups; g += (long long)gridDim.x * 256) {
        const long long s0 = g << 3;
        uint32_t c[16];
        if (s0 + 8 <= ns) {
            const float4 *src = reinterpret_cast<const float4 *>(in + 2 * s0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 f = src[k];
                c[4 * k] = quant24(f.x); c[4 * k + 1] = quant24(f.y);
                c[4 * k + 2] = quant24(f.z); c[4 * k + 3] = quant24(f.w);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                c[e] = (s0 + e / 2 < ns) ? quant24(in[2 * s0 + e]) : 0u;
        }
        uint32_t w[12];
#pragma unroll
        for (int h = 0; h < 4; ++h) {       /* 4 codes (I0 Q0 I1 Q1) -> 3 dwords */
            const uint32_t a = c[4 * h], b = c[4 * h + 1], d = c[4 * h + 2], e = c[4 * h + 3];
            w[3 * h]     = a | (b << 24);
            w[3 * h + 1] = (b >> 8) | (d << 16);
            w[3 * h + 2] = (d >> 16) | (e << 8);
        }
        if (s0 + 8 <= ns) {
            uint4 *dst = reinterpret_cast<uint4 *>(out + s0 * 6);
#pragma unroll
            for (int k = 0; k < 3; ++k)
                dst[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
        } else {
            const long long nb = (ns - s0) * 6;
            for (int b = 0; b < 48 && b < nb; ++b)
                out[s0 * 6 + b] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
        }
    }
}

hipError_t launch_pack24(const float *in, long long ns, void *out, hipStream_t s)
{
    if (ns <= 0)
        return hipSuccess;
    long long blocks = (((ns + 7) >> 3) + 255) / 256;
    if (blocks > 4096)
        blocks = 4096;
    hipLaunchKernelGGL(k_pack24, dim3((unsigned)blocks), dim3(256), 0, s, in, static_cast<uint8_t *>(out), ns);
    return hipGetLastError();
}

/* ======================================================================== */
/* k_hist_update                                                            */
/* ======================================================================== */
__global__ __launch_bounds__(256) void k_hist_update(uint32_t *dst, const uint32_t *hist, int Hw,
                                                      const uint32_t *batch, long long nw)
{
    /* words; new[i] = concat(hist, batch)[i + nw], i < Hw.  Single block:
     * gather everything into registers before the first store.             */
    constexpr int MAXPT = 16;
    uint32_t v[MAXPT];
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < MAXPT; ++k) {
        const int i = tid + 256 * k;
        if (i < Hw) {
            const long long j = (long long)i + nw;
            v[k] = j < Hw ? hist[j] : batch[j - Hw];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MAXPT; ++k) {
        const int i = tid + 256 * k;
        if (i < Hw)
            dst[i] = v[k];
    }
}

hipError_t launch_hist_update(void *dst, const void *hist, int H, const void *batch, long long n,
                              int elem_bytes, hipStream_t s)
{
    if (H <= 0 || n <= 0)
        return hipSuccess;
    if ((elem_bytes * H) % 4 != 0 || ((long long)elem_bytes * n) % 4 != 0)
        return hipErrorInvalidValue;
    const int Hw = elem_bytes * H / 4;
    if (Hw > 256 * 16)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_hist_update, dim3(1), dim3(256), 0, s, static_cast<uint32_t *>(dst),
                       static_cast<const uint32_t *>(hist), Hw, static_cast<const uint32_t *>(batch),
                       (long long)elem_bytes * n / 4);
    return hipGetLastError();
}

/* ======================================================================== */
/* k_stream_copy : the measured copy ceiling (bench.py roofline.copy_ceiling) */
/* ======================================================================== */
/* 16 bytes per lane, four loads in flight per thread, nontemporal stores: the plain
 * streaming copy the HBM figures of MI355X_MICROARCH.md are quoted for.            */
__global__ __launch_bounds__(256) void k_stream_copy(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst,
                                                      long long n16)
{
    const long long stride = (long long)gridDim.x * 1024;
    for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < n16; i += stride) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            v[u] = (i + 256 * u < n16) ? src[i + 256 * u] : u32x4{ 0u, 0u, 0u, 0u };
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + 256 * u < n16)
                __builtin_nontemporal_store(v[u], dst + i + 256 * u);
    }
}

/* read stream + write stream of equal size over two buffers of any sizes (both wrap): the probe of
 * pddc_malloc_apart -- how well do THESE two buffers stream against each other?                    */
__global__ __launch_bounds__(256) void k_stream_probe(const u32x4 *__restrict__ src, long long src16,
                                                       u32x4 *__restrict__ dst, long long dst16, long long total16)
{
    const long long stride = (long long)gridDim.x * 1024;
    for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < total16; i += stride) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            v[u] = src[(i + 256 * u) % src16];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            __builtin_nontemporal_store(v[u], dst + (i + 256 * u) % dst16);
    }
}

hipError_t launch_stream_probe(const void *src, size_t src_bytes, void *dst, size_t dst_bytes, size_t total_bytes,
                               hipStream_t s)
{
    const long long s16 = (long long)(src_bytes / 16), d16 = (long long)(dst_bytes / 16), t16 = (long long)(total_bytes / 16);
    if (s16 <= 0 || d16 <= 0 || t16 <= 0)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_stream_probe, dim3(4096), dim3(256), 0, s, static_cast<const u32x4 *>(src), s16,
                       static_cast<u32x4 *>(dst), d16, t16);
    return hipGetLastError();
}

hipError_t launch_stream_copy(const void *src, void *dst, size_t nbytes, hipStream_t s)
{
    const long long n16 = (long long)(nbytes / 16);
    if (n16 <= 0)
        return hipSuccess;
    long long blocks = (n16 + 1023) / 1024;
    if (blocks > 4096)
        blocks = 4096;
    hipLaunchKernelGGL(k_stream_copy, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const u32x4 *>(src),
                       static_cast<u32x4 *>(dst), n16);
    return hipGetLastError();
}

/* ======================================================================== */
/* k_synth_lcg                                                              */
/* ======================================================================== */
/* the affine map of `steps` LCG steps: s -> A*s + C (mod 2^32), by squaring */
__device__ __forceinline__ void lcg_jump(unsigned long long steps, uint32_t &A, uint32_t &C)
{
    A = 1u;
    C = 0u;
    uint32_t a = 1664525u, cc = 1013904223u;      /* map for 2^b steps */
    while (steps) {
        if (steps & 1ull) {
            A = a * A;
            C = a * C + cc;
        }
        cc = (a + 1u) * cc;
        a = a * a;
        steps >>= 1;
    }
}

/* The generator, for one stream or several (blockIdx.y): 16 bytes per thread and iteration.  The jump ladders run per
 * THREAD, not per chunk: to the thread's first chunk -- split into the ladder to the BLOCK's first chunk, the same for
 * all its threads (scalar unit, up to 64 rounds), and the thread's own 16*tid steps behind it (12 rounds) -- and for the
 * grid stride, again the same map for every thread.  A chunk then costs the 16 steps of its bytes plus one multiply-add
 * (2^22 samples: 31 -> 8 us with the split ladder; 2^28: the write stream's own time).                            */
__global__ __launch_bounds__(256) void k_synth_lcg(SynthMany m, unsigned long long nbytes)
{
    uint8_t *dst = static_cast<uint8_t *>(m.dst[blockIdx.y]);
    const uint32_t seed = m.seed[blockIdx.y];
    const unsigned long long nch = (nbytes + 15) >> 4;
    const unsigned long long stride = (unsigned long long)gridDim.x * 256;
    const unsigned long long cb = (unsigned long long)blockIdx.x * 256;
    unsigned long long c = cb + threadIdx.x;
    if (c >= nch)
        return;
    uint32_t A, C, At, Ct, As, Cs;
    lcg_jump(m.byte_offset[blockIdx.y] + (cb << 4), A, C);      /* uniform */
    lcg_jump((unsigned long long)threadIdx.x << 4, At, Ct);
    lcg_jump((stride - 1) << 4, As, Cs);                        /* uniform */
    uint32_t st = At * (A * seed + C) + Ct;
    for (; c < nch; c += stride) {
        uint32_t w[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                st = st * 1664525u + 1013904223u;
                v |= (st >> 24) << (8 * b);
            }
            w[d] = v;
        }
        const unsigned long long o = c << 4;
        if (o + 16 <= nbytes) {
            *reinterpret_cast<uint4 *>(dst + o) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            for (int b = 0; b < 16 && o + b < nbytes; ++b)
                dst[o + b] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
        }
        st = As * st + Cs;
    }
}

hipError_t launch_synth_lcg_many(const SynthMany &m, int n, size_t nbytes, hipStream_t s)
{
    if (n < 1 || n > kFir8ManyMax)
        return hipErrorInvalidValue;
    if (nbytes == 0)
        return hipSuccess;
    const unsigned long long nch = ((unsigned long long)nbytes + 15) >> 4;
    unsigned long long blocks = (nch + 255) / 256;
    const unsigned long long cap = 256ull * 16 / (unsigned)n > 256 ? 256ull * 16 / (unsigned)n : 256;
    if (blocks > cap)
        blocks = cap;
    hipLaunchKernelGGL(k_synth_lcg, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0, s, m, (unsigned long long)nbytes);
    return hipGetLastError();
}

hipError_t launch_synth_lcg(void *dst, size_t nbytes, uint32_t seed, uint64_t byte_offset, hipStream_t s)
{
    SynthMany m = {};
    m.dst[0] = dst;
    m.byte_offset[0] = byte_offset;
    m.seed[0] = seed;
    return launch_synth_lcg_many(m, 1, nbytes, s);
}

} // namespace pddc
