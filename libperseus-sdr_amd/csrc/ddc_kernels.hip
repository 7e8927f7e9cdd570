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
 *   k_fir_generic  any-D decimating FIR on float2 (later cascade stages).
 *   k_resample     rational L/M polyphase resampler (non-integer rates).
 *   k_pack24       float32 -> 24-bit packed (inverse of the unpack).
 *   k_hist_update  carries the FIR history between batches (generic path).
 *   k_synth_lcg    device-side synthetic source.
 *
 * k_fir8 design (DESIGN.md "Kernels"):
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
 *   No MFMA: 9 flop/B, a banded single-filter FIR would waste 2/3 of a matrix
 *   op, and the stream is memory- and power-bound.
 */
#include "ddc_kernels.h"

#include <cstdio>
#include <cstdlib>

#define PDDC_CONSTANT __attribute__((address_space(4)))

#ifndef PDDC_PRIO_U
#define PDDC_PRIO_U 0
#endif
#ifndef PDDC_PRIO_F
#define PDDC_PRIO_F 2
#endif

namespace pddc {

static constexpr int kFused3MaxChunks = 2048;      /* seam slots / flag words of a fused-cascade launch */
static constexpr int kPend3 = 4;                   /* open seams a block of the fused cascade may carry along */

/* float = (float)(v24*256) * RN(1/2147483392): the int->float convert is exact
 * (24 significant bits) and the product is bit-identical to the reference's
 * (float)int32 / (float)(INT_MAX-256) for all 2^24 codes (tests). */
static constexpr float kUnpackScale = 0x1.000002p-31f;   /* RN(1/8388607) / 256 */

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

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

/* exp(-j*2*pi*phase/2^32) from the exact 32-bit phase: quadrant reduction in
 * integers, then minimax polynomials on [-pi/4, pi/4] (abs error < 1e-7). */
__device__ __forceinline__ void nco_lo(uint32_t phase, float &c, float &s)
{
    const uint32_t q = (phase + 0x20000000u) >> 30;
    const int32_t  r = (int32_t)(phase - (q << 30));
    const float t  = (float)r * 1.4629180792671596e-9f;          /* pi / 2^31 */
    const float t2 = t * t;
    float sn = fmaf(t2, fmaf(t2, fmaf(t2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), 0.0f);
    sn = fmaf(sn, t, t);
    float cs = fmaf(t2, fmaf(t2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f);
    cs = fmaf(t2 * t2, cs, fmaf(t2, -0.5f, 1.0f));
    /* theta = q*pi/2 + t.  Branch-free quadrant fix-up (a switch here costs four divergent
     * regions per call): odd quadrants swap sin and cos, the signs are XORed in          */
    const bool odd = (q & 1u) != 0;
    const float cc = odd ? sn : cs;
    const float ss = odd ? cs : sn;
    const uint32_t neg_c = ((q + 1u) & 2u) << 30;              /* cos < 0 in quadrants 1, 2 */
    const uint32_t neg_s = ((q & 2u) << 30) ^ 0x80000000u;     /* sin < 0 in 2, 3; and exp(-j theta) */
    c = __uint_as_float(__float_as_uint(cc) ^ neg_c);
    s = __uint_as_float(__float_as_uint(ss) ^ neg_s);
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
    static const int cap = getenv("PDDC_UNPACK_BLOCKS") ? atoi(getenv("PDDC_UNPACK_BLOCKS")) : 512;   /* 2 per CU measured best (0.66 vs 0.70 ms) */
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
         * k_fir8); the NCO takes ONE sin/cos per thread -- a thread's pairs lie 512 samples apart, the next one's phasor
         * is the last one's times LO(512); only the batch's first block can meet samples of the previous batch, mixed
         * with the previous tuning word (exact phase per pair there).                                             */
        const uint8_t *inb = static_cast<const uint8_t *>(a.in);
        const uint8_t *hb8 = static_cast<const uint8_t *>(a.hist);
        const GenMixArgs &mx = a.mx;
        const long long xa = s0 & ~1LL;                        /* pairs start at even sample indices (4-byte aligned) */
        const int shift = (int)(s0 - xa);                      /* 0 or 1 */
        const int npairs = (span + shift + 1) >> 1;
        constexpr int NPF = (G::PD * (256 + 16) / 2 + 1 + 255) / 256;       /* pairs per thread */
        struct W3 { uint32_t a, b, c; };
        if (xa >= 0 && xa + 2LL * 256 * NPF <= a.n_batch) {
            /* interior block (uniform): no history, nothing beyond the batch, and every staged index a pair can
             * touch (-1 .. span) has a slot thanks to the spare segments: straight-line code                  */
            const uint8_t *src0 = inb + xa * 6 + 12 * tid;
            W3 rw[NPF];
#pragma unroll
            for (int u = 0; u < NPF; ++u)
                rw[u] = *reinterpret_cast<const W3 *>(src0 + 12 * 256 * u);
            float gc = 1.0f, gs = 0.0f, kc = 1.0f, ks = 0.0f;
            if (MIX) {
                nco_lo((uint32_t)(mx.n0 + (unsigned long long)(xa + 2LL * tid)) * mx.freg + mx.phase_off, gc, gs);
                nco_lo(512u * mx.freg, kc, ks);
            }
            const float stc = mx.lo_c[1], sts = mx.lo_s[1];
#pragma unroll
            for (int u = 0; u < NPF; ++u) {
                float x0i = (float)(int32_t)__builtin_amdgcn_perm(rw[u].a, rw[u].a, 0x0201000cu);
                float x0q = (float)(int32_t)__builtin_amdgcn_perm(rw[u].b, rw[u].a, 0x0504030cu);
                float x1i = (float)(int32_t)__builtin_amdgcn_perm(rw[u].c, rw[u].b, 0x0403020cu);
                float x1q = (float)(int32_t)(rw[u].c & 0xffffff00u);
                if (MIX) {
                    if (u > 0)
                        cmul(gc, gs, kc, ks);
                    const float c1 = gc * stc - gs * sts, s1 = gc * sts + gs * stc;
                    cmul(x0i, x0q, gc, gs);
                    cmul(x1i, x1q, c1, s1);
                }
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
        float gc = 1.0f, gs = 0.0f, kc = 1.0f, ks = 0.0f;
        const bool has_old = MIX && xa < 0;                    /* uniform: the batch's first block only */
        const uint32_t off_old = mx.phase_off + (uint32_t)mx.n0 * (mx.freg - mx.freg_hist);
        if (MIX) {
            nco_lo((uint32_t)(mx.n0 + (unsigned long long)(xa + 2LL * tid)) * mx.freg + mx.phase_off, gc, gs);
            nco_lo(512u * mx.freg, kc, ks);
        }
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int j = tid + 256 * u;
            if (u > 0 && MIX)
                cmul(gc, gs, kc, ks);
            if (j >= npairs)
                continue;
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
template <int NTB2, int R>
struct Fir8Geom2 {
    static constexpr int TO2   = 16 * R;                 /* stage-2 outputs per tile (TO/8)   */
    static constexpr int R2    = TO2 / 64;               /* per lane of waves 0 (I) and 1 (Q) */
    static constexpr int GT2   = 16 * R;                 /* new input groups per tile (TO/8)  */
    static constexpr int NG2   = GT2 + NTB2;
    /* plain layout (offset 8 + position): the second stage is ~3 % of the work, a
     * 2-way bank conflict on its reads is cheaper than the LDS a pad would cost  */
    static constexpr int PLANE = NTB2 > 0 ? 8 + 8 * NG2 + 8 : 0;
    /* two plane SETS, alternating tile by tile (the history of tile t+1 is carried into the other set while all four
     * waves may still be reading tile t's), and two staging areas for the two halves of the tap sum */
    static constexpr int LDS_FLT = NTB2 > 0 ? 4 * PLANE + 4 * TO2 : 0;
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

#ifdef PDDC_CLOCK_PROBE
/* development: per-block 100 MHz wall-clock (s_memrealtime) and shader-clock (s_memtime)
 * stamps plus the hardware placement, to see the engine clock the kernel ran at and how
 * evenly the persistent blocks finish (make HIPFLAGS+=-DPDDC_CLOCK_PROBE; the report is
 * printed by pddc_pipeline_time_stage0, PDDC_PROBE_VERBOSE=1 lists every block)          */
struct ProbeRec { unsigned long long w0, w1, c0, c1; unsigned hw, xcc; };
__device__ ProbeRec g_probe[4096];
void fir8_probe_dump()
{
    static ProbeRec h[4096];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_probe), sizeof(h)) != hipSuccess)
        return;
    int nb = 0;
    while (nb < 4096 && h[nb].w1 != 0)
        ++nb;
    if (nb == 0)
        return;
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < nb; ++b) {
        if (h[b].w0 < t0) t0 = h[b].w0;
        if (h[b].w1 > t1) t1 = h[b].w1;
    }
    static double dur[4096], st[4096], en[4096];
    for (int b = 0; b < nb; ++b) {
        st[b] = (h[b].w0 - t0) / 100.0;
        en[b] = (h[b].w1 - t0) / 100.0;
        dur[b] = en[b] - st[b];
    }
    if (getenv("PDDC_PROBE_VERBOSE"))
        for (int b = 0; b < nb; ++b)
            fprintf(stderr, "[blk] %d xcc %u se %u sh %u cu %u simd %u wave %u  start %.1f end %.1f\n", b, h[b].xcc & 15,
                    (h[b].hw >> 13) & 7, (h[b].hw >> 12) & 1, (h[b].hw >> 8) & 15, (h[b].hw >> 4) & 3, h[b].hw & 15,
                    st[b], en[b]);
    auto srt = [&](double *v) { for (int i = 1; i < nb; ++i) { double x = v[i]; int j = i - 1; while (j >= 0 && v[j] > x) { v[j + 1] = v[j]; --j; } v[j + 1] = x; } };
    const double mhz = (double)(h[0].c1 - h[0].c0) / ((double)(h[0].w1 - h[0].w0) / 100.0);
    srt(st); srt(en); srt(dur);
    fprintf(stderr, "[probe] %d blocks, span %.1f us, block 0 at %.0f MHz\n", nb, (t1 - t0) / 100.0, mhz);
    fprintf(stderr, "[probe] start  min %.1f  p50 %.1f  p90 %.1f  max %.1f us\n", st[0], st[nb / 2], st[nb * 9 / 10], st[nb - 1]);
    fprintf(stderr, "[probe] end    min %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f us\n", en[0], en[nb / 10], en[nb / 2], en[nb * 9 / 10], en[nb - 1]);
    for (int b = 0; b < nb; ++b)
        h[b] = ProbeRec{};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_probe), h, sizeof(h));
}
#endif

/* The same sliding window with the loops exchanged: tap block j outermost, the R groups
 * that meet it (ub = r + NTB-1-j) held in VGPRs and shifted by one group per step.  Only one
 * tap block is live at a time (8 SGPRs + the next s_load) instead of R of them: the R=8
 * kernels otherwise keep 64 tap SGPRs live and spill (119 v_readlane per tile at 255 taps).  */
#ifndef PDDC_TAP_OUTER_R8
#define PDDC_TAP_OUTER_R8 1
#endif
static constexpr bool kTapOuterR8 = PDDC_TAP_OUTER_R8 != 0;
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
 * The first tile of a chunk takes its history from global memory (the previous
 * 8*NTB input samples, or p.hist for tile 0), prefetched with the tile itself.
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
/* (PDDC_ABLATE_LOADS / _FIR / _STORES: timing-only builds with one part of the kernel removed,
 * tools/ablate.sh; their outputs are garbage.)                                              */
/* NT = threads per block.  256 (4 waves: two per plane) is the default; 128 (R = 8 only: one wave per
 * plane, half the tile, half the LDS) lets four independent blocks share a CU instead of two.       */
/* SL3 > 0 (FUSE3): a third stage (plain decimate-by-d3 FIR, struct Fir8Stage3) runs on the second stage's outputs in LDS
 * as well, SL3 taps per wave and tile (its work is sliced into the tile loop, see "third stage" below).          */
template <int NTB, int R, int INFMT, bool MIX, int NTB2, int NT = 256, int SL3 = 0>
__global__ __launch_bounds__(NT, 2) void k_fir8(Fir8Args p, int ntiles, int S, int K)
{
    constexpr bool FUSE3 = SL3 > 0;
    constexpr int SLN = SL3 > 0 ? SL3 : 1;
    /* CARRY (the fused pair): the grid's last p.tail.nblocks blocks are not part of the pair -- they run the generic
     * decimator on the PREVIOUS batch's second-stage outputs (its tail, Fir8Args::tail).  They are dealt out behind the
     * pair's persistent blocks and live on the waves those leave idle (the pair issues vector instructions 38 % of the
     * time and holds two of a SIMD's three possible waves).                                                     */
    constexpr bool CARRY = SL3 == 0 && NT == 256 && R == 4 && INFMT == IN_PACKED24;
    if (CARRY && (int)blockIdx.x >= (int)gridDim.x - p.tail.nblocks) {
        extern __shared__ __attribute__((aligned(16))) float2 sd_tail[];
        run_tail_block(p.tail, (int)blockIdx.x - ((int)gridDim.x - p.tail.nblocks), sd_tail);
        return;
    }
    static_assert(NT == 256 || (NT == 128 && R == 8 && NTB2 == 0), "128-thread blocks: R = 8, no fused second stage");
    static_assert(!FUSE3 || (NTB2 > 0 && NT == 256), "the third stage sits behind the fused pair");
#ifdef PDDC_CLOCK_PROBE
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        g_probe[blockIdx.x].c0 = clock64();
        g_probe[blockIdx.x].w0 = wall_clock64();
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_probe[blockIdx.x].hw = hw;
        g_probe[blockIdx.x].xcc = xcc;
    }
#endif
    using G = Fir8Geom<NTB, R, NT>;
    using G2 = Fir8Geom2<NTB2, R>;
    constexpr bool FUSE2 = NTB2 > 0;     /* a second decimate-by-8 stage runs on the tile's outputs in LDS */
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sI = smem;
    float *sQ = smem + G::PLANE;
    float *ot = smem + 2 * G::PLANE;     /* stage-1 output staging (unfused) ...                            */
    /* ... or (fused) the second stage's input planes -- set s: I at ot + 2s*PLANE2, Q behind it -- and its staging */
    auto pl2_of = [&](int set, int q) { return ot + (2 * set + q) * G2::PLANE; };
    float *ot2 = ot + 4 * G2::PLANE;
    int cur2 = 0;                        /* the plane set of the tile in work (uniform)                      */
    /* third stage (FUSE3), all in (I, Q) pairs: the four waves' partial sums, two ring sets -- [padf zeros | h3 history |
     * g3 tiles of second-stage outputs] each, alternating group by group like the second stage's planes -- and the
     * held-back outputs of the first groups of up to kPend3 chunks                                                */
    const int RING = FUSE3 ? p.s3.padf + p.s3.h + p.s3.g * G2::TO2 : 0;
    f32x2 *part3  = reinterpret_cast<f32x2 *>(ot2 + 4 * G2::TO2);      /* [4 waves][64]: partial sums of a group */
    f32x2 *ring3  = part3 + 4 * 64;                   /* sets 0 and 1; set 2 is the scratch of the seam fix-ups */
    f32x2 *pend3  = ring3 + 3 * RING;                 /* [kPend3][64] held-back outputs of chunks whose seam is open */
    int   *pendh  = reinterpret_cast<int *>(pend3 + kPend3 * 64);   /* [kPend3][4] chunk id, samples, first output */
    int   *poll3  = pendh + 4 * kPend3;               /* one word: thread 0's poll result for the block          */
    /* plane offsets 0..6 (slots 0..6 of "group -1") never hold a sample: offset 0 of the
     * I plane (smem[0], as raw bits) carries the next chunk index from thread 0 to the
     * block.  Accessed as smem[0] so it stays an LDS access (a cast pointer becomes a
     * flat load whose vmcnt(0) wait would also wait for the tile's stores).             */

    constexpr int NW = (INFMT == IN_PACKED24) ? 3 : 4;             /* 16-byte words per group */
    constexpr int ES = (INFMT == IN_PACKED24) ? 6 : 8;             /* bytes per sample        */

    const int tid = threadIdx.x;
    /* group handled by this thread in the load/unpack phases: lane bits 2 and 3
     * swapped, so that the 8-lane groups of ds_write_b128/_b96 hit 8 distinct
     * 4-bank sets (32-byte group stride + the 16-byte pad every R groups)      */
    const int gtid = (tid & ~12) | ((tid & 4) << 1) | ((tid & 8) >> 1);
    const int nblk = (int)gridDim.x - (CARRY ? p.tail.nblocks : 0);      /* the persistent blocks */
    const int dyn0 = nblk * S;                           /* first tile of the dynamic part */
    const int ND   = (ntiles - dyn0 + K - 1) / K;        /* number of dynamic chunks       */

    /* leaving: the last block out resets the schedule for the next launch (and, FUSE3, the chunks' flags: every
     * other block has made its last poll by then) */
    auto leave = [&]() {
        bool last_out = false;
        if (tid == 0 && atomicAdd(p.sched + 1, 1u) == (unsigned)(nblk - 1)) {
            atomicExch(p.sched, 0u);
            atomicExch(p.sched + 1, 0u);
            last_out = true;
        }
        if (FUSE3) {
            __syncthreads();                              /* nobody still reads smem[0] as a chunk index */
            if (tid == 0)
                smem[0] = last_out ? 1.0f : 0.0f;
            __syncthreads();
            if (smem[0] != 0.0f) {
                const int nchunks = (S > 0 ? nblk : 0) + ND;
                for (int i = tid; i < nchunks; i += NT)
                    __hip_atomic_store(p.s3.flags + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#ifdef PDDC_CLOCK_PROBE
        if (threadIdx.x == 0 && blockIdx.x < 4096) {
            g_probe[blockIdx.x].c1 = clock64();
            g_probe[blockIdx.x].w1 = wall_clock64();
        }
#endif
    };

    /* ---- first chunk: [c_lo, c_hi) are the tiles whose outputs the block writes ---- */
    int c_lo, c_hi;
    if (S > 0) {
        c_lo = (int)blockIdx.x * S;
        c_hi = c_lo + S;
    } else {
        if (tid == 0)
            smem[0] = __int_as_float((int)atomicAdd(p.sched, 1u));
        __syncthreads();
        const int j = __builtin_amdgcn_readfirstlane(__float_as_int(smem[0]));
        __syncthreads();
        if (j >= ND) {
            leave();
            return;
        }
        c_lo = dyn0 + j * K;
        c_hi = min(c_lo + K, ntiles);
    }
    /* fused second stage: its history is 8*NTB2 stage-1 outputs, i.e. the tile
     * in front of a chunk is recomputed as a warm-up (no output) -- except for
     * tile 0, whose stage-2 history comes from the previous call               */
    if (FUSE2 && c_lo == 0 && tid < NTB2) {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(static_cast<const float *>(p.hist2) + 16 * tid);
        const u32x4 h2raw[4] = { src[0], src[1], src[2], src[3] };
        float xi[8], xq[8];
        group_to_float<IN_F32C, false, 4>(h2raw, xi, xq, 0ull, p);
        if (MIX) {      /* stored values are final; inside tile 0 they must become final by * phi_0 */
            float c0, s0;
            nco_lo((uint32_t)p.n0 * p.freg + p.phase_off, c0, s0);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                cmul(xi[e], xq[e], c0, -s0);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {            /* position p2 = 8*tid + e - 1, offset 8 + p2 */
            pl2_of(cur2, 0)[7 + 8 * tid + e] = xi[e];
            pl2_of(cur2, 1)[7 + 8 * tid + e] = xq[e];
        }
    }

    /* one tile of prefetched input in registers: the next tile is requested while
     * this one is filtered.  (A second tile in flight was measured no faster; it
     * needs asm loads with hand-counted vmcnt because hipcc waits vmcnt(0) for
     * loop-carried loads, and that form is fragile under register pressure --
     * DESIGN.md 5.)  rawH: the NTB history groups of a chunk's first tile.       */
    u32x4 rawA[G::GPT][NW];
    u32x4 rawH[NW];
    auto prefetch = [&](int tile, bool with_hist) {
        const long long tin0 = (long long)tile * G::TI;
        const u32x4 *src0 = reinterpret_cast<const u32x4 *>(static_cast<const uint8_t *>(p.in) +
                                                            (tin0 + 8LL * gtid) * ES);
        if (tin0 + G::TI <= p.n_in) {                         /* whole tile in range (wave-uniform) */
#pragma unroll
            for (int k = 0; k < G::GPT; ++k)
#pragma unroll
                for (int w = 0; w < NW; ++w)
#ifdef PDDC_ABLATE_LOADS
                    rawA[k][w] = u32x4{ (unsigned)tile * 2654435761u + tid, (unsigned)(k + w) << 20, (unsigned)tile << 9, 77u * tid };
#else
                    rawA[k][w] = src0[(NT * k * 8 * ES) / 16 + w];
#endif
        } else {                                        /* ragged last tile */
#pragma unroll
            for (int k = 0; k < G::GPT; ++k) {
                const bool have = tin0 + 8LL * (gtid + NT * k) < p.n_in;
#pragma unroll
                for (int w = 0; w < NW; ++w)
                    rawA[k][w] = have ? src0[(NT * k * 8 * ES) / 16 + w] : u32x4{ 0u, 0u, 0u, 0u };
            }
        }
        if (with_hist && tid < NTB) {                   /* groups 0..NTB-1: the 8*NTB samples before the tile */
            const long long s_abs = tin0 + 8LL * tid - 8 * NTB;
            const uint8_t *src = (s_abs < 0) ? static_cast<const uint8_t *>(p.hist) + (s_abs + 8 * NTB) * ES
                                             : static_cast<const uint8_t *>(p.in) + s_abs * ES;
#pragma unroll
            for (int k = 0; k < NW; ++k)
                rawH[k] = (s_abs < p.n_in) ? reinterpret_cast<const u32x4 *>(src)[k] : u32x4{ 0u, 0u, 0u, 0u };
        }
    };

    const int wave  = __builtin_amdgcn_readfirstlane(tid >> 6);   /* provably wave-uniform */
    const int lane  = tid & 63;
    const int plane = wave & 1;
    /* segment (R outputs, 8R inputs) owned by this lane, 0..127 */
    const int par   = (R == 4) ? (wave >> 1) : 0;                  /* R=4: even / odd half segments */
    const int L     = (R == 4) ? (2 * lane + par) : ((wave >> 1) * 64 + lane);
    const float *base = (plane ? sQ : sI) + 8 + 8 * R * L + 4 * ((R * L) >> 3);   /* goff(R*L) */
    const float PDDC_CONSTANT *hb = (const float PDDC_CONSTANT *)p.taps_blk;
    const long long n_out = p.n_in >> 3;

    /* S (fused): the tile's TO2 second-stage outputs, 16 bytes per thread */
    auto store_tile2 = [&](int tile, float pc, float ps) {
        constexpr int NCH2 = G2::TO2 / 2;
        if (tid < NCH2) {
            const long long m = (long long)tile * G2::TO2 + 2LL * tid;     /* no ragged tiles when fused */
            f32x4 v = *reinterpret_cast<const f32x4 *>(ot2 + 4 * tid);
            v += *reinterpret_cast<const f32x4 *>(ot2 + 2 * G2::TO2 + 4 * tid);      /* the other half of the taps */
            if (MIX)                        /* tile-relative NCO: the tile's phasor goes on at the very end */
                v = cmul2(v, pc, ps);
            float *dstp = p.out + 2 * m;
            asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(dstp), "v"(v) : "memory");
        }
    };
    const float PDDC_CONSTANT *hb2 = (const float PDDC_CONSTANT *)p.taps2_blk;

    /* ---- third stage (FUSE3) ------------------------------------------------------------------------------------
     * APPEND: a tile's second-stage outputs are summed and leave the staging area after the next barrier A, where
     * store_tile2 would have written them to HBM; they go into the ring set of the GROUP being filled (g tiles).
     * JOB: when a group is complete its outputs are due -- lane j output j, wave w the taps [w*seglen, (w+1)*seglen),
     * seglen = g*SL3.  Doing that on the spot stalls the whole block on a chain of latencies (LDS round trips, a
     * barrier for the partial sums, parameters from the argument segment): measured 0.8 us per group, 26 groups per
     * block, +27 us on a 0.29 ms launch, although the arithmetic is a tenth of the first stage's.  So the job is SLICED
     * into the next g tiles: each F phase issues SL3 LDS reads before the first-stage FIR and takes SL3 taps -- two
     * VGPRs hold the wave's taps, one per lane, read with v_readlane, so a tap has no memory latency -- into two packed
     * accumulators after it; the last slice leaves the wave's partial sums in LDS and they are combined behind the
     * tile's barrier B.  The group's ring set stays untouched meanwhile (the next group fills the other set).
     * SEAMS: a chunk of tiles starts with an unknown history: the outputs of its first group are held back (pending
     * list), its last h second-stage outputs are published for the chunk behind it (write-through stores, then a flag:
     * MI355X_MICROARCH.md, inter-workgroup visibility), and at later chunk ends -- at the exit at the latest -- the
     * block looks whether the chunk in front has published, and completes the held-back outputs (resolve3).     */
    int rs3 = 0;                         /* ring set of the group being filled                                       */
    int g_left = FUSE3 ? p.s3.g : 0;     /* tiles the group still takes                                              */
    int npend = 0;                       /* chunks of this block whose seam with the chunk in front is still open    */
    const Fir8Args PDDC_CONSTANT *kp = (const Fir8Args PDDC_CONSTANT *)__builtin_amdgcn_kernarg_segment_ptr();
    auto ring_of = [&](int set) { return ring3 + set * RING; };
    f32x2 *ap3 = FUSE3 ? ring_of(0) + p.s3.padf + p.s3.h + 2 * tid : nullptr;   /* where this thread's next pair goes */
    /* chunk [lo, ..) in tile order: the static runs first, then the dynamic chunks */
    auto chunk_id = [&](int lo) { return lo < dyn0 ? lo / S : (S > 0 ? nblk : 0) + (lo - dyn0) / K; };
    float tapv0 = 0.0f, tapv1 = 0.0f;    /* the wave's taps: lane k holds taps k and 64 + k of its segment           */
    if (FUSE3) {
        const int seglen = p.s3.seglen;
        const float *tp = p.s3.taps + wave * seglen;
        if (lane < seglen)
            tapv0 = tp[lane];
        if (64 + lane < seglen)
            tapv1 = tp[64 + lane];
    }
    /* the job in work */
    const f32x2 *xr3 = ring3 + RING;     /* this lane's newest sample of the next slice (reads xr3[0 .. -(SL3-1)])   */
    int sl_k = 0, sl_left = 0;           /* the next slice's first tap; slices to go (0: no job, the slices idle)    */
    f32x2 acc3a = { 0.0f, 0.0f }, acc3b = { 0.0f, 0.0f };
    bool sum_ready = false;              /* the waves' partial sums are in part3, to be combined behind a barrier    */
    int job_nnew = 0, job_cid = -1;      /* the group's samples; >= 0: its outputs are held back for chunk job_cid   */
    long long job_m0 = 0;                /* its first output                                                         */
    /* SL3 plain ds_read_b64 (2 LDS cycles each; hipcc pairs neighbouring loads into ds_read2_b64, which runs at half
     * that rate: MI355X_MICROARCH.md, LDS table), issued from inline asm: the compiler's own counted lgkmcnt waits stay
     * correct because these reads are OLDER than anything it waits for (LDS returns in order); slice_taps waits.   */
    auto slice_load = [&](f32x2 (&xs)[SLN], const f32x2 *xr) {
        const unsigned lo = (unsigned)(size_t)(xr - (SLN - 1));       /* the LDS offset: low half of a flat LDS address */
#pragma unroll
        for (int u = 0; u < SLN; ++u)
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(xs[u]) : "v"(lo), "n"(8 * (SLN - 1 - u)));
    };
    /* taps k0 .. k0+SL3-1 of the wave's segment (a slice never straddles lane 64: fir8_fused3_geometry) */
    auto slice_taps = [&](f32x2 (&xs)[SLN], int k0, f32x2 &a, f32x2 &b) {
        const int tv = __float_as_int(k0 < 64 ? tapv0 : tapv1);          /* uniform */
        float hh[SLN];
#pragma unroll
        for (int u = 0; u < SLN; ++u)
            hh[u] = __int_as_float(__builtin_amdgcn_readlane(tv, (k0 + u) & 63));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xs[0]) : : "memory");
#pragma unroll
        for (int u = 1; u < SLN; ++u)
            asm volatile("" : "+v"(xs[u]));                              /* not before the wait */
#pragma unroll
        for (int u = 0; u < SLN; ++u) {
            if (u & 1)
                b = __builtin_elementwise_fma(f32x2{ hh[u], hh[u] }, xs[u], b);
            else
                a = __builtin_elementwise_fma(f32x2{ hh[u], hh[u] }, xs[u], a);
        }
    };
    /* second half of a slice (the first is slice_load); with no job in work it accumulates nothing anybody reads */
    auto slice_fma = [&](f32x2 (&xs)[SLN]) {
        slice_taps(xs, sl_k, acc3a, acc3b);
        if (sl_left > 0) {
            xr3 -= SL3;
            sl_k += SL3;
            if (--sl_left == 0) {
                part3[64 * wave + lane] = acc3a + acc3b;
                sum_ready = true;
            }
        }
    };
    auto store_out3 = [&](float *out3, long long m, f32x2 v) {
        float *dstp = out3 + 2 * m;
        asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(dstp), "v"(v) : "memory");   /* see store_tile */
    };
    /* all outputs of the group in ring set `rb` at once (the seam fix-ups; a handful per block and launch):
     * returns output `tid` to the threads tid < 64, behind a barrier                                              */
    auto visit3 = [&](const Fir8Stage3 PDDC_CONSTANT &q, const f32x2 *rb) {
        const int seglen = q.seglen, ng = q.ng;
        const int j = lane < ng ? lane : ng - 1;
        const f32x2 *xr = rb + q.padf + q.h + q.off + j * q.d - wave * seglen;
        f32x2 a = { 0.0f, 0.0f }, b = a;
        for (int k0 = 0; k0 < seglen; k0 += SLN) {
            f32x2 xs[SLN];
            slice_load(xs, xr - k0);
            slice_taps(xs, k0, a, b);
        }
        part3[64 * wave + lane] = a + b;
        __syncthreads();
        f32x2 sum = { 0.0f, 0.0f };
        if (tid < 64)
            sum = (part3[tid] + part3[64 + tid]) + (part3[128 + tid] + part3[192 + tid]);
        return sum;
    };
    /* Settle open seams.  Entry i of the pending list is a chunk of this block whose first outputs wait for the last
     * h second-stage outputs of the chunk in front of it (the previous call's for the batch's first chunk).  Thread 0
     * looks at that chunk's flag -- once; or, `blocking`, until it is up (bounded) -- and when it is, the block loads
     * the tail into the scratch ring set ([zeros | tail | zeros]), runs the third stage on it and adds the result to
     * the held-back outputs.  Never waiting at a chunk's end matters: the two blocks of a CU run 15-25 % apart, a
     * block that waited for its neighbour's static run would idle through exactly the time the dynamic tail is there
     * to fill.  What is still open when the block runs out of work is waited for at the exit.                     */
    auto resolve3 = [&](bool blocking) {
        asm volatile("" : "+s"(kp));
        const Fir8Stage3 PDDC_CONSTANT &q = kp->s3;
        const int h3 = q.h, padf = q.padf;
        f32x2 *fx = ring_of(2);
        int keep = 0;
        __syncthreads();                            /* the list as the last push left it */
        for (int i = 0; i < npend; ++i) {
            const int id = __builtin_amdgcn_readfirstlane(pendh[4 * i]);
            if (tid == 0) {
                int ok = 1;
                if (id > 0) {
                    unsigned spins = 0;
                    while ((ok = __hip_atomic_load(q.flags + id - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) == 0 &&
                           blocking) {
                        __builtin_amdgcn_s_sleep(8);
                        if (++spins > (1u << 20)) {                   /* ~0.3 s: something is badly wrong */
                            atomicOr(p.sched + 2, 1u);
                            ok = 1;
                            break;
                        }
                    }
                    if (ok)
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                *poll3 = ok;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const int ok = __builtin_amdgcn_readfirstlane(*poll3);
            if (ok) {
                if (tid < h3 / 2) {
                    f32x4 v;
                    if (id == 0) {
                        v = *(reinterpret_cast<const f32x4 *>(q.hist) + tid);
                    } else {
                        const uint8_t *srcp = static_cast<const uint8_t *>(q.seam) + (size_t)(id - 1) * q.seam_stride + 16 * tid;
                        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(srcp) : "memory");
                    }
                    *reinterpret_cast<f32x4 *>(fx + padf + 2 * tid) = v;
                }
                __syncthreads();
                const f32x2 c = visit3(q, fx);
                const int nnew = pendh[4 * i + 1];
                const long long m0 = ((long long)pendh[4 * i + 3] << 32) | (unsigned)pendh[4 * i + 2];
                if (tid < q.ng && q.off + tid * q.d < nnew && m0 + tid < q.n_out)
                    store_out3(q.out, m0 + tid, pend3[64 * i + tid] + c);
            } else {
                if (keep != i) {                    /* stays open: move it down */
                    if (tid < 64)
                        pend3[64 * keep + tid] = pend3[64 * i + tid];
                    if (tid < 4)
                        pendh[4 * keep + tid] = pendh[4 * i + tid];
                }
                ++keep;
            }
            __syncthreads();                        /* poll3, the scratch ring set and the list are reused */
        }
        npend = keep;
    };
    /* the waves' partial sums of the finished job (in part3 since before the last barrier) -> its outputs: to HBM,
     * or, for a chunk's first group, onto the pending list                                                         */
    auto combine3 = [&]() {
        asm volatile("" : "+s"(kp));
        const Fir8Stage3 PDDC_CONSTANT &q = kp->s3;
        f32x2 y = { 0.0f, 0.0f };
        if (tid < 64)
            y = (part3[tid] + part3[64 + tid]) + (part3[128 + tid] + part3[192 + tid]);
        sum_ready = false;
        if (job_cid >= 0) {
            if (npend == kPend3)
                resolve3(true);                     /* the list is full: wait for its oldest seams */
            if (tid < 64)
                pend3[64 * npend + tid] = y;
            if (tid == 0) {
                pendh[4 * npend] = job_cid;
                pendh[4 * npend + 1] = job_nnew;
                pendh[4 * npend + 2] = (int)(unsigned)(job_m0 & 0xffffffffLL);
                pendh[4 * npend + 3] = (int)(job_m0 >> 32);
            }
            ++npend;
        } else if (tid < q.ng && q.off + tid * q.d < job_nnew && job_m0 + tid < q.n_out) {
            store_out3(q.out, job_m0 + tid, y);
        }
    };
    /* finish the job in work on the spot (the block is about to leave, or groups follow each other faster than g tiles:
     * a short last chunk) */
    auto flush3 = [&]() {
        while (sl_left > 0) {
            f32x2 xs[SLN];
            slice_load(xs, xr3);
            slice_fma(xs);
        }
        if (sum_ready) {
            __syncthreads();
            combine3();
        }
    };
    /* S (FUSE3): the tile's TO2 second-stage outputs go into the ring instead of HBM */
    auto append3 = [&](float pc, float ps) {
        if (tid < G2::TO2 / 2) {
            f32x4 v = *reinterpret_cast<const f32x4 *>(ot2 + 4 * tid);
            v += *reinterpret_cast<const f32x4 *>(ot2 + 2 * G2::TO2 + 4 * tid);
            if (MIX)
                v = cmul2(v, pc, ps);
            *reinterpret_cast<f32x4 *>(ap3) = v;
        }
        ap3 += G2::TO2;
        --g_left;
    };
    /* a group is complete (g tiles, or the chunk [a_lo, ..) ends with tile t_last): its job starts; unless the chunk
     * ends, its last h samples become the history of the next group in the other ring set                         */
    auto group_done3 = [&](int a_lo, int t_last, bool chunk_ends) {
        flush3();                             /* normally long finished */
        asm volatile("" : "+s"(kp));
        const Fir8Stage3 PDDC_CONSTANT &q = kp->s3;
        const int g_cnt = q.g - g_left;
        const int g_t0 = t_last + 1 - g_cnt;
        const int n_new = g_cnt * G2::TO2;
        const int h3 = q.h, padf = q.padf;
        acc3a = acc3b = f32x2{ 0.0f, 0.0f };
        sl_k = 0;
        sl_left = q.g;
        xr3 = ring_of(rs3) + padf + h3 + q.off + (lane < q.ng ? lane : q.ng - 1) * q.d - wave * q.seglen;
        job_nnew = n_new;
        job_m0 = (long long)(g_t0 / q.g) * q.ng;
        job_cid = g_t0 == a_lo ? chunk_id(a_lo) : -1;       /* a chunk's first group has no history yet: held back */
        if (!chunk_ends) {
            for (int i = tid; i < h3; i += NT)
                ring_of(rs3 ^ 1)[padf + i] = ring_of(rs3)[padf + n_new + i];
            rs3 ^= 1;
            g_left = q.g;
            ap3 = ring_of(rs3) + padf + h3 + 2 * tid;
        }
    };
    /* the chunk [a_lo, a_hi) has ended (its last group's job has just started in ring set rs3): publish its last h
     * second-stage outputs for the chunk behind it, settle whatever seams can be settled, and -- the batch's last chunk
     * -- leave the next call's history.  The next chunk fills the other ring set, from a zero history.            */
    auto chunk_end3 = [&](int a_lo, int a_hi, bool at_exit) {
        asm volatile("" : "+s"(kp));
        const Fir8Stage3 PDDC_CONSTANT &q = kp->s3;
        const int h3 = q.h, padf = q.padf;
        const int g_cnt = q.g - g_left;
        const int n_new = g_cnt * G2::TO2;
        const bool single = a_hi - g_cnt == a_lo;            /* the chunk's last group is also its first */
        const int id = chunk_id(a_lo);
        const bool batch_last = a_hi == ntiles;
        if (!batch_last) {
            if (tid < h3 / 2) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(ring_of(rs3) + padf + n_new + 2 * tid);
                uint8_t *dstp = static_cast<uint8_t *>(q.seam) + (size_t)id * q.seam_stride + 16 * tid;
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dstp), "v"(v) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          /* every storing wave drains ...          */
            __syncthreads();                                          /* ... before ONE lane raises the flag    */
            if (tid == 0)
                __hip_atomic_store(q.flags + id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (at_exit)
            flush3();                                /* nothing left to hide the last job behind */
        resolve3(at_exit);
        if (at_exit && batch_last && q.hist_out != nullptr) {
            /* the stream's last h3 second-stage outputs: from the last group's ring set; where that group is the chunk's
             * first and shorter than h3 (a tiny batch), from the tail of the chunk in front (settled just above)       */
            const float2 *pred = id == 0 ? static_cast<const float2 *>(q.hist)
                                         : reinterpret_cast<const float2 *>(static_cast<const uint8_t *>(q.seam) +
                                                                            (size_t)(id - 1) * q.seam_stride);
            for (int i = tid; i < h3; i += NT) {
                const int rel = n_new - h3 + i;              /* relative to the first new sample of set rs3 */
                float2 v;
                if (rel >= 0 || !single) {
                    const f32x2 r = ring_of(rs3)[padf + h3 + rel];
                    v = make_float2(r.x, r.y);
                } else {
                    const float2 *srcp = pred + (h3 + rel);
                    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(srcp) : "memory");
                }
                static_cast<float2 *>(q.hist_out)[i] = v;
            }
        }
        rs3 ^= 1;                                            /* the job keeps reading the old set */
        for (int i = tid; i < h3; i += NT)
            ring_of(rs3)[padf + i] = f32x2{ 0.0f, 0.0f };
        g_left = q.g;
        ap3 = ring_of(rs3) + padf + h3 + 2 * tid;
    };
    if (FUSE3) {
        for (int i = tid; i < 3 * RING; i += NT)
            ring3[i] = f32x2{ 0.0f, 0.0f };
    }

    /* S: coalesced stores of one finished tile from the staging area */
    auto store_tile = [&](int tile, float pc, float ps) {
        const long long tile_o0 = (long long)tile * G::TO;
        constexpr int NCH = G::TO / 2;                             /* 16-byte chunks */
        if (tile_o0 + G::TO <= n_out) {                            /* whole tile in range (uniform) */
#pragma unroll
            for (int it = 0; it < NCH / NT; ++it) {
                const int q = tid + NT * it;
                const int qs = q ^ ((q >> 3) & 7);
                /* streaming (nt) store: measured 0.349 vs 0.371 ms for this 6:1
                 * read/write mix (tools/ubench/stream_mix.hip) */
                f32x4 v = *reinterpret_cast<const f32x4 *>(ot + 4 * qs);
                if (MIX)
                    v = cmul2(v, pc, ps);
                float *dstp = p.out + 2 * (tile_o0 + 2LL * q);
                /* Issued from inline asm on purpose: hipcc then does not count the store
                 * in its vmcnt bookkeeping, so the waits it places for the prefetched
                 * loads stay COUNTED (vmcnt(N)) instead of collapsing to vmcnt(0) as they
                 * do whenever loads and stores are both pending.  Memory operations
                 * retire in issue order on gfx9, so a counted wait computed without
                 * these stores is only ever stronger than needed, never weaker.  The
                 * trailing s_nop 1 is the wait state a 128-bit store needs before the next
                 * instruction may overwrite its data registers (hipcc pads nothing inside
                 * or after an asm string).                                             */
#ifdef PDDC_ABLATE_STORES
                if (v.x == 1.2345e-30f)
#endif
                asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(dstp), "v"(v) : "memory");
            }
        } else {
#pragma unroll
            for (int it = 0; it < NCH / NT; ++it) {
                const int q = tid + NT * it;
                const int qs = q ^ ((q >> 3) & 7);
                float4 v = *reinterpret_cast<const float4 *>(ot + 4 * qs);
                if (MIX) {
                    cmul(v.x, v.y, pc, ps);
                    cmul(v.z, v.w, pc, ps);
                }
                const long long m = tile_o0 + 2LL * q;
                if (m + 1 < n_out)
                    *reinterpret_cast<float4 *>(p.out + 2 * m) = v;
                else if (m < n_out)
                    *reinterpret_cast<float2 *>(p.out + 2 * m) = make_float2(v.x, v.y);
            }
        }
    };

    /* NCO, tile-relative.  LO(n) of sample i of tile t factors as phi_t * w(i) with
     * phi_t = LO(n0 + t*TI) and w(i) = exp(-j*2*pi*freg*i/2^32), i the index inside the tile
     * (the phase is linear in the sample index).  U multiplies by w(i) only -- per-thread
     * constants, no sin/cos and no rotation chain per group -- and everything downstream is
     * linear, so phi_t is applied once per OUTPUT, at the stores.  A sample carried to the
     * next tile as FIR history is rotated by conj(D), D = phi_(t+1)/phi_t = LO(TI).
     * R=4: w for the thread's 8*GPT samples sits in registers; R=8 (no VGPRs to spare) keeps
     * w of each group's first sample and steps through the group with the host's phasors.   */
    constexpr bool WTAB = MIX && R == 4;
    float w_c[WTAB ? G::GPT : 1][8], w_s[WTAB ? G::GPT : 1][8];
    float wg_c[G::GPT], wg_s[G::GPT];
    float d_c = 1.0f, d_s = 0.0f;
#pragma unroll
    for (int k = 0; k < G::GPT; ++k) {
        wg_c[k] = 1.0f;
        wg_s[k] = 0.0f;
        if (MIX && !WTAB)
            nco_lo((uint32_t)(8 * (gtid + NT * k)) * p.freg, wg_c[k], wg_s[k]);
        if (WTAB) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                nco_lo((uint32_t)(8 * (gtid + NT * k) + e) * p.freg, w_c[k][e], w_s[k][e]);
        }
    }
    if (MIX)
        nco_lo((uint32_t)G::TI * p.freg, d_c, d_s);
    /* The tile's phasor is the same for every lane: lane l computes the one of tile ph_base + l, once per 64
     * consecutive tiles, and each tile fetches its own with v_readlane -- the sin/cos polynomial is ~25 VALU
     * instructions, a tenth of what a wave of the fused pair issues per tile (same-box A/B -0.4 %).           */
    /* (R = 4 only: the R = 8 mixing variants sit at 256 VGPRs and the two table registers would spill) */
    constexpr bool PHTAB = MIX && R == 4;
    float ph_c = 1.0f, ph_s = 0.0f;
    int ph_base = -0x40000000;
    auto tile_phasor = [&](int tile, float &c, float &sn) {
        c = 1.0f;
        sn = 0.0f;
        if (MIX && !PHTAB)
            nco_lo((uint32_t)(p.n0 + (unsigned long long)((long long)tile * G::TI)) * p.freg + p.phase_off, c, sn);
        if (PHTAB) {
            if (tile < ph_base || tile >= ph_base + 64) {            /* uniform */
                ph_base = tile;
                nco_lo(((uint32_t)p.n0 + (uint32_t)(tile + (tid & 63)) * (uint32_t)G::TI) * p.freg + p.phase_off, ph_c,
                       ph_s);
            }
            c  = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ph_c), tile - ph_base));
            sn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ph_s), tile - ph_base));
        }
    };

    int  t = (FUSE2 && c_lo > 0) ? c_lo - 1 : c_lo;   /* tile in work (a fused chunk starts one tile early) */
    bool first = true;                                /* t opens a chunk: history comes from rawH            */
    int  tprev = -1;                                  /* tile whose outputs are staged, not yet stored        */
    bool prev_out2 = false;                           /* ... and (fused) whether it produced stage-2 outputs  */
    int  pv_lo = 0, pv_hi = 0;                        /* ... and (FUSE3) the chunk it belongs to              */
    unsigned grabv = 0;                               /* thread 0: the chunk taken for after this one         */
    float pp_c = 1.0f, pp_s = 0.0f;                   /* NCO phasor of tile tprev                             */
    prefetch(t, true);
    if (t + 1 == c_hi && tid == 0)
        grabv = atomicAdd(p.sched, 1u);

    for (;;) {
        /* wave issue priority: the FIR phase runs at PDDC_PRIO_F (2), everything else at PDDC_PRIO_U (0).
         * When the two waves of a SIMD both want to issue, the one inside its FMA run goes first and the
         * other's loads / LDS traffic fill the gaps: 255 taps 0.481 -> 0.466 ms, 127 taps 0.3669 -> 0.3650,
         * x320 cascade unchanged (same-box A/B, profiles/r02/ab_prio.txt); the reverse (loads first) gains nothing */
        __builtin_amdgcn_s_setprio(PDDC_PRIO_U);
        const bool last = (t + 1 == c_hi);            /* last tile of its chunk */
        /* ---- U: registers -> LDS planes (groups NTB ..; a chunk's first tile also 0..NTB-1) ---- */
        float pt_c, pt_s;                                /* this tile's phasor, used when its outputs leave */
        tile_phasor(t, pt_c, pt_s);
#pragma unroll
        for (int k = 0; k < G::GPT; ++k) {
            const int v = NTB + gtid + NT * k;
            float xi[8], xq[8];
            group_to_float<INFMT, false, NW>(rawA[k], xi, xq, 0ull, p);
            if (WTAB) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    cmul(xi[e], xq[e], w_c[WTAB ? k : 0][e], w_s[WTAB ? k : 0][e]);
            } else if (MIX) {
                mix8_lo(xi, xq, wg_c[k], wg_s[k], p);
            }
            group_to_lds<R>(sI, sQ, v, xi, xq);
        }
        if (first && tid < NTB) {          /* after the tile's own groups: rawH was requested last */
            float xi[8], xq[8];
            group_to_float<INFMT, false, NW>(rawH, xi, xq, 0ull, p);
            if (MIX) {
                /* tile-relative index i = 8*tid - 8*NTB < 0 (the 32-bit phase wraps correctly).  The
                 * samples in front of tile 0 belong to the previous batch: in the first batch after a
                 * retune they were mixed with the OLD tuning word -- the switch is sample-accurate at
                 * the batch boundary and phase-continuous there, like the FPGA's phase accumulator --
                 * so relative to this tile's phasor they carry exp(-j*2*pi*i*freg_old/2^32).  The old
                 * word and its step phasors come with the arguments (== the current ones otherwise). */
                const bool old = (t == 0);                       /* uniform */
                float cb, sb;
                nco_lo((uint32_t)(8 * tid - 8 * NTB) * (old ? p.freg_hist : p.freg), cb, sb);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float sc = old ? p.lo_c_hist[e] : p.lo_c[e];
                    const float ss = old ? p.lo_s_hist[e] : p.lo_s[e];
                    cmul(xi[e], xq[e], cb * sc - sb * ss, cb * ss + sb * sc);
                }
            }
            group_to_lds<R>(sI, sQ, tid, xi, xq);
        }
        /* ---- S (deferred): the PREVIOUS tile's stores go out here, behind this
         * tile's load wait.  gfx9 has one vmcnt for loads and stores and hipcc
         * waits vmcnt(0) whenever both kinds are pending, so stores issued just
         * before the wait for prefetched loads would stall every tile on the
         * write acknowledgements (measured: +0.12 ms per 2^28 samples).        */
        /* publish the next chunk BEFORE the stores are issued: reading grabv waits
         * for everything outstanding (vmcnt(0)), which here is nothing new -- after the
         * stores it would wait for their acknowledgements                             */
        if (last && tid == 0)
            smem[0] = __int_as_float((int)grabv);
        if (!FUSE2 && tprev >= 0)
            store_tile(tprev, pp_c, pp_s);
        __syncthreads();                                           /* A */
        const bool appended = FUSE3 && prev_out2;
        if (FUSE3) {
            if (prev_out2)             /* into the third stage's ring */
                append3(pp_c, pp_s);
        } else if (FUSE2 && prev_out2) /* written by waves 0/1 after the previous barrier B */
            store_tile2(tprev, pp_c, pp_s);

        /* ---- P: next tile's loads (and, one tile before a chunk ends, the next chunk) ---- */
        int tn = t + 1, n_lo = c_lo, n_hi = c_hi;
        if (last) {
            const int j = __builtin_amdgcn_readfirstlane(__float_as_int(smem[0]));
            if (j < ND) {
                n_lo = dyn0 + j * K;
                n_hi = min(n_lo + K, ntiles);
                tn = FUSE2 ? n_lo - 1 : n_lo;                        /* dynamic chunks never start at tile 0... */
                if (FUSE2 && n_lo == 0)
                    tn = 0;                                          /* ... unless S == 0                      */
            } else {
                tn = -1;
            }
        }
        if (tn >= 0) {
            prefetch(tn, last);
            if (tn + 1 == n_hi && tid == 0)
                grabv = atomicAdd(p.sched, 1u);
        }

        /* ---- F: FIR ------------------------------------------------------ */
        /* Packed fp32: every VALU op costs ~4 cycles per wave64 on gfx950, and
         * v_pk_fma_f32 does two FMAs in that slot (measured 68 vs 33 TFMA/s,
         * tools/ubench/fma_issue.hip).  The dot product of one output is split
         * into its even and odd terms: acc.x += h[k]*x[i], acc.y += h[k-1]*x[i+1]
         * -- both operands are natural adjacent pairs (SGPR pair of taps, VGPR
         * pair of samples from one ds_read_b128), no broadcast, no shuffles.   */
        __builtin_amdgcn_s_setprio(PDDC_PRIO_F);
        asm volatile("" : "+s"(hb));      /* keep the tap s_loads inside the tile loop (no SGPR spills) */
        if (FUSE2)
            asm volatile("" : "+s"(hb2));
        f32x2 xs3[SLN];
        if (FUSE3)                        /* third stage: this tile's slice of the job in work, loads first ... */
            slice_load(xs3, xr3);
        constexpr int NA = FirAcc<NTB>::N;
        f32x2 accp[R][NA];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int a = 0; a < NA; ++a)
                accp[r][a] = f32x2{ 0.0f, 0.0f };
#ifdef PDDC_ABLATE_FIR
#pragma unroll
        for (int r = 0; r < R; ++r)
            accp[r][0] = *reinterpret_cast<const f32x2 *>(base + 8 * r);
#else
        if (R == 4 && par)
            fir_window<NTB, R, 1>(base, hb, accp);
        else if (R == 8 && kTapOuterR8)
            fir_window_tap_outer<NTB, R, 0>(base, hb, accp);
        else
            fir_window<NTB, R, 0>(base, hb, accp);
#endif
        f32x2 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r)
            acc[r] = NA == 2 ? accp[r][0] + accp[r][NA - 1] : accp[r][0];
        if (FUSE3)                        /* ... taps behind the first-stage FIR, which hid the loads' latency */
            slice_fma(xs3);
        if (FUSE2) {
            /* results -> the second stage's input plane, rotated like the first:
             * position p2 = m + 8*NTB2 - 1 for tile-relative output m = R*L + r,
             * float offset 8 + p2                                                  */
            float *pl2 = pl2_of(cur2, plane);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                pl2[8 + (R * L + r + 8 * NTB2 - 1)] = acc[r].x + acc[r].y;
            }
        } else {
            /* results -> staging (XOR-swizzled 16-byte chunks, interleaved I/Q) */
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int f  = 2 * (R * L + r) + plane;
                const int q  = f >> 2;
                const int qs = q ^ ((q >> 3) & 7);
                ot[4 * qs + (f & 3)] = acc[r].x + acc[r].y;
            }
        }
        __syncthreads();                                           /* B */

        /* ---- C: tail groups -> history of the next tile of the chunk -------- */
        if (!last && gtid >= NT - NTB) {
            const int gd = gtid - (NT - NTB);                       /* 0..NTB-1 */
            const int os = goff<R>(G::GT + gd), od = goff<R>(gd);
            float4 i0 = *reinterpret_cast<const float4 *>(sI + os);
            float4 i1 = *reinterpret_cast<const float4 *>(sI + os + 4);
            float4 q0 = *reinterpret_cast<const float4 *>(sQ + os);
            float4 q1 = *reinterpret_cast<const float4 *>(sQ + os + 4);
            if (MIX) {          /* into the next tile's frame: * conj(D) */
                cmul(i0.x, q0.x, d_c, -d_s);
                cmul(i0.y, q0.y, d_c, -d_s);
                cmul(i0.z, q0.z, d_c, -d_s);
                cmul(i0.w, q0.w, d_c, -d_s);
                cmul(i1.x, q1.x, d_c, -d_s);
                cmul(i1.y, q1.y, d_c, -d_s);
                cmul(i1.z, q1.z, d_c, -d_s);
                cmul(i1.w, q1.w, d_c, -d_s);
            }
            *reinterpret_cast<float4 *>(sI + od) = i0;
            *reinterpret_cast<float4 *>(sQ + od) = q0;
            *reinterpret_cast<float2 *>(sI + od + 4) = make_float2(i1.x, i1.y);
            *reinterpret_cast<float2 *>(sQ + od + 4) = make_float2(q1.x, q1.y);
            sI[od + 6] = i1.z;
            sQ[od + 6] = q1.z;
            if (gd != NTB - 1) {          /* slot 7 of the last group belongs to the next tile's first sample */
                sI[od + 7] = i1.w;
                sQ[od + 7] = q1.w;
            }
        }
        /* ---- F2 / C2 (fused): the second decimator on the TO stage-1 outputs now in LDS, all four waves: waves
         * 0 / 1 take the OLDER half of the tap blocks of plane I / Q, waves 2 / 3 the newer half (the two partial
         * sums meet in store_tile2) -- the 64 taps on two waves left SIMDs 0 and 1 with half as much FIR work again
         * as SIMDs 2 and 3 (same-box A/B 0.2897 -> 0.2875 ms).  Then waves 0 / 1 carry the history INTO THE OTHER
         * PLANE SET, which nobody reads before the next barrier B, so waves 2 / 3 may still be reading this one.
         * Ordering needs no extra barrier: the inputs were written before B, the outputs (ot2) are read after the
         * next A, and the next stage-1 results go into the other set after the next A as well.                  */
        if (FUSE2) {
            const float *pl2r = pl2_of(cur2, wave & 1);
            const int newer = wave >> 1;
            if (t >= c_lo) {
                constexpr int R2 = G2::R2 > 0 ? G2::R2 : 1;
                constexpr int HB = NTB2 > 1 ? NTB2 / 2 : 1;
                static_assert(NTB2 == 0 || NTB2 % 2 == 0, "the split second stage needs an even number of tap blocks");
                f32x2 acc2[R2][1];                 /* HB <= 4 tap blocks: one packed accumulator per output */
#pragma unroll
                for (int r = 0; r < R2; ++r)
                    acc2[r][0] = f32x2{ 0.0f, 0.0f };
                fir_window<HB, R2, 0, false>(pl2r + 8 + 8 * R2 * lane + (newer ? 8 * HB : 0), hb2 + (newer ? 0 : 8 * HB),
                                             acc2);
#pragma unroll
                for (int r = 0; r < R2; ++r)
                    ot2[newer * 2 * G2::TO2 + 2 * (R2 * lane + r) + (wave & 1)] = acc2[r][0].x + acc2[r][0].y;
            }
            if (wave < 2 && !last && lane < NTB2) {
                float *nx2 = pl2_of(cur2 ^ 1, wave);                /* the next tile's plane of this wave */
                const int os = 8 + 8 * (G2::GT2 + lane), od = 8 + 8 * lane;
                float4 a0 = *reinterpret_cast<const float4 *>(pl2r + os);
                float4 a1 = *reinterpret_cast<const float4 *>(pl2r + os + 4);
                if (MIX) {
                    /* * conj(D), like the first stage's history.  The rotation needs the other
                     * plane's tail too: read-only here (written before B)                      */
                    const float *ol2 = pl2_of(cur2, wave ^ 1);
                    const float4 b0 = *reinterpret_cast<const float4 *>(ol2 + os);
                    const float4 b1 = *reinterpret_cast<const float4 *>(ol2 + os + 4);
                    /* wave 0: I' = I*dc + Q*ds ; wave 1: Q' = Q*dc - I*ds */
                    const float sg = wave ? -d_s : d_s;
                    a0.x = a0.x * d_c + b0.x * sg;
                    a0.y = a0.y * d_c + b0.y * sg;
                    a0.z = a0.z * d_c + b0.z * sg;
                    a0.w = a0.w * d_c + b0.w * sg;
                    a1.x = a1.x * d_c + b1.x * sg;
                    a1.y = a1.y * d_c + b1.y * sg;
                    a1.z = a1.z * d_c + b1.z * sg;
                    a1.w = a1.w * d_c + b1.w * sg;
                }
                *reinterpret_cast<float4 *>(nx2 + od) = a0;
                *reinterpret_cast<float2 *>(nx2 + od + 4) = make_float2(a1.x, a1.y);
                nx2[od + 6] = a1.z;
                if (lane != NTB2 - 1)
                    nx2[od + 7] = a1.w;
            }
        }
        /* ---- F3 (FUSE3): the ring holds tprev's outputs since before barrier B.  A full group, or the end of tprev's
         * chunk: the third stage runs on it; a chunk's end also settles the seam with the chunk in front of it     */
        if (FUSE3 && sum_ready)
            combine3();
        if (appended) {
            const bool a_last = tprev + 1 == pv_hi;
            if (g_left == 0 || a_last) {
                group_done3(pv_lo, tprev, a_last);
                if (a_last)
                    chunk_end3(pv_lo, pv_hi, false);
            }
        }
        tprev = t;
        pp_c = pt_c;
        pp_s = pt_s;
        prev_out2 = FUSE2 && t >= c_lo;
        pv_lo = c_lo;
        pv_hi = c_hi;
        if (tn < 0)
            break;
        if (FUSE2)
            cur2 ^= 1;                    /* the next tile's stage-1 results and history live in the other set */
        first = last;
        t = tn;
        c_lo = n_lo;
        c_hi = n_hi;
    }

    if (FUSE2) {
        __syncthreads();                 /* ot2 and the stage-2 planes of the last tile are complete */
        if (FUSE3) {
            if (prev_out2) {
                append3(pp_c, pp_s);
                __syncthreads();
                group_done3(pv_lo, tprev, true);
                chunk_end3(pv_lo, pv_hi, true);
            }
        } else if (prev_out2)
            store_tile2(tprev, pp_c, pp_s);
        /* the last 8*NTB2 stage-1 outputs are the second stage's next history */
        if (p.hist2_out != nullptr && tprev == ntiles - 1 && tid < 8 * NTB2) {
            const int o = 8 + (G::TO + tid - 1);       /* position of stage-1 output TO - 8*NTB2 + tid */
            float hi = pl2_of(cur2, 0)[o], hq = pl2_of(cur2, 1)[o];
            if (MIX)
                cmul(hi, hq, pp_c, pp_s);              /* stored in final form */
            static_cast<float2 *>(p.hist2_out)[tid] = make_float2(hi, hq);
        }
    } else {
        store_tile(tprev, pp_c, pp_s);
    }

    /* the block that ran the last tile leaves the batch's last 8*NTB input
     * samples as the next call's history (the host alternates two buffers, so
     * tile 0 of THIS launch never sees them)                                  */
    if (p.hist_out != nullptr && tprev == ntiles - 1 && p.n_in >= 8 * NTB) {
        constexpr int HCH = 8 * NTB * ES / 16;                     /* 16-byte chunks */
        const uint4 *src = reinterpret_cast<const uint4 *>(static_cast<const uint8_t *>(p.in) +
                                                           (p.n_in - 8 * NTB) * ES);
        for (int c = tid; c < HCH; c += NT)
            static_cast<uint4 *>(p.hist_out)[c] = src[c];
    }
    leave();
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
 * tile on one address).  The fused pair pays a warm-up tile per chunk: round 1 saw no
 * gain and kept it static; with the buffers placed (round 2) 8 % in chunks of 8 is worth 1 %.
 * Development overrides: PDDC_FIR8_DYN_PCT (share of the tiles handed out
 * dynamically), PDDC_FIR8_CHUNK (K).                                              */
struct Fir8Sched {
    int nblocks, S, K;
};

static Fir8Sched fir8_schedule(int ntiles, int R, bool fused, int NT = 256, int group = 0)
{
    /* read per launch (two getenv calls against a launch of several microseconds), so a
     * test can switch schedules inside one process */
    const char *e = getenv("PDDC_FIR8_DYN_PCT");
    const int v = e ? atoi(e) : -1;
    const int dyn_pct = v < 0 ? -1 : (v > 100 ? 100 : v);
    e = getenv("PDDC_FIR8_CHUNK");
    const int chunk = e ? atoi(e) : 0;
    Fir8Sched sc;
    /* 128-thread blocks: four per CU, tiles half as long (chunks of twice as many) */
    /* small batches (BASELINE config 5's low end): one block per tile up to one block per CU, then about three tiles
     * per block -- a block with a single tile overlaps nothing, its load, filter and store phases just follow each other
     * (tools/small_batch.sh, 127 taps: 2^22 samples 19.0 -> 14.4 us with 256 instead of 512 blocks, 2^23 22.4 -> 20.2
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
    /* fused pair: a dynamic chunk starts with a warm-up tile, so only a small share pays (same-box sweep under the
     * arena placement, tools/sched_sweep_c320.sh: static 0.2897 ms, 5-10 % in chunks of 8 0.2863-0.2867, 20 % 0.293) */
    const int pct = dyn_pct >= 0 ? dyn_pct : (fused ? 8 : 20);
    sc.S = (int)((long long)ntiles * (100 - pct) / 100 / sc.nblocks);
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
    using G2 = Fir8Geom2<8, R>;
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
    const int plane2 = 8 + 8 * (16 * R + 8) + 8;
    return (size_t)(2 * plane + 4 * plane2 + 4 * 16 * R) * sizeof(float);
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
    using G2 = Fir8Geom2<8, R>;
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

void fir8_set_grid_blocks(int nblocks) { g_fir8_blocks = nblocks > 0 ? nblocks : 0; }

bool fir8_nt_supported(int ntb, int R, int NT)
{
#ifdef PDDC_EXPERIMENT_NT128
    if (NT == 128 && R == 8 && (ntb == 8 || ntb == 16 || ntb == 32))
        return true;
#endif
    (void)ntb;
    (void)R;
    return NT == 256;
}

hipError_t launch_fir8(int ntb, int R, InFmt fmt, bool mix, const Fir8Args &a, hipStream_t s, int NT)
{
    if (fmt == IN_F32C && mix)
        return hipErrorInvalidValue;
#ifdef PDDC_EXPERIMENT_NT128
    /* 128-thread blocks (four independent blocks per CU instead of two) -- measured, not faster: 127 taps
     * 0.377 vs 0.367 ms, and at 255 taps only three such blocks fit the LDS (0.613 vs 0.482 ms); kept
     * behind this macro for the record, not instantiated in the product build          */
    if (NT == 128) {
        if (fmt != IN_PACKED24 || R != 8)
            return hipErrorInvalidValue;
        if (ntb == 8) return launch_fir8_t<8, 8, 128>(fmt, mix, a, s);
        if (ntb == 16) return launch_fir8_t<16, 8, 128>(fmt, mix, a, s);
        if (ntb == 32) return launch_fir8_t<32, 8, 128>(fmt, mix, a, s);
        return hipErrorInvalidValue;
    }
#else
    if (NT != 256)
        return hipErrorInvalidValue;
#endif
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
    int force_nt = 0, force_p = 0;                     /* development: PDDC_GEN_SHAPE=NT,P */
    if (const char *e = getenv("PDDC_GEN_SHAPE"))
        (void)sscanf(e, "%d,%d", &force_nt, &force_p);
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

/* which D are built: first stages /10 and /5, tails /4 /5 /10 (the reference's rate plans) */
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
    if (getenv("PDDC_NO_FIRP"))
        return false;
    if (!(D == 4 || D == 5 || D == 10) || ntaps < 1)
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
    static const int pp = getenv("PDDC_FIRP_PACKED_P") ? atoi(getenv("PDDC_FIRP_PACKED_P")) : 0;
    if (D == 10 && infmt == IN_PACKED24 && pp == 1) {
        a.nbq = (ntaps + D - 1) / D;
        return launch_firp_t<10, 1>(infmt, mix, a, ntaps, s);
    }
    if (D == 4) return launch_firp_t<4>(infmt, mix, a, ntaps, s);
    if (D == 5) return launch_firp_t<5>(infmt, mix, a, ntaps, s);
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

/* 16 bytes per thread and iteration.  The jump ladder (up to 64 rounds) runs twice per THREAD -- to the thread's first
 * chunk, and for the grid stride, which is the same map for every thread -- not once per chunk: a chunk then costs the
 * 16 steps of its bytes plus one multiply-add (2^28 samples: 0.84 -> the write stream's own time).              */
__global__ __launch_bounds__(256) void k_synth_lcg(uint8_t *dst, unsigned long long nbytes, uint32_t seed,
                                                    unsigned long long byte_offset)
{
    const unsigned long long nch = (nbytes + 15) >> 4;
    const unsigned long long stride = (unsigned long long)gridDim.x * 256;
    unsigned long long c = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (c >= nch)
        return;
    uint32_t A, C, As, Cs;
    lcg_jump(byte_offset + (c << 4), A, C);          /* state before byte (byte_offset + 16c) */
    lcg_jump((stride - 1) << 4, As, Cs);             /* from the end of one chunk to the start of the thread's next */
    uint32_t st = A * seed + C;
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

hipError_t launch_synth_lcg(void *dst, size_t nbytes, uint32_t seed, uint64_t byte_offset, hipStream_t s)
{
    if (nbytes == 0)
        return hipSuccess;
    unsigned long long nch = ((unsigned long long)nbytes + 15) >> 4;
    unsigned long long blocks = (nch + 255) / 256;
    if (blocks > 256 * 32)
        blocks = 256 * 32;
    hipLaunchKernelGGL(k_synth_lcg, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<uint8_t *>(dst),
                       (unsigned long long)nbytes, seed, (unsigned long long)byte_offset);
    return hipGetLastError();
}

} // namespace pddc
