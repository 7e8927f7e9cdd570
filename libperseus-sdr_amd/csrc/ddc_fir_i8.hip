/*
 * ddc_fir_i8.hip -- the first stages on the INT8 matrix cores of gfx950 (MI355X, CDNA4): k_fir_i8x.
 *
 * The wire bytes are the operand planes (a 24-bit sample IS three bytes), the taps four planes of balanced base-256 digits,
 * int32 accumulation is exact.  Without NCO one tap table; with it the NCO is folded into the TAPS:
 *       y[m] = LO(n0 + 8 m) * sum_k (h[k] e^{+j theta k}) x_raw[8 m - k],   theta = 2 pi freg / 2^32,
 * i.e. complex taps on the raw integer planes (four real band products instead of one) and ONE float rotation per output,
 * with the exact 32-bit phase; optionally a second decimate-by-8 stage fused behind it (the cascades' pair: the 1 B/sample
 * intermediate never reaches HBM); a decimate-by-10 form.  Tiles are handed round the blocks in chunks of C (1:
 * tile-interleaved, what streams best; the pair: 4, or 8 with two matrix + two finishing waves from 2^23 samples on), NOTEBOOK.md R4.2.
 * (Round 3's k_fir_i8 -- the same product without NCO, eight matrix waves -- lived here until round 5; what it alone
 * offered, binary16-STORED taps quantised by the matrix waves themselves, is now the plain form's FirI8xArgs::taps16.)
 *
 * Reference anchors: the samples are the 24-bit wire format of examples/perseustest.c:449-455, the tuning word is
 * perseus-sdr.c:584; the arithmetic itself has no reference source (FPGA bitstreams), DESIGN.md 3.
 */
#include "ddc_kernels.h"
#include "ddc_dev.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace pddc {

/* ---- the formulation (round 3; what every form below shares) ----------------------------------------------------------
 * int8 MFMA has thirty times the vector unit's multiply-add peak, and this data fits it exactly: a 24-bit sample is three
 * bytes, a tap quantised to 2^-E (E = 30 - ceil(log2 max|h|), i.e. 31 significant bits on the largest tap) is four
 * balanced base-256 digits, every digit x byte-plane product sum over 256 taps stays below 2^24, so int32 accumulation is
 * EXACT; products of equal weight 256^(i+j) share an accumulator, the three lightest (i + j < 2: below 1.2e-7 of full
 * scale even if every term had the same sign, 2e-9 typical) are dropped, and the four sums are recombined in fp32 once
 * per output.  The unpack is gone: the loaders only de-interleave bytes (v_perm) into six planes (planes 0 and 1 xor
 * 0x80: unsigned -> signed, the offset comes back as one constant per filter).
 *   out[16 n + r] = sum_c T[r][c] X[c][n],  T[r][c] = h[HIST - (c - 8 r)] (banded Toeplitz, 16 x (120 + HIST)),
 *   X[c][n] = xp[8192 tile + 128 n + c],    xp = the HIST history samples followed by the batch.
 * v_mfma_i32_16x16x64_i8: 16 output rows, 16 columns, k-steps of 64; 9 plane products per step.  LDS rows are padded (lane
 * stride 144 B / 20 floats): conflict-free.  Measured as stand-alone prototypes first (tools/ubench/fir_i8_planes*.hip). */
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));
namespace i8 {
__device__ __forceinline__ int swz(int p) { return p + 16 * (p >> 7); }

/* the 8 bytes at offsets 6 s + O (s = 0..7) of the 48 bytes w[0..11] */
template <int O>
__device__ __forceinline__ void plane_bytes(const uint32_t (&w)[12], uint32_t &lo, uint32_t &hi)
{
    uint32_t out[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        uint32_t pair[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int b0 = 6 * (4 * half + 2 * q) + O, b1 = b0 + 6;
            const uint32_t sel = (uint32_t)(b0 & 3) | ((uint32_t)(4 + (b1 & 3)) << 8) | 0x0c0c0000u;
            pair[q] = __builtin_amdgcn_perm(w[b1 >> 2], w[b0 >> 2], sel);
        }
        out[half] = __builtin_amdgcn_perm(pair[1], pair[0], 0x05040100u);
    }
    lo = out[0];
    hi = out[1];
}

} // namespace i8

/* host: the binary16 array the plain form reads with FirI8xArgs::taps16 (values must be binary16-representable):
 * G[128 + tt] = h[hist - tt] for tt = 1 .. hist, zeros elsewhere */
void fir_i8_taps16(const float *taps, int ntaps, int hist, uint16_t *out)
{
    for (int i = 0; i < kFirI8Taps16Len; ++i)
        out[i] = 0;
    for (int tt = 1; tt <= hist; ++tt) {
        const int k = hist - tt;
        if (k < ntaps) {
            const _Float16 hv = (_Float16)taps[k];
            __builtin_memcpy(&out[128 + tt], &hv, 2);
        }
    }
}


/* ======================================================================== */
/* k_fir_i8x : decimate by 8 on the int8 matrix cores WITH the NCO, [+ a fused second decimate-by-8 stage]              */
/* ======================================================================== */
/* The mix commutes with the filter once it is moved into the taps (file header): what k_fir_i8 multiplies by h[k] this
 * kernel multiplies by gc[k] = h[k] cos(theta k) and gs[k] = h[k] sin(theta k),
 *     uI = gc * xI - gs * xQ,   uQ = gs * xI + gc * xQ     (* = the band product of k_fir_i8, exact in int32),
 * and y[m] = (uI + j uQ)[m] (cos phi_m - j sin phi_m), phi_m = the exact 32-bit phase of input sample n0 + 8 m, one
 * nco_lo and four multiplies per OUTPUT.  The samples stay the integers of the wire: no per-sample mix, no rounding
 * before the accumulation.  Three forms of one kernel (MODE):
 *   0  no NCO            waves = component x half of the columns, one tap set                    (k_fir_i8's arithmetic)
 *   1  NCO, 129..256 taps waves = tap set (c / s) x half of the columns, BOTH components each: four partial products
 *                        P[c|s][I|Q] meet in LDS, uI = P[c][I] - P[s][Q], uQ = P[s][I] + P[c][Q] at the stores
 *                        (two tap sets in one wave would need 2 x 96 registers)
 *   2  NCO, <= 128 taps  ONE operand holds both tap sets: rows 0..7 the band of eight outputs for one set, rows 8..15 the
 *                        band of the same eight outputs for the other (columns of 8 outputs, 64 samples apart; the band is
 *                        56 + HIST wide: 2 k-steps up to 64 taps, 3 up to 128).  [c ; s] meets the I planes, [-s ; c] the Q
 *                        planes, both into the SAME int32 accumulators: rows 0..7 come out as uI, rows 8..15 as uQ --
 *                        complete, one rounding -- and every wave makes both rails of its columns.  A third fewer matrix
 *                        instructions than two 16-row bands per rail (the dead corners of a band shrink with its height):
 *                        same box, 2^28 samples, 48 taps 0.3406 -> 0.3272 ms, 127 taps 0.3865 -> 0.3513 (it replaced a form
 *                        with both 16-row tables per wave for <= 64 taps and took 65..128 taps over from mode 1)
 * The tuned form 2 also DECIMATES BY 10 (template D = 10, hist = 64; the 1.6 / 2 / 1 MS/s plans' first stage,
 * launch_fir_i8x_d10): columns of 8 outputs 80 samples apart, a band of 70 + 64 samples = 3 k-steps, tiles of 10240 samples =
 * 1024 outputs, 1280 loader groups (two and a half rounds).  A stream's decimation phase makes a batch's first window end on
 * any sample 0 .. 9; the loaders' groups sit on multiples of 8, so the taps are delayed by (-first) mod 8 samples (one table
 * set per delay, built on the host) and the windows end on in_off = first + delay; the rotation takes that sample's phase.
 * Walk: the batch is cut into chunks of C tiles and block b takes chunks b, b + G, b + 2 G, ...  C = 1 is k_fir_i8's
 * tile-interleaved walk -- all CUs read one compact window of the batch, which is what this chip's HBM likes (same
 * kernel, 2^28 samples: contiguous ranges per block 0.380 ms, interleaved 0.332) --; inside a chunk the HIST samples in
 * front of a tile are the last ones of the tile before it and come over from the previous plane set inside LDS, only a
 * chunk's first tile loads them.
 * FUSE2 (modes 0 and 2): the tile's 1024 first-stage values u stay in LDS behind a 64-entry porch that holds the 64 values
 * in front of the tile; thread p of waves 0/1 computes second-stage output p of the tile,
 *     z = LO(n0 + 64 P) sum_k (h2[k] e^{+j 8 theta k}) u[8 P - k],
 * from float4 reads of both rails (packed FMAs, taps through the scalar cache) and stores it; still ONE barrier per tile.
 * Inside a chunk the porch is the previous tile's tail (copied by the two waves that do not filter); a chunk's first tile
 * computes it itself: four more columns in front of the tile (one extra pass of the matrix waves 2/3 over 512 + HIST more
 * samples, loaded by 72 loader lanes) -- no tile depends on another block, no warm-up tile; the batch's first tile takes
 * the stream's second-stage history (the 64 mixed first-stage outputs k_fir8's fused pair keeps too) instead.        */
namespace i8x {
using i8::plane_bytes;
using i8::swz;
constexpr int TILE = 8192;
constexpr int kTaps2Len = 68;                  /* floats per rail of the second stage's tap table (65 used) */

/* D: the decimation -- 8, or 10 for the tuned form with paired rows (mode 2): a tile is 1024 outputs = 1024 D samples */
template <int HIST, int MODE, bool FUSE2, int D = 8>
struct Geo {
    static constexpr int EXTRA = FUSE2 ? 512 : 0;                   /* samples in front of the history: the porch's columns */
    static constexpr int FRONT = EXTRA + HIST;
    static constexpr int TILE_S = 1024 * D;                         /* samples per tile */
    static constexpr int SPAN = TILE_S + FRONT, PLANE = SPAN + 16 * ((SPAN + 127) / 128);
    /* the band of a row block: 16 outputs (modes 0, 1) or 8 outputs x the two tap sets (mode 2), D samples apart, + HIST */
    static constexpr int KSTEPS = ((MODE == 2 ? 7 : 15) * D + HIST + 63) / 64;
    static constexpr int HG = HIST / 8, FG = FRONT / 8;              /* history groups of 8 samples; with the extra ones */
    static constexpr int NARR = MODE == 1 ? 4 : 2;
    static constexpr int PORCH = FUSE2 ? 64 : 0;
    static constexpr int AS = 20 * ((1024 + PORCH) / 16);            /* floats per output array (16 values per 20) */
    static constexpr size_t LDS_BYTES = 12 * (size_t)PLANE + 2 * (size_t)NARR * AS * sizeof(float);
    static constexpr int NTAB = MODE == 0 ? 1 : 2;
    static constexpr int TABV = 4 * KSTEPS * 64;                     /* v4i entries per tap table */
    static_assert(MODE != 2 || HIST <= 128, "two paired tables of 5 k-steps do not fit the registers");
    static_assert(D == 8 || (D == 10 && MODE == 2 && !FUSE2), "decimate-by-10: the tuned paired-rows form only");
    static_assert(!(FUSE2 && MODE == 1), "the fused second stage reads complete u values");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(FG <= 128, "front groups are loaded by the first loader waves");
};

/* a block's tile sequence: chunk b, b + G, ... of C tiles each */
struct Cursor {
    long long chunk;
    int k;
};
struct Walk {
    long long ntiles, G;
    int C;
    __device__ __forceinline__ long long tile(const Cursor &c) const
    {
        const long long t = c.chunk * C + c.k;
        return t < ntiles ? t : -1;
    }
    __device__ __forceinline__ Cursor next(Cursor c) const
    {
        if (++c.k == C) {
            c.k = 0;
            c.chunk += G;
        }
        return c;
    }
};

/* 8 samples (48 bytes) -> 8 bytes in each of the six planes at (swizzled) position `at` */
__device__ __forceinline__ void put_planes(const uint4 (&r)[3], uint8_t *plane, int PLANE, int at)
{
    const uint32_t w[12] = { r[0].x, r[0].y, r[0].z, r[0].w, r[1].x, r[1].y, r[1].z, r[1].w, r[2].x, r[2].y, r[2].z, r[2].w };
    uint32_t lo, hi;
#define PDDC_PL(C, I, O, X)                                                                       \
    plane_bytes<O>(w, lo, hi);                                                                    \
    *reinterpret_cast<uint2 *>(plane + (3 * C + I) * PLANE + at) = make_uint2(lo ^ X, hi ^ X);
    PDDC_PL(0, 0, 0, 0x80808080u)
    PDDC_PL(0, 1, 1, 0x80808080u)
    PDDC_PL(0, 2, 2, 0u)
    PDDC_PL(1, 0, 3, 0x80808080u)
    PDDC_PL(1, 1, 4, 0x80808080u)
    PDDC_PL(1, 2, 5, 0u)
#undef PDDC_PL
}

/* the loads of group g = lt + 512 q (q = 0, 1) of tile t: batch samples 8192 t + 8 g .. + 8, zeros behind the batch.
 * (Plain loads: a lane's three 16-byte loads share their cache lines with its neighbours' -- as NONTEMPORAL loads the lines
 * are fetched again and again: 0.3205 -> 0.4293 ms for the untuned 127-tap stage, same box, tools/ab_i8x.sh.)          */
template <int TILE_S = TILE>
__device__ __forceinline__ void issue_group(const FirI8xArgs &a, long long t, int g, uint4 (&r)[3])
{
    const long long b = t * TILE_S + 8LL * g + a.in_off;
    if (b + 8 <= a.n_in) {
        const uint4 *p = reinterpret_cast<const uint4 *>(static_cast<const uint8_t *>(a.in) + b * 6);
        r[0] = p[0];
        r[1] = p[1];
        r[2] = p[2];
    } else {
        r[0] = r[1] = r[2] = make_uint4(0u, 0u, 0u, 0u);
    }
}

/* the groups in front of a chunk's first tile t: front group g (0 .. FG) = samples 8192 t - FRONT + 8 g ..; groups below
 * `g0` are not needed (no porch columns: not FUSE2, or the batch's first tile).  From the stream's history for t = 0. */
template <int HIST, int FRONT, int TILE_S = TILE>
__device__ __forceinline__ void issue_front(const FirI8xArgs &a, long long t, int g, uint4 (&r)[3])
{
    const long long b = t * TILE_S - FRONT + 8LL * g + a.in_off;
    const int hl = a.hist_len ? a.hist_len : HIST;                   /* samples the history buffer holds (in front of a.in) */
    if (b < -(long long)hl) {                                       /* in front of the history: only zero taps reach there */
        r[0] = r[1] = r[2] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    const uint4 *p = b < 0 ? reinterpret_cast<const uint4 *>(static_cast<const uint8_t *>(a.hist) + (b + hl) * 6)
                           : reinterpret_cast<const uint4 *>(static_cast<const uint8_t *>(a.in) + b * 6);
    r[0] = p[0];
    r[1] = p[1];
    r[2] = p[2];
}

/* A float2 store whose data registers stay untouched for two more issue slots.  Measured on MI355X (ROCm 7.2): when a wave
 * stores (global_store_dwordx2) while ANOTHER wave on its SIMD is issuing matrix instructions, and the wave's next vector
 * instruction overwrites one of the store's data registers, the last quarter of the wave (lanes 48..63) now and then stores
 * the new value of that register -- once per few thousand tiles with post waves beside two matrix waves, never seen without
 * a matrix wave beside the storing one.  hipcc pads only stores of more than 64 bits; the pad has to sit inside the statement,
 * or the scheduler moves vector instructions in front of it.
 * All result stores of this file are NONTEMPORAL: the outputs are written once and read by another kernel much later, and
 * keeping them out of the L2's way is worth 5.7 % on the untuned 127-tap stage (same box, two rounds: 0.3383 / 0.3359 ->
 * 0.3191 / 0.3166 ms), 2 % on the tuned 48-tap one, nothing where the matrix work bounds (tools/ab_i8x.sh).           */
__device__ __forceinline__ void store_f2_padded(float2 *p, float x, float y)
{
    const f32x2 v = { x, y };
    asm volatile("global_store_dwordx2 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

/* y = sum_s acc[s] 256^(s+2) as floats */
__device__ __forceinline__ float recombine(const v4i_t (&acc)[4], int v)
{
    return ((float)acc[0][v] * 65536.0f + (float)acc[1][v] * 16777216.0f) +
           ((float)acc[2][v] * 4294967296.0f + (float)acc[3][v] * 1099511627776.0f);
}
} // namespace i8x

/* Who does what (LAYOUT).  The matrix waves' chain per tile -- operand reads, matrix instructions, recombination, and then
 * the tile's finish (rotation and stores, or the second stage) -- was the longest in the block: the loaders sat in the
 * barrier for a third to a half of every tile while waves 0..3 finished it (in-kernel clock probe: tuned 48 taps, matrix pass
 * 2800 cycles + finish 2700 against 3650..4300 of loader work; the finish is slow because its stores queue behind the
 * loaders' loads in the CU's one memory pipeline).
 *   0  matrix waves 0..3 finish their tile themselves, behind its barrier (k_fir_i8's scheme)
 *   1  the LOADERS finish the tile before while the matrix waves work on this one: every wave of the block then has about
 *      the same work per tile; stores from a loader wave sit beside a matrix wave's instructions on its SIMD and need the
 *      padded form (store_f2_padded)
 *   2  TWO matrix waves (0, 1: one component -- or tap set -- each, all four column blocks) and two FINISHING waves (2, 3:
 *      the tile before, while the matrix waves work on this one).  Waves w, w + 4, w + 8 share a SIMD
 *      (tools/ubench/wave_simd.hip): the finishing waves have no matrix wave beside them (packed fp32 is safe there) and,
 *      without tap operands, the registers to keep a second stage's LDS reads in flight
 * (Tried: matrix and dedicated post waves on two SIMDs, six loader waves with the vector port to themselves on the other
 * two -- a matrix instruction holds its SIMD's vector issue port for 8 of its 16 cycles, which costs the loaders beside it
 * 0.05 ms per 2^28 samples -- but six loader waves stream worse than eight: 0.39 ms for every form.)                 */
template <int LAYOUT, int D = 8>
struct Roles {
    static constexpr int NLT = 512;                            /* loader threads */
    static constexpr int NPT = LAYOUT == 0 ? 256 : LAYOUT == 1 ? 512 : 128;       /* threads that finish a tile */
    static constexpr int NGRP = 128 * D;                       /* main groups of 8 samples per tile */
    static constexpr int NQ = (NGRP + NLT - 1) / NLT;          /* ... per loader thread (the last round may be partial) */
};

/* the block's work; `nblk` blocks walk this stream's tiles, this is block `blk` of them (k_fir_i8x: the grid; k_fir_i8x_many:
 * the grid's x dimension, one stream per y) */
template <int HIST, int MODE, bool FUSE2, int LAYOUT, int D = 8>
__device__ __forceinline__ void fir_i8x_block(const FirI8xArgs &a, long long ntiles, int C, long long nblk, long long blk)
{
    using namespace i8x;
    using G = Geo<HIST, MODE, FUSE2, D>;
    using R = Roles<LAYOUT, D>;
    constexpr int TILE_S = G::TILE_S, NGRP = R::NGRP;
    constexpr bool PART = NGRP % R::NLT != 0;                  /* the last round of main groups is a partial one */
    constexpr int PLANE = G::PLANE, KSTEPS = G::KSTEPS, AS = G::AS, NARR = G::NARR, PORCH = G::PORCH, EXTRA = G::EXTRA,
                  FRONT = G::FRONT, NLT = R::NLT, NPT = R::NPT, NQ = R::NQ;
    constexpr bool MIX = MODE != 0;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_i8x[];
    /* [2 buffers][2 components][3 planes][PLANE], then [2 buffers][NARR][AS] first-stage values */
    float *arr_base = reinterpret_cast<float *>(lds_i8x + 12 * PLANE);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   /* (an SGPR: role branches are uniform) */
    const Walk wk{ ntiles, nblk, C };
    Cursor cur{ blk, 0 };
    const uint32_t n0lo = (uint32_t)a.n0;
    const long long n_out = a.n_out ? a.n_out : a.n_in >> 3;

    /* ---- finishing a tile whose values are in `arr` (after its barrier): non-FUSE2 combine, rotate, store float2 (thread
     * pt of NPT); FUSE2 the second stage and the porch copy.  GUARD: the batch's last tile (ragged end, hist2_out). */
    /* (u + j v) (c + j s) in place.  NO packed fp32 in code that loader waves run (this TU is compiled without the SLP
     * vectoriser, and the finishing code below uses no 2-vectors): on MI355X a v_pk_fma_f32 / v_pk_mul_f32 result consumed
     * one or two instructions later came out wrong in lanes 48..63 now and then -- only in waves that share their SIMD with
     * a wave issuing matrix instructions (tools/i8x_debug.py, 20..27 of 40 runs of 600 tiles; none with plain v_fma_f32) */
    auto rotate = [](float &u, float &v, float c, float s) __attribute__((always_inline)) {
        const float x = __builtin_fmaf(-v, s, u * c), y = __builtin_fmaf(v, c, u * s);
        u = x;
        v = y;
    };
    auto put_f2 = [&](float2 *p, float x, float y) __attribute__((always_inline)) {
        if (LAYOUT == 0) {
            const f32x2 v = { x, y };
            __builtin_nontemporal_store(v, reinterpret_cast<f32x2 *>(p));
        } else
            store_f2_padded(p, x, y);
    };
    auto post_store = [&](long long t, const float *arr, int pt, auto guard_c) __attribute__((always_inline)) {
        constexpr bool GUARD = decltype(guard_c)::value;
        float2 *dst = reinterpret_cast<float2 *>(a.out) + t * 1024;
        const long long left = n_out - t * 1024;
#pragma unroll
        for (int o4 = 0; o4 < 1024 / NPT; ++o4) {
            const int o = pt + NPT * o4;
            const int q = 20 * (o >> 4) + (o & 15);
            float uI, uQ;
            if (MODE == 1) {
                uI = (arr[q] - arr[3 * AS + q]) + a.ct[0];
                uQ = (arr[2 * AS + q] + arr[AS + q]) + a.ct[1];
            } else {
                uI = arr[q];
                uQ = arr[AS + q];
            }
            if (MIX) {
                float c, s;
                nco_lo((n0lo + (uint32_t)D * (uint32_t)(t * 1024 + o)) * a.freg + a.phase_off, c, s);
                rotate(uI, uQ, c, s);
            }
            if (!GUARD || o < left)
                put_f2(dst + o, uI, uQ);
        }
    };
    /* second stage, output p of the tile from u[8 p - 64 .. 8 p] (array positions 8 p .. 8 p + 64 behind the porch); the tap
     * tables are in descending order, gre[i] = Re g2[64 - i], so that a float4 of u meets a float4 of taps.
     * WHOLE: one thread forms the complex output.  Otherwise a PAIR of neighbouring lanes forms it: lane `comp` = 0 the real
     * part gr * uI - gi * uQ, lane 1 the imaginary part gr * uQ + gi * uI (the same code on swapped rails), one quad
     * permute brings the halves together for the rotation, and each lane stores its own float.                        */
    auto post_stage2 = [&](long long t, const float *arr, int p) __attribute__((always_inline)) {
        constexpr bool WHOLE = true;
        const float *uA = arr, *uB = arr + AS;
        const f32x2 PDDC_CONSTANT *gre = (const f32x2 PDDC_CONSTANT *)a.taps2;
        const f32x2 PDDC_CONSTANT *gim = (const f32x2 PDDC_CONSTANT *)(a.taps2 + kTaps2Len);
        /* Packed fp32 only where the wave that runs this issues the matrix instructions itself (LAYOUT 0).  The finishing
         * waves of LAYOUT 2 take ONE float per instruction, like the loaders of LAYOUT 1: that they have no matrix wave on
         * their SIMD rests on waves w, w + 4, w + 8 sharing one (tools/ubench/wave_simd.hip) -- measured, not guaranteed
         * (round 5 advisor) -- and packed fp32 beside a matrix wave is the code shape that delivered wrong lanes 48..63
         * (`rotate` above).  The same four partial sums in the same order: the same bits either way.                     */
        constexpr bool PK = LAYOUT == 0;
        f32x2 ra_[2] = { { 0.f, 0.f }, { 0.f, 0.f } }, ib_[2] = { { 0.f, 0.f }, { 0.f, 0.f } };
        f32x2 ia_[2] = { { 0.f, 0.f }, { 0.f, 0.f } }, rb_[2] = { { 0.f, 0.f }, { 0.f, 0.f } };
        float sra[4] = { 0.f, 0.f, 0.f, 0.f }, srb[4] = { 0.f, 0.f, 0.f, 0.f }, sia[4] = { 0.f, 0.f, 0.f, 0.f },
              sib[4] = { 0.f, 0.f, 0.f, 0.f };
        const float PDDC_CONSTANT *gref = (const float PDDC_CONSTANT *)a.taps2;
        const float PDDC_CONSTANT *gimf = (const float PDDC_CONSTANT *)(a.taps2 + kTaps2Len);
        /* (hipcc waits for every float4 pair: sixteen exposed LDS latencies, 4200 cycles per tile on the clock probe -- the
         * reason why LAYOUT 1 gives this work to waves that can keep the reads in flight) */
        constexpr int S2B = LAYOUT == 2 ? 8 : 1;  /* (matrix waves: no registers for more -- batches of 4 spill 12, of 8 47) */
#pragma unroll
        for (int jb = 0; jb < 16; jb += S2B) {
            float4 xA[S2B], xB[S2B];
#pragma unroll
            for (int jj = 0; jj < S2B; ++jj) {
                const int op = 8 * p + 4 * (jb + jj), q = 20 * (op >> 4) + (op & 15);
                xA[jj] = *reinterpret_cast<const float4 *>(uA + q);
                xB[jj] = *reinterpret_cast<const float4 *>(uB + q);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 0; jj < S2B; ++jj) {
                const int j = jb + jj;
                if (PK) {
                    const f32x2 xa[2] = { { xA[jj].x, xA[jj].y }, { xA[jj].z, xA[jj].w } },
                                xb[2] = { { xB[jj].x, xB[jj].y }, { xB[jj].z, xB[jj].w } };
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const f32x2 gr = gre[2 * j + e];
                        ra_[e] = __builtin_elementwise_fma(gr, xa[e], ra_[e]);
                        if (WHOLE)
                            rb_[e] = __builtin_elementwise_fma(gr, xb[e], rb_[e]);
                        if (MIX) {
                            const f32x2 gi = gim[2 * j + e];
                            ib_[e] = __builtin_elementwise_fma(gi, xb[e], ib_[e]);
                            if (WHOLE)
                                ia_[e] = __builtin_elementwise_fma(gi, xa[e], ia_[e]);
                        }
                    }
                } else {
                    const float *xa = &xA[jj].x, *xb = &xB[jj].x;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gr = gref[4 * j + e];
                        sra[e] = __builtin_fmaf(gr, xa[e], sra[e]);
                        srb[e] = __builtin_fmaf(gr, xb[e], srb[e]);
                        if (MIX) {
                            const float gi = gimf[4 * j + e];
                            sib[e] = __builtin_fmaf(gi, xb[e], sib[e]);
                            sia[e] = __builtin_fmaf(gi, xa[e], sia[e]);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!PK) {
            ra_[0] = f32x2{ sra[0], sra[1] };
            ra_[1] = f32x2{ sra[2], sra[3] };
            rb_[0] = f32x2{ srb[0], srb[1] };
            rb_[1] = f32x2{ srb[2], srb[3] };
            ia_[0] = f32x2{ sia[0], sia[1] };
            ia_[1] = f32x2{ sia[2], sia[3] };
            ib_[0] = f32x2{ sib[0], sib[1] };
            ib_[1] = f32x2{ sib[2], sib[3] };
        }
        const int op = 8 * p + 64, q = 20 * (op >> 4) + (op & 15);
        const float g0r = a.taps2[64], g0i = a.taps2[kTaps2Len + 64];
        const float x0A = uA[q], x0B = uB[q];
        const uint32_t ph = (n0lo + 64u * (uint32_t)(t * 128 + p)) * a.freg + a.phase_off;
        if (WHOLE) {
            float zr = ((ra_[0].x + ra_[0].y) + (ra_[1].x + ra_[1].y)) + g0r * x0A;
            float zi = ((rb_[0].x + rb_[0].y) + (rb_[1].x + rb_[1].y)) + g0r * x0B;
            if (MIX) {
                zr -= ((ib_[0].x + ib_[0].y) + (ib_[1].x + ib_[1].y)) + g0i * x0B;
                zi += ((ia_[0].x + ia_[0].y) + (ia_[1].x + ia_[1].y)) + g0i * x0A;
                float c, s;
                nco_lo(ph, c, s);
                const float yr = zr * c - zi * s, yi = zr * s + zi * c;
                zr = yr;
                zi = yi;
            }
            put_f2(reinterpret_cast<float2 *>(a.out) + t * 128 + p, zr, zi);
        }
    };
    /* the same output by a PAIR of neighbouring loader lanes (LAYOUT 1): lane `comp` = 0 the real part gr * uI - gi * uQ,
     * lane 1 the imaginary part gr * uQ + gi * uI -- the same code on swapped rails --, scalar FMAs (see rotate()), the
     * window's LDS reads in batches that are in flight before the first FMA (these waves have the registers the matrix waves
     * lack), one quad permute brings the halves together for the rotation, each lane stores its own float.  A wave takes
     * outputs of ONE parity (B: p = 2 (32 h + i) + B for lane pair i of wave 2 h + B): the window of output p starts at
     * array position 8 p = 16 a + 8 B, so with the parity fixed every read is the lane's base 20 a plus a constant (the 16
     * values of a block lie 20 floats apart), and the taps stay wave-uniform (scalar loads).                           */
    auto post_stage2_pair = [&](long long t, const float *arr, int h, int lane_, auto parity_c) __attribute__((always_inline)) {
        constexpr int B = decltype(parity_c)::value;
        const int comp = lane_ & 1, ap = 32 * h + (lane_ >> 1), p = 2 * ap + B;
        const float *uA = arr + comp * AS + 20 * ap, *uB = arr + (comp ^ 1) * AS + 20 * ap;
        const float PDDC_CONSTANT *gre = (const float PDDC_CONSTANT *)a.taps2;
        const float PDDC_CONSTANT *gim = (const float PDDC_CONSTANT *)(a.taps2 + kTaps2Len);
        float s1[4] = { 0.f, 0.f, 0.f, 0.f }, s2[4] = { 0.f, 0.f, 0.f, 0.f };
        constexpr int NB = 4;
#pragma unroll
        for (int jb = 0; jb < 16; jb += NB) {
            float4 xA[NB], xB[NB];
#pragma unroll
            for (int jj = 0; jj < NB; ++jj) {
                constexpr int dummy = 0;
                (void)dummy;
                const int m = jb + jj + 2 * B, q = 4 * m + 4 * (m >> 2);          /* float4 m of the lane's blocks */
                xA[jj] = *reinterpret_cast<const float4 *>(uA + q);
                if (MIX)
                    xB[jj] = *reinterpret_cast<const float4 *>(uB + q);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 0; jj < NB; ++jj) {
                const int i = 4 * (jb + jj);
                const float *xa = &xA[jj].x, *xb = &xB[jj].x;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s1[e] = __builtin_fmaf(gre[i + e], xa[e], s1[e]);
                    if (MIX)
                        s2[e] = __builtin_fmaf(gim[i + e], xb[e], s2[e]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        constexpr int ml = 16 + 2 * B, ql = 4 * ml + 4 * (ml >> 2);
        float z = ((s1[0] + s1[1]) + (s1[2] + s1[3])) + gre[64] * uA[ql];
        if (MIX) {
            const float w = ((s2[0] + s2[1]) + (s2[2] + s2[3])) + gim[64] * uB[ql];
            z = comp ? z + w : z - w;
            const float other = __builtin_bit_cast(
                float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, z), 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true));
            const float zr = comp ? other : z, zi = comp ? z : other;
            float c, s;
            nco_lo((n0lo + 64u * (uint32_t)(t * 128 + p)) * a.freg + a.phase_off, c, s);
            z = comp ? __builtin_fmaf(zi, c, zr * s) : __builtin_fmaf(-zi, s, zr * c);
        }
        float *dst = a.out + 2 * (t * 128 + p) + comp;
        asm volatile("global_store_dword %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(z) : "memory");
    };
    auto post_porch = [&](long long t, const float *arr, float *other, int c, bool chain, auto guard_c) __attribute__((always_inline)) {
        constexpr bool GUARD = decltype(guard_c)::value;
        const int comp = c >> 6, e = c & 63;
        const int qs = 20 * (64 + (e >> 4)) + (e & 15), qd = 20 * (e >> 4) + (e & 15);
        if (chain)                                           /* inside a chunk the tile's last 64 values are the next one's porch */
            other[comp * AS + qd] = arr[comp * AS + qs];
        if (GUARD && t == ntiles - 1 && a.hist2_out && comp == 0) {
            /* ... and the batch's last 64 the next call's second-stage history: mixed floats, y = u LO */
            float uI = arr[qs], uQ = arr[AS + qs];
            if (MIX) {
                float c, s;
                nco_lo((n0lo + 8u * (uint32_t)(t * 1024 + 960 + e)) * a.freg + a.phase_off, c, s);
                rotate(uI, uQ, c, s);
            }
            put_f2(static_cast<float2 *>(a.hist2_out) + e, uI, uQ);
        }
    };
    /* `next_first`: the tile behind t starts a chunk (it makes its own porch, or takes the stream's history); pt: the
     * finishing thread, 0 .. NPT */
    auto post = [&](long long t, float *arr, float *other, bool next_first, int pt, auto guard_c) __attribute__((always_inline)) {
        const int pw = __builtin_amdgcn_readfirstlane(pt >> 6);
        if (!FUSE2) {
            post_store(t, arr, pt, guard_c);
        } else if (LAYOUT == 0) {
            if (pw < 2)
                post_stage2(t, arr, pt);
            else
                post_porch(t, arr, other, pt - 128, !next_first, guard_c);
        } else if (LAYOUT == 2) {
            post_stage2(t, arr, pt);
            post_porch(t, arr, other, pt, !next_first, guard_c);
        } else {
            if (pw < 4) {
                if (pw & 1)
                    post_stage2_pair(t, arr, pw >> 1, pt & 63, std::integral_constant<int, 1>{});
                else
                    post_stage2_pair(t, arr, pw >> 1, pt & 63, std::integral_constant<int, 0>{});
            }
            else if (pw < 6)
                post_porch(t, arr, other, pt - 256, !next_first, guard_c);
        }
    };
    if (wave >= 4) {
        /* ---- loaders (waves 4..11): main groups two tiles ahead in two register sets, front groups one tile ahead;
         * LAYOUT 1: they also finish the tile before the one the matrix waves are working on */
        const int lt = tid - 256;
        if (blk == 0 && a.hist_out) {             /* the next call's history: the batch's last HIST (hist_len) samples */
            const int hl = a.hist_len ? a.hist_len : HIST;
            const uint4 *src = reinterpret_cast<const uint4 *>(static_cast<const uint8_t *>(a.in) + (a.n_in - hl) * 6);
            if (lt < hl * 6 / 16)
                static_cast<uint4 *>(a.hist_out)[lt] = src[lt];
        }
        /* front groups a chunk's first tile t needs: all FG with the porch columns (FUSE2, t > 0), else the last HG */
        auto front_lo = [&](long long t) __attribute__((always_inline)) { return FUSE2 && t > 0 ? 0 : G::FG - G::HG; };
        uint4 ra[NQ][3], rb[NQ][3], rf[3];
        Cursor c0 = cur, c1 = wk.next(c0), c2 = wk.next(c1);
        long long t = wk.tile(c0), t1 = wk.tile(c1), t2 = wk.tile(c2), tp = -1;
        auto has = [&](int q) __attribute__((always_inline)) { return !PART || q < NQ - 1 || lt + NLT * q < NGRP; };
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (has(q))
                issue_group<TILE_S>(a, t, lt + NLT * q, ra[q]);
        if (lt < G::FG && lt >= front_lo(t)) {
            issue_front<HIST, FRONT, TILE_S>(a, t, lt, rf);
            put_planes(rf, lds_i8x, PLANE, swz(8 * lt));
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (has(q))
                put_planes(ra[q], lds_i8x, PLANE, swz(FRONT + 8 * (lt + NLT * q)));
        if (t1 >= 0) {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (has(q))
                    issue_group<TILE_S>(a, t1, lt + NLT * q, ra[q]);
        }
        __syncthreads();
        /* one step: tile `tn` (already in `cur_r`) goes into plane set `dst` while the matrix waves work on `src`; the loads
         * of the tile after it (`tnn`, if any) go out group by group BETWEEN the conversions (k_fir_i8's pacing) */
        auto step = [&](long long tn, bool tn_first, long long tnn, uint4 (&nxt)[NQ][3], const uint4 (&cur_r)[NQ][3], uint8_t *dst,
                        const uint8_t *src) {
            const bool ff = tn_first && lt < G::FG && lt >= front_lo(tn);
            if (ff)
                issue_front<HIST, FRONT, TILE_S>(a, tn, lt, rf);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int g = lt + NLT * q;
                if (!has(q))
                    continue;
                if (tnn >= 0)
                    issue_group<TILE_S>(a, tnn, g, nxt[q]);
                __builtin_amdgcn_sched_barrier(0);
                put_planes(cur_r[q], dst, PLANE, swz(FRONT + 8 * g));
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ff) {
                put_planes(rf, dst, PLANE, swz(8 * lt));
            } else if (!tn_first && lt < G::HG) {
                /* inside a chunk: the history comes over from the plane set of the tile before */
                const int s_at = swz(FRONT + TILE_S - HIST + 8 * lt), d_at = swz(EXTRA + 8 * lt);
#pragma unroll
                for (int pl = 0; pl < 6; ++pl)
                    *reinterpret_cast<uint2 *>(dst + pl * PLANE + d_at) = *reinterpret_cast<const uint2 *>(src + pl * PLANE + s_at);
            }
        };
        float *arr0 = arr_base, *arr1 = arr_base + NARR * AS;
        int lastbuf;
        for (;;) {
            /* the matrix waves compute tile t from plane set 0 into value set 0; t1 (in ra) goes to plane set 1, t2 starts
             * towards rb; the tile before t (values in set 1) is finished */
            if (t1 >= 0)
                step(t1, c1.k == 0, t2, rb, ra, lds_i8x + 6 * PLANE, lds_i8x);
            if (LAYOUT == 1 && tp >= 0)
                post(tp, arr1, arr0, c0.k == 0, lt, std::false_type{});
            __syncthreads();
            lastbuf = 0;
            if (t1 < 0)
                break;
            tp = t;
            c0 = c1;
            c1 = c2;
            c2 = wk.next(c2);
            t = t1;
            t1 = t2;
            t2 = wk.tile(c2);
            if (t1 >= 0)
                step(t1, c1.k == 0, t2, ra, rb, lds_i8x, lds_i8x + 6 * PLANE);
            if (LAYOUT == 1)
                post(tp, arr0, arr1, c0.k == 0, lt, std::false_type{});
            __syncthreads();
            lastbuf = 1;
            if (t1 < 0)
                break;
            tp = t;
            c0 = c1;
            c1 = c2;
            c2 = wk.next(c2);
            t = t1;
            t1 = t2;
            t2 = wk.tile(c2);
        }
        if (LAYOUT == 1)
            post(t, lastbuf ? arr1 : arr0, lastbuf ? arr0 : arr1, true, lt, std::true_type{});
        return;
    }
    if (LAYOUT == 2 && wave >= 2) {
        /* ---- finishing waves (LAYOUT 2): tile t - 1 is finished while the matrix waves work on tile t */
        const int pt = tid - 128;
        if (FUSE2 && blk == 0 && pt < 64) {
            /* the batch's first tile: the stream's second-stage history -- the 64 first-stage outputs in front of this batch,
             * mixed floats -- taken back into the frame of this call's u values: u = y conj(LO) */
            const float2 y = static_cast<const float2 *>(a.hist2)[pt];
            float uI = y.x, uQ = y.y;
            if (MIX) {
                float c, s;
                nco_lo((n0lo + 8u * (uint32_t)(pt - 64)) * a.freg + a.phase_off, c, s);
                uI = y.x * c + y.y * s;
                uQ = y.y * c - y.x * s;
            }
            const int q = 20 * (pt >> 4) + (pt & 15);
            arr_base[q] = uI;
            arr_base[AS + q] = uQ;
        }
        __syncthreads();
        int buf = 0;
        long long t = wk.tile(cur), tprev = -1;
        while (t >= 0) {
            const bool first = cur.k == 0;
            cur = wk.next(cur);
            if (tprev >= 0)
                post(tprev, arr_base + (buf ^ 1) * NARR * AS, arr_base + buf * NARR * AS, first, pt, std::false_type{});
            __syncthreads();
            tprev = t;
            t = wk.tile(cur);
            buf ^= 1;
        }
        post(tprev, arr_base + (buf ^ 1) * NARR * AS, arr_base + buf * NARR * AS, true, pt, std::true_type{});
        return;
    }
    /* ---- matrix waves (0..3; LAYOUT 2: 0 and 1, both halves of the columns each) */
    const int w0 = wave & 1, half = wave >> 1;
    const int n = lane & 15, kq = lane >> 4;
    const v4i_t *atab = static_cast<const v4i_t *>(a.atab);
    /* tap operand(s): mode 0 the one table; mode 1 table w0 (c or s); mode 2 both paired tables -- [c ; s] meets the I planes,
     * [-s ; c] the Q planes */
    v4i_t A0[KSTEPS][4];
    v4i_t A1[MODE == 2 ? KSTEPS : 1][4];
    if (MODE == 0 && a.taps16 != nullptr) {                 /* uniform */
        /* binary16 tap storage: this lane's 16 columns of every k-step, quantised here exactly as build_tables_core does on
         * the host (H = llround(h 2^E), four balanced base-256 digits).  The device array is laid out for this read:
         * G[128 + tt] = h[HIST - tt] for tt = 1 .. HIST, zeros around it (kFirI8Taps16Len entries), so that the 16 values of
         * a k-step -- T[r][c] = G[128 + c - 8 r] -- are two aligned 16-byte loads */
        const uint4 *g16 = static_cast<const uint4 *>(a.taps16);
        const float two_e = a.two_e;                        /* a power of two: h 2^E is exact in binary32 */
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int i0 = (128 + 64 * ks + 16 * kq - 8 * n) >> 3;      /* in units of 8 values */
            const uint4 lo = g16[i0], hi = g16[i0 + 1];
            const uint32_t hw[8] = { lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w };
            int w[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                w[0][q] = w[1][q] = w[2][q] = w[3][q] = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int jj = 4 * q + b;
                    const uint32_t bits = (jj & 1) ? hw[jj >> 1] >> 16 : hw[jj >> 1] & 0xffffu;
                    _Float16 hv;
                    const uint16_t b16 = (uint16_t)bits;
                    __builtin_memcpy(&hv, &b16, 2);
                    const float x = (float)hv * two_e;
                    int r = (int)(x + __builtin_copysignf(0.5f, x));   /* llround: exact, |x| <= 2^30 and 11 bits wide */
                    /* balanced digits: d = the low byte, signed; what is left is (r - d) / 256 = (r + 128) >> 8.  Byte b of
                     * plane j's word takes the low byte as it is (one v_perm_b32) */
                    const uint32_t sel = 0x03020100u ^ ((0x04u ^ (uint32_t)b) << (8 * b));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        w[j][q] = (int)__builtin_amdgcn_perm((uint32_t)r, (uint32_t)w[j][q], sel);
                        r = (r + 128) >> 8;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                A0[ks][j] = v4i_t{ w[j][0], w[j][1], w[j][2], w[j][3] };
        }
    } else {
        const int tab0 = MODE == 1 ? w0 : 0;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                A0[ks][j] = atab[tab0 * G::TABV + (j * KSTEPS + ks) * 64 + lane];
                if (MODE == 2)
                    A1[ks][j] = atab[G::TABV + (j * KSTEPS + ks) * 64 + lane];
            }
    }
    if (LAYOUT != 2 && FUSE2 && blk == 0 && tid < 64) {
        /* the batch's first tile: the stream's second-stage history -- the 64 first-stage outputs in front of this batch,
         * mixed floats -- taken back into the frame of this call's u values: u = y conj(LO) */
        const float2 y = static_cast<const float2 *>(a.hist2)[tid];
        float uI = y.x, uQ = y.y;
        if (MIX) {
            float c, s;
            nco_lo((n0lo + 8u * (uint32_t)(tid - 64)) * a.freg + a.phase_off, c, s);
            uI = y.x * c + y.y * s;
            uQ = y.y * c - y.x * s;
        }
        const int q = 20 * (tid >> 4) + (tid & 15);
        arr_base[q] = uI;
        arr_base[AS + q] = uQ;
    }
    /* one band pass of this wave over 16 columns whose operand bytes start at plane position `pos` (per lane: + 16 kq + 64 ks) */
    auto band = [&](const uint8_t *pb, int pos, float *dst) __attribute__((always_inline)) {
        if (MODE == 1) {
#pragma unroll
            for (int comp = 0; comp < 2; ++comp) {
                v4i_t acc[4];
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc[s] = v4i_t{ 0, 0, 0, 0 };
                /* The operand reads run ONE k-step ahead of the matrix instructions that use them, in two register sets:
                 * hipcc left to itself issues a k-step's reads three or four matrix instructions (50-60 cycles) before their
                 * first use and waits -- the 216 matrix instructions of a tile filled 57 % of these waves' 6000 ticks of work
                 * (clock probe, NOTEBOOK R5.7), and these waves are the chain the block waits for in this form */
                constexpr bool PF1 = LAYOUT == 1;               /* (the form's default layout; the others have no registers left) */
                v4i_t B[PF1 ? 2 : 1][3];
                auto read_b1 = [&](int ks, v4i_t (&b)[3]) __attribute__((always_inline)) {
                    const int at = swz(pos + 64 * ks);
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        b[i] = *reinterpret_cast<const v4i_t *>(pb + (3 * comp + i) * PLANE + at);
                };
                if (PF1)
                    read_b1(0, B[0]);
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) {
                    if (!PF1)
                        read_b1(ks, B[0]);
                    else if (ks + 1 < KSTEPS)
                        read_b1(ks + 1, B[(ks + 1) & 1]);
                    if (PF1)
                        __builtin_amdgcn_sched_barrier(0);
                    const v4i_t(&b)[3] = B[PF1 ? ks & 1 : 0];
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (i + j >= 2)
                                acc[i + j - 2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A0[ks][j], b[i], acc[i + j - 2], 0, 0, 0);
                    if (PF1)
                        __builtin_amdgcn_sched_barrier(0);
                }
                float4 y;
                float *yp = &y.x;
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    yp[v] = recombine(acc, v) * a.scale;
                *reinterpret_cast<float4 *>(dst + (2 * w0 + comp) * AS) = y;
            }
        } else {
            v4i_t acc[4];
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc[s] = v4i_t{ 0, 0, 0, 0 };
            /* (operand reads one k-step ahead of their matrix instructions, two register sets: see mode 1 above) -- for the long
             * untuned first stage only (six k-steps: there these waves' chain is what the block waits for, 255 taps 0.3451 ->
             * 0.3369 ms).  Where the loaders are the longer chain the pinned order costs 2 % (two-k-step tuned form 0.3260 ->
             * 0.3320), and the three-k-step tuned form has no registers for a second set (profiles/r05/f_ab_operand_prefetch.txt) */
            constexpr bool PF = MODE == 0 && KSTEPS > 4 && !FUSE2;
            if (PF) {
                v4i_t B[2][3];
                auto read_b = [&](int ks, v4i_t (&b)[3]) __attribute__((always_inline)) {
                    const int at = swz(pos + 64 * ks);
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        b[i] = *reinterpret_cast<const v4i_t *>(pb + (3 * w0 + i) * PLANE + at);
                };
                read_b(0, B[0]);
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) {
                    if (ks + 1 < KSTEPS)
                        read_b(ks + 1, B[(ks + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (i + j >= 2)
                                acc[i + j - 2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A0[ks][j], B[ks & 1][i], acc[i + j - 2], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const int at = swz(pos + 64 * ks);
                if (MODE == 0) {
                    v4i_t B[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        B[i] = *reinterpret_cast<const v4i_t *>(pb + (3 * w0 + i) * PLANE + at);
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (i + j >= 2)
                                acc[i + j - 2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A0[ks][j], B[i], acc[i + j - 2], 0, 0, 0);
                } else {
                    v4i_t BI[3], BQ[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        BI[i] = *reinterpret_cast<const v4i_t *>(pb + i * PLANE + at);
                        BQ[i] = *reinterpret_cast<const v4i_t *>(pb + (3 + i) * PLANE + at);
                    }
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (i + j >= 2) {
                                acc[i + j - 2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A0[ks][j], BI[i], acc[i + j - 2], 0, 0, 0);
                                acc[i + j - 2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A1[MODE == 2 ? ks : 0][j], BQ[i], acc[i + j - 2], 0, 0, 0);
                            }
                }
            }
            }
            /* this lane: its column, rows 4 kq + v -> four consecutive values; mode 2: rows 0..7 are uI, rows 8..15 uQ of the
             * column's eight outputs */
            const int rail = MODE == 2 ? kq >> 1 : w0;
            float4 y;
            float *yp = &y.x;
#pragma unroll
            for (int v = 0; v < 4; ++v)
                yp[v] = recombine(acc, v) * a.scale + (rail ? a.ct[1] : a.ct[0]);
            *reinterpret_cast<float4 *>(dst + rail * AS) = y;
        }
    };
    __syncthreads();
    int buf = 0;
    long long t = wk.tile(cur);
    while (t >= 0) {
        const bool first = cur.k == 0;
        cur = wk.next(cur);
        const uint8_t *pb = lds_i8x + buf * 6 * PLANE;
        float *arr = arr_base + buf * NARR * AS, *arr_o = arr_base + (buf ^ 1) * NARR * AS;
        if (MODE == 2) {
            /* columns of 8 outputs, 8 D = 64 (80) samples apart: 128 of them in a tile, 16 per pass; every matrix wave makes both rails.
             * Output o = 8 col + 4 (kq & 1) + v lies at array position o + PORCH. */
            constexpr int NMW = LAYOUT == 2 ? 2 : 4;
            if (FUSE2 && first && t > 0 && wave == NMW - 1) {
                /* a chunk's first tile computes its own porch: columns -8 .. -1 in lanes 8..15; the other lanes repeat column
                 * -8 (same operand bytes, same results, same address: the whole wave runs the matrix instructions) */
                const int colx = n < 8 ? -8 : n - 16, pp = 64 + 8 * colx + 4 * (kq & 1);
                band(pb, EXTRA + 8 * D * colx + 16 * kq, arr + 20 * (pp >> 4) + (pp & 15));
            }
#pragma unroll
            for (int cb = 0; cb < 8 / NMW; ++cb) {
                const int col = 16 * ((8 / NMW) * wave + cb) + n, pp = PORCH + 8 * col + 4 * (kq & 1);
                band(pb, EXTRA + 8 * D * col + 16 * kq, arr + 20 * (pp >> 4) + (pp & 15));
            }
        } else {
        if (FUSE2 && first && t > 0 && (LAYOUT == 2 || half == 1)) {
            /* a chunk's first tile computes its own porch: columns -4 .. -1 in lanes 12..15; the other lanes repeat column -4
             * (same operand bytes, same results, same address: the whole wave runs the matrix instructions) */
            const int colx = n < 12 ? -4 : n - 16;
            band(pb, EXTRA + 128 * colx + 16 * kq, arr + 20 * (colx + 4) + 4 * kq);
        }
#pragma unroll
        for (int cb = 0; cb < (LAYOUT == 2 ? 4 : 2); ++cb) {
            const int col = 16 * (LAYOUT == 2 ? cb : 2 * half + cb) + n;
            band(pb, EXTRA + 128 * col + 16 * kq, arr + 20 * (col + PORCH / 16) + 4 * kq);
        }
        }
        __syncthreads();                 /* ONE barrier per tile: the next tile's planes are written, this tile's values are in LDS */
        if (LAYOUT == 0)                 /* ... and these waves finish it themselves */
            post(t, arr, arr_o, cur.k == 0, tid, std::true_type{});
        t = wk.tile(cur);
        buf ^= 1;
    }
}

template <int HIST, int MODE, bool FUSE2, int LAYOUT, int D = 8>
__global__ __launch_bounds__(768, 1) void k_fir_i8x(FirI8xArgs a, long long ntiles, int C)
{
    fir_i8x_block<HIST, MODE, FUSE2, LAYOUT, D>(a, ntiles, C, (long long)gridDim.x, (long long)blockIdx.x);
}

/* several streams, one launch (the gang: receivers that share a GPU): blockIdx.y is the stream, every record its own --
 * buffers, history, tuning word, tap tables; the same block body, so the same bits as a launch of its own */
template <int HIST, int MODE, bool FUSE2, int LAYOUT>
__global__ __launch_bounds__(768, 1) void k_fir_i8x_many(FirI8xMany m, long long ntiles, int C)
{
    /* this stream's record, field by field through the scalar cache (indexing the by-value array puts a copy on the stack) */
    const FirI8xArgs PDDC_CONSTANT &r = ((const FirI8xArgs PDDC_CONSTANT *)__builtin_amdgcn_kernarg_segment_ptr())[blockIdx.y];
    (void)m;
    FirI8xArgs a;
    a.in = r.in;
    a.hist = r.hist;
    a.hist_out = r.hist_out;
    a.out = r.out;
    a.atab = r.atab;
    a.n_in = r.n_in;
    a.scale = r.scale;
    a.ct[0] = r.ct[0];
    a.ct[1] = r.ct[1];
    a.n0 = r.n0;
    a.freg = r.freg;
    a.phase_off = r.phase_off;
    a.taps2 = r.taps2;
    a.hist2 = r.hist2;
    a.hist2_out = r.hist2_out;
    a.taps16 = r.taps16;
    a.two_e = r.two_e;
    fir_i8x_block<HIST, MODE, FUSE2, LAYOUT>(a, ntiles, C, (long long)gridDim.x, (long long)blockIdx.x);
}

/* ---- host side: the tap operands ------------------------------------------------------------------------------------ */
/* H[k] (k = 0 .. hist-1, |H| <= 2^30) -> four planes of balanced base-256 digits in the matrix instruction's lane order:
 * lane l of k-step ks holds A[row l & 15][k = 16 (l >> 4) + jj], T[r][c] = H[hist - (c - 8 r)] */
static bool i8_fill_table(const long long *H, int hist, int8_t *table)
{
    std::vector<int8_t> dig(4 * (size_t)hist);
    for (int k = 0; k < hist; ++k) {
        long long r = H[k];
        for (int j = 0; j < 4; ++j) {
            const long long d = j == 3 ? r : ((r + 128) & 255) - 128;
            if (d < -128 || d > 127)
                return false;
            dig[(size_t)j * hist + k] = (int8_t)d;
            r = (r - d) / 256;
        }
    }
    const int ksteps = (120 + hist + 63) / 64;
    for (int j = 0; j < 4; ++j)
        for (int ks = 0; ks < ksteps; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int jj = 0; jj < 16; ++jj) {
                    const int r = l & 15, c = 64 * ks + 16 * (l >> 4) + jj, tt = c - 8 * r;
                    table[(((size_t)j * ksteps + ks) * 64 + l) * 16 + jj] =
                        (tt >= 1 && tt <= hist) ? dig[(size_t)j * hist + (hist - tt)] : 0;
                }
    return true;
}

/* mode 2: the same digits with the two tap sets of a row block in ONE operand -- rows 0..7 the band of `top` over eight
 * outputs, rows 8..15 the band of `bot` over the same eight: lane l of k-step ks holds A[row l & 15][k = 16 (l >> 4) + jj],
 * T[r][c] = H_(r >> 3)[hist - (c - 8 (r & 7))] */
static bool i8_fill_table_paired(const long long *top, const long long *bot, int hist, int8_t *table, int D = 8)
{
    std::vector<int8_t> dig(8 * (size_t)hist);
    for (int set = 0; set < 2; ++set)
        for (int k = 0; k < hist; ++k) {
            long long r = (set ? bot : top)[k];
            for (int j = 0; j < 4; ++j) {
                const long long d = j == 3 ? r : ((r + 128) & 255) - 128;
                if (d < -128 || d > 127)
                    return false;
                dig[((size_t)set * 4 + j) * hist + k] = (int8_t)d;
                r = (r - d) / 256;
            }
        }
    const int ksteps = (7 * D + hist + 63) / 64;
    for (int j = 0; j < 4; ++j)
        for (int ks = 0; ks < ksteps; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int jj = 0; jj < 16; ++jj) {
                    const int r = l & 15, c = 64 * ks + 16 * (l >> 4) + jj, tt = c - D * (r & 7);
                    table[(((size_t)j * ksteps + ks) * 64 + l) * 16 + jj] =
                        (tt >= 1 && tt <= hist) ? dig[((size_t)(r >> 3) * 4 + j) * hist + (hist - tt)] : 0;
                }
    return true;
}

int fir_i8x_mode(int hist, bool mix) { return !mix ? 0 : hist <= 128 ? 2 : 1; }

static int fir_i8x_ksteps(int hist, int mode) { return ((mode == 2 ? 56 : 120) + hist + 63) / 64; }

size_t fir_i8x_table_bytes(int hist, bool mix)
{
    const int mode = fir_i8x_mode(hist, mix), ksteps = fir_i8x_ksteps(hist, mode);
    return (size_t)(mode == 0 ? 1 : 2) * 4 * ksteps * 64 * 16;
}

static bool build_tables_core(const float *taps, int ntaps, int hist, bool mix, uint32_t freg, int8_t *tables, float *scale,
                              float ct[2], int *exp2, int D, int delay);

bool fir_i8x_build_tables(const float *taps, int ntaps, int hist, bool mix, uint32_t freg, int8_t *tables, float *scale,
                          float ct[2], int *exp2)
{
    return build_tables_core(taps, ntaps, hist, mix, freg, tables, scale, ct, exp2, 8, 0);
}

size_t fir_i8x_d10_table_bytes() { return (size_t)2 * 4 * ((70 + kFirI8xD10Hist + 63) / 64) * 64 * 16; }

bool fir_i8x_d10_build_tables(const float *taps, int ntaps, int delay, uint32_t freg, int8_t *tables, float *scale, float ct[2])
{
    return build_tables_core(taps, ntaps, kFirI8xD10Hist, true, freg, tables, scale, ct, nullptr, 10, delay);
}

/* taps h[0 .. ntaps) delayed by `delay` samples: g[k] = h[k - delay] e^{+j theta k} (D = 8: delay = 0) */
static bool build_tables_core(const float *taps, int ntaps, int hist, bool mix, uint32_t freg, int8_t *tables, float *scale,
                              float ct[2], int *exp2, int D, int delay)
{
    if (!taps || ntaps < 1 || (hist != 32 && hist != 64 && hist != 128 && hist != 256) || delay < 0 || ntaps + delay > hist || !tables)
        return false;
    if (D != 8 && !(D == 10 && mix && hist == kFirI8xD10Hist))
        return false;
    double hmax = 0.0;
    for (int k = 0; k < ntaps; ++k)
        hmax = std::fmax(hmax, std::fabs((double)taps[k]));
    if (!(hmax > 0.0) || !std::isfinite(hmax))
        return false;
    const int E = 30 - (int)std::ceil(std::log2(hmax));            /* |H| <= 2^30 for h, a fortiori for h cos, h sin */
    if (exp2)
        *exp2 = E;
    const int mode = fir_i8x_mode(hist, mix);
    const size_t tb = (size_t)4 * (D == 8 ? fir_i8x_ksteps(hist, mode) : (7 * D + hist + 63) / 64) * 64 * 16;
    /* sample = v24 / 8388607 (perseustest.c:466-502); planes 0 and 1 are stored minus 128: V = planes + 32896 */
    const double unit = std::ldexp(1.0, -E) / 8388607.0;
    *scale = (float)unit;
    std::vector<long long> Hc((size_t)hist, 0), Hs((size_t)hist, 0), Hm((size_t)hist, 0);
    long long sc = 0, ss = 0;
    const double w = 6.283185307179586476925286766559 / 4294967296.0;
    for (int kk = 0; kk < ntaps; ++kk) {
        const int k = kk + delay;
        const double hk = std::ldexp((double)taps[kk], E);
        if (mix) {
            const double th = w * (double)(uint32_t)((uint64_t)k * freg);
            Hc[k] = std::llround(hk * std::cos(th));
            Hs[k] = std::llround(hk * std::sin(th));
        } else {
            Hc[k] = std::llround(hk);
        }
        Hm[k] = -Hs[k];
        sc += Hc[k];
        ss += Hs[k];
    }
    if (mode == 2) {
        /* [c ; s] meets the I planes, [-s ; c] the Q planes, both into the same accumulators: rows 0..7 uI, rows 8..15 uQ */
        if (!i8_fill_table_paired(Hc.data(), Hs.data(), hist, tables, D) ||
            !i8_fill_table_paired(Hm.data(), Hc.data(), hist, tables + tb, D))
            return false;
    } else {
        if (!i8_fill_table(Hc.data(), hist, tables))
            return false;
        if (mix && !i8_fill_table(Hs.data(), hist, tables + tb))
            return false;
    }
    ct[0] = (float)((double)(sc - ss) * 32896.0 * unit);           /* uI = gc xI - gs xQ */
    ct[1] = (float)((double)(ss + sc) * 32896.0 * unit);           /* uQ = gs xI + gc xQ */
    return true;
}

void fir_i8x_taps2(const float *taps2, int ntaps2, bool mix, uint32_t freg, float *out)
{
    /* out[i] = Re g2[64 - i], out[kFirI8xTaps2Len / 2 + i] = Im g2[64 - i]; g2[k] = h2[k] e^{+j 8 theta k}: a first-stage
     * output is 8 input samples */
    for (int i = 0; i < kFirI8xTaps2Len; ++i)
        out[i] = 0.0f;
    const double w = 6.283185307179586476925286766559 / 4294967296.0;
    for (int k = 0; k < ntaps2 && k < 64; ++k) {
        double c = 1.0, s = 0.0;
        if (mix) {
            const double th = w * (double)(uint32_t)((uint64_t)(8 * k) * freg);
            c = std::cos(th);
            s = std::sin(th);
        }
        out[64 - k] = (float)((double)taps2[k] * c);
        out[kFirI8xTaps2Len / 2 + 64 - k] = (float)((double)taps2[k] * s);
    }
}

template <int HIST, int MODE, bool FUSE2, int LAYOUT, int D = 8>
static hipError_t launch_fir_i8x_l(const FirI8xArgs &a, int max_blocks, int chunk, hipStream_t s, const FirI8xMany *many = nullptr,
                                   int nmany = 0)
{
    using G = i8x::Geo<HIST, MODE, FUSE2, D>;
    const long long ntiles = D == 8 ? (a.n_in + i8x::TILE - 1) / i8x::TILE : (a.n_out + 1023) / 1024;
    int dev = 0;
    (void)hipGetDevice(&dev);
    static int cus[64] = { 0 };
    if (cus[dev & 63] == 0) {
        int v = 0;
        hipError_t e = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess)
            return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fir_i8x<HIST, MODE, FUSE2, LAYOUT, D>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
        if (e != hipSuccess)
            return e;
        if (D == 8)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fir_i8x_many<HIST, MODE, FUSE2, LAYOUT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
        if (e != hipSuccess)
            return e;
        cus[dev & 63] = v > 0 ? v : 256;
    }
    /* chunks of C tiles go round the blocks: C = 1 (tile-interleaved) unless asked otherwise; never more than a block's
     * fair share, so that every CU has work */
    long long nblk = cus[dev & 63];
    if (many && nmany > 1)
        nblk = (nblk + nmany - 1) / nmany;        /* the streams of a round share the CUs */
    if (max_blocks > 0 && nblk > max_blocks)
        nblk = max_blocks;
    long long C = chunk > 0 ? chunk : FUSE2 ? (LAYOUT == 2 ? 8 : 4) : 1;
    if (C > (ntiles + nblk - 1) / nblk)
        C = (ntiles + nblk - 1) / nblk;
    const long long nchunks = (ntiles + C - 1) / C;
    const long long grid = nchunks < nblk ? nchunks : nblk;
    if (many && D == 8)
        hipLaunchKernelGGL((k_fir_i8x_many<HIST, MODE, FUSE2, LAYOUT>), dim3((unsigned)grid, (unsigned)nmany), dim3(768), G::LDS_BYTES, s,
                           *many, ntiles, (int)C);
    else
        hipLaunchKernelGGL((k_fir_i8x<HIST, MODE, FUSE2, LAYOUT, D>), dim3((unsigned)grid), dim3(768), G::LDS_BYTES, s, a, ntiles, (int)C);
    return hipGetLastError();
}

struct I8xManyRef {
    const FirI8xMany *m = nullptr;
    int n = 0;
};
static thread_local I8xManyRef t_many;       /* (set by launch_fir_i8x_many around its dispatch through the same switch) */
template <int HIST, int MODE, bool FUSE2>
static hipError_t launch_fir_i8x_t(const FirI8xArgs &a, int max_blocks, int chunk, int layout, hipStream_t s)
{
    return layout == 0   ? launch_fir_i8x_l<HIST, MODE, FUSE2, 0>(a, max_blocks, chunk, s, t_many.m, t_many.n)
           : layout == 1 ? launch_fir_i8x_l<HIST, MODE, FUSE2, 1>(a, max_blocks, chunk, s, t_many.m, t_many.n)
                         : launch_fir_i8x_l<HIST, MODE, FUSE2, 2>(a, max_blocks, chunk, s, t_many.m, t_many.n);
}

template <int HIST>
static hipError_t launch_fir_i8x_h(const FirI8xArgs &a, bool mix, bool fuse2, int max_blocks, int chunk, int layout, hipStream_t s)
{
    constexpr int MM = HIST <= 128 ? 2 : 1;
    if (!mix)
        return fuse2 ? launch_fir_i8x_t<HIST, 0, true>(a, max_blocks, chunk, layout, s) : launch_fir_i8x_t<HIST, 0, false>(a, max_blocks, chunk, layout, s);
    if (!fuse2)
        return launch_fir_i8x_t<HIST, MM, false>(a, max_blocks, chunk, layout, s);
    if constexpr (MM == 2)
        return launch_fir_i8x_t<HIST, 2, true>(a, max_blocks, chunk, layout, s);
    else
        return hipErrorInvalidValue;
}


bool fir_i8x_supported(int hist, bool mix, bool fuse2)
{
    if (hist != 32 && hist != 64 && hist != 128 && hist != 256)
        return false;
    return !(fuse2 && mix && hist > 128);
}

hipError_t launch_fir_i8x(const FirI8xArgs &a, int hist, bool mix, bool fuse2, hipStream_t s, int max_blocks, int chunk, int layout)
{
    if (a.n_in <= 0)
        return hipSuccess;
    if ((a.n_in & 7) || !a.in || !a.hist || !a.out || (!a.atab && !(a.taps16 && !mix)) || (a.hist_out && a.n_in < hist) ||
        !fir_i8x_supported(hist, mix, fuse2))
        return hipErrorInvalidValue;
    if (a.taps16 && (mix || !(a.two_e > 0.0f)))            /* binary16-stored taps: the untuned form only (tuned tables come from the host) */
        return hipErrorInvalidValue;
    if (fuse2 && ((a.n_in % i8x::TILE) || !a.taps2 || !a.hist2))
        return hipErrorInvalidValue;
    /* by form: the loaders finish the tile where the finish is heavy -- the four partial products of 129..256 tuned taps, and
     * the fused pair's second stage for SMALL batches; from 1024 tiles on (2^23 samples: four tiles per block and more) the
     * pair runs on two matrix + two finishing waves with chunks of 8 tiles.  Same box, the API's 250 kS/s pair, layout 1 /
     * chunks of 4 against layout 2 / chunks of 8: 2^22 7.6 / 7.7 us, 2^24 23.9 / 21.4, 2^26 99.7 / 88.5 (k_fir8's pair: 93.7),
     * 2^28 374 / 336 (profiles/r05/c_pair_layouts.txt) */
    if (layout < 0)
        layout = fuse2 ? (a.n_in >= (1LL << 23) ? 2 : 1) : (mix && hist > 128) ? 1 : 0;
    switch (hist) {
    case 32:
        return launch_fir_i8x_h<32>(a, mix, fuse2, max_blocks, chunk, layout, s);
    case 64:
        return launch_fir_i8x_h<64>(a, mix, fuse2, max_blocks, chunk, layout, s);
    case 128:
        return launch_fir_i8x_h<128>(a, mix, fuse2, max_blocks, chunk, layout, s);
    default:
        return launch_fir_i8x_h<256>(a, mix, fuse2, max_blocks, chunk, layout, s);
    }
}

hipError_t launch_fir_i8x_d10(const FirI8xArgs &a, hipStream_t s, int max_blocks, int chunk, int layout)
{
    if (a.n_out <= 0)
        return hipSuccess;
    if ((a.n_in & 7) || (a.in_off & 7) || a.in_off < 0 || !a.in || !a.hist || !a.out || !a.atab || a.taps2 ||
        (a.hist_out && a.n_in < a.hist_len) ||
        a.hist_len < 8 || (a.hist_len & 7) || (long long)a.in_off - 7 + 10 * (a.n_out - 1) >= a.n_in)
        return hipErrorInvalidValue;
    return layout == 1   ? launch_fir_i8x_l<kFirI8xD10Hist, 2, false, 1, 10>(a, max_blocks, chunk, s)
           : layout == 2 ? launch_fir_i8x_l<kFirI8xD10Hist, 2, false, 2, 10>(a, max_blocks, chunk, s)
                         : launch_fir_i8x_l<kFirI8xD10Hist, 2, false, 0, 10>(a, max_blocks, chunk, s);
}

hipError_t launch_fir_i8x_many(const FirI8xMany &m, int n, int hist, bool mix, bool fuse2, hipStream_t s, int max_blocks, int chunk,
                               int layout)
{
    if (n < 1 || n > kFir8ManyMax)
        return hipErrorInvalidValue;
    for (int i = 1; i < n; ++i)
        if (m.a[i].n_in != m.a[0].n_in)
            return hipErrorInvalidValue;
    t_many = I8xManyRef{ &m, n };
    const hipError_t e = launch_fir_i8x(m.a[0], hist, mix, fuse2, s, max_blocks, chunk, layout);
    t_many = I8xManyRef{};
    return e;
}

} // namespace pddc
