/*
 * ddc_fir_i8.hip -- the decimate-by-8 first stages on the INT8 matrix cores of gfx950 (MI355X, CDNA4).
 *
 *   k_fir_i8    65..256 taps, no NCO: the wire bytes are the operand planes, the taps four planes of balanced
 *               base-256 digits, int32 accumulation is exact (round 3; BASELINE configs 2 and 5).
 *   k_fir_i8x   the same product with the NCO folded into the TAPS (round 4):
 *                   y[m] = LO(n0 + 8 m) * sum_k (h[k] e^{+j theta k}) x_raw[8 m - k],   theta = 2 pi freg / 2^32,
 *               i.e. complex taps on the raw integer planes (four real band products instead of one) and ONE float
 *               rotation per output, with the exact 32-bit phase; optionally a second decimate-by-8 stage fused behind
 *               it (the cascades' pair: the 1 B/sample intermediate never reaches HBM); blocks walk contiguous tile
 *               ranges and carry the filter history from tile to tile inside LDS.
 *
 * Reference anchors: the samples are the 24-bit wire format of examples/perseustest.c:449-455, the tuning word is
 * perseus-sdr.c:584; the arithmetic itself has no reference source (FPGA bitstreams), DESIGN.md 3.
 */
#include "ddc_kernels.h"
#include "ddc_dev.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace pddc {

/* ======================================================================== */
/* k_fir_i8 : 129..256 taps, decimate by 8, packed input, no NCO -- int8 MFMA */
/* ======================================================================== */
/* The 255-tap first stage is the one configuration that is bound by vector issue, not by HBM: 17 G
 * multiply-adds per 2^28 samples on a power-capped clock.  fp32 MFMA has the vector unit's own peak; int8 MFMA has
 * thirty times that, and this data fits it exactly: a 24-bit sample is three bytes, a tap quantised to 2^-E (E = 30 -
 * ceil(log2 max|h|), i.e. 31 significant bits on the largest tap) is four balanced base-256 digits, every digit x
 * byte-plane product sum over 256 taps stays below 2^24, so int32 accumulation is EXACT; products of equal weight
 * 256^(i+j) share an accumulator, the three lightest (i + j < 2: below 1.2e-7 of full scale even if every term had the
 * same sign, 2e-9 typical) are dropped, and the four sums are recombined in fp32 once per output.  The unpack is gone:
 * the loader only de-interleaves bytes (v_perm) into six planes (planes 0 and 1 xor 0x80: unsigned -> signed, the
 * offset comes back as one constant per filter).
 *   out[16 n + r] = sum_c T[r][c] X[c][n],  T[r][c] = h[256 - (c - 8 r)] (banded Toeplitz, 16 x 384: two thirds full),
 *   X[c][n] = xp[8192 tile + 128 n + c],    xp = the 256 history samples followed by the batch.
 * v_mfma_i32_16x16x64_i8: 16 output rows, 16 columns, 6 k-steps of 64; 9 plane products per step.  (The first version
 * used 32x32x32: a 32 x 512 band that is half zeros, the k range split over two waves and their partial sums added
 * through LDS: 0.409 ms; this one 0.37.)  A tile = 64 columns x 16 outputs = 8192 inputs (+256).  Persistent block of 12
 * waves per CU: waves 8..11 load -- two tiles ahead, two register sets used alternately so that no register copy waits
 * for a load -- and write the planes of the next tile; waves 0..7 (component x block of 16 columns; ALL 24 tap
 * fragments, 96 VGPRs, resident in registers) run 54 MFMAs per tile, recombine, scale and leave their outputs in LDS;
 * ONE barrier per tile; then they store float2.  Planes and outputs exist twice.  LDS rows are padded (lane stride
 * 144 B / 20 floats): conflict-free.  Measured as stand-alone prototypes first (tools/ubench/fir_i8_planes*.hip).     */
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));
namespace i8 {
#ifndef PDDC_I8_WIDE
#define PDDC_I8_WIDE 1
#endif
/* MFMA waves + loader threads per block.  The loaders (global loads, byte de-interleave, plane writes) are the half the
 * kernel sits on, the matrix work has room: 8 + 256 (4 loader waves) 0.3608 ms, 4 + 512 0.3454 ms for 255 taps,
 * 0.3378 -> 0.3250 ms for 127 (same-box A/B, tools/ab_libs.sh). */
constexpr int NMW = PDDC_I8_WIDE == 2 ? 2 : PDDC_I8_WIDE == 1 ? 4 : 8, NLT = PDDC_I8_WIDE == 2 ? 640 : PDDC_I8_WIDE == 1 ? 512 : 256,
              NB = 8 / NMW;
/* HIST = 256 (129..256 taps) or 128 (65..128 taps): history samples in front of the batch = the filter's reach */
template <int HIST>
struct Geo {
    static constexpr int TILE = 8192, SPAN = TILE + HIST, PLANE = SPAN + 16 * ((SPAN + 127) / 128), NG = SPAN / 8;
    static constexpr int KSTEPS = (120 + HIST + 63) / 64;          /* the band is 16 x (8 * 15 + HIST) wide */
    static constexpr int NQ = (NG + NLT - 1) / NLT;
    static constexpr size_t LDS_BYTES = 12 * (size_t)PLANE + 4 * (size_t)(20 * 64) * sizeof(float);
};
constexpr int OS = 20 * 64, TILE = 8192;

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

/* the loads of one tile: group g of tile t is xp[8192 t + 8 g ..+8) -- history, batch, or (behind the batch) zeros.
 * (Measured and NOT kept, same-box A/B of whole library builds, tools/ab_libs.sh, profiles/r03/i_fir_i8_prototype.txt: a
 * branch-free path for interior tiles -- one base pointer, constant strides -- 0.378 -> 0.399 ms; on top of it the byte
 * de-interleave in 24 instead of 36 v_perm -- 0.409 ms, although the stand-alone prototype gains 2.7 % from it.  With the
 * address arithmetic between them the loads leave spread out; as one burst they are slower.)                      */
template <int HIST>
__device__ __forceinline__ void issue_tile(const FirI8Args &a, long long t, uint4 (&raw)[Geo<HIST>::NQ][3], int lt)
{
    constexpr int NQ = Geo<HIST>::NQ, NG = Geo<HIST>::NG;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int g = lt + NLT * q;
        if (g < NG) {
            const long long b = t * TILE + 8LL * g - HIST;         /* first sample of the group, relative to the batch */
            const uint4 *p = b < 0 ? reinterpret_cast<const uint4 *>(static_cast<const uint8_t *>(a.hist) + (b + HIST) * 6)
                                   : reinterpret_cast<const uint4 *>(static_cast<const uint8_t *>(a.in) + b * 6);
            if (b + 8 <= a.n_in) {
                raw[q][0] = p[0];
                raw[q][1] = p[1];
                raw[q][2] = p[2];
            } else {
                raw[q][0] = raw[q][1] = raw[q][2] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    }
}

template <int HIST>
__device__ __forceinline__ void planes_from(const uint4 (&raw)[Geo<HIST>::NQ][3], uint8_t *plane, int lt)
{
    constexpr int NQ = Geo<HIST>::NQ, NG = Geo<HIST>::NG, PLANE = Geo<HIST>::PLANE;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int g = lt + NLT * q;
        if (g < NG) {
            const uint32_t w[12] = { raw[q][0].x, raw[q][0].y, raw[q][0].z, raw[q][0].w, raw[q][1].x, raw[q][1].y,
                                     raw[q][1].z, raw[q][1].w, raw[q][2].x, raw[q][2].y, raw[q][2].z, raw[q][2].w };
            const int at = swz(8 * g);
            uint32_t lo, hi;
#define PDDC_PL(C, I, O, X)                                                                       \
            plane_bytes<O>(w, lo, hi);                                                            \
            *reinterpret_cast<uint2 *>(plane + (3 * C + I) * PLANE + at) = make_uint2(lo ^ X, hi ^ X);
            PDDC_PL(0, 0, 0, 0x80808080u)
            PDDC_PL(0, 1, 1, 0x80808080u)
            PDDC_PL(0, 2, 2, 0u)
            PDDC_PL(1, 0, 3, 0x80808080u)
            PDDC_PL(1, 1, 4, 0x80808080u)
            PDDC_PL(1, 2, 5, 0u)
#undef PDDC_PL
        }
    }
}
/* one loader step: the loads of tile `tn` (if any) go out group by group BETWEEN the conversions of the tile that has
 * arrived -- spread over the whole step instead of one burst */
template <int HIST>
__device__ __forceinline__ void load_and_convert(const FirI8Args &a, long long tn, bool have_next,
                                                 uint4 (&nxt)[Geo<HIST>::NQ][3], const uint4 (&cur)[Geo<HIST>::NQ][3],
                                                 uint8_t *plane, int lt)
{
    constexpr int NQ = Geo<HIST>::NQ, NG = Geo<HIST>::NG, PLANE = Geo<HIST>::PLANE;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int g = lt + NLT * q;
        if (g < NG) {
            if (have_next) {
                const long long b = tn * TILE + 8LL * g - HIST;
                const uint4 *p = b < 0 ? reinterpret_cast<const uint4 *>(static_cast<const uint8_t *>(a.hist) + (b + HIST) * 6)
                                       : reinterpret_cast<const uint4 *>(static_cast<const uint8_t *>(a.in) + b * 6);
                if (b + 8 <= a.n_in) {
                    nxt[q][0] = p[0];
                    nxt[q][1] = p[1];
                    nxt[q][2] = p[2];
                } else {
                    nxt[q][0] = nxt[q][1] = nxt[q][2] = make_uint4(0u, 0u, 0u, 0u);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t w[12] = { cur[q][0].x, cur[q][0].y, cur[q][0].z, cur[q][0].w, cur[q][1].x, cur[q][1].y,
                                     cur[q][1].z, cur[q][1].w, cur[q][2].x, cur[q][2].y, cur[q][2].z, cur[q][2].w };
            const int at = swz(8 * g);
            uint32_t lo, hi;
#define PDDC_PL(C, I, O, X)                                                                       \
            plane_bytes<O>(w, lo, hi);                                                            \
            *reinterpret_cast<uint2 *>(plane + (3 * C + I) * PLANE + at) = make_uint2(lo ^ X, hi ^ X);
            PDDC_PL(0, 0, 0, 0x80808080u)
            PDDC_PL(0, 1, 1, 0x80808080u)
            PDDC_PL(0, 2, 2, 0u)
            PDDC_PL(1, 0, 3, 0x80808080u)
            PDDC_PL(1, 1, 4, 0x80808080u)
            PDDC_PL(1, 2, 5, 0u)
#undef PDDC_PL
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
} // namespace i8

template <int HIST>
__global__ __launch_bounds__(64 * i8::NMW + i8::NLT, 1) void k_fir_i8(FirI8Args a, long long ntiles)
{
    using namespace i8;
    constexpr int PLANE = Geo<HIST>::PLANE, KSTEPS = Geo<HIST>::KSTEPS, NQ = Geo<HIST>::NQ;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_i8[];
    /* [2 buffers][2 components][3 planes][PLANE], then [2 buffers][2 components][OS] outputs */
    float *osum_base = reinterpret_cast<float *>(lds_i8 + 12 * PLANE);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long G = gridDim.x;
    long long t = blockIdx.x;
    if (wave >= NMW) {
        /* ---- loaders (waves 8..11) */
        const int lt = tid - 64 * NMW;
        if (blockIdx.x == 0 && a.hist_out) {             /* the next call's history: the batch's last 256 samples */
            const uint4 *src = reinterpret_cast<const uint4 *>(static_cast<const uint8_t *>(a.in) + (a.n_in - HIST) * 6);
            if (lt < HIST * 6 / 16)
                static_cast<uint4 *>(a.hist_out)[lt] = src[lt];
        }
        uint4 ra[NQ][3], rb[NQ][3];
        issue_tile<HIST>(a, t, ra, lt);
        planes_from<HIST>(ra, lds_i8, lt);
        if (t + G < ntiles)
            issue_tile<HIST>(a, t + G, ra, lt);
        __syncthreads();
        for (;;) {
            /* tile t is computed from buffer 0; t + G (in ra) goes to buffer 1, t + 2 G starts towards rb */
            if (t + G < ntiles)
                load_and_convert<HIST>(a, t + 2 * G, t + 2 * G < ntiles, rb, ra, lds_i8 + 6 * PLANE, lt);
            __syncthreads();
            t += G;
            if (t >= ntiles)
                break;
            if (t + G < ntiles)
                load_and_convert<HIST>(a, t + 2 * G, t + 2 * G < ntiles, ra, rb, lds_i8, lt);
            __syncthreads();
            t += G;
            if (t >= ntiles)
                break;
        }
        return;
    }
    /* ---- MFMA waves (0..7): component x block of 16 columns; the whole tap operand stays in registers */
    const int comp = wave & 1, nb0 = (wave >> 1) * NB;
    const int n = lane & 15, kq = lane >> 4;
    const v4i_t *atab = static_cast<const v4i_t *>(a.atab);
    v4i_t A[KSTEPS][4];
    if (a.taps16 == nullptr) {                             /* uniform */
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                A[ks][j] = atab[(j * KSTEPS + ks) * 64 + lane];
    } else {
        /* binary16 tap storage: this lane's 16 columns of every k-step, quantised here exactly as fir_i8_build_table
         * does on the host (H = llround(h 2^E), four balanced base-256 digits).  The device array is laid out for this
         * read: G[128 + tt] = h[HIST - tt] for tt = 1 .. HIST, zeros around it (kFirI8Taps16Len entries), so that the
         * 16 values of a k-step -- T[r][c] = G[128 + c - 8 r] -- are two aligned 16-byte loads */
        const uint4 *g16 = static_cast<const uint4 *>(a.taps16);
        const float two_e = (float)a.two_e;                 /* a power of two: h 2^E is exact in binary32 */
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
                A[ks][j] = v4i_t{ w[j][0], w[j][1], w[j][2], w[j][3] };
        }
    }
    const long long n_out = a.n_in >> 3;
    __syncthreads();
    int buf = 0;
    for (; t < ntiles; t += G, buf ^= 1) {
        const uint8_t *pb = lds_i8 + buf * 6 * PLANE + 3 * comp * PLANE;
        float *osum = osum_base + buf * 2 * OS;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int col = 16 * (nb0 + b) + n;
            v4i_t acc[4];
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc[s] = v4i_t{ 0, 0, 0, 0 };
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const int at = swz(128 * col + 64 * ks + 16 * kq);
                v4i_t B[3];
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    B[i] = *reinterpret_cast<const v4i_t *>(pb + i * PLANE + at);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (i + j >= 2)
                            acc[i + j - 2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[ks][j], B[i], acc[i + j - 2], 0, 0, 0);
            }
            /* y = sum_s acc[s] 256^(s+2), as floats (every acc[s] is below 2^24: the conversions are exact).  This lane:
             * column `col`, rows 4 kq + v -> outputs 16 col + 4 kq + v of the tile, four consecutive ones */
            float4 y;
            float *yp = &y.x;
#pragma unroll
            for (int v = 0; v < 4; ++v)
                yp[v] = (((float)acc[0][v] * 65536.0f + (float)acc[1][v] * 16777216.0f) +
                         ((float)acc[2][v] * 4294967296.0f + (float)acc[3][v] * 1099511627776.0f)) * a.scale + a.cterm;
            *reinterpret_cast<float4 *>(osum + comp * OS + 20 * col + 4 * kq) = y;
        }
        __syncthreads();                 /* ONE barrier per tile: the next tile's planes are written, this tile's outputs are in LDS */
        float2 *dst = reinterpret_cast<float2 *>(a.out) + t * 1024;
        const long long left = n_out - t * 1024;
        for (int o = tid; o < 1024; o += 64 * NMW) {
            const int q = 20 * (o >> 4) + (o & 15);
            if (o < left)
                dst[o] = make_float2(osum[q], osum[OS + q]);
        }
    }
}

bool fir_i8_build_table(const float *taps, int ntaps, int hist, int8_t *table, float *scale, float *cterm, int *exp2)
{
    if (!taps || ntaps < 1 || (hist != 128 && hist != 256) || ntaps > hist || !table)
        return false;
    double hmax = 0.0;
    for (int k = 0; k < ntaps; ++k)
        hmax = std::fmax(hmax, std::fabs((double)taps[k]));
    if (!(hmax > 0.0) || !std::isfinite(hmax))
        return false;
    const int E = 30 - (int)std::ceil(std::log2(hmax));            /* |H| <= 2^30: the top digit stays within +-64 */
    if (exp2)
        *exp2 = E;
    int8_t dig[4][256];
    long long hsum = 0;
    for (int k = 0; k < hist; ++k) {
        long long r = k < ntaps ? std::llround(std::ldexp((double)taps[k], E)) : 0;
        hsum += r;
        for (int j = 0; j < 4; ++j) {
            const long long d = j == 3 ? r : ((r + 128) & 255) - 128;
            if (d < -128 || d > 127)
                return false;
            dig[j][k] = (int8_t)d;
            r = (r - d) / 256;
        }
    }
    /* lane l of k-step ks holds A[row l & 15][k = 16 (l >> 4) + jj]: T[r][c] = h[hist - (c - 8 r)] */
    const int ksteps = (120 + hist + 63) / 64;
    for (int j = 0; j < 4; ++j)
        for (int ks = 0; ks < ksteps; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int jj = 0; jj < 16; ++jj) {
                    const int r = l & 15, c = 64 * ks + 16 * (l >> 4) + jj, tt = c - 8 * r;
                    table[(((size_t)j * ksteps + ks) * 64 + l) * 16 + jj] = (tt >= 1 && tt <= hist) ? dig[j][hist - tt] : 0;
                }
    /* sample = (v24 << 8) / (INT_MAX - 256): the reference's float (perseustest.c:466-502) */
    const double unit = std::ldexp(1.0, -E) * 256.0 / 2147483391.0;
    *scale = (float)unit;
    *cterm = (float)((double)hsum * 32896.0 * unit);               /* planes 0 and 1 are stored minus 128: 128 + 128*256 */
    return true;
}

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

template <int HIST>
static hipError_t launch_fir_i8_t(const FirI8Args &a, hipStream_t s)
{
    const long long ntiles = (a.n_in + i8::TILE - 1) / i8::TILE;
    int dev = 0;
    (void)hipGetDevice(&dev);
    static int cus[64] = { 0 };
    if (cus[dev & 63] == 0) {
        int v = 0;
        hipError_t e = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess)
            return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fir_i8<HIST>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)i8::Geo<HIST>::LDS_BYTES);
        if (e != hipSuccess)
            return e;
        cus[dev & 63] = v > 0 ? v : 256;
    }
    const long long grid = ntiles < cus[dev & 63] ? ntiles : cus[dev & 63];
    hipLaunchKernelGGL(k_fir_i8<HIST>, dim3((unsigned)grid), dim3(64 * i8::NMW + i8::NLT), i8::Geo<HIST>::LDS_BYTES, s, a, ntiles);
    return hipGetLastError();
}

hipError_t launch_fir_i8(const FirI8Args &a, int hist, hipStream_t s)
{
    if (a.n_in <= 0)
        return hipSuccess;
    if ((a.n_in & 7) || !a.in || !a.hist || !a.out || (!a.atab && !a.taps16) || (a.hist_out && a.n_in < hist))
        return hipErrorInvalidValue;
    if (hist == 256)
        return launch_fir_i8_t<256>(a, s);
    if (hist == 128)
        return launch_fir_i8_t<128>(a, s);
    return hipErrorInvalidValue;
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
 *   1  NCO, 65..256 taps waves = tap set (c / s) x half of the columns, BOTH components each: four partial products
 *                        P[c|s][I|Q] meet in LDS, uI = P[c][I] - P[s][Q], uQ = P[s][I] + P[c][Q] at the stores
 *                        (two tap sets in one wave would need 2 x 96 registers)
 *   2  NCO, <= 64 taps   waves = component x half, each holds the two tap sets it needs (c and -s, or s and c: 96
 *                        registers at 3 k-steps) and adds both band products into the SAME int32 accumulators: u leaves
 *                        the wave complete, one rounding
 * Walk: block b owns the contiguous tiles [b T / G, (b + 1) T / G); the HIST samples in front of a tile are the last
 * ones of the tile before it, so the loaders copy them from the previous plane set inside LDS (8..32 lanes, 6 x 8 bytes
 * each) instead of re-reading them from HBM; only a block's first tile loads them.
 * FUSE2 (modes 0 and 2): the tile's 1024 first-stage values u stay in LDS behind a 64-entry porch that holds the last 64 of
 * the tile before (copied there by two otherwise idle waves while the other two filter); thread p of waves 0/1 computes
 * second-stage output p of the tile, z = LO(n0 + 64 P) sum_k (h2[k] e^{+j 8 theta k}) u[8 P - k], from float4 reads of
 * both rails (packed FMAs, taps through the scalar cache) and stores it; still ONE barrier per tile.  A block whose
 * range does not start the batch first runs the tile in front of it silently (only the columns the porch needs).    */
namespace i8x {
using i8::NLT;
using i8::NMW;
using i8::plane_bytes;
using i8::swz;
static_assert(NMW == 4 && NLT == 512, "k_fir_i8x is written for 4 matrix waves + 8 loader waves");
constexpr int TILE = 8192;
constexpr int kTaps2Len = 68;                  /* floats per rail of the second stage's tap table (65 used) */

template <int HIST, int MODE, bool FUSE2>
struct Geo {
    static constexpr int SPAN = TILE + HIST, PLANE = SPAN + 16 * ((SPAN + 127) / 128);
    static constexpr int KSTEPS = (120 + HIST + 63) / 64;
    static constexpr int HG = HIST / 8;                              /* history groups of 8 samples */
    static constexpr int NARR = MODE == 1 ? 4 : 2;
    static constexpr int PORCH = FUSE2 ? 64 : 0;
    static constexpr int AS = 20 * ((1024 + PORCH) / 16);            /* floats per output array (16 values per 20) */
    static constexpr size_t LDS_BYTES = 12 * (size_t)PLANE + 2 * (size_t)NARR * AS * sizeof(float);
    static constexpr int NTAB = MODE == 0 ? 1 : MODE == 1 ? 2 : 3;
    static constexpr int TABV = 4 * KSTEPS * 64;                     /* v4i entries per tap table */
    static_assert(MODE != 2 || HIST <= 64, "two tap sets per wave only fit at 3 k-steps");
    static_assert(!(FUSE2 && MODE == 1), "the fused second stage reads complete u values");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
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

/* the loads of group g = lt + 512 q (q = 0, 1) of tile t: batch samples 8192 t + 8 g .. + 8, zeros behind the batch */
__device__ __forceinline__ void issue_group(const FirI8xArgs &a, long long t, int g, uint4 (&r)[3])
{
    const long long b = t * TILE + 8LL * g;
    if (b + 8 <= a.n_in) {
        const uint4 *p = reinterpret_cast<const uint4 *>(static_cast<const uint8_t *>(a.in) + b * 6);
        r[0] = p[0];
        r[1] = p[1];
        r[2] = p[2];
    } else {
        r[0] = r[1] = r[2] = make_uint4(0u, 0u, 0u, 0u);
    }
}

/* one loader step: the loads of tile `tn` (if any) go out group by group BETWEEN the conversions of the tile that has
 * arrived (k_fir_i8's pacing); then the history groups come over from the plane set of the tile before */
template <int HIST, int PLANE>
__device__ __forceinline__ void load_and_convert(const FirI8xArgs &a, long long tn, bool have_next, uint4 (&nxt)[2][3],
                                                 const uint4 (&cur)[2][3], uint8_t *plane, const uint8_t *prev, int lt)
{
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int g = lt + NLT * q;
        if (have_next)
            issue_group(a, tn, g, nxt[q]);
        __builtin_amdgcn_sched_barrier(0);
        put_planes(cur[q], plane, PLANE, swz(HIST + 8 * g));
        __builtin_amdgcn_sched_barrier(0);
    }
    if (lt < HIST / 8) {
        const int src = swz(TILE + 8 * lt), dst = swz(8 * lt);
#pragma unroll
        for (int pl = 0; pl < 6; ++pl)
            *reinterpret_cast<uint2 *>(plane + pl * PLANE + dst) = *reinterpret_cast<const uint2 *>(prev + pl * PLANE + src);
    }
}

/* y = sum_s acc[s] 256^(s+2) as floats */
__device__ __forceinline__ float recombine(const v4i_t (&acc)[4], int v)
{
    return ((float)acc[0][v] * 65536.0f + (float)acc[1][v] * 16777216.0f) +
           ((float)acc[2][v] * 4294967296.0f + (float)acc[3][v] * 1099511627776.0f);
}
} // namespace i8x

template <int HIST, int MODE, bool FUSE2>
__global__ __launch_bounds__(64 * i8::NMW + i8::NLT, 1) void k_fir_i8x(FirI8xArgs a, long long ntiles)
{
    using namespace i8x;
    using G = Geo<HIST, MODE, FUSE2>;
    constexpr int PLANE = G::PLANE, KSTEPS = G::KSTEPS, AS = G::AS, NARR = G::NARR, PORCH = G::PORCH;
    constexpr bool MIX = MODE != 0;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_i8x[];
    /* [2 buffers][2 components][3 planes][PLANE], then [2 buffers][NARR][AS] first-stage values */
    float *arr_base = reinterpret_cast<float *>(lds_i8x + 12 * PLANE);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long nblk = gridDim.x, blk = blockIdx.x;
    const long long t0 = blk * ntiles / nblk, t1 = (blk + 1) * ntiles / nblk;
    const long long ts = (FUSE2 && t0 > 0) ? t0 - 1 : t0;          /* the tile in front primes the second stage's history */
    if (wave >= NMW) {
        /* ---- loaders (waves 4..11) */
        const int lt = tid - 64 * NMW;
        if (blk == 0 && a.hist_out) {                    /* the next call's history: the batch's last HIST samples */
            const uint4 *src = reinterpret_cast<const uint4 *>(static_cast<const uint8_t *>(a.in) + (a.n_in - HIST) * 6);
            if (lt < HIST * 6 / 16)
                static_cast<uint4 *>(a.hist_out)[lt] = src[lt];
        }
        uint4 ra[2][3], rb[2][3];
        issue_group(a, ts, lt, ra[0]);
        issue_group(a, ts, lt + NLT, ra[1]);
        if (lt < G::HG) {
            /* this block's first tile takes the samples in front of it from memory: the stream's history or the batch */
            const long long b = ts * TILE - HIST + 8LL * lt;
            const uint4 *p = b < 0 ? reinterpret_cast<const uint4 *>(static_cast<const uint8_t *>(a.hist) + (b + HIST) * 6)
                                   : reinterpret_cast<const uint4 *>(static_cast<const uint8_t *>(a.in) + b * 6);
            const uint4 rh[3] = { p[0], p[1], p[2] };
            put_planes(rh, lds_i8x, PLANE, swz(8 * lt));
        }
        put_planes(ra[0], lds_i8x, PLANE, swz(HIST + 8 * lt));
        put_planes(ra[1], lds_i8x, PLANE, swz(HIST + 8 * (lt + NLT)));
        long long t = ts;
        if (t + 1 < t1) {
            issue_group(a, t + 1, lt, ra[0]);
            issue_group(a, t + 1, lt + NLT, ra[1]);
        }
        __syncthreads();
        for (;;) {
            /* tile t is computed from plane set 0; t + 1 (in ra) goes to set 1, t + 2 starts towards rb */
            if (t + 1 < t1)
                load_and_convert<HIST, PLANE>(a, t + 2, t + 2 < t1, rb, ra, lds_i8x + 6 * PLANE, lds_i8x, lt);
            __syncthreads();
            if (++t >= t1)
                break;
            if (t + 1 < t1)
                load_and_convert<HIST, PLANE>(a, t + 2, t + 2 < t1, ra, rb, lds_i8x, lds_i8x + 6 * PLANE, lt);
            __syncthreads();
            if (++t >= t1)
                break;
        }
        return;
    }
    /* ---- matrix waves (0..3) */
    const int w0 = wave & 1, half = wave >> 1;
    const int n = lane & 15, kq = lane >> 4;
    const v4i_t *atab = static_cast<const v4i_t *>(a.atab);
    /* tap operand(s): mode 0 the one table; mode 1 table w0 (c or s); mode 2 the set that meets the I planes and the one
     * that meets the Q planes: c and -s for component I, s and c for component Q (tables c, s, -s) */
    v4i_t A0[KSTEPS][4];
    v4i_t A1[MODE == 2 ? KSTEPS : 1][4];
    {
        const int tab0 = MODE == 0 ? 0 : w0, tab1 = w0 == 0 ? 2 : 0;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                A0[ks][j] = atab[tab0 * G::TABV + (j * KSTEPS + ks) * 64 + lane];
                if (MODE == 2)
                    A1[ks][j] = atab[tab1 * G::TABV + (j * KSTEPS + ks) * 64 + lane];
            }
    }
    const uint32_t n0lo = (uint32_t)a.n0;
    if (FUSE2 && ts == 0 && tid < 64) {
        /* the stream's second-stage history: the 64 first-stage outputs in front of this batch (mixed floats, the state
         * k_fir8's fused pair keeps too), taken back into the frame of this call's u values: u = y conj(LO) */
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
    const long long n_out = a.n_in >> 3;
    __syncthreads();
    int buf = 0;
    for (long long t = ts; t < t1; ++t, buf ^= 1) {
        const bool silent = FUSE2 && t < t0;
        const uint8_t *pb = lds_i8x + buf * 6 * PLANE;
        float *arr = arr_base + buf * NARR * AS;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            if (FUSE2 && silent && !(half == 1 && cb == 1))
                continue;                                 /* the porch needs the tile's last 64 values only */
            const int col = 16 * (2 * half + cb) + n;
            float *dst = arr + 20 * (col + PORCH / 16) + 4 * kq;
            if (MODE == 1) {
#pragma unroll
                for (int comp = 0; comp < 2; ++comp) {
                    v4i_t acc[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        acc[s] = v4i_t{ 0, 0, 0, 0 };
#pragma unroll
                    for (int ks = 0; ks < KSTEPS; ++ks) {
                        const int at = swz(128 * col + 64 * ks + 16 * kq);
                        v4i_t B[3];
#pragma unroll
                        for (int i = 0; i < 3; ++i)
                            B[i] = *reinterpret_cast<const v4i_t *>(pb + (3 * comp + i) * PLANE + at);
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (i + j >= 2)
                                    acc[i + j - 2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A0[ks][j], B[i], acc[i + j - 2], 0, 0, 0);
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
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) {
                    const int at = swz(128 * col + 64 * ks + 16 * kq);
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
                /* this lane: column `col`, rows 4 kq + v -> values 16 col + 4 kq + v of the tile, four consecutive ones */
                float4 y;
                float *yp = &y.x;
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    yp[v] = recombine(acc, v) * a.scale + a.ct[w0];
                *reinterpret_cast<float4 *>(dst + w0 * AS) = y;
            }
        }
        __syncthreads();                 /* ONE barrier per tile: the next tile's planes are written, this tile's values are in LDS */
        if (!FUSE2) {
            float2 *dst = reinterpret_cast<float2 *>(a.out) + t * 1024;
            const long long left = n_out - t * 1024;
            for (int o = tid; o < 1024; o += 64 * NMW) {
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
                    nco_lo((n0lo + 8u * (uint32_t)(t * 1024 + o)) * a.freg + a.phase_off, c, s);
                    const float yI = uI * c - uQ * s, yQ = uI * s + uQ * c;
                    uI = yI;
                    uQ = yQ;
                }
                if (o < left)
                    dst[o] = make_float2(uI, uQ);
            }
        } else if (tid < 128) {
            /* second stage: output p of the tile from u[8 p - 64 .. 8 p] (array positions 8 p .. 8 p + 64 behind the porch);
             * the tap tables are in descending order, gre[i] = Re g2[64 - i], so that a float4 of u meets a float4 of taps */
            if (!silent) {
                const int p = tid;
                const float *uIp = arr, *uQp = arr + AS;
                const f32x2 PDDC_CONSTANT *gre = (const f32x2 PDDC_CONSTANT *)a.taps2;
                const f32x2 PDDC_CONSTANT *gim = (const f32x2 PDDC_CONSTANT *)(a.taps2 + kTaps2Len);
                f32x2 rr[2] = { { 0.f, 0.f }, { 0.f, 0.f } }, ii[2] = { { 0.f, 0.f }, { 0.f, 0.f } };
                f32x2 ir[2] = { { 0.f, 0.f }, { 0.f, 0.f } }, ri[2] = { { 0.f, 0.f }, { 0.f, 0.f } };
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int op = 8 * p + 4 * j, q = 20 * (op >> 4) + (op & 15);
                    const float4 xI = *reinterpret_cast<const float4 *>(uIp + q);
                    const float4 xQ = *reinterpret_cast<const float4 *>(uQp + q);
                    const f32x2 xi[2] = { { xI.x, xI.y }, { xI.z, xI.w } }, xq[2] = { { xQ.x, xQ.y }, { xQ.z, xQ.w } };
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const f32x2 gr = gre[2 * j + e];
                        rr[e] = __builtin_elementwise_fma(gr, xi[e], rr[e]);
                        ri[e] = __builtin_elementwise_fma(gr, xq[e], ri[e]);
                        if (MIX) {
                            const f32x2 gi = gim[2 * j + e];
                            ii[e] = __builtin_elementwise_fma(gi, xq[e], ii[e]);
                            ir[e] = __builtin_elementwise_fma(gi, xi[e], ir[e]);
                        }
                    }
                }
                const int op = 8 * p + 64, q = 20 * (op >> 4) + (op & 15);
                const float g0r = a.taps2[64], g0i = a.taps2[kTaps2Len + 64];
                const float x0I = uIp[q], x0Q = uQp[q];
                float zr = ((rr[0].x + rr[0].y) + (rr[1].x + rr[1].y)) + g0r * x0I;
                float zi = ((ri[0].x + ri[0].y) + (ri[1].x + ri[1].y)) + g0r * x0Q;
                if (MIX) {
                    zr -= ((ii[0].x + ii[0].y) + (ii[1].x + ii[1].y)) + g0i * x0Q;
                    zi += ((ir[0].x + ir[0].y) + (ir[1].x + ir[1].y)) + g0i * x0I;
                    float c, s;
                    nco_lo((n0lo + 64u * (uint32_t)(t * 128 + p)) * a.freg + a.phase_off, c, s);
                    const float yr = zr * c - zi * s, yi = zr * s + zi * c;
                    zr = yr;
                    zi = yi;
                }
                reinterpret_cast<float2 *>(a.out)[t * 128 + p] = make_float2(zr, zi);
            }
        } else {
            /* waves 2, 3: the tile's last 64 values become the porch of the other set */
            const int idx = tid - 128, comp = idx >> 6, e = idx & 63;
            const int qs = 20 * (64 + (e >> 4)) + (e & 15), qd = 20 * (e >> 4) + (e & 15);
            float *other = arr_base + (buf ^ 1) * NARR * AS;
            other[comp * AS + qd] = arr[comp * AS + qs];
            if (t == ntiles - 1 && a.hist2_out && comp == 0) {
                /* ... and the next call's second-stage history: mixed floats, y = u LO */
                float uI = arr[qs], uQ = arr[AS + qs];
                if (MIX) {
                    float c, s;
                    nco_lo((n0lo + 8u * (uint32_t)(t * 1024 + 960 + e)) * a.freg + a.phase_off, c, s);
                    const float yI = uI * c - uQ * s, yQ = uI * s + uQ * c;
                    uI = yI;
                    uQ = yQ;
                }
                static_cast<float2 *>(a.hist2_out)[e] = make_float2(uI, uQ);
            }
        }
    }
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

int fir_i8x_mode(int hist, bool mix) { return !mix ? 0 : hist <= 64 ? 2 : 1; }

size_t fir_i8x_table_bytes(int hist, bool mix)
{
    const int ksteps = (120 + hist + 63) / 64, mode = fir_i8x_mode(hist, mix);
    return (size_t)(mode == 0 ? 1 : mode == 1 ? 2 : 3) * 4 * ksteps * 64 * 16;
}

bool fir_i8x_build_tables(const float *taps, int ntaps, int hist, bool mix, uint32_t freg, int8_t *tables, float *scale,
                          float ct[2], int *exp2)
{
    if (!taps || ntaps < 1 || (hist != 32 && hist != 64 && hist != 128 && hist != 256) || ntaps > hist || !tables)
        return false;
    double hmax = 0.0;
    for (int k = 0; k < ntaps; ++k)
        hmax = std::fmax(hmax, std::fabs((double)taps[k]));
    if (!(hmax > 0.0) || !std::isfinite(hmax))
        return false;
    const int E = 30 - (int)std::ceil(std::log2(hmax));            /* |H| <= 2^30 for h, a fortiori for h cos, h sin */
    if (exp2)
        *exp2 = E;
    const size_t tb = (size_t)4 * ((120 + hist + 63) / 64) * 64 * 16;
    /* sample = v24 / 8388607 (perseustest.c:466-502); planes 0 and 1 are stored minus 128: V = planes + 32896 */
    const double unit = std::ldexp(1.0, -E) / 8388607.0;
    *scale = (float)unit;
    std::vector<long long> Hc((size_t)hist, 0), Hs((size_t)hist, 0), Hm((size_t)hist, 0);
    long long sc = 0, ss = 0;
    const double w = 6.283185307179586476925286766559 / 4294967296.0;
    for (int k = 0; k < ntaps; ++k) {
        const double hk = std::ldexp((double)taps[k], E);
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
    if (!i8_fill_table(Hc.data(), hist, tables))
        return false;
    if (mix) {
        if (!i8_fill_table(Hs.data(), hist, tables + tb))
            return false;
        if (fir_i8x_mode(hist, true) == 2 && !i8_fill_table(Hm.data(), hist, tables + 2 * tb))
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

template <int HIST, int MODE, bool FUSE2>
static hipError_t launch_fir_i8x_t(const FirI8xArgs &a, int max_blocks, hipStream_t s)
{
    using G = i8x::Geo<HIST, MODE, FUSE2>;
    const long long ntiles = (a.n_in + i8x::TILE - 1) / i8x::TILE;
    int dev = 0;
    (void)hipGetDevice(&dev);
    static int cus[64] = { 0 };
    if (cus[dev & 63] == 0) {
        int v = 0;
        hipError_t e = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess)
            return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fir_i8x<HIST, MODE, FUSE2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
        if (e != hipSuccess)
            return e;
        cus[dev & 63] = v > 0 ? v : 256;
    }
    long long grid = ntiles < cus[dev & 63] ? ntiles : cus[dev & 63];
    if (max_blocks > 0 && grid > max_blocks)
        grid = max_blocks;
    hipLaunchKernelGGL((k_fir_i8x<HIST, MODE, FUSE2>), dim3((unsigned)grid), dim3(64 * i8::NMW + i8::NLT), G::LDS_BYTES, s, a,
                       ntiles);
    return hipGetLastError();
}

template <int HIST>
static hipError_t launch_fir_i8x_h(const FirI8xArgs &a, bool mix, bool fuse2, int max_blocks, hipStream_t s)
{
    constexpr int MM = HIST <= 64 ? 2 : 1;
    if (!mix)
        return fuse2 ? launch_fir_i8x_t<HIST, 0, true>(a, max_blocks, s) : launch_fir_i8x_t<HIST, 0, false>(a, max_blocks, s);
    if (!fuse2)
        return launch_fir_i8x_t<HIST, MM, false>(a, max_blocks, s);
    if constexpr (MM == 2)
        return launch_fir_i8x_t<HIST, 2, true>(a, max_blocks, s);
    else
        return hipErrorInvalidValue;
}

bool fir_i8x_supported(int hist, bool mix, bool fuse2)
{
    if (hist != 32 && hist != 64 && hist != 128 && hist != 256)
        return false;
    return !(fuse2 && mix && hist > 64);
}

hipError_t launch_fir_i8x(const FirI8xArgs &a, int hist, bool mix, bool fuse2, hipStream_t s, int max_blocks)
{
    if (a.n_in <= 0)
        return hipSuccess;
    if ((a.n_in & 7) || !a.in || !a.hist || !a.out || !a.atab || (a.hist_out && a.n_in < hist) || !fir_i8x_supported(hist, mix, fuse2))
        return hipErrorInvalidValue;
    if (fuse2 && ((a.n_in % i8x::TILE) || !a.taps2 || !a.hist2))
        return hipErrorInvalidValue;
    switch (hist) {
    case 32:
        return launch_fir_i8x_h<32>(a, mix, fuse2, max_blocks, s);
    case 64:
        return launch_fir_i8x_h<64>(a, mix, fuse2, max_blocks, s);
    case 128:
        return launch_fir_i8x_h<128>(a, mix, fuse2, max_blocks, s);
    default:
        return launch_fir_i8x_h<256>(a, mix, fuse2, max_blocks, s);
    }
}

} // namespace pddc
