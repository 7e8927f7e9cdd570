// Prototype / feasibility measurement (NOTEBOOK.md rounds 1-3 5 (v)): the long first-stage FIR (255 taps, decimate by 8, no NCO) on the
// INT8 matrix cores.  A 24-bit sample IS three int8 planes -- the wire bytes themselves (I0 I1 I2 Q0 Q1 Q2) -- and taps
// quantised to 32-bit integers are four balanced base-256 digits; v_mfma_i32_32x32x32_i8 forms the byte-plane products
// EXACTLY in int32, the planes are recombined once per output.  The only error is the tap quantisation (2^-31 of the
// largest tap) plus the three lowest-order plane products that are dropped (<= 1.2e-7 of full scale, typically 2e-9).
//
//   y[m] = sum_{t=0}^{255} g[t] * xp[8m + t]        (xp = 256 history samples + the batch; g = the taps, zero padded)
//
// VERSION 2: v_mfma_i32_16x16x64_i8 -- 16 output rows per column, so the banded Toeplitz matrix is 16 x 384 (two thirds
// full instead of half), a wave keeps the WHOLE tap operand (24 fragments, 96 VGPRs) and four 16x16 accumulators: no split
// of the k range, no partial sums to add across waves.
// One tile = 64 columns x 16 outputs per component (8192 input samples + 256 of history); a persistent block per CU:
//   load   packed bytes -> six byte planes in LDS (v_perm de-interleave; planes 0/1 xor 0x80: unsigned -> signed)
//   MFMA   D[r][n] = sum_c T[r][c] * X[c][n],  T[r][c] = g[c - 8r] (banded Toeplitz, 32 x 512), X[c][n] = xp[256 n + c]
//          wave w: component w >> 1, half (w & 1) of the 16 k-steps; 9 plane products per k-step into 4 accumulators
//          (products with the same power of 256 share one)
//   out    int32 -> float, the two halves added through LDS, I/Q interleaved, coalesced float2 stores
//
// Checks itself against a double-precision CPU reference on windows of the output, then times full-size launches.
// build: hipcc --offload-arch=gfx950 -O3 -o fir_i8_planes fir_i8_planes.hip      run: ./fir_i8_planes [log2 samples]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#ifdef ABL_NOMFMA
#define ABL_MFMA_COND && (ks == 0 && i == 2 && j == 0)
#else
#define ABL_MFMA_COND
#endif
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int TILE = 8192;                 // input samples per tile (32 columns of 256)
constexpr int SPAN = TILE + 256;           // with the history in front
constexpr int PLANE = SPAN + 16 * (SPAN / 128);      // bytes of one plane in LDS: 16 B of padding per 128 (lane stride 144 B)
constexpr int NG = SPAN / 8;               // groups of 8 samples per tile
constexpr int KSTEPS = 6;                  // 384-wide window / 64  (16 rows: 8 * 15 + 256 = 376)
constexpr int OS = 20 * 64;                // outputs of one component: 64 columns x 16 rows, column stride 20 floats
#ifndef NMW
#define NMW 8                              // MFMA waves per block: 8 (component x column block) or 4 (two column blocks each)
#endif

__device__ __forceinline__ int swz(int p) { return p + 16 * (p >> 7); }

// the 8 bytes at offsets 6s + O (s = 0..7) of the 48 bytes w[0..11]
template <int O>
__device__ __forceinline__ void plane_bytes(const uint32_t (&w)[12], uint32_t &lo, uint32_t &hi)
{
    uint32_t out[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        constexpr int dummy = 0;
        (void)dummy;
        uint32_t pair[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int b0 = 6 * (4 * half + 2 * q) + O, b1 = b0 + 6;          // two bytes, 6 apart
            const int d0 = b0 >> 2, d1 = b1 >> 2;
            // select byte (b0 & 3) of w[d0] into byte 0 and byte (b1 & 3) of w[d1] into byte 1
            const uint32_t sel = (uint32_t)(b0 & 3) | ((uint32_t)(4 + (b1 & 3)) << 8) | 0x0c0c0000u;
            pair[q] = __builtin_amdgcn_perm(w[d1], w[d0], sel);
        }
        out[half] = __builtin_amdgcn_perm(pair[1], pair[0], 0x05040100u);
    }
    lo = out[0];
    hi = out[1];
}

constexpr int NQ = (NG + 255) / 256;       // groups per loader thread

__device__ __forceinline__ void issue_tile(const uint8_t *__restrict__ src, uint4 (&raw)[NQ][3], int lt)
{
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int g = lt + 256 * q;
        if (g < NG) {
            const uint4 *p = reinterpret_cast<const uint4 *>(src + (size_t)g * 48);
            raw[q][0] = p[0];
            raw[q][1] = p[1];
            raw[q][2] = p[2];
        }
    }
}

__device__ __forceinline__ void planes_from(const uint4 (&raw)[NQ][3], uint8_t *plane, int lt)
{
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int g = lt + 256 * q;
        if (g < NG) {
            const uint32_t w[12] = { raw[q][0].x, raw[q][0].y, raw[q][0].z, raw[q][0].w, raw[q][1].x, raw[q][1].y,
                                     raw[q][1].z, raw[q][1].w, raw[q][2].x, raw[q][2].y, raw[q][2].z, raw[q][2].w };
            const int at = swz(8 * g);
#ifndef PERM36
            // 24 v_perm_b32 per 8 samples: three per PAIR of samples put two bytes of two planes into one dword each --
            // (I0,I1), (I2,Q0), (Q1,Q2) --, two per plane gather four samples' bytes from two of those
            uint32_t a[4], b[4], c[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a[t] = __builtin_amdgcn_perm(w[3 * t + 1], w[3 * t], 0x07010600u);
                b[t] = __builtin_amdgcn_perm(w[3 * t + 2], w[3 * t], 0x05030402u);
                c[t] = __builtin_amdgcn_perm(w[3 * t + 2], w[3 * t + 1], 0x07010600u);
            }
#pragma unroll
            for (int q6 = 0; q6 < 6; ++q6) {
                const uint32_t *src = q6 < 2 ? a : q6 < 4 ? b : c;
                const uint32_t sel = (q6 & 1) ? 0x07060302u : 0x05040100u, x = (q6 % 3) == 2 ? 0u : 0x80808080u;
                *reinterpret_cast<uint2 *>(plane + q6 * PLANE + at) =
                    make_uint2(__builtin_amdgcn_perm(src[1], src[0], sel) ^ x, __builtin_amdgcn_perm(src[3], src[2], sel) ^ x);
            }
            continue;
#endif
            uint32_t lo, hi;
#define PL(C, I, O, X)                                                                            \
            plane_bytes<O>(w, lo, hi);                                                            \
            *reinterpret_cast<uint2 *>(plane + (3 * C + I) * PLANE + at) = make_uint2(lo ^ X, hi ^ X);
            PL(0, 0, 0, 0x80808080u)
            PL(0, 1, 1, 0x80808080u)
            PL(0, 2, 2, 0u)
            PL(1, 0, 3, 0x80808080u)
            PL(1, 1, 4, 0x80808080u)
            PL(1, 2, 5, 0u)
#undef PL
        }
    }
}

// Atab: [4 tap planes][6 k-steps][64 lanes][16 bytes]; lane l holds A[row l & 15][k = 16 (l >> 4) + jj]
__global__ __launch_bounds__(256 + 64 * NMW, 1) void k_fir_i8(const uint8_t *__restrict__ in, const v4i *__restrict__ atab,
                                                              float2 *__restrict__ out, long long ntiles, float scale, float cterm)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    float *osum_base = reinterpret_cast<float *>(lds + 12 * PLANE);      // [2 buffers][2 comps][OS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long G = gridDim.x;
    long long t = blockIdx.x;
    if (t >= ntiles)
        return;
    if (wave >= NMW) {
        const int lt = tid - 64 * NMW;
        uint4 ra[NQ][3], rb[NQ][3];
        issue_tile(in + (size_t)t * TILE * 6, ra, lt);
        planes_from(ra, lds, lt);
        if (t + G < ntiles)
            issue_tile(in + (size_t)(t + G) * TILE * 6, ra, lt);
        __syncthreads();
        for (;;) {
            if (t + G < ntiles) {
                if (t + 2 * G < ntiles)
                    issue_tile(in + (size_t)(t + 2 * G) * TILE * 6, rb, lt);
#ifndef ABL_NOPLANES
                planes_from(ra, lds + 6 * PLANE, lt);
#endif
            }
            __syncthreads();
            t += G;
            if (t >= ntiles)
                break;
            if (t + G < ntiles) {
                if (t + 2 * G < ntiles)
                    issue_tile(in + (size_t)(t + 2 * G) * TILE * 6, ra, lt);
#ifndef ABL_NOPLANES
                planes_from(rb, lds, lt);
#endif
            }
            __syncthreads();
            t += G;
            if (t >= ntiles)
                break;
        }
        return;
    }
    // ---- MFMA waves: component and column block(s); the whole tap operand in registers
    const int comp = wave & 1;
    constexpr int NB = 8 / NMW;                      // column blocks (16 columns) per wave
    const int nb0 = (wave >> 1) * NB;
    const int n = lane & 15, kq = lane >> 4;
    v4i A[KSTEPS][4];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            A[ks][j] = atab[(j * KSTEPS + ks) * 64 + lane];
    __syncthreads();
    int buf = 0;
    for (; t < ntiles; t += G, buf ^= 1) {
        const uint8_t *pb = lds + buf * 6 * PLANE + 3 * comp * PLANE;
        float *osum = osum_base + buf * 2 * OS;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int col = 16 * (nb0 + b) + n;
            v4i acc[4];
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc[s] = v4i{ 0, 0, 0, 0 };
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const int at = swz(128 * col + 64 * ks + 16 * kq);
                v4i B[3];
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    B[i] = *reinterpret_cast<const v4i *>(pb + i * PLANE + at);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (i + j >= 2 ABL_MFMA_COND)
                            acc[i + j - 2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[ks][j], B[i], acc[i + j - 2], 0, 0, 0);
            }
            // this lane: column `col`, rows 4 kq + v  ->  output 16 col + 4 kq + v of the tile: four consecutive outputs
            float4 y;
            float *yp = &y.x;
#pragma unroll
            for (int v = 0; v < 4; ++v)
                yp[v] = (((float)acc[0][v] * 65536.0f + (float)acc[1][v] * 16777216.0f) +
                         ((float)acc[2][v] * 4294967296.0f + (float)acc[3][v] * 1099511627776.0f)) * scale + cterm;
            *reinterpret_cast<float4 *>(osum + comp * OS + 20 * col + 4 * kq) = y;
        }
        __syncthreads();                 // ONE barrier per tile: the next tile's planes are written, this tile's outputs are in LDS
        float2 *dst = out + (size_t)t * 1024;
        for (int o = tid; o < 1024; o += 64 * NMW) {
            const int q = 20 * (o >> 4) + (o & 15);
            dst[o] = make_float2(osum[q], osum[OS + q]);
        }
    }
}

int main(int argc, char **argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 28;
    const size_t ns = (size_t)1 << log2n;
    const long long ntiles = (long long)(ns / TILE);
    // taps: a 255-tap low-pass (Hamming-windowed sinc), g[t] for window position t (zero at t = 255)
    double hd[256];
    {
        double sum = 0;
        for (int k = 0; k < 255; ++k) {
            const double u = k - 127.0, x = 2 * 0.045 * u;
            const double sinc = u == 0 ? 1.0 : sin(M_PI * x) / (M_PI * x);
            hd[k] = sinc * (0.54 - 0.46 * cos(2 * M_PI * k / 254.0));
            sum += hd[k];
        }
        for (int k = 0; k < 255; ++k)
            hd[k] /= sum;
        hd[255] = 0;
    }
    std::vector<float> hf(256);
    for (int k = 0; k < 256; ++k)
        hf[k] = (float)hd[k];                              // what the fp32 kernel would use
    // integer taps: H = round(h * 2^E), |H| < 2^31 - 2^23 (room for the balanced digits), four digits in [-128, 127]
    double hmax = 0;
    for (int k = 0; k < 256; ++k)
        hmax = fmax(hmax, fabs((double)hf[k]));
    int E = 30 - (int)ceil(log2(hmax));
    std::vector<long long> H(256);
    int8_t dig[4][256];
    long long hsum = 0;
    for (int k = 0; k < 256; ++k) {
        H[k] = llround(ldexp((double)hf[k], E));
        hsum += H[k];
        long long r = H[k];
        for (int j = 0; j < 4; ++j) {
            long long d = ((r + 128) & 255) - 128;
            if (j == 3)
                d = r;
            if (d < -128 || d > 127) {
                printf("digit overflow at tap %d\n", k);
                return 1;
            }
            dig[j][k] = (int8_t)d;
            r = (r - d) / 256;
        }
    }
    // A operand table: lane l holds A[row l & 31][k = 16 (l >> 5) + jj], jj = 0..15
    std::vector<int8_t> atab((size_t)4 * KSTEPS * 64 * 16);      // lane l: A[row l & 15][k = 16 (l >> 4) + jj]
    for (int j = 0; j < 4; ++j)
        for (int ks = 0; ks < KSTEPS; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int jj = 0; jj < 16; ++jj) {
                    const int r = l & 15, c = 64 * ks + 16 * (l >> 4) + jj, tt = c - 8 * r;
                    atab[(((size_t)j * KSTEPS + ks) * 64 + l) * 16 + jj] = (tt >= 0 && tt < 256) ? dig[j][tt] : 0;
                }
    // input: LCG bytes, 256 samples of history in front
    const size_t nbytes = (ns + 256) * 6;
    std::vector<uint8_t> hin((size_t)1 << 22);              // only the first 4 MiB are checked on the host; the rest repeats
    uint32_t st = 12345;
    for (auto &b : hin) {
        st = st * 1664525u + 1013904223u;
        b = (uint8_t)(st >> 24);
    }
    uint8_t *d_in;
    v4i *d_atab;
    float2 *d_out;
    CHECK(hipMalloc(&d_in, nbytes + 64));
    CHECK(hipMalloc(&d_atab, atab.size()));
    CHECK(hipMalloc(&d_out, (size_t)ntiles * 1024 * sizeof(float2)));
    for (size_t off = 0; off < nbytes; off += hin.size())
        CHECK(hipMemcpy(d_in + off, hin.data(), std::min(hin.size(), nbytes - off), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_atab, atab.data(), atab.size(), hipMemcpyHostToDevice));
    const float scale = (float)(ldexp(1.0, -E) * 256.0 / 2147483391.0);
    const float cterm = (float)((double)hsum * 32896.0 * ldexp(1.0, -E) * 256.0 / 2147483391.0);
    const size_t ldsb = 12 * PLANE + 2 * 2 * OS * sizeof(float);
    const unsigned grid = (unsigned)std::min<long long>(ntiles, argc > 2 ? atoi(argv[2]) : 256);        // persistent: one block per CU
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fir_i8), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipLaunchKernelGGL(k_fir_i8, dim3(grid), dim3(256 + 64 * NMW), ldsb, 0, d_in, d_atab, d_out, ntiles, scale, cterm);
    CHECK(hipDeviceSynchronize());
    // ---- check: outputs of the first tiles against a double reference on the float taps
    const int ncheck = 3 * 1024 + 77;
    std::vector<float2> hout(ncheck);
    CHECK(hipMemcpy(hout.data(), d_out, ncheck * sizeof(float2), hipMemcpyDeviceToHost));
    auto sample = [&](size_t p, int c) {                    // sample p of xp, component c, as the reference's float
        const uint8_t *b = &hin[(p * 6 + 3 * c) % hin.size()];
        const int32_t v = (int32_t)((uint32_t)b[0] << 8 | (uint32_t)b[1] << 16 | (uint32_t)b[2] << 24);      // (v24 << 8)
        return (double)v / 2147483391.0;
    };
    double worst = 0, ref_max = 0;
    for (int m = 0; m < ncheck; ++m)
        for (int c = 0; c < 2; ++c) {
            double acc = 0;
            for (int tt = 0; tt < 256; ++tt)
                acc += (double)hf[tt] * sample((size_t)8 * m + tt, c);
            const double got = c ? hout[m].y : hout[m].x;
            worst = fmax(worst, fabs(got - acc));
            ref_max = fmax(ref_max, fabs(acc));
        }
    printf("check: %d outputs x 2, max|y - ref| / max|ref| = %.3e (tolerance of the product: 1e-6)   tap scale 2^%d\n", ncheck,
           worst / ref_max, E);
    // ---- time
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i)
        hipLaunchKernelGGL(k_fir_i8, dim3(grid), dim3(256 + 64 * NMW), ldsb, 0, d_in, d_atab, d_out, ntiles, scale, cterm);
    CHECK(hipEventRecord(e0, 0));
    const int reps = 50;
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL(k_fir_i8, dim3(grid), dim3(256 + 64 * NMW), ldsb, 0, d_in, d_atab, d_out, ntiles, scale, cterm);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("2^%d samples, 255 taps / 8 on the int8 matrix cores (16x16x64): %.4f ms per launch = %.1f GS/s = %.1f %% of 8 TB/s at 7 B/sample"
           "   (k_fir8, fp32 vector FMAs: 0.465 ms = 50.6 %%)\n", log2n, ms, ns / (ms * 1e-3) / 1e9,
           100.0 * 7.0 * ns / (ms * 1e-3) / 8e12);
    return 0;
}
