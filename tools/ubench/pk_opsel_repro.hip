// Stand-alone attempt at the packed-fp32 fault of NOTEBOOK R6.4, shaped like ONE block of the product's layout 1:
//   a VOP3P packed-fp32 instruction whose SECOND source takes its low half from the HIGH register (op_sel:[0,1]) delivers a
//   low-half result of 0.0 in lanes 48..63 now and then -- in a wave that shares its SIMD with a wave issuing matrix
//   instructions.  In the product one busy CU is enough (tools/i8x_debug.py with I8X_BLOCKS=1).
// The block (768 threads, one barrier per "tile", like k_fir_i8x):
//   waves 0..3   matrix waves: per tile two column blocks x three k-steps of six ds_read_b128 operand reads and 18
//                v_mfma_i32_16x16x64_i8 (two tap tables in 96 registers, two dependent products into each accumulator), then
//                the recombination (v_cvt_f32_i32, scale, add) and a ds_write_b128 of the values -- the product's band pass
//   waves 4..11  loader waves: per tile two groups of three global_load_dwordx4, the byte de-interleave (v_perm_b32) and
//                ds_write_b64 into the operand planes; then, for the PREVIOUS tile's values, what the finishing code does:
//                ds_read2st64_b32 of (u, v), the rotation's multiply  v_pk_mul_f32 P, (u, v), (s, c) op_sel:[0,1] op_sel_hi:[0,0]
//                (P.lo = u*c, P.hi = u*s) and -- the check -- the same two products by v_mul_f32; a low half that differs is counted
//                by SIMD and quarter of the wave, and whether it is exactly 0.0
// usage: ./pk_opsel_repro [blocks] [tiles] [matrix 0|1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int PLANE = 9360;                    // bytes per operand plane (the product's: 8192 + 128 samples, padded rows)
constexpr int AS = 1280;                       // floats per value array

__global__ __launch_bounds__(768, 1) void k(const uint4 *__restrict__ in, unsigned long long *bad, int tiles, int matrix, int *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float *arr = reinterpret_cast<float *>(lds + 12 * PLANE);            // [2 buffers][2 rails][AS]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < (12 * PLANE + 4 * AS * 4) / 4; i += 768)
        reinterpret_cast<unsigned *>(lds)[i] = 0x3f800000u + 977u * (unsigned)i;      // some floats / bytes
    __syncthreads();
    if (wave < 4) {
        // ---- matrix waves
        v4i A0[3][4], A1[3][4];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                A0[ks][j] = *reinterpret_cast<const v4i *>(lds + 16 * ((lane + 64 * (4 * ks + j)) % 500));
                A1[ks][j] = *reinterpret_cast<const v4i *>(lds + 16 * ((lane + 64 * (4 * ks + j) + 77) % 500));
            }
        const int n = lane & 15, kq = lane >> 4;
        int acc_sink = 0;
        for (int t = 0; t < tiles; ++t) {
            const unsigned char *pb = lds + (t & 1) * 6 * PLANE;
            float *dst = arr + (t & 1) * 2 * AS;
            if (matrix) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    v4i acc[4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
                    const int pos = 128 * (16 * (2 * (wave & 1) + cb) + n) % 8000 + 16 * kq;
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks) {
                        v4i BI[3], BQ[3];
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            BI[i] = *reinterpret_cast<const v4i *>(pb + i * PLANE + ((pos + 64 * ks) & ~15));
                            BQ[i] = *reinterpret_cast<const v4i *>(pb + (3 + i) * PLANE + ((pos + 64 * ks) & ~15));
                        }
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (i + j >= 2) {
                                    acc[i + j - 2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A0[ks][j], BI[i], acc[i + j - 2], 0, 0, 0);
                                    acc[i + j - 2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A1[ks][j], BQ[i], acc[i + j - 2], 0, 0, 0);
                                }
                    }
                    f4 y;
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        y[v] = (((float)acc[0][v] * 65536.0f + (float)acc[1][v] * 16777216.0f) +
                                ((float)acc[2][v] * 4294967296.0f + (float)acc[3][v] * 1099511627776.0f)) * 1e-12f + 0.25f;
                    const int col = 16 * (2 * (wave >> 1) + cb) + n, pp = 8 * col + 4 * (kq & 1);
                    *reinterpret_cast<f4 *>(dst + (kq >> 1) * AS + (20 * (pp >> 4) + (pp & 15)) % (AS - 4)) = y;
                    acc_sink += acc[0][0];
                }
            }
            __syncthreads();
        }
        if (acc_sink == 0x7fffffff)
            *sink = 1;
        return;
    }
    // ---- loader waves: the next tile's bytes into the planes, then the finishing code's multiply on the previous tile's values
    const int lt = tid - 256;
    unsigned long long wrong_lo = 0, wrong_hi = 0, zero_lo = 0;
    for (int t = 0; t < tiles; ++t) {
        unsigned char *dstp = lds + ((t + 1) & 1) * 6 * PLANE;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int g = lt + 512 * q;
            const uint4 *p = in + ((size_t)((t * 1024 + g) & 0xfffff)) * 3;
            const uint4 r0 = p[0], r1 = p[1], r2 = p[2];
            const unsigned w[12] = { r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w };
#pragma unroll
            for (int pl = 0; pl < 6; ++pl) {
                unsigned lo = __builtin_amdgcn_perm(w[(pl + 3) % 12], w[pl], 0x0c0c0400u | (pl << 16));
                unsigned hi = __builtin_amdgcn_perm(w[(pl + 9) % 12], w[pl + 6], 0x05040100u);
                lo = __builtin_amdgcn_perm(hi, lo, 0x07030602u);
                *reinterpret_cast<uint2 *>(dstp + pl * PLANE + ((8 * g) % (PLANE - 8) & ~7)) = make_uint2(lo ^ 0x80808080u, hi);
            }
        }
        const float *av = arr + ((t + 1) & 1) * 2 * AS;
#pragma unroll
        for (int o4 = 0; o4 < 2; ++o4) {
            const int o = lt + 512 * o4, qidx = (20 * (o >> 4) + (o & 15)) % (AS - 1);
            // (c, s) by the product's kind of arithmetic: a phase, a polynomial, selects -- plain vector instructions
            const unsigned ph = (unsigned)(t * 1024 + o) * 381178347u;
            const float x = (float)(int)(ph << 2) * 3.6e-10f, x2 = x * x;
            float sn = x * (1.0f + x2 * (-0.16666667f + x2 * 0.0083333f)), cs = 1.0f + x2 * (-0.5f + x2 * 0.041666667f);
            if (ph & 0x40000000u) {
                const float tmp = sn;
                sn = cs;
                cs = -tmp;
            }
            // the product's registers, by name (v[40:41] = (u, v) from LDS, v[78:79] = (s, c), v[96:97] = the products)
            float plo, phi, u, e_lo, e_hi;
            // (a sweep of the multiply's issue time against the matrix waves' instruction stream: 0 .. 31 idle cycles, by tile)
            for (int d = (t + 5 * o4) & 31; d > 0; --d)
                asm volatile("s_nop 0");
            asm volatile("v_mov_b32 v78, %3\n\tv_mov_b32 v79, %4\n\t"
                         "ds_read2st64_b32 v[40:41], %5 offset1:20\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "v_pk_mul_f32 v[96:97], v[40:41], v[78:79] op_sel:[0,1] op_sel_hi:[0,0]\n\t"
                         "v_mov_b32 %0, v96\n\tv_mov_b32 %1, v97\n\tv_mov_b32 %2, v40"
                         : "=&v"(plo), "=&v"(phi), "=&v"(u) : "v"(sn), "v"(cs), "v"((unsigned)(size_t)(av + qidx))
                         : "memory", "v40", "v41", "v78", "v79", "v96", "v97");
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e_lo) : "v"(u), "v"(cs));      // u * c: what the low half must be
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e_hi) : "v"(u), "v"(sn));      // u * s: what the high half must be
            if (__builtin_bit_cast(unsigned, plo) != __builtin_bit_cast(unsigned, e_lo)) {
                ++wrong_lo;
                zero_lo += plo == 0.0f;
            }
            wrong_hi += __builtin_bit_cast(unsigned, phi) != __builtin_bit_cast(unsigned, e_hi);
        }
        __syncthreads();
    }
    if (wrong_lo | wrong_hi) {
        const int slot = ((wave & 3) * 4 + (lane >> 4)) * 3;
        atomicAdd(&bad[slot], wrong_lo);
        atomicAdd(&bad[slot + 1], zero_lo);
        atomicAdd(&bad[slot + 2], wrong_hi);
    }
}

int main(int argc, char **argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 256, tiles = argc > 2 ? atoi(argv[2]) : 20000;
    uint4 *in;
    unsigned long long *bad, h[48];
    int *sink;
    const size_t lds = 12 * PLANE + 4 * AS * 4;
    if (hipMalloc(&in, (size_t)(1 << 20) * 48 + 4096) != hipSuccess || hipMalloc(&bad, sizeof h) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess)
        return 2;
    (void)hipMemset(in, 0x5a, (size_t)(1 << 20) * 48 + 4096);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int total = 0;
    for (int matrix = 1; matrix >= 0; --matrix) {
        if (argc > 3 && atoi(argv[3]) != matrix)
            continue;
        (void)hipMemset(bad, 0, sizeof h);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(768), lds, 0, in, bad, tiles, matrix, sink);
        if (hipDeviceSynchronize() != hipSuccess) {
            printf("launch failed: %s\n", hipGetErrorString(hipGetLastError()));
            return 3;
        }
        (void)hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost);
        printf("matrix waves %s, %d block(s), %d tiles: %.3g products per loader wave\n", matrix ? "ISSUING" : "idle   ", blocks, tiles,
               2.0 * tiles * 64 * blocks);
        for (int w = 0; w < 4; ++w) {
            printf("  loader waves %d, %d: low half wrong by quarter", w + 4, w + 8);
            for (int q = 0; q < 4; ++q)
                printf(" %llu (zero: %llu)", h[(w * 4 + q) * 3], h[(w * 4 + q) * 3 + 1]);
            printf("   high half wrong");
            for (int q = 0; q < 4; ++q) {
                printf(" %llu", h[(w * 4 + q) * 3 + 2]);
                total += h[(w * 4 + q) * 3] != 0 || h[(w * 4 + q) * 3 + 2] != 0;
            }
            printf("\n");
        }
    }
    printf("%s\n", total ? "REPRODUCED: a packed product differs from the scalar one" : "clean");
    return 0;
}
