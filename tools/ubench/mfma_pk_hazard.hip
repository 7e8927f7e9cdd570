// Reproducer for the second hazard libperseus-sdr_amd/csrc/ddc_fir_i8.hip avoids (no packed fp32 in code that runs beside
// a matrix wave): v_pk_mul_f32 -> v_pk_fma_f32 chains in a wave that shares its SIMD with a wave issuing matrix
// instructions -- do lanes 48..63 now and then come out wrong?
// One 768-thread block per CU, the product's shape: waves 0..3 run the product's band pass (tap tables in 96 registers,
// operands from LDS, v_mfma_i32_16x16x64_i8) or idle (the control), waves 4..11 (two beside every matrix wave) run the chain
// and compare every result, in registers, with the same arithmetic done one float per instruction (v_mul_f32 /
// v_fma_f32); mismatches are counted by SIMD and quarter of the wave.
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_pk_hazard mfma_pk_hazard.hip      run: ./mfma_pk_hazard [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// FORM 0: one float per instruction (the reference form); 1: the packed sequence hipcc's SLP vectoriser made of the product's
// rotation (x = u c - v s, y = v c + u s), register for register, INCLUDING the instructions behind it that overwrite the
// sources of the v_pk_fma_f32 (v78 = its src1.lo, then v96 = its src2.lo) at once; 2: the same with `s_nop 4` between the
// v_pk_fma_f32 and the first overwrite
template <int FORM>
__global__ __launch_bounds__(768, 1) void k_chain(unsigned *hwid, unsigned long long *bad, int iters, int mfma, int *sink)
{
    __shared__ v4i lds[2048];
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (lane == 0)
        hwid[blockIdx.x * 12 + wave] = hw;
    for (int i = threadIdx.x; i < 2048; i += 768)
        lds[i] = v4i{ i, 1, 2, 3 };
    __syncthreads();
    if (wave < 4) {
        // the product's band pass: two tap tables of 3 k-steps x 4 digit planes in registers (96 VGPRs), the six byte planes'
        // operands from LDS, 18 matrix instructions per k-step into four accumulators
        v4i A0[3][4], A1[3][4], acc[4] = {};
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                A0[ks][j] = lds[(lane + 64 * (4 * ks + j)) & 2047];
                A1[ks][j] = lds[(lane + 64 * (4 * ks + j) + 777) & 2047];
            }
        if (mfma)
            for (int i = 0; i < iters / 2; ++i) {
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    v4i BI[3], BQ[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        BI[k] = lds[(lane + 64 * k + 16 * (i & 31) + 200 * ks) & 2047];
                        BQ[k] = lds[(lane + 64 * k + 16 * (i & 31) + 200 * ks + 1000) & 2047];
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (k + j >= 2) {
                                acc[k + j - 2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A0[ks][j], BI[k], acc[k + j - 2], 0, 0, 0);
                                acc[k + j - 2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A1[ks][j], BQ[k], acc[k + j - 2], 0, 0, 0);
                            }
                }
            }
        if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 0x7fffffff)
            *sink = 1;
        return;
    }
    unsigned wrong_x = 0, wrong_y = 0;
    float u = 1.0f + lane * 0.001f, v = 2.0f - lane * 0.002f;
    for (int it = 0; it < iters; ++it) {
        const float ang = 0.37f * (float)((it * 7 + lane) & 255), c = __builtin_cosf(ang), sn = __builtin_sinf(ang);
        float ex, ey, t0, t1, rx, ry;
        asm volatile("v_mul_f32 %2, %4, %6\n\tv_fma_f32 %0, -%5, %7, %2\n\tv_mul_f32 %3, %4, %7\n\tv_fma_f32 %1, %5, %6, %3"
                     : "=&v"(ex), "=&v"(ey), "=&v"(t0), "=&v"(t1) : "v"(u), "v"(v), "v"(c), "v"(sn));
        if (FORM == 0)
            asm volatile("v_mul_f32 %2, %4, %6\n\tv_fma_f32 %0, -%5, %7, %2\n\tv_mul_f32 %3, %4, %7\n\tv_fma_f32 %1, %5, %6, %3"
                         : "=&v"(rx), "=&v"(ry), "=&v"(t0), "=&v"(t1) : "v"(u), "v"(v), "v"(c), "v"(sn));
        else
#define PDDC_SEQ(GAP)                                                                                                        \
            asm volatile("v_mov_b32 v40, %2\n\tv_mov_b32 v41, %3\n\tv_mov_b32 v78, %5\n\tv_mov_b32 v79, %4\n\tv_mov_b32 v39, %6\n\ts_nop 4\n\t" \
                         "v_pk_mul_f32 v[96:97], v[40:41], v[78:79] op_sel:[0,1] op_sel_hi:[0,0]\n\t"                            \
                         "v_mov_b32 v40, v41\n\t"                                                                              \
                         "v_add_u32 v95, 0x20000000, v39\n\t"                                                                  \
                         "v_pk_fma_f32 v[40:41], v[40:41], v[78:79], v[96:97] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t" GAP          \
                         "v_and_b32 v78, -2.0, v95\n\t"                                                                         \
                         "v_sub_u32 v78, v39, v78\n\t"                                                                          \
                         "v_cvt_f32_i32 v96, v78\n\t"                                                                           \
                         "s_nop 4\n\tv_mov_b32 %0, v40\n\tv_mov_b32 %1, v41"                                                     \
                         : "=&v"(rx), "=&v"(ry) : "v"(u), "v"(v), "v"(c), "v"(sn), "v"(it * 0x01234567 + lane)                    \
                         : "v39", "v40", "v41", "v78", "v79", "v95", "v96", "v97")
        if (FORM == 1)
            PDDC_SEQ("");
        else if (FORM == 3) {
            float2 *slot = reinterpret_cast<float2 *>(&lds[1024]) + threadIdx.x;          // this lane's u at [0], v 20 * 64 dwords on
            float *sl = reinterpret_cast<float *>(&lds[1024]);
            sl[threadIdx.x] = u;
            sl[threadIdx.x + 20 * 64] = v;
            (void)slot;
            asm volatile("v_mov_b32 v78, %4\n\tv_mov_b32 v79, %3\n\tv_mov_b32 v39, %5\n\ts_waitcnt lgkmcnt(0)\n\t"
                         "ds_read2st64_b32 v[40:41], %2 offset1:20\n\t"
                         "v_xor_b32 v78, 0x80000000, v78\n\tv_xor_b32 v78, 0x80000000, v78\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "v_pk_mul_f32 v[96:97], v[40:41], v[78:79] op_sel:[0,1] op_sel_hi:[0,0]\n\t"
                         "v_mov_b32 v40, v41\n\t"
                         "v_add_u32 v95, 0x20000000, v39\n\t"
                         "v_pk_fma_f32 v[40:41], v[40:41], v[78:79], v[96:97] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"
                         "v_and_b32 v78, -2.0, v95\n\t"
                         "v_sub_u32 v78, v39, v78\n\t"
                         "v_cvt_f32_i32 v96, v78\n\t"
                         "s_nop 4\n\tv_mov_b32 %0, v40\n\tv_mov_b32 %1, v41"
                         : "=&v"(rx), "=&v"(ry) : "v"((unsigned)(size_t)(sl + threadIdx.x)), "v"(c), "v"(sn), "v"(it * 0x01234567 + lane)
                         : "memory", "v39", "v40", "v41", "v78", "v79", "v95", "v96", "v97");
        } else
            PDDC_SEQ("s_nop 4\n\t");
        wrong_x += __builtin_bit_cast(unsigned, rx) != __builtin_bit_cast(unsigned, ex);
        wrong_y += __builtin_bit_cast(unsigned, ry) != __builtin_bit_cast(unsigned, ey);
        u = ex * 0.5f + 1.0f;
        v = ey * 0.5f - 1.0f;
    }
    if (wrong_x)
        atomicAdd(&bad[(((wave - 4) & 3) * 4 + (lane >> 4)) * 2], (unsigned long long)wrong_x);
    if (wrong_y)
        atomicAdd(&bad[(((wave - 4) & 3) * 4 + (lane >> 4)) * 2 + 1], (unsigned long long)wrong_y);
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 100000, nblk = 256;
    unsigned *hwid, h_hw[12];
    unsigned long long *bad, h_bad[32];
    int *sink;
    if (hipMalloc(&hwid, nblk * 12 * 4) != hipSuccess || hipMalloc(&bad, sizeof(h_bad)) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess)
        return 2;
    int clean_forms_wrong = 0;
    for (int mfma = 1; mfma >= 0; --mfma)
        for (int form = 1; form <= 4; ++form) {
            (void)hipMemset(bad, 0, sizeof(h_bad));
            for (int rep = 0; rep < 8; ++rep) {
                if (form == 1)
                    hipLaunchKernelGGL(k_chain<1>, dim3(nblk), dim3(768), 0, 0, hwid, bad, iters, mfma, sink);
                else if (form == 2)
                    hipLaunchKernelGGL(k_chain<2>, dim3(nblk), dim3(768), 0, 0, hwid, bad, iters, mfma, sink);
                else if (form == 4)
                    hipLaunchKernelGGL(k_chain<3>, dim3(nblk), dim3(768), 0, 0, hwid, bad, iters, mfma, sink);
                else
                    hipLaunchKernelGGL(k_chain<0>, dim3(nblk), dim3(768), 0, 0, hwid, bad, iters, mfma, sink);
            }
            if (hipMemcpy(h_bad, bad, sizeof(h_bad), hipMemcpyDeviceToHost) != hipSuccess)
                return 3;
            (void)hipMemcpy(h_hw, hwid, sizeof(h_hw), hipMemcpyDeviceToHost);
            printf("matrix wave %s, %s: %llu results per wave\n", mfma ? "ISSUING" : "idle   ",
                   form == 1 ? "packed, sources overwritten at once " : form == 2 ? "packed, s_nop 4 before the overwrite" : form == 4 ? "packed, inputs through an LDS read  " : "one float per instruction          ",
                   8ull * nblk * iters * 64);
            for (int w = 0; w < 4; ++w) {
                printf("  waves %d, %d (simd %u): wrong x by quarter of the wave", w + 4, w + 8, (h_hw[w + 4] >> 4) & 3);
                for (int q = 0; q < 4; ++q)
                    printf(" %llu", h_bad[(w * 4 + q) * 2]);
                printf("   wrong y");
                for (int q = 0; q < 4; ++q) {
                    printf(" %llu", h_bad[(w * 4 + q) * 2 + 1]);
                    if (form != 1 && form != 4)
                        clean_forms_wrong += (h_bad[(w * 4 + q) * 2] | h_bad[(w * 4 + q) * 2 + 1]) != 0;
                }
                printf("\n");
            }
        }
    printf("%s\n", clean_forms_wrong ? "A FORM EXPECTED CLEAN IS WRONG" : "scalar and padded forms clean");
    return clean_forms_wrong ? 1 : 0;
}
