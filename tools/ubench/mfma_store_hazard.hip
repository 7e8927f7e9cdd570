// Reproducer for the hazard libperseus-sdr_amd/csrc/ddc_fir_i8.hip (store_f2_padded) works around:
//   a wave stores 64 bits per lane (global_store_dwordx2) and its NEXT vector instruction overwrites the store's data
//   registers, while ANOTHER wave on the same SIMD issues matrix instructions -> do the last lanes store the NEW value?
// One 768-thread block per CU, the product's shape: waves 0..3 issue v_mfma_i32_16x16x64_i8 with operands read from LDS (or
// idle: the control), waves 4..11 (two beside every matrix wave: waves w, w + 4, w + 8 share a SIMD) load, write LDS and
// store.  Every store goes to its own address; a second kernel counts words that hold the poison instead of the value, by
// SIMD and by quarter of the wave (lanes 0-15 .. 48-63).  The data registers' bank (number mod 4) is swept.
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_store_hazard mfma_store_hazard.hip   run: ./mfma_store_hazard [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
constexpr unsigned POISON = 0xDEADBEEFu;
__device__ __forceinline__ unsigned value(unsigned blk, unsigned w, unsigned it, unsigned lane) { return (blk * 8u + w) * 0x9E3779B1u + it * 64u + lane + 1u; }

// The store's data registers are v[100:101].  OVER: what overwrites them behind the store -- 0: one v_lshlrev_b64, 1: v_mov_b32
// of x then of y, 2: the RETURN of an LDS read (ds_read2st64_b32 v[100:101], what the product's next loop iteration does);
// NT: the product's nontemporal store.  PAD: s_nop count between the store and the overwriting instruction (-1: none)
template <int PAD, int OVER, bool NT>
__global__ __launch_bounds__(768, 1) void k_store(u2 *out, const uint4 *src, unsigned *hwid, int iters, int mfma, int *sink)
{
    __shared__ v4i lds[1024];
    __shared__ unsigned pois[2048];
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (lane == 0)
        hwid[blockIdx.x * 12 + wave] = hw;
    lds[threadIdx.x] = v4i{ (int)threadIdx.x, 1, 2, 3 };
    for (int i = threadIdx.x; i < 2048; i += 768)
        pois[i] = POISON;
    __syncthreads();
    if (wave < 4) {                                    // the matrix waves: operand reads from LDS + 12 MFMAs per round, like a k-step
        v4i a[4] = { { (int)lane, 1, 2, 3 }, { 4, 5, 6, (int)lane }, { 1, 1, 1, 1 }, { 2, 2, 2, 2 } }, c[4] = {};
        if (mfma)
            for (int i = 0; i < iters * 8; ++i) {
                v4i b[3];
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    b[k] = lds[(lane + 64 * k + 16 * (i & 31)) & 1023];
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        c[(k + j) & 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[j], b[k], c[(k + j) & 3], 0, 0, 0);
            }
        if (c[0][0] + c[1][1] + c[2][2] + c[3][3] == 0x7fffffff)
            *sink = 1;
        return;
    }
    const unsigned w = wave - 4;
    u2 *dst = out + ((size_t)(blockIdx.x * 8 + w) * iters) * 64 + lane;
    const uint4 *ld = src + (size_t)(blockIdx.x * 8 + w) * 64 + lane;
    const u2 poison = { POISON, POISON };
    const unsigned paddr = (unsigned)(size_t)(pois + lane);            // LDS byte address of this lane's poison words
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        const uint4 g0 = ld[(size_t)it * 256 * 8 * 64], g1 = ld[(size_t)it * 256 * 8 * 64 + 256 * 8 * 64 / 2];   // loads in flight beside the stores
        const unsigned val = value(blockIdx.x, w, it, lane);
        u2 v = { val, ~val };
#define PDDC_ST2(NTS, P, OV)                                                                                                    \
        asm volatile("v_mov_b32 v100, %1\n\tv_mov_b32 v101, %2\n\tglobal_store_dwordx2 %0, v[100:101], " NTS "\n\t" P OV              \
                     "\n\ts_waitcnt lgkmcnt(0)"                                                                                  \
                     :: "v"(dst), "v"(v.x), "v"(v.y), "v"(poison), "v"(poison.x), "v"(paddr) : "memory", "v100", "v101")
#define PDDC_ST1(NTS, P)                                                                                                        \
        if (OVER == 0)                                                                                                          \
            PDDC_ST2(NTS, P, "v_lshlrev_b64 v[100:101], 0, %3");                                                                \
        else if (OVER == 1)                                                                                                     \
            PDDC_ST2(NTS, P, "v_mov_b32 v100, %4\n\tv_mov_b32 v101, %4");                                                        \
        else                                                                                                                    \
            PDDC_ST2(NTS, P, "ds_read2st64_b32 v[100:101], %5 offset1:4");
#define PDDC_ST(P)                                                                                                              \
        if (NT) {                                                                                                               \
            PDDC_ST1("off nt", P)                                                                                               \
        } else {                                                                                                                \
            PDDC_ST1("off", P)                                                                                                  \
        }
        if (PAD < 0) {
            PDDC_ST("")
        } else if (PAD == 0) {
            PDDC_ST("s_nop 0\n\t")
        } else {
            PDDC_ST("s_nop 1\n\t")
        }
        acc += g0.x ^ g1.w;
        lds[(threadIdx.x + it) & 1023][0] = (int)acc;           // (an LDS write per round, as the plane writes are)
        dst += 64;
    }
    if (acc == 0x12345678u)
        *sink = 2;
}

__global__ void k_check(const u2 *out, int iters, unsigned nblk, unsigned long long *bad)     // bad[w][quarter][x|y]
{
    const size_t n = (size_t)nblk * 8 * iters * 64;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned lane = i & 63, it = (i >> 6) % iters, bw = (unsigned)((i >> 6) / iters), w = bw & 7, blk = bw >> 3;
        const unsigned e = value(blk, w, it, lane);
        const u2 g = out[i];
        if (g.x != e)
            atomicAdd(&bad[((w & 3) * 4 + (lane >> 4)) * 2 + 0], 1ull);
        if (g.y != ~e)
            atomicAdd(&bad[((w & 3) * 4 + (lane >> 4)) * 2 + 1], 1ull);
    }
}

template <int PAD, int VAR = 5>
static void launch(int var, int nblk, u2 *out, const uint4 *src, unsigned *hwid, int iters, int mfma, int *sink)
{
    if (var == VAR)
        hipLaunchKernelGGL((k_store<PAD, VAR % 3, (VAR >= 3)>), dim3(nblk), dim3(768), 0, 0, out, src, hwid, iters, mfma, sink);
    else if constexpr (VAR > 0)
        launch<PAD, VAR - 1>(var, nblk, out, src, hwid, iters, mfma, sink);
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2048, nblk = 256, reps = argc > 2 ? atoi(argv[2]) : 4;
    u2 *out;
    uint4 *src;
    unsigned *hwid, h_hw[12];
    unsigned long long *bad, h_bad[32];
    int *sink;
    const size_t nbytes = (size_t)nblk * 8 * iters * 64 * sizeof(u2), sbytes = ((size_t)iters + 1) * 256 * 8 * 64 * sizeof(uint4);
    if (hipMalloc(&out, nbytes) != hipSuccess || hipMalloc(&src, sbytes) != hipSuccess || hipMalloc(&hwid, nblk * 12 * 4) != hipSuccess ||
        hipMalloc(&bad, sizeof(h_bad)) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess)
        return 2;
    (void)hipMemset(src, 1, sbytes);
    int worst_padded = 0;
    for (int mfma = 1; mfma >= 0; --mfma)
        for (int pad = -1; pad <= 1; ++pad)
            for (int bank = 0; bank < 6; ++bank) {
                unsigned long long tot[32] = {};
                for (int rep = 0; rep < reps; ++rep) {
                    (void)hipMemset(out, 0, nbytes);
                    (void)hipMemset(bad, 0, sizeof(h_bad));
                    if (pad < 0)
                        launch<-1>(bank, nblk, out, src, hwid, iters, mfma, sink);
                    else if (pad == 0)
                        launch<0>(bank, nblk, out, src, hwid, iters, mfma, sink);
                    else
                        launch<1>(bank, nblk, out, src, hwid, iters, mfma, sink);
                    hipLaunchKernelGGL(k_check, dim3(1024), dim3(256), 0, 0, out, iters, (unsigned)nblk, bad);
                    if (hipMemcpy(h_bad, bad, sizeof(h_bad), hipMemcpyDeviceToHost) != hipSuccess)
                        return 3;
                    for (int i = 0; i < 32; ++i)
                        tot[i] += h_bad[i];
                }
                (void)hipMemcpy(h_hw, hwid, sizeof(h_hw), hipMemcpyDeviceToHost);
                printf("matrix waves %s, %s, %s store of v[100:101] then %s: %llu stores per SIMD; wrong x | y by SIMD and quarter of the wave:", mfma ? "ISSUING" : "idle   ",
                       pad < 0 ? "no pad " : pad == 0 ? "s_nop 0" : "s_nop 1", bank >= 3 ? "nt   " : "plain",
                       bank % 3 == 0 ? "v_lshlrev_b64   " : bank % 3 == 1 ? "2 v_mov_b32     " : "ds_read2st64_b32", 2ull * reps * nblk * iters * 64);
                for (int w = 0; w < 4; ++w) {
                    printf("  [simd %u:", (h_hw[w + 4] >> 4) & 3);
                    for (int q = 0; q < 4; ++q)
                        printf(" %llu", tot[(w * 4 + q) * 2]);
                    printf(" |");
                    for (int q = 0; q < 4; ++q)
                        printf(" %llu", tot[(w * 4 + q) * 2 + 1]);
                    printf("]");
                    if (pad == 1)
                        for (int q = 0; q < 8; ++q)
                            worst_padded += tot[w * 8 + q] != 0;
                }
                printf("\n");
            }
    printf("%s\n", worst_padded ? "PADDED FORM CORRUPTED" : "padded form clean");
    return worst_padded ? 1 : 0;
}
