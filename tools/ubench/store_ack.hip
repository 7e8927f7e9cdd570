// Microbenchmark (round 6): WHY does a write stream of 2 % of the traffic cost k_fir8's fused pair 13 % when it lies in the
// same HBM extent class as the read stream (NOTEBOOK rounds 1-3 5 (p)), while k_fir_i8x hardly cares (1-3 %)?
// The pair's traffic with no arithmetic: 512 persistent blocks of 256 threads, tiles of 4096 samples (24 KB: two 48-byte
// groups per thread, prefetched one tile ahead in registers), 512 B of result stores per tile.
//
// Hypothesis 1 (REFUTED by this file's first version, profiles/r06/a_store_ack.txt): gfx9 has one vmcnt for loads and stores,
//   so a wave that stores and then waits for its prefetched loads also waits for the store's acknowledgement.  A fifth
//   wave that only stores (mode 1) is no faster, and the store -> ack time is 2900 cycles in the same class against 2400
//   in another: not the difference.
// Hypothesis 2: it is the DRAM itself -- every small write lands in a bank whose open row belongs to the read stream
//   (precharge, activate, write recovery, activate again), and what decides the cost is how many such EVENTS there are and
//   whether consecutive ones share a row.  Knobs:
//     walk   0: block b owns one contiguous run of tiles (k_fir8's static schedule: 512 write positions 64 KB apart)
//            C: chunks of C tiles handed round the blocks (all blocks write into one window of 512*C*512 B)
//     burst  B: a block keeps the results of B tiles and writes B*512 B at once
//     flavour: nt / plain / sc1 / sc0 sc1
// One arena; input at its start, output in slot `o` (8 GiB slots): o = 1 is "first come".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0 coupled, 1 decoupled (fifth wave), 2 no stores
template <int MODE, int B, int FL>
__global__ __launch_bounds__(MODE == 1 ? 320 : 256) void k(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, int ntiles,
                                                         int tpb, int C, unsigned long long *lat)
{
    __shared__ u32x4 stage[B][32];
    const int tid = threadIdx.x;
    const bool loader = tid < 256;
    const int nblk = gridDim.x;
    /* the block's tiles in walk order: contiguous run, or chunks of C tiles round the blocks */
    auto tile_of = [&](int i) { return C == 0 ? (int)blockIdx.x * tpb + i : ((i / C) * nblk + (int)blockIdx.x) * C + i % C; };
    auto count = [&]() {
        if (C == 0)
            return max(0, min(tpb, ntiles - (int)blockIdx.x * tpb));
        int n = 0;
        for (int ch = blockIdx.x; ch * C < ntiles; ch += nblk)
            n += min(C, ntiles - ch * C);
        return n;
    };
    const int n = count();
    u32x4 raw[2][3];
    auto prefetch = [&](int t) {
        const u32x4 *src = in + ((size_t)t * 512 + tid) * 3;
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int w = 0; w < 3; ++w)
                raw[g][w] = src[(size_t)g * 256 * 3 + w];
    };
    if (loader && n > 0)
        prefetch(tile_of(0));
    unsigned long long lsum = 0, lmax = 0, ln = 0;
    for (int i = 0; i < n; ++i) {
        const int t = tile_of(i);
        u32x4 acc = { 0, 0, 0, 0 };
        if (loader) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int w = 0; w < 3; ++w)
                    acc ^= raw[g][w];                 /* consumes the prefetched tile: the compiler's vmcnt wait sits here */
            if (tid < 32)
                stage[i % B][tid] = acc;
        }
        __syncthreads();
        /* results leave: the tiles i-B+1 .. i of a burst are consecutive in memory (B divides C and tpb) */
        const int sb = MODE == 1 ? tid - 256 : tid;
        const bool storer = MODE != 2 && (i % B == B - 1 || i == n - 1) && sb >= 0 && sb < 32 * (i % B + 1);
        if (storer) {
            const u32x4 v = stage[sb >> 5][sb & 31];
            u32x4 *d = out + (size_t)(t - i % B) * 32 + sb;
            unsigned long long c0 = 0;
            if (MODE == 1)
                c0 = __builtin_readcyclecounter();
            if (FL == 0) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(d), "v"(v) : "memory");
            else if (FL == 1) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(d), "v"(v) : "memory");
            else if (FL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(d), "v"(v) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(d), "v"(v) : "memory");
            if (MODE == 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned long long c1 = __builtin_readcyclecounter();
                lsum += c1 - c0;
                lmax = c1 - c0 > lmax ? c1 - c0 : lmax;
                ++ln;
            }
        }
        if (loader && i + 1 < n)
            prefetch(tile_of(i + 1));
        __syncthreads();
    }
    if (MODE == 1 && tid == 256 && lat) {
        atomicAdd(lat, lsum);
        atomicAdd(lat + 1, ln);
        atomicMax(lat + 2, lmax);
    }
}

template <int MODE, int B = 1, int FL = 0>
static float run(const void *in, void *out, size_t ns, int C = 0, unsigned long long *lat = nullptr, int reps = 10)
{
    const int blocks = 512, ntiles = (int)(ns / 4096);
    const int tpb = (ntiles + blocks - 1) / blocks;
    static hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!e0) {
        hipEventCreate(&e0);
        hipEventCreate(&e1);
    }
    std::vector<float> v;
    for (int rep = 0; rep < reps + 4; ++rep) {
        if (lat)
            hipMemsetAsync(lat, 0, 24, 0);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, B, FL>), dim3(blocks), dim3(MODE == 1 ? 320 : 256), 0, 0, (const u32x4 *)in, (u32x4 *)out, ntiles,
                           tpb, C, lat);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 4)
            v.push_back(ms);
    }
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main(int argc, char **argv)
{
    const size_t ns = (size_t)1 << 28, slot = (size_t)8 << 30;
    const int nslots = argc > 1 ? atoi(argv[1]) : 9;
    char *arena;
    unsigned long long *lat, hl[3];
    CHECK(hipMalloc(&arena, slot * nslots));
    CHECK(hipMalloc(&lat, 24));
    CHECK(hipMemset(arena, 1, ns * 6));
    for (int i = 0; i < 200; ++i)       /* settle the clocks */
        run<2>(arena, arena + slot, ns, 0, nullptr, 1);
    printf("pair-like traffic, 2^28 samples, 512 B of stores per 24 KB tile; ms (median of 10); ack = store -> vmcnt(0), shader cycles\n");
    int worst = 1, best = 1;
    float tw = 0, tb = 1e9;
    for (int o = 1; o < nslots; ++o) {
        char *out = arena + o * slot + ((size_t)2 << 30);
        const float none = run<2>(arena, out, ns);
        const float coupled = run<0>(arena, out, ns);
        const float dec = run<1>(arena, out, ns, 0, lat);
        CHECK(hipMemcpy(hl, lat, 24, hipMemcpyDeviceToHost));
        printf("out slot %d: no stores %.4f  coupled %.4f  decoupled (5th wave) %.4f   ack mean %.0f max %llu cycles (n=%llu)\n", o, none, coupled,
               dec, hl[1] ? (double)hl[0] / hl[1] : 0.0, hl[2], hl[1]);
        if (coupled > tw) { tw = coupled; worst = o; }
        if (coupled < tb) { tb = coupled; best = o; }
    }
    printf("slow slot %d, fast slot %d\n", worst, best);
    for (int pass = 0; pass < 2; ++pass)
        for (int o : { worst, best }) {
            char *out = arena + o * slot + ((size_t)2 << 30);
            printf("pass %d slot %d  no stores: contiguous %.4f  round robin C=1 %.4f C=8 %.4f\n", pass, o, run<2>(arena, out, ns, 0), run<2>(arena, out, ns, 1),
                   run<2>(arena, out, ns, 8));
            printf("pass %d slot %d  contiguous runs, burst 1/2/4/8/16 tiles:  nt %.4f %.4f %.4f %.4f %.4f\n", pass, o, run<0, 1>(arena, out, ns, 0),
                   run<0, 2>(arena, out, ns, 0), run<0, 4>(arena, out, ns, 0), run<0, 8>(arena, out, ns, 0), run<0, 16>(arena, out, ns, 0));
            printf("pass %d slot %d  contiguous runs, flavour (burst 1 / 8): plain %.4f %.4f  sc1 %.4f %.4f  sc0sc1 %.4f %.4f\n", pass, o,
                   run<0, 1, 1>(arena, out, ns, 0), run<0, 8, 1>(arena, out, ns, 0), run<0, 1, 2>(arena, out, ns, 0), run<0, 8, 2>(arena, out, ns, 0),
                   run<0, 1, 3>(arena, out, ns, 0), run<0, 8, 3>(arena, out, ns, 0));
            printf("pass %d slot %d  round robin, burst 1: C=1 %.4f C=2 %.4f C=4 %.4f C=8 %.4f C=16 %.4f C=32 %.4f\n", pass, o, run<0, 1>(arena, out, ns, 1),
                   run<0, 1>(arena, out, ns, 2), run<0, 1>(arena, out, ns, 4), run<0, 1>(arena, out, ns, 8), run<0, 1>(arena, out, ns, 16),
                   run<0, 1>(arena, out, ns, 32));
            printf("pass %d slot %d  round robin, burst = C: C=2 %.4f C=4 %.4f C=8 %.4f C=16 %.4f\n", pass, o, run<0, 2>(arena, out, ns, 2),
                   run<0, 4>(arena, out, ns, 4), run<0, 8>(arena, out, ns, 8), run<0, 16>(arena, out, ns, 16));
            printf("pass %d slot %d  round robin plain stores, burst 1: C=1 %.4f C=8 %.4f ; burst = C = 8: %.4f\n", pass, o, run<0, 1, 1>(arena, out, ns, 1),
                   run<0, 1, 1>(arena, out, ns, 8), run<0, 8, 1>(arena, out, ns, 8));
        }
    return 0;
}
