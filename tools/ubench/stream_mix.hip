// Microbenchmark: achievable HBM rate for the hot path's traffic mix on MI355X:
// read 6 B / sample (48-byte groups per lane, like k_fir8's loads) and write
// 1 B / sample (coalesced 16-byte stores), no compute.  Also plain copy / read-only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ inline uint4 ntload(const uint4 *p) { u32x4 v = __builtin_nontemporal_load((const u32x4 *)p); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ inline void ntstore(uint4 a, uint4 *p) { u32x4 v = { a.x, a.y, a.z, a.w }; __builtin_nontemporal_store(v, (u32x4 *)p); }
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// mode 0: read-only (sum), 1: read 6 + write 1 per sample, 2: float4 copy
template <int MODE, int NTL, int NTS>
__global__ __launch_bounds__(256) void k(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n16_in, size_t n16_out)
{
    size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
    size_t nth = (size_t)gridDim.x * 256;
    if (MODE == 2) {
        for (size_t i = tid; i < n16_in; i += nth) out[i] = in[i];
        return;
    }
    uint4 acc = make_uint4(0, 0, 0, 0);
    // each thread: groups of 3 consecutive uint4 (48 B), 4 groups in flight
    size_t ngroups = n16_in / 3;
    for (size_t g = tid; g < ngroups; g += 4 * nth) {
        uint4 v[4][3];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            size_t gg = g + k * nth;
            if (gg < ngroups) {
#pragma unroll
                for (int w = 0; w < 3; ++w) v[k][w] = NTL ? ntload(&in[gg * 3 + w]) : in[gg * 3 + w];
            } else {
#pragma unroll
                for (int w = 0; w < 3; ++w) v[k][w] = make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int w = 0; w < 3; ++w) { acc.x ^= v[k][w].x; acc.y += v[k][w].y; acc.z ^= v[k][w].z; acc.w += v[k][w].w; }
        if (MODE == 1) {
            // 4 groups = 32 samples -> 32 B of output = 2 uint4 per thread ... write 1 uint4 per 2 groups
            size_t o = (g / nth) * nth / 2 + tid / 1;   // keep it simple: coalesced region per sweep
            size_t oi = (g / (4 * nth)) * (2 * nth) + tid;
            if (oi < n16_out) { if (NTS) ntstore(acc, &out[oi]); else out[oi] = acc; }
            if (oi + nth < n16_out) { if (NTS) ntstore(acc, &out[oi + nth]); else out[oi + nth] = acc; }
            (void)o;
        }
    }
    if (MODE == 0 && acc.x == 0x12345678u) out[tid] = acc;
}

int main()
{
    const size_t ns = (size_t)1 << 28;
    const size_t in_bytes = ns * 6, out_bytes = ns;       // 6 B in, 1 B out per sample
    uint4 *in, *out;
    CHECK(hipMalloc(&in, in_bytes));
    CHECK(hipMalloc(&out, in_bytes));
    CHECK(hipMemset(in, 1, in_bytes));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int cfg = 0; cfg < 6; ++cfg)
        for (int blocks : { 512, 1024 }) {
            float best = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0);
                dim3 g(blocks), b(256);
                size_t a = in_bytes / 16, c = out_bytes / 16;
                switch (cfg) {
                case 0: hipLaunchKernelGGL((k<0, 0, 0>), g, b, 0, 0, in, out, a, c); break;
                case 1: hipLaunchKernelGGL((k<0, 1, 0>), g, b, 0, 0, in, out, a, c); break;
                case 2: hipLaunchKernelGGL((k<1, 0, 0>), g, b, 0, 0, in, out, a, c); break;
                case 3: hipLaunchKernelGGL((k<1, 1, 0>), g, b, 0, 0, in, out, a, c); break;
                case 4: hipLaunchKernelGGL((k<1, 0, 1>), g, b, 0, 0, in, out, a, c); break;
                case 5: hipLaunchKernelGGL((k<1, 1, 1>), g, b, 0, 0, in, out, a, c); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const char *nm[6] = { "read-only plain", "read-only nt   ", "r+w plain/plain", "r+w ntload     ", "r+w ntstore    ", "r+w nt/nt      " };
            double bytes = cfg < 2 ? in_bytes : in_bytes + out_bytes;
            printf("%s blocks=%4d  %.3f ms  %.2f TB/s\n", nm[cfg], blocks, best, bytes / best / 1e9);
        }
    return 0;
}
