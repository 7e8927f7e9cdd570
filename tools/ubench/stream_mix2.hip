// Does the 48-byte-per-lane load shape (TA-heavy) cost anything against a
// perfectly lane-contiguous load shape, read-only and with the 1 B/sample nt writes?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// SHAPE 0: lane loads 3 consecutive uint4 (48 B, like k_fir8); SHAPE 1: each load instruction is lane-contiguous
template <int SHAPE, int WR>
__global__ __launch_bounds__(256) void k(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, size_t n16_in, size_t n16_out)
{
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
    const size_t ngroups = n16_in / 3;
    u32x4 acc = { 0, 0, 0, 0 };
    size_t sweep = 0;
    for (size_t g = tid; g < ngroups; g += 4 * nth, ++sweep) {
        u32x4 v[12];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                size_t idx = SHAPE == 0 ? (g + k * nth) * 3 + w                       // 48-B stride between lanes
                                        : (sweep * 12 + k * 3 + w) * nth + tid;       // contiguous across lanes
                v[k * 3 + w] = idx < n16_in ? in[idx] : u32x4{ 0, 0, 0, 0 };
            }
#pragma unroll
        for (int i = 0; i < 12; ++i) { acc.x ^= v[i].x; acc.y += v[i].y; acc.z ^= v[i].z; acc.w += v[i].w; }
        if (WR) {
            size_t oi = sweep * (2 * nth) + tid;
            if (oi < n16_out) __builtin_nontemporal_store(acc, &out[oi]);
            if (oi + nth < n16_out) __builtin_nontemporal_store(acc, &out[oi + nth]);
        }
    }
    if (!WR && acc.x == 0x12345678u) out[tid] = acc;
}

int main()
{
    const size_t ns = (size_t)1 << 28, in_bytes = ns * 6, out_bytes = ns;
    u32x4 *in, *out;
    CHECK(hipMalloc(&in, in_bytes)); CHECK(hipMalloc(&out, in_bytes)); CHECK(hipMemset(in, 1, in_bytes));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *nm[4] = { "48B/lane read-only", "contig   read-only", "48B/lane r+w nt   ", "contig   r+w nt   " };
    for (int cfg = 0; cfg < 4; ++cfg)
        for (int blocks : { 512, 1024, 2048 }) {
            float best = 1e9;
            for (int rep = 0; rep < 8; ++rep) {
                dim3 g(blocks), b(256); size_t a = in_bytes / 16, c = out_bytes / 16;
                hipEventRecord(e0);
                if (cfg == 0) hipLaunchKernelGGL((k<0, 0>), g, b, 0, 0, in, out, a, c);
                if (cfg == 1) hipLaunchKernelGGL((k<1, 0>), g, b, 0, 0, in, out, a, c);
                if (cfg == 2) hipLaunchKernelGGL((k<0, 1>), g, b, 0, 0, in, out, a, c);
                if (cfg == 3) hipLaunchKernelGGL((k<1, 1>), g, b, 0, 0, in, out, a, c);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            double bytes = cfg < 2 ? in_bytes : in_bytes + out_bytes;
            printf("%s blocks=%4d  %.3f ms  %.2f TB/s\n", nm[cfg], blocks, best, bytes / best / 1e9);
        }
    return 0;
}
