/*
 * ddc_dev.h -- device-side helpers shared by the kernel translation units (ddc_kernels.hip, ddc_fir_i8.hip).
 * Internal; gfx950 only.
 */
#ifndef PDDC_DDC_DEV_H
#define PDDC_DDC_DEV_H

#include <hip/hip_runtime.h>
#include <stdint.h>

/* loads through a pointer of this address space go through the scalar cache (s_load) when the address is uniform */
#define PDDC_CONSTANT __attribute__((address_space(4)))

namespace pddc {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

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

} // namespace pddc
#endif
