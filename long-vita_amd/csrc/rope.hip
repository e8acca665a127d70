// Rotary position embedding on bf16 rows (non-interleaved "rotate half" form), forward and
// backward, for gfx950.  Pure data movement plus six roundings per element pair: HBM-bound.
//
//   out[.., c]        = bf16( bf16(x[c]      * cos[c])      + bf16(-x[c + r/2] * sin[c]) )        c < r/2
//   out[.., c + r/2]  = bf16( bf16(x[c+r/2]  * cos[c+r/2])  + bf16( x[c]       * sin[c+r/2]) )
//   out[.., c >= r]   = x[c]                                   (columns past the rotary width pass through)
//
// Every product and the sum are rounded to bf16 on their own: that is what the reference's eager
// bf16 tensor ops do (rotary_pos_embedding.py:200-203), and it makes the result bit-identical to
// theirs given the same cos / sin tables.  `conj` selects the transpose of the rotation (the
// gradient): out[c] = bf16(bf16(g[c] cos[c]) + bf16(g[c+r/2] sin[c+r/2])),
//            out[c+r/2] = bf16(bf16(g[c+r/2] cos[c+r/2]) - bf16(g[c] sin[c])).
#include "internal.h"

// Round to the nearest bf16 (ties to even) and return it as f32.  Done on the bit pattern: written as a
// float -> __bf16 -> float cast pair the compiler may keep the unrounded value ("excess precision") and
// fuse the product into the following add, which changes results in the last bit.
__device__ __forceinline__ float rb(float v) {
    uint32_t u = __float_as_uint(v);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return __uint_as_float(u & 0xFFFF0000u);
}
__device__ __forceinline__ __bf16 as_bf16(float rounded) {  // `rounded` already has zero low bits
    return __builtin_bit_cast(__bf16, (uint16_t)(__float_as_uint(rounded) >> 16));
}
// The same rounding by the hardware converter, two values per instruction (round to nearest even is the
// default mode).  As asm so that the value really passes through bf16 here.
__device__ __forceinline__ uint32_t rb2(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float lo_f32(uint32_t pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float hi_f32(uint32_t pk) { return __uint_as_float(pk & 0xFFFF0000u); }

template <bool CONJ>
__global__ __launch_bounds__(256) void lv_rope_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ out,
                                                       const __bf16* __restrict__ cos_tab, const __bf16* __restrict__ sin_tab,
                                                       const int32_t* __restrict__ pos_ids, int64_t rows, int heads, int d, int rot,
                                                       int64_t x_rs, int64_t x_hs, int64_t o_rs, int64_t o_hs, int64_t tab_rows,
                                                       int tab_div) {
    // one thread: 8 columns of the first rotary half and the matching 8 of the second half
    // (two 16-byte loads, two 16-byte stores), or 8 pass-through columns
    const int half = rot / 2;
    const int rot_chunks = half / 8;               // chunk pairs inside the rotary width
    const int pass_chunks = (d - rot) / 8;         // plain 8-column chunks after it
    const int per_head = rot_chunks + pass_chunks;
    const int64_t total = rows * heads * per_head;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % per_head);
        const int64_t rh = idx / per_head;
        const int head = (int)(rh % heads);
        const int64_t row = rh / heads;
        const __bf16* xp = x + row * x_rs + (int64_t)head * x_hs;
        __bf16* op = out + row * o_rs + (int64_t)head * o_hs;
        if (ch >= rot_chunks) {
            const int c0 = rot + 8 * (ch - rot_chunks);
            *reinterpret_cast<bf16x8*>(op + c0) = *reinterpret_cast<const bf16x8*>(xp + c0);
            continue;
        }
        int64_t trow = pos_ids ? (int64_t)pos_ids[row] : row / tab_div;
        trow = trow < 0 ? 0 : (trow >= tab_rows ? tab_rows - 1 : trow);  // host checks the range; never read outside the table
        const int c0 = 8 * ch;
        const bf16x8 x1 = *reinterpret_cast<const bf16x8*>(xp + c0);
        const bf16x8 x2 = *reinterpret_cast<const bf16x8*>(xp + c0 + half);
        const bf16x8 c1 = *reinterpret_cast<const bf16x8*>(cos_tab + trow * rot + c0);
        const bf16x8 c2 = *reinterpret_cast<const bf16x8*>(cos_tab + trow * rot + c0 + half);
        const bf16x8 s1 = *reinterpret_cast<const bf16x8*>(sin_tab + trow * rot + c0);
        const bf16x8 s2 = *reinterpret_cast<const bf16x8*>(sin_tab + trow * rot + c0 + half);
        u32x4 o1, o2;
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            float r1[2], r2[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float a = (float)x1[i + e], b = (float)x2[i + e];
                // each product is rounded on its own (two per converter instruction), then the sum
                const uint32_t p1 = CONJ ? rb2(a * (float)c1[i + e], b * (float)s2[i + e]) : rb2(a * (float)c1[i + e], b * (float)s1[i + e]);
                const uint32_t p2 = CONJ ? rb2(b * (float)c2[i + e], a * (float)s1[i + e]) : rb2(b * (float)c2[i + e], a * (float)s2[i + e]);
                r1[e] = CONJ ? lo_f32(p1) + hi_f32(p1) : lo_f32(p1) - hi_f32(p1);
                r2[e] = CONJ ? lo_f32(p2) - hi_f32(p2) : lo_f32(p2) + hi_f32(p2);
            }
            o1[i >> 1] = rb2(r1[0], r1[1]);
            o2[i >> 1] = rb2(r2[0], r2[1]);
        }
        *reinterpret_cast<u32x4*>(op + c0) = o1;
        *reinterpret_cast<u32x4*>(op + c0 + half) = o2;
    }
}

int lv_launch_rope(const void* x, void* out, const void* cos_tab, const void* sin_tab, const int32_t* pos_ids, int64_t rows,
                   int heads, int d, int rot, int64_t x_rs, int64_t x_hs, int64_t o_rs, int64_t o_hs, int64_t tab_rows, int tab_div,
                   int conj, hipStream_t stream) {
    const int64_t total = rows * heads * (int64_t)((rot / 2 + (d - rot)) / 8);
    if (total == 0) return 0;
    // enough workgroups to fill 256 CUs several times over; the grid-stride loop takes the rest
    int64_t grid = (total + 255) / 256;
    if (grid > 256 * 32) grid = 256 * 32;
    if (conj)
        hipLaunchKernelGGL((lv_rope_kernel<true>), dim3((unsigned)grid), dim3(256), 0, stream, (const __bf16*)x, (__bf16*)out,
                           (const __bf16*)cos_tab, (const __bf16*)sin_tab, pos_ids, rows, heads, d, rot, x_rs, x_hs, o_rs, o_hs, tab_rows,
                           tab_div);
    else
        hipLaunchKernelGGL((lv_rope_kernel<false>), dim3((unsigned)grid), dim3(256), 0, stream, (const __bf16*)x, (__bf16*)out,
                           (const __bf16*)cos_tab, (const __bf16*)sin_tab, pos_ids, rows, heads, d, rot, x_rs, x_hs, o_rs, o_hs, tab_rows,
                           tab_div);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
