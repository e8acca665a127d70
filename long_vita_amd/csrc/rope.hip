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
            for (int e = 0; e < 2; ++e)   // each product is rounded on its own (two per converter instruction), then the sum
                rope_pair<CONJ>((float)x1[i + e], (float)x2[i + e], (float)c1[i + e], (float)s1[i + e], (float)c2[i + e],
                                (float)s2[i + e], r1[e], r2[e]);
            o1[i >> 1] = rb2(r1[0], r1[1]);
            o2[i >> 1] = rb2(r2[0], r2[1]);
        }
        *reinterpret_cast<u32x4*>(op + c0) = o1;
        *reinterpret_cast<u32x4*>(op + c0 + half) = o2;
    }
}

// Rotary embedding of K fused with the packing of K and V into ONE message buffer kv [2][rows][heads][d] (the block a
// context-parallel rank sends round the ring): kv[0] = rope(k), kv[1] = v.  One thread: one rotary chunk pair of K, or
// one pass-through chunk of K, or one 16-byte chunk of V.
__global__ __launch_bounds__(256) void lv_rope_pack_kv_kernel(const __bf16* __restrict__ k, const __bf16* __restrict__ v,
                                                               __bf16* __restrict__ kv, const __bf16* __restrict__ cos_tab,
                                                               const __bf16* __restrict__ sin_tab, const int32_t* __restrict__ pos_ids,
                                                               int64_t rows, int heads, int d, int rot, int64_t k_rs, int64_t k_hs,
                                                               int64_t v_rs, int64_t v_hs, int64_t tab_rows, int tab_div) {
    const int half = rot / 2;
    const int rot_chunks = half / 8, pass_chunks = (d - rot) / 8, v_chunks = d / 8;
    const int per_head = rot_chunks + pass_chunks + v_chunks;
    const int64_t total = rows * heads * per_head;
    const int64_t o_rs = (int64_t)heads * d;
    __bf16* const vout = kv + rows * o_rs;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % per_head);
        const int64_t rh = idx / per_head;
        const int head = (int)(rh % heads);
        const int64_t row = rh / heads;
        const __bf16* xp = k + row * k_rs + (int64_t)head * k_hs;
        __bf16* op = kv + row * o_rs + (int64_t)head * d;
        if (ch >= rot_chunks + pass_chunks) {
            const int c0 = 8 * (ch - rot_chunks - pass_chunks);
            *reinterpret_cast<bf16x8*>(vout + row * o_rs + (int64_t)head * d + c0) =
                *reinterpret_cast<const bf16x8*>(v + row * v_rs + (int64_t)head * v_hs + c0);
            continue;
        }
        if (ch >= rot_chunks) {
            const int c0 = rot + 8 * (ch - rot_chunks);
            *reinterpret_cast<bf16x8*>(op + c0) = *reinterpret_cast<const bf16x8*>(xp + c0);
            continue;
        }
        int64_t trow = pos_ids ? (int64_t)pos_ids[row] : row / tab_div;
        trow = trow < 0 ? 0 : (trow >= tab_rows ? tab_rows - 1 : trow);
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
            for (int e = 0; e < 2; ++e)
                rope_pair<false>((float)x1[i + e], (float)x2[i + e], (float)c1[i + e], (float)s1[i + e], (float)c2[i + e],
                                 (float)s2[i + e], r1[e], r2[e]);
            o1[i >> 1] = rb2(r1[0], r1[1]);
            o2[i >> 1] = rb2(r2[0], r2[1]);
        }
        *reinterpret_cast<u32x4*>(op + c0) = o1;
        *reinterpret_cast<u32x4*>(op + c0 + half) = o2;
    }
}

int lv_launch_rope_pack_kv(const void* k, const void* v, void* kv, const void* cos_tab, const void* sin_tab, const int32_t* pos_ids,
                           int64_t rows, int heads, int d, int rot, int64_t k_rs, int64_t k_hs, int64_t v_rs, int64_t v_hs,
                           int64_t tab_rows, int tab_div, hipStream_t stream) {
    const int64_t total = rows * heads * (int64_t)((rot / 2 + (d - rot) + d) / 8);
    if (total == 0) return 0;
    int64_t grid = (total + 255) / 256;
    if (grid > 256 * 32) grid = 256 * 32;
    hipLaunchKernelGGL(lv_rope_pack_kv_kernel, dim3((unsigned)grid), dim3(256), 0, stream, (const __bf16*)k, (const __bf16*)v,
                       (__bf16*)kv, (const __bf16*)cos_tab, (const __bf16*)sin_tab, pos_ids, rows, heads, d, rot, k_rs, k_hs, v_rs,
                       v_hs, tab_rows, tab_div);
    return hipGetLastError() == hipSuccess ? 0 : -2;
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
