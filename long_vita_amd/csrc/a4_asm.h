// Inline-asm building blocks of the forward kernels that lay the accumulator half of the register file out by hand
// (one wave per SIMD, 512 registers per lane).  The compiler neither counts nor pads what is inside an asm statement: the
// callers place the waits and the wait states (a vector write -> MFMA operand: 2; an MFMA result -> any other reader: 12).
#pragma once
#include "internal.h"
#include <type_traits>

typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((ext_vector_type(4))) uint32_t a4_u32x4;

// One staging request: piece J of a tile (1 KiB per wave).  M0 = wave_base + DST (the LDS destination), the lane's
// global offset is its offset in piece 0 with XORC XORed in (the piece's share of the LDS swizzle) plus jstep bytes; the address instruction sits
// in the wait state that the M0 write needs before the request.  M0 is not restored: nothing else here uses it.
template <int DST, int J, int XORC>
__device__ __forceinline__ void a4_dma(a4_u32x4 rs, int voff0, int jstep, uint32_t wave_base) {
    if constexpr (J == 0) {
        asm volatile("s_add_i32 m0, %2, %c3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                     :: "v"(voff0), "s"(rs), "s"(wave_base), "i"(DST) : "memory", "scc");
    } else {
        int voff;
        asm volatile("s_add_i32 m0, %4, %c5\n\tv_xad_u32 %0, %1, %c6, %2\n\tbuffer_load_dwordx4 %0, %3, 0 offen lds"
                     : "=&v"(voff) : "v"(voff0), "s"(jstep), "s"(rs), "s"(wave_base), "i"(DST), "i"(XORC) : "memory", "scc");
    }
}

// a[A .. A+3] = one 16-byte fragment held in vector registers
template <int A>
__device__ __forceinline__ void a4_put4(a4_u32x4 x) {
    asm volatile("v_accvgpr_write_b32 a[%c4], %0\n\tv_accvgpr_write_b32 a[%c5], %1\n\tv_accvgpr_write_b32 a[%c6], %2\n\tv_accvgpr_write_b32 a[%c7], %3"
                 :: "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "i"(A), "i"(A + 1), "i"(A + 2), "i"(A + 3));
}
template <int A>
__device__ __forceinline__ void a4_zero16() {
    asm volatile("v_accvgpr_write_b32 a[%c0], 0" :: "i"(A));
    if constexpr ((A & 15) != 15) a4_zero16<A + 1>();
}
template <int A>
__device__ __forceinline__ float a4_get() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "i"(A));
    return x;
}
// a[A] *= alpha for 16 consecutive registers (a rare path: the running maximum grew by more than the threshold)
template <int A, int N>
__device__ __forceinline__ void a4_scale(float alpha) {
    float t;
    asm volatile("v_accvgpr_read_b32 %0, a[%c2]\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a[%c2], %0"
                 : "=&v"(t) : "v"(alpha), "i"(A));
    if constexpr (N > 1) a4_scale<A + 1, N - 1>(alpha);
}
// K fragment (st, ks) <- LDS
template <int A, int OFF>
__device__ __forceinline__ void a4_read_k(uint32_t lds_addr) {
    asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%c3" :: "v"(lds_addr), "i"(A), "i"(A + 3), "i"(OFF) : "memory");
}
// V fragment half <- LDS (transposed 8-byte read).  The destination is a vector register the compiler allocates, but it
// does not count this read: the MFMA that uses it carries the s_waitcnt with the right count (a4_mfma_o).
template <int OFF>
__device__ __forceinline__ void a4_read_v(bf16x4& dst, uint32_t lds_addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%c2" : "=v"(dst) : "v"(lds_addr), "i"(OFF) : "memory");
}
// S^T tile += K fragment x Q fragment (both in the accumulator half); FIRST: the product starts from zero
template <int KA, int QA, bool FIRST>
__device__ __forceinline__ void a4_mfma_s(f32x16& s) {
    if constexpr (FIRST)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], 0" : "=v"(s) : "i"(KA), "i"(KA + 3), "i"(QA), "i"(QA + 3));
    else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], %0" : "+v"(s) : "i"(KA), "i"(KA + 3), "i"(QA), "i"(QA + 3));
}
// O^T tile += V fragment x P fragment (vector registers); WAIT >= 0: first let all but WAIT LDS reads land
template <int OA, int WAIT>
__device__ __forceinline__ void a4_mfma_o(bf16x8 vf, bf16x8 pf) {
    if constexpr (WAIT >= 0)
        asm volatile("s_waitcnt lgkmcnt(%c4)\n\tv_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]"
                     :: "v"(vf), "v"(pf), "i"(OA), "i"(OA + 15), "i"(WAIT) : "memory");
    else
        asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(vf), "v"(pf), "i"(OA), "i"(OA + 15));
}
// two fp32 -> one dword of two bf16 (round to nearest even), the first in the low half
__device__ __forceinline__ uint32_t a4_pack(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
    bf16x2_ v;
    v[0] = (__bf16)a; v[1] = (__bf16)b;
    return __builtin_bit_cast(uint32_t, v);
}
// The same written as a read-modify-write of the destination register (used on the rare paths, which overwrite P in the
// registers the optimistic pass filled).
__device__ __forceinline__ void a4_pack_into(uint32_t& dst, float a, float b) {
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "+v"(dst) : "v"(a), "v"(b));
}
template <int N>
__device__ __forceinline__ void a4_nop() { asm volatile("s_nop %c0" :: "i"(N - 1)); }

template <int I, int N, typename F>
__device__ __forceinline__ void a4_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        a4_static_for<I + 1, N>(f);
    }
}

