// Shared device/host helpers for the gfx950 attention kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

// Swizzle of a [rows][128 x bf16] LDS tile (256-byte rows, sixteen 16-byte chunks):
// chunk c of row r lives at 256*r + 16*(c ^ f(r)).  With f(r) = ((r&3)<<2)|((r>>2)&3)
// both the ds_read_b128 row reads of the 32x32x16 MFMA operand (lane = row) and the
// ds_read_b64_tr_b16 transposed reads (4 rows x 16 columns per 16-lane group) touch
// every bank exactly once per wave-instruction.
__device__ __forceinline__ int swz128(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }
__device__ __forceinline__ int lds_off128(int row, int chunk) { return row * 256 + ((chunk ^ swz128(row)) << 4); }

// Same idea for [rows][64 x bf16] tiles (128-byte rows, eight chunks): two rows share
// one 256-byte bank row (the row's low bit selects the half), so the XOR uses three
// bits taken from row bits 1..3; conflict-free for the same two kinds of read.
__device__ __forceinline__ int swz64(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }
__device__ __forceinline__ int lds_off64(int row, int chunk) { return row * 128 + ((chunk ^ swz64(row)) << 4); }

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Lanes l and l^32 hold the two halves of one query row after a 32x32 MFMA.  One
// v_permlane32_swap (lanes 32-63 of the first register <-> lanes 0-31 of the second)
// gives every lane both halves' values.  Written as inline asm: with hipcc (ROCm 7.2)
// __builtin_amdgcn_permlane32_swap was observed to reuse its first operand for the
// second one.  The s_nop covers the VALU-write -> permlane-read hazard.
__device__ __forceinline__ void pair32(float x, float& a, float& b) {
    a = x;
    b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float pair32_max(float x) { float a, b; pair32(x, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float pair32_sum(float x) { float a, b; pair32(x, a, b); return a + b; }

// ---- bf16 rounding helpers of the rotary embedding (shared by rope.hip and the attention prologue) ----
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

// One rotary pair in the reference's eager bf16 op order (rotary_pos_embedding.py:200-203): x1 / x2 are the
// elements at column c and c + rot/2, (c1, s1) and (c2, s2) the table values at those columns.  Every product and
// each sum is rounded to bf16 on its own.  CONJ = the transposed rotation (the gradient).
template <bool CONJ>
__device__ __forceinline__ void rope_pair(float a, float b, float c1, float s1, float c2, float s2, float& r1, float& r2) {
    const uint32_t p1 = CONJ ? rb2(a * c1, b * s2) : rb2(a * c1, b * s1);
    const uint32_t p2 = CONJ ? rb2(b * c2, a * s1) : rb2(b * c2, a * s2);
    r1 = CONJ ? lo_f32(p1) + hi_f32(p1) : lo_f32(p1) - hi_f32(p1);
    r2 = CONJ ? lo_f32(p2) - hi_f32(p2) : lo_f32(p2) + hi_f32(p2);
}

struct AttnDesc {
    int a_row0;  // fwd: q_row0   bwd: k_row0
    int a_rows;  // fwd: q_rows   bwd: k_rows
    int b_row0;  // fwd: k_row0   bwd: q_row0
    int b_len;   // fwd: k_len    bwd: q_len
    int shift;
    int out_delta;  // results go to row (a_row0 + out_delta + i) of the output tensors instead of row (a_row0 + i): work
                    // items that split one block's sweep write their partial results to workspace rows of their own
    int band_p1;    // 0: plain causal rule.  b + 1: a band as well, key j visible to row i iff i + shift - b <= j <= i + shift
                    // (taken by the ping-pong forward, the ping-pong dQ and the role-split dK/dV kernels, which a launch
                    // with causal == 2 selects)
    int pad2;
};
// lower end of a row's visible keys, as an offset to add to the row index: `i + lv_band_low(dsc)` is the first visible key
// (far below zero without a band)
__device__ __forceinline__ int lv_band_low(const AttnDesc& d) { return d.band_p1 > 0 ? d.shift - (d.band_p1 - 1) : -(1 << 29); }
