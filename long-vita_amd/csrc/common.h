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

struct AttnDesc {
    int a_row0;  // fwd: q_row0   bwd: k_row0
    int a_rows;  // fwd: q_rows   bwd: k_rows
    int b_row0;  // fwd: k_row0   bwd: q_row0
    int b_len;   // fwd: k_len    bwd: q_len
    int shift;
    int pad0, pad1, pad2;
};
