// fp32 products on the bf16 matrix pipe: the pieces shared by mlp_wide.hip (split_nt_kernel, split_tn_kernel) and mlp_res.hip
// (split_bwd_res_kernel).  x = hi + mid + lo EXACTLY with three bf16 pieces; a b = six bf16 x bf16 products accumulated in fp32
// (see the header comment of split_nt_kernel for the arithmetic and its measured error).
#pragma once
#include "pn2_common.h"

typedef __bf16 pn2_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pn2_bf16x8 __attribute__((ext_vector_type(8)));
typedef float pn2_f32x2 __attribute__((ext_vector_type(2)));
union SplitFrag { pn2_bf16x8 v; unsigned u[4]; uint4 q; };

// two fp32 values -> three packed bf16 pairs (v_cvt_pk_bf16_f32, v_and / v_lshl, v_pk_add_f32: nine instructions)
__device__ __forceinline__ void split2(float a, float b, unsigned &hi, unsigned &mid, unsigned &lo) {
    pn2_f32x2 v = {a, b};
    const pn2_bf16x2 h = __builtin_convertvector(v, pn2_bf16x2);
    v = v - __builtin_convertvector(h, pn2_f32x2);
    const pn2_bf16x2 m = __builtin_convertvector(v, pn2_bf16x2);
    v = v - __builtin_convertvector(m, pn2_f32x2);
    const pn2_bf16x2 l = __builtin_convertvector(v, pn2_bf16x2);
    hi = __builtin_bit_cast(unsigned, h); mid = __builtin_bit_cast(unsigned, m); lo = __builtin_bit_cast(unsigned, l);
}


typedef short pn2_s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned tr_img_off(int row, int ch) {      // byte offset of 16-byte chunk ch of row `row` inside a panel
    return (unsigned)(256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))));
}

// tags for code instantiated per (wave-uniform) role: `auto body = [&](auto tag) { constexpr bool X = decltype(tag)::value; ... }`
struct pn2_true { static constexpr bool value = true; };
struct pn2_false { static constexpr bool value = false; };
