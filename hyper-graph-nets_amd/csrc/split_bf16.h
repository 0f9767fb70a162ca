// The three-term bf16 split of fp32 operands (x = hi + mid + lo, each term round-to-nearest-even of the remainder before it), two
// values at a time: v_cvt_pk_bf16_f32 converts a pair, the pair's fp32 images come back with one shift and one mask, and the
// remainders are one v_pk_add_f32.  Nine VALU instructions per pair (4.5 per value; the element-wise form the compiler was given
// before cost 6.7), same bits: the conversions are the same RNE conversions and the subtractions are exact in fp32.
#pragma once
#include <hip/hip_runtime.h>

namespace hgn_split {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void pair(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
  const f2 x = {x0, x1};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, b2));
  const f2 h = {__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
  const f2 r1 = x - h;
  mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, b2));
  const f2 m = {__uint_as_float(mid << 16), __uint_as_float(mid & 0xffff0000u)};
  const f2 r2 = r1 - m;
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, b2));
}

// eight values -> the three operand vectors (element j of each vector belongs to v[j])
__device__ __forceinline__ void eight(const float (&v)[8], b8 (&s)[3]) {
  u4 w[3];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    unsigned a, b, c;
    pair(v[2 * p], v[2 * p + 1], a, b, c);
    w[0][p] = a; w[1][p] = b; w[2][p] = c;
  }
  s[0] = __builtin_bit_cast(b8, w[0]); s[1] = __builtin_bit_cast(b8, w[1]); s[2] = __builtin_bit_cast(b8, w[2]);
}

// ---- two-term fp16 split (product mode 3): hi = rne(x), lo = rne(x - hi), on operands scaled by a power of two so that their largest
// magnitude lies in [2^14, 2^15) -- fp16 keeps 11 significant bits per term (hi + lo: the fp32 value to 2^-24 relative) but only 5
// exponent bits.  Three products hi*hi + hi*lo + lo*hi then carry what six products of three bf16 terms carry.
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
constexpr int SCALE_TOP = 15;       // scaled operands: largest magnitude in [2^14, 2^15) (fp16 overflows at 2^16)
constexpr int SCALE_CLAMP = 120;    // |exponent| of ONE scale (2^+-120 is a normal fp32 number): rows / blocks down to 2^-105 keep full precision.
                                    // A product's accumulators carry the SUM of two exponents (up to +-240): scale_pair splits it when one
                                    // factor cannot hold it.

__device__ __forceinline__ float pow2f(int e) { return __builtin_amdgcn_ldexpf(1.0f, e); }
// x * 2^e for |e| <= 2 SCALE_CLAMP: one multiplication while 2^e is a normal number (|e| <= 120), two beyond (exact either way, barring
// the under- / overflow of x itself).  `e` must be UNIFORM where the branch matters for speed; it is correct for any e.
__device__ __forceinline__ float mul_pow2(float x, int e) {
  const int e1 = e < -SCALE_CLAMP ? -SCALE_CLAMP : (e > SCALE_CLAMP ? SCALE_CLAMP : e);
  return (x * pow2f(e1)) * pow2f(e - e1);
}
// the exponent s with maxabs * 2^s in [2^14, 2^15), clamped (0 / inf / nan: frexp gives exponent 0, s = 15: "magnitude 1")
__device__ __forceinline__ int scale_exp_of(float maxabs) {
  const int s = SCALE_TOP - __builtin_amdgcn_frexp_expf(maxabs);
  return s < -SCALE_CLAMP ? -SCALE_CLAMP : (s > SCALE_CLAMP ? SCALE_CLAMP : s);
}
// the same where several groups SHARE one scale, the minimum of their exponents (csrc/fused_bwd3.hip): an all-zero group -- the
// padding rows of a tail tile, a dead row -- takes the LARGEST exponent, the limit of ever smaller data, so that it never decides
// the minimum (as "magnitude 1" it would crush a block of data at 1e-12 to nothing)
__device__ __forceinline__ int scale_exp_shared(float maxabs) { return maxabs == 0.f ? SCALE_CLAMP : scale_exp_of(maxabs); }
// Accumulators that already hold something are carried at the products' scale 2^T through a block (x 2^T before, x 2^-T after).
// When T is large (tiny operand rows) that something must not overflow: T above SCALE_EASY is capped at acc_room() of what the
// accumulator holds (a wave-uniform test: ordinary data never takes the branch).  Below SCALE_EASY an overflow needs accumulator
// contents of 2^62 = 4.6e18 and more.
constexpr int SCALE_EASY = 64;
__device__ __forceinline__ int acc_room(float acc_maxabs) { return 125 - __builtin_amdgcn_frexp_expf(acc_maxabs); }
__device__ __forceinline__ void pair16(float x0, float x1, unsigned& hi, unsigned& lo) {
  const f2 x = {x0, x1};
  const h2 h = __builtin_convertvector(x, h2);
  const f2 r = x - __builtin_convertvector(h, f2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, h2));
}
// eight values times the power of two `sc` -> the two operand vectors (fp16 bit patterns in bf16x8 containers)
__device__ __forceinline__ void eight16(const float (&v)[8], float sc, b8 (&s)[3]) {
  u4 w[2];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    unsigned a, b;
    pair16(v[2 * p] * sc, v[2 * p + 1] * sc, a, b);
    w[0][p] = a; w[1][p] = b;
  }
  s[0] = __builtin_bit_cast(b8, w[0]); s[1] = __builtin_bit_cast(b8, w[1]);
}

}  // namespace hgn_split
