// "Split rows": the activation format between the encoders and the MLP kernels (include/dns_hip.h, DNS split-row format).
//
// A row of K fp32 values v[0..K) travels as  hi[j] = f16(v[j] 2^e),  lo[j] = f16(v[j] 2^e - hi[j])  -- K halfs of hi followed by
// K halfs of lo (the lo plane is absent in the half-width mode) -- plus ONE integer exponent e per row: exactly the two operand
// parts the MLP kernels used to derive from fp32 rows in every launch (mlp_split.hpp: max |row| -> e = scale_exp(max), split8),
// now produced ONCE by whoever writes the row.  A lane of a 32-point MFMA tile then loads its B-operand fragment (8 consecutive
// k of one point) as one 16-byte global load per part: no LDS staging, no row maximum, no conversion in the consumers.
// The bytes per row are those of the fp32 row (2 x 2 x K); half of that in the half-width mode.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dns {
namespace sr {

constexpr int TGT_EXP = 14;                  // scaled maxima lie in [2^13, 2^14): f16's top binades, clear of 65504

// exponent e such that m 2^e lies in [2^(TGT_EXP-1), 2^TGT_EXP); 0 for m = 0 or a non-finite m (which then propagates)
__device__ __forceinline__ int scale_exp(float m) {
  if (!(m > 0.f) || !(m < INFINITY)) return 0;
  int ex;
  (void)frexpf(m, &ex);                      // m = f 2^ex, f in [0.5, 1)
  return min(max(TGT_EXP - ex, -110), 110);
}

// two scaled values -> (hi pair, lo pair) as packed halfs (low half = the first value)
__device__ __forceinline__ void split_pair(float a, float b, float s, uint32_t& hi, uint32_t& lo) {
  const float xa = a * s, xb = b * s;
  const _Float16 ha = (_Float16)xa, hb = (_Float16)xb;
  const _Float16 la = (_Float16)(xa - (float)ha), lb = (_Float16)(xb - (float)hb);
  typedef _Float16 half2v __attribute__((ext_vector_type(2)));
  half2v h, l;
  h[0] = ha; h[1] = hb;
  l[0] = la; l[1] = lb;
  hi = *reinterpret_cast<uint32_t*>(&h);
  lo = *reinterpret_cast<uint32_t*>(&l);
}

}  // namespace sr
}  // namespace dns
