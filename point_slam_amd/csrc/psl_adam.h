// Adam arithmetic of the optimiser launches (psl_ray.hip).
#pragma once
#include "psl_common.h"
#include "psl_device.h"

namespace psl {

// torch.optim.Adam (defaults, no weight decay / amsgrad):
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= (lr/(1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// Dense over every element each step (the reference steps all frustum-selected rows, Mapper.py:394-402).
// Written in the operation order of torch's single-tensor path (lerp_, mul_/addcmul_, sqrt/div/add_, addcdiv_).
__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, float lr_bc1, float sqrt_bc2,
                                            float b1, float b2, float eps) {
  m = m + (1.0f - b1) * (g - m);
  v = v * b2 + ((1.0f - b2) * g) * g;
  float denom = sqrtf(v) / sqrt_bc2 + eps;
  p = p + ((-lr_bc1) * m) / denom;
}

// A replayed step of the lazy Adam: the gradient is zero.  m and v are the expressions of adam_update with g = 0, bit for
// bit.  The parameter increment -lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps) uses the hardware reciprocal and square root
// (1 ulp each) instead of the correctly rounded quotient and root: the replay loop is a serial chain per row, and the
// IEEE sequences made one step cost ~1 us on a lone wavefront.  The increment is ~1e-3 of the step size itself
// ~1e-3 |p|; two ulp of it are ~1e-13 |p|, far below the rounding of the sum p + increment.
__device__ __forceinline__ void adam_replay(float& p, float& m, float& v, float lr_bc1, float inv_sqrt_bc2, float b1,
                                            float b2, float eps) {
  m = m + (1.0f - b1) * (0.f - m);
  v = v * b2;
  const float denom = fmaf(__builtin_amdgcn_sqrtf(v), inv_sqrt_bc2, eps);
  p = fmaf((-lr_bc1) * m, __builtin_amdgcn_rcpf(denom), p);
}

// Feature rows, lazily.  torch.optim.Adam steps every selected row in every iteration, also the ones without a gradient
// (m decays, p keeps moving): ~10^5 rows x 2 groups x 7 accesses of 128 B per iteration, 30-50 us of pure HBM time,
// although an iteration reads and writes only the ~2x10^4 rows next to its samples.  A row's update depends on nothing
// but its own (p, g, m, v) and the step's constants, so the steps a row missed are replayed later IN REGISTERS, in order
// (m and v bit-identical to the dense sweep, p to ~1e-13 relative: adam_replay).  A row is brought up to date when
//  (a) this iteration's neighbour lists name it (it may have received a gradient: `touched`), or
//  (b) the next iteration's lists name it (the forward will read it) -- both sets come from the prefetched lists as a
//      de-duplicated work list (k_adam_worklist), or
//  (c) dense pass (list == null): the next lists are not known yet (end of a k-NN prefetch block) or the call ends.
// upto[row] = number of this call's iterations already applied; -1 = never had a gradient (m = v = 0, every missed
// step is exactly +0).  The replay is a serial chain per channel and the slowest row of a launch sets its duration.
// the feature-row part of a lazy Adam launch for workgroup blk0 of nb_rows (group `is_col`).
// LPR lanes per row, 32 / LPR channels per lane (one 4-, 8- or 16-byte access per stream and lane): a 256-thread workgroup
// steps 256 / LPR rows per trip.  The launch is a chain of dependent memory round trips, not a bandwidth problem (45 MB in
// 25 us with one channel per lane): fewer, wider trips shorten the chain, while the replay of the missed steps becomes
// 32 / LPR independent serial chains per lane (and a wavefront waits for the longest replay of 64 / LPR rows).
// Every load that does not need the list's length is requested before the length is known: the first list entry of the
// workgroup (the list is allocated to its capacity; entries past the length are never dereferenced) and the constants table.
template <int LPR>
__device__ __forceinline__ void adam_lazy_rows_block(const AdamRowsSeg& sg, bool is_col, int blk0, int nb_rows, float b1, float b2,
                                                     float eps, const AdamLazy& lz, float2* stab) {
  static_assert(LPR == 8 || LPR == 16 || LPR == 32, "lanes per row");
  constexpr int V = 32 / LPR;                             // channels per lane
  constexpr int RPT = 256 / LPR;                          // rows per workgroup trip
  const int sub = threadIdx.x / LPR, e = (threadIdx.x % LPR) * V;
  int row_first = -1;
  if (lz.list && (long long)blk0 * RPT + sub < lz.list_cap) row_first = lz.list[(long long)blk0 * RPT + sub];
  const int n_work = lz.list ? *lz.count : sg.n_rows;
  // per-iteration constants since the last dense pass, staged once per workgroup (a global load per replayed step
  // made each step a full memory round trip)
  const int nt = lz.it - lz.base + 1;
  for (int t = threadIdx.x; t < nt && t < kAdamTabLds; t += blockDim.x) {
    const float4 v = lz.tab[lz.base + t];
    stab[t] = is_col ? make_float2(v.z, v.w) : make_float2(v.x, v.y);
  }
  __syncthreads();
  auto consts = [&](int t) -> float2 {
    const int k = t - lz.base;
    if (k >= 0 && k < kAdamTabLds) return stab[k];
    const float4 v = lz.tab[t];
    return is_col ? make_float2(v.z, v.w) : make_float2(v.x, v.y);
  };
  unsigned long long done = 0;
  // the grid is a fixed number of workgroups per group: each walks the work list with a stride (a grid sized to the
  // list's CAPACITY -- 10^4 workgroups per group, most of them past its length -- cost more to dispatch than to run)
  for (long long blk = blk0; blk * RPT < n_work; blk += nb_rows) {
    const long long ridx = blk * RPT + sub;
    int row = -1;
    if (ridx < n_work) row = !lz.list ? (int)ridx : (blk == blk0 ? row_first : lz.list[ridx]);
    bool has_g = false, work = false;
    int u = -1;
    if (row >= 0) {
      has_g = sg.touched[row] != 0;
      u = sg.upto[row];
      work = has_g || (u >= 0 && u <= lz.it);
    }
    if (!work) continue;
    if (u < 0) u = lz.it;                                 // first gradient of this row: the missed steps were +0
    float* pptr = sg.feats + (size_t)sg.rows[row] * C + e;
    const size_t k = (size_t)row * C + e;
    float* gp = reinterpret_cast<float*>(sg.g) + k;
    float* mp = reinterpret_cast<float*>(sg.m) + k;
    float* vp = reinterpret_cast<float*>(sg.v) + k;
    using vec = float __attribute__((ext_vector_type(V)));
    vec pp = *reinterpret_cast<vec*>(pptr), mm = *reinterpret_cast<vec*>(mp), vv = *reinterpret_cast<vec*>(vp), gg = 0.f;
    if (has_g) { gg = *reinterpret_cast<vec*>(gp); *reinterpret_cast<vec*>(gp) = 0.f; }
    for (int t = u; t < lz.it; ++t) {                     // replay of the steps without a gradient
      const float2 ab = consts(t);
      const float ib = __builtin_amdgcn_rcpf(ab.y);
#pragma unroll
      for (int c = 0; c < V; ++c) { float p1 = pp[c], m1 = mm[c], v1 = vv[c]; adam_replay(p1, m1, v1, ab.x, ib, b1, b2, eps); pp[c] = p1; mm[c] = m1; vv[c] = v1; }
    }
    const float2 ab = consts(lz.it);
#pragma unroll
    for (int c = 0; c < V; ++c) { float p1 = pp[c], m1 = mm[c], v1 = vv[c]; adam_update(p1, gg[c], m1, v1, ab.x, ab.y, b1, b2, eps); pp[c] = p1; mm[c] = m1; vv[c] = v1; }
    *reinterpret_cast<vec*>(pptr) = pp; *reinterpret_cast<vec*>(mp) = mm; *reinterpret_cast<vec*>(vp) = vv;
    // the lanes of a row sit in one wavefront and have all read touched/upto above
    if (e == 0) { sg.upto[row] = lz.it + 1; if (has_g) sg.touched[row] = 0; ++done; }
  }
  if (lz.rows_done) {                                     // one atomic per wavefront, spread over 256 cache lines
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) done += __shfl_xor(done, o);
    if ((threadIdx.x & 63) == 0 && done) atomicAdd(lz.rows_done + 8 * (blockIdx.x & 255), done);
  }
}

}  // namespace psl
