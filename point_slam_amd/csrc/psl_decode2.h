// Small device helpers shared by the register-chained decode kernels (psl_decode_fwd2.hip, psl_decode_geo.hip).
#pragma once
#include "psl_decode.h"
#include "psl_frag.h"

namespace psl {

__device__ __forceinline__ f32x4 ldfrag(const float* __restrict__ WF, int frag, int lane) {
  return *reinterpret_cast<const f32x4*>(WF + (size_t)frag * FRAG + lane * 4);
}
__device__ __forceinline__ f32x4 ldbias(const float* __restrict__ WF, int layer_bias_off, int nt, int g) {
  return *reinterpret_cast<const f32x4*>(WF + layer_bias_off + nt * 16 + 4 * g);
}
// four k-steps: acc += A-fragment (4 floats) x B registers (4 floats)
__device__ __forceinline__ void mma4(f32x4& acc, const f32x4& a, const f32x4& b) {
  acc = mfma16(a[0], b[0], acc);
  acc = mfma16(a[1], b[1], acc);
  acc = mfma16(a[2], b[2], acc);
  acc = mfma16(a[3], b[3], acc);
}
// nothing is scheduled across this point: pins the hand-written order "request the fragments of step s + D, then issue
// the MFMAs of step s" (left alone, the scheduler hoists every data-independent weight load to the top and spills)
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// sum over the 8 consecutive lanes that hold the 8 neighbours of one sample, as three DPP moves (full-rate VALU; the
// __shfl_xor form goes through ds_bpermute: ~100 cycles of LDS latency per step, 24 steps per wave here)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float group8_sum(float v) {
  v += dpp_mov<0xB1>(v);     // quad_perm [1,0,3,2]: lane ^ 1
  v += dpp_mov<0x4E>(v);     // quad_perm [2,3,0,1]: lane ^ 2
  v += dpp_mov<0x141>(v);    // row_half_mirror: lane i <-> 7 - i inside each group of 8 (both quads hold their own sum)
  return v;
}

// ------------------------------------------------------------------------------------------------ geometry role
// The geometry decoder of one tile as a flat list of "steps" (one k-group = 4 k-steps for both 16-column output tiles):
// fragment indices, which registers feed the B operand, which accumulator pair receives.  The list is walked fully
// unrolled with the weight fragments of step s + GEO_AHEAD requested before the MFMAs of step s (the weights do not
// depend on the data, so the prefetch runs across layer boundaries).
struct GStep { int f0, f1, bsel, dst, layer_end; };   // bsel: 0..5 embedding groups, 6..7 hidden, 8..9 interpolated feature
struct GSteps { GStep s[40]; int n; };
constexpr GSteps make_geo_steps() {
  GSteps g{};
  int n = 0;
  const int FLs[5] = {FL_G0, FL_G1, FL_G2, FL_G3, FL_G4};
  const int FLf[5] = {FL_GF0, FL_GF1, FL_GF2, FL_GF3, FL_GF4};
  for (int i = 0; i < 5; ++i) {
    const int L = FLs[i], nq = kFLayers[L].ngroups, first = ffirst(L);
    for (int q = 0; q < nq; ++q) {
      int bsel = 0;
      if (i == 0) bsel = q;
      else if (i == 3) bsel = q < 6 ? q : 6 + (q - 6);
      else bsel = 6 + q;
      g.s[n++] = GStep{first + q, first + nq + q, bsel, 0, 0};
    }
    const int ff = ffirst(FLf[i]);
    for (int q = 0; q < 2; ++q) g.s[n++] = GStep{ff + q, ff + 2 + q, 8 + q, 1, q == 1 ? i + 1 : 0};
  }
  for (int q = 0; q < 2; ++q) g.s[n++] = GStep{ffirst(FL_GOUT) + q, -1, 6 + q, 2, 0};
  g.n = n;
  return g;
}
constexpr GSteps kGeo = make_geo_steps();


// ------------------------------------------------------------------------------------------------ coalesced gradient scatter
// The backward adds  w_k * dC[sample][32]  (interpolation) or dX[pair][32] (F_theta) to feature-gradient rows chosen by the
// neighbour lists.  Straight from the accumulator layout -- lane (sample, g) holds channels 16 jt + 4 g + r -- one atomic
// instruction touches 16 different rows with four scattered dwords each, and the eight instructions of a neighbour slot
// return to the same 16 lines eight times: phase stamps put 40 % of the one-launch geometry iteration and 15-30 % of the
// colour backward into this scatter.  Transposed through a per-wave LDS tile instead, an instruction covers TWO WHOLE ROWS
// (lanes = 2 x 32 consecutive channels, two 128-byte lines): an eighth of the line operations at the L2 atomic units, the
// same products, the same (unordered) sums.
struct ScatterLds { float dc[TILE * C]; float w[TILE * K]; int row[TILE * K]; };      // 3 KB per wavefront
// Round 4: the loop is software-pipelined by hand.  As one `for` over the 64 pair slots (rounds 2-3) hipcc emitted, per slot,
// [ds_read row -> wait -> branch -> ds_read w, dc -> wait -> atomic -> branch -> byte store]: 128 serial LDS round trips,
// 14 k cycles of the one-launch geometry iteration's 58 k (its phase stamps).  Now a lane owns one channel and the FOUR
// neighbour slots 4 half .. 4 half + 3 of every sample: the rows / weights of a sample's slots are one 16-byte LDS read each,
// four samples (12 reads) are requested together, and the sixteen atomics of the group issue back to back.
// Round 6: a slot used to cost ~17 instructions on the lone wavefront of a geometry tile (64-bit row address, a nested branch for the
// work-list tag): the rows are addressed as a 32-bit BYTE offset from the uniform base (psl_create bounds the capacity to 2^25 rows
// for exactly this) and the tags are written once per list entry by the lanes that hold the lists.
__device__ __forceinline__ void scatter_interp_rows(ScatterLds& L, float* __restrict__ g_feat, unsigned char* __restrict__ touched,
                                                    const f32x4 (&dc)[2], const float (&w)[K], const int (&dst)[K]) {
  const int lane = threadIdx.x & 63, rl = lane & 15, g = lane >> 4;
  *reinterpret_cast<f32x4*>(L.dc + rl * C + 4 * g) = dc[0];
  *reinterpret_cast<f32x4*>(L.dc + rl * C + 16 + 4 * g) = dc[1];
  if (g == 0) {
    *reinterpret_cast<float4*>(L.w + rl * K) = make_float4(w[0], w[1], w[2], w[3]);
    *reinterpret_cast<float4*>(L.w + rl * K + 4) = make_float4(w[4], w[5], w[6], w[7]);
    *reinterpret_cast<int4*>(L.row + rl * K) = make_int4(dst[0], dst[1], dst[2], dst[3]);
    *reinterpret_cast<int4*>(L.row + rl * K + 4) = make_int4(dst[4], dst[5], dst[6], dst[7]);
    if (touched) {
#pragma unroll
      for (int k = 0; k < K; ++k)
        if (dst[k] >= 0) touched[dst[k]] = 1;
    }
  }
  wave_lds_sync();
  const int ch = lane & 31, half = lane >> 5;
  char* __restrict__ base = reinterpret_cast<char*>(g_feat);
  const unsigned ch4 = 4u * (unsigned)ch;
#pragma unroll
  for (int s0 = 0; s0 < TILE; s0 += 4) {
    int4 rows[4];
    float4 ws[4];
    float dv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      rows[j] = *reinterpret_cast<const int4*>(L.row + (s0 + j) * K + 4 * half);
      ws[j] = *reinterpret_cast<const float4*>(L.w + (s0 + j) * K + 4 * half);
      dv[j] = L.dc[(s0 + j) * C + ch];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) pin(dv[j]);      // keeps the reads of dC up here (sunk into the first branch that uses them otherwise)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r4[4] = {rows[j].x, rows[j].y, rows[j].z, rows[j].w};
      const float w4[4] = {ws[j].x, ws[j].y, ws[j].z, ws[j].w};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (r4[k] >= 0) atomic_add_f32(reinterpret_cast<float*>(base + ((unsigned)r4[k] * (4u * C) + ch4)), w4[k] * dv[j]);
    }
  }
}
// per-pair rows: lane (pair rl, g) holds dX[pair][16 jt + 4 g + r]; dst = gradient row of the lane's pair (-1: none).
// The row lookups (a cross-lane read each) and tile reads of four pairs are requested before their atomics.
__device__ __forceinline__ void scatter_pair_rows(float* tile /*[16][32] per-wave LDS*/, float* __restrict__ g_feat, const f32x4 (&dx)[2],
                                                  int dst) {
  const int lane = threadIdx.x & 63, rl = lane & 15, g = lane >> 4;
  *reinterpret_cast<f32x4*>(tile + rl * C + 4 * g) = dx[0];
  *reinterpret_cast<f32x4*>(tile + rl * C + 16 + 4 * g) = dx[1];
  wave_lds_sync();
  const int ch = lane & 31, half = lane >> 5;
#pragma unroll
  for (int j0 = 0; j0 < TILE / 2; j0 += 4) {             // four at a time: eight registers (the pose-gradient instantiation has none to spare)
    int row[4];
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pair = 2 * (j0 + j) + half;
      row[j] = __shfl(dst, pair);                       // lanes 0..15 (g = 0) hold the 16 pairs' rows
      v[j] = tile[pair * C + ch];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (row[j] >= 0) atomic_add_f32(reinterpret_cast<float*>(reinterpret_cast<char*>(g_feat) + ((unsigned)row[j] * (4u * C) + 4u * (unsigned)ch)), v[j]);
  }
  wave_lds_sync();
}

}  // namespace psl
