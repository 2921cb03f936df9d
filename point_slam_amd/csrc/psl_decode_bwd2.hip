// Fused per-sample decode, backward, register-chained form (operand algebra: psl_frag.h; forward: psl_decode_fwd2.hip).
//
// Mirrors autograd through MLP_color / MLP_geometry / get_feature_at_pos (src/conv_onet/models/decoder.py:130-222,
// 341-449): gradients w.r.t. the interpolated features (scatter-added to the neural-point feature rows), w.r.t. the
// sample positions (-> camera pose, tracker) and the per-layer dZ / G tiles that the parameter-gradient GEMM
// (psl_dw.hip) contracts over all samples.
//
// dX^T[in][sample] = sum_out W[out][in] dZ^T[out][sample] is the same MFMA with A = W^T (backward fragments, k walks the
// layer's OUTPUT channels in accumulator order) and B = dZ^T straight from the registers that hold it.  Roles as in
// the forward:
//  * colour role: one 512-thread workgroup per 16-sample tile; wavefront w owns hidden channels 16 w .. 16 w + 15 of
//    every dL/dh tile (one LDS exchange of dZ and one barrier per layer); d/dc is accumulated K-split (each wavefront
//    contracts its own 16 channels, the eight partial tiles meet once at the end); F_theta's backward is again private
//    to the wavefront that owns the 16 (sample, neighbour) rows.
//  * geometry role: one wavefront per tile, registers only.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <type_traits>
#include "psl_decode.h"
#include "psl_frag.h"
#include "psl_decode2.h"

namespace psl {

struct TrunkPlan { int n2, n1; };
TrunkPlan trunk_plan(int tiles);   // psl_decode_fwd2.hip

__device__ __forceinline__ f32x4 ldfragb(const float* __restrict__ WB, int frag, int lane) {
  return *reinterpret_cast<const f32x4*>(WB + (size_t)frag * FRAG + lane * 4);
}
__device__ __forceinline__ void mma4b(f32x4& acc, const f32x4& a, const f32x4& b) {
  acc = mfma16(a[0], b[0], acc);
  acc = mfma16(a[1], b[1], acc);
  acc = mfma16(a[2], b[2], acc);
  acc = mfma16(a[3], b[3], acc);
}
__device__ __forceinline__ void sched_fence_b() { __builtin_amdgcn_sched_barrier(0); }

struct Bwd2Out {
  float* g_geo; float* g_col; const int* row_map;
  float* g_brel;     // [30] accumulated with atomics (pre-zeroed)
  float* g_affine;   // [12] accumulated with atomics (pre-zeroed)
  unsigned char *t_geo, *t_col;   // per compact row: "has received a gradient" (psl_map_iters' lazy Adam), or null
};

constexpr int LD_E2 = 44;   // colour d(embedding) tile [16][40] (PTSG)
constexpr int LD_X2 = 22;   // rel-pos part of F_theta's dX1, per wave [16][20]
// F_theta's backward weight fragments live in LDS (see NbrStage in psl_decode_fwd2.hip for the why): linear2^T (16) then
// linear1^T (32), contiguous at the start of the backward fragment buffer.
constexpr int kNbrFragsB = 48;
constexpr int NBR_WG_B = 256;     // threads of a k_nbr_bwd workgroup (four wavefronts)
static_assert(bfirst(BL_N2) == 0 && bfirst(BL_N1) == 16 && bfirst(BL_C1) == kNbrFragsB, "F_theta fragments lead the backward buffer");
// LDS-DMA copy (glds16, psl_device.h), 1 KiB per wave-instruction.  Wave 2 issues the global stores / atomics of the d(logits)
// set-up (ray outputs, loss slots) before the first barrier and would have to drain them with the copy, so the six waves
// 0, 1, 3..6 carry eight fragments each.
__device__ __forceinline__ void nbr_stage_dma_b(const float* __restrict__ W, float* sW, int wave, int lane) {
  if (wave == 2 || wave == 7) return;
  const int slot = wave < 2 ? wave : wave - 1;
#pragma unroll
  for (int j = 0; j < kNbrFragsB / 6; ++j) glds16(W + ((size_t)(j * 6 + slot) * 64 + lane) * 4, sW + (j * 6 + slot) * FRAG);
}
__device__ __forceinline__ f32x4 ldsfragb(const float* sW, int frag, int lane) {
  return *reinterpret_cast<const f32x4*>(sW + frag * FRAG + lane * 4);
}
// The 16 KiB dz exchange buffers of the trunk are dead once the last layer's barrier has been passed: the per-wave partial
// tiles of dL/dc (sDccP) and, after their reduction, the rel-pos tiles of F_theta's backward (sXe) reuse them, which keeps
// the workgroup at 73 KB -- two per CU -- with the 48 KiB of weight fragments resident.
struct Bwd2Lds {
  static constexpr int oI = 0, oW = 128, oRel = 256, oPts = 640, oHas = 704, oDO = 720, oDB = 784, oAffP = 816,
                       oGW = oAffP + 16 * 12, oDP = oGW + 128, oDcc = oDP + 64, oDZ = oDcc + 2 * FRAG,
                       oDccP = oDZ, oXe = oDZ, oDE = oDZ + 2 * 8 * FRAG,
                       total = oDE + 16 * LD_E2,           // ~6.3 K floats = 25 KB
                       oWn = (total + 3) / 4 * 4, total_nbr = oWn + kNbrFragsB * FRAG;   // 73 KB
};
static_assert(8 * TILE * C <= 2 * 8 * FRAG && 16 * LD_X2 <= TILE * C, "the per-wave scatter / rel-pos tiles fit the dz buffers");

// ------------------------------------------------------------------------------------------------ geometry role
// One wavefront per tile.  d_occ flows for masked samples too (straight-through of the -100 write, Renderer.py:189-190).
template <bool PTSG>
__device__ __forceinline__ void geo_tile_bwd(const DecodeArgs& a, const Bwd2Out& o, const float* __restrict__ WB, int p0,
                                             ScatterLds& sl, const RayFuse* rf = nullptr) {
  const int lane = threadIdx.x & 63, rl = lane & 15, g = lane >> 4;
  const int p = min(p0 + rl, a.P - 1);
  const bool live = p0 + rl < a.P;
  const float* __restrict__ M = a.master;
  const bool featg = (a.flags & PSL_FEAT_GRAD) != 0;
  const SampleGeom sg = sample_geom(a, p);
  int nb[K];
  {
    const int4 i0 = *reinterpret_cast<const int4*>(a.ws.I + (size_t)p * K);
    const int4 i1 = *reinterpret_cast<const int4*>(a.ws.I + (size_t)p * K + 4);
    nb[0] = i0.x; nb[1] = i0.y; nb[2] = i0.z; nb[3] = i0.w; nb[4] = i1.x; nb[5] = i1.y; nb[6] = i1.z; nb[7] = i1.w;
  }
  float w[K];
  {
    const float4 w0 = *reinterpret_cast<const float4*>(a.ws.w + (size_t)p * K), w1 = *reinterpret_cast<const float4*>(a.ws.w + (size_t)p * K + 4);
    w[0] = w0.x; w[1] = w0.y; w[2] = w0.z; w[3] = w0.w; w[4] = w1.x; w[5] = w1.y; w[6] = w1.z; w[7] = w1.w;
  }
  const bool has = live && a.ws.cnt[p] >= a.min_nn;
  float docc = 0.f;
  if (rf && rf->on) {      // ray stage inside this kernel: every lane evaluates the ray of its own sample (outputs and loss
    double u0, u1, u2;     // belong to the colour role)
    docc = live ? ray_cotangent(a, *rf, p, false, u0, u1, u2).w : 0.f;
  } else {
    docc = live ? a.ws.d_raw[(size_t)p * 4 + 3] : 0.f;
  }
  // G = d_occ * w_out (output_linear.weight [1][32]), channel 16 nt + 4 g + r
  f32x4 G[2], dcg[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) G[nt][r] = docc * M[MO(PI_G_OUT) + nt * 16 + 4 * g + r];
  f32x4 dE[6];
  if constexpr (PTSG) {
#pragma unroll
    for (int q = 0; q < 6; ++q) dE[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int i = 4; i >= 0; --i) {
    constexpr int BLs[5] = {BL_G0, BL_G1, BL_G2, BL_G3, BL_G4};
    constexpr int BLf[5] = {BL_GF0, BL_GF1, BL_GF2, BL_GF3, BL_GF4};
    sched_fence_b();
    f32x4 dz[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const f32x4 y = *reinterpret_cast<const f32x4*>(a.ws.g_y + ((size_t)i * a.ws.Ppad + p0 + rl) * HG + nt * 16 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) dz[nt][r] = (y[r] > 0.f) ? G[nt][r] : 0.f;      // ReLU
    }
    // dL/dc += Wc_i^T G   (fc_c.i.weight [32][32])
    const int ff = bfirst(BLf[i]);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int it = 0; it < 2; ++it) mma4b(dcg[it], ldfragb(WB, ff + it * 2 + q, lane), G[q]);
    // dL/d(input of layer i) = W_i^T dz
    const int fb = bfirst(BLs[i]);
    if (i > 0) {
      f32x4 Gn[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int it = 0; it < 2; ++it) mma4b(Gn[it], ldfragb(WB, fb + it * 2 + q, lane), dz[q]);   // hidden tiles come first
      if (PTSG && i == 3) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int it = 0; it < 6; ++it) mma4b(dE[it], ldfragb(WB, fb + (2 + it) * 2 + q, lane), dz[q]);
      }
      G[0] = Gn[0]; G[1] = Gn[1];
    } else if (PTSG) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int it = 0; it < 6; ++it) mma4b(dE[it], ldfragb(WB, fb + it * 2 + q, lane), dz[q]);
    }
  }
  sched_fence_b();
  // ---- scatter w_k * dC into the geometry feature rows (coalesced: psl_decode2.h); dL/dw_k for the pose gradient
  float gw[K];
  if (featg) {
    int dst[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int i = nb[k];
      dst[k] = (i >= 0 && has && w[k] != 0.f) ? (o.row_map ? o.row_map[i] : i) : -1;
    }
    scatter_interp_rows(sl, o.g_geo, o.t_geo, dcg, w, dst);
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    gw[k] = 0.f;
    const int i = nb[k];
    if (i >= 0 && has) {
      if constexpr (PTSG) {
        const float* frow = a.geo_feats + (size_t)i * C + 4 * g;
        const f32x4 f0 = *reinterpret_cast<const f32x4*>(frow), f1 = *reinterpret_cast<const f32x4*>(frow + 16);
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) v += f0[r] * dcg[0][r] + f1[r] * dcg[1][r];
        gw[k] = v;
      }
    }
  }
  if constexpr (PTSG) {
    // (1) interpolation weights: w = a/S, a = [D<=r2]/(D+1e-10), D = |x_k - p|^2   (decoder.py:143-160)
    float px = 0.f, py = 0.f, pz = 0.f;
    float av[K], rx[K], ry[K], rz[K];
    float S1 = 0.f, dot = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      gw[k] += __shfl_xor(gw[k], 16); gw[k] += __shfl_xor(gw[k], 32);       // over the four channel groups
      const float4 q = a.pos[max(nb[k], 0)];
      rx[k] = (nb[k] >= 0) ? __fsub_rn(q.x, sg.x) : 0.f; ry[k] = (nb[k] >= 0) ? __fsub_rn(q.y, sg.y) : 0.f;
      rz[k] = (nb[k] >= 0) ? __fsub_rn(q.z, sg.z) : 0.f;
      const float D = (nb[k] >= 0) ? __fadd_rn(__fadd_rn(__fmul_rn(rx[k], rx[k]), __fmul_rn(ry[k], ry[k])), __fmul_rn(rz[k], rz[k]))
                                   : __int_as_float(0x7F800000);
      av[k] = (D > sg.r2) ? 0.f : 1.0f / (D + 1e-10f);
      S1 += av[k];
      gw[k] = has ? gw[k] : 0.f;
      dot += gw[k] * w[k];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float da = (gw[k] - dot) / fmaxf(S1, 1e-12f);
      const float dD = -da * av[k] * av[k];
      px += -2.f * dD * rx[k]; py += -2.f * dD * ry[k]; pz += -2.f * dD * rz[k];
    }
    // (2) Fourier embedding sin(2 pi p . B) (93 frequencies), this lane's 24 channels
    const float* __restrict__ Bg = M + MO(PI_G_B);
    float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 16 * q + 4 * g + r;
        if (f < EG) {
          const float dy2 = TWO_PI * dE[q][r] * fast_cosf(fourier_phase(sg.x, sg.y, sg.z, Bg, EG, f));
          ax += dy2 * Bg[f]; ay += dy2 * Bg[EG + f]; az += dy2 * Bg[2 * EG + f];
        }
      }
    ax += __shfl_xor(ax, 16); ax += __shfl_xor(ax, 32);
    ay += __shfl_xor(ay, 16); ay += __shfl_xor(ay, 32);
    az += __shfl_xor(az, 16); az += __shfl_xor(az, 32);
    if (g == 0 && live) reinterpret_cast<float4*>(a.ws.dp2)[p] = make_float4(px + ax, py + ay, pz + az, 0.f);
  }
}

// ------------------------------------------------------------------------------------------------ geometry role, pose gradient
// The tracker's instantiation of the geometry role (d/d(sample position): through the 93 Fourier features of layers 0 and 3
// and through the interpolation weights, decoder.py:143-160,175-222).  Round 3 ran it through geo_tile_bwd<true>: 188 VGPRs,
// which put the whole colour-stage kernel at 189 -> ONE 512-thread workgroup per CU.  Same products, same order of every
// sum, but written for a 128-register budget (two colour tiles per CU):
//  * the neighbour lists / weights are loaded AFTER the layer chain (they are only needed by the epilogue);
//  * the 12 embedding fragments of a layer are walked one input tile at a time, the next tile's pair in flight (the
//    compiler otherwise hoists all twelve loads: 48 registers);
//  * the Fourier epilogue consumes dE before the interpolation-weight epilogue builds its per-neighbour arrays.
__device__ __forceinline__ void geo_tile_bwd_ptsg(const DecodeArgs& a, const Bwd2Out& o, const float* __restrict__ WB, int p0,
                                                  ScatterLds& sl, const TrackFuse& tf, float thr_track) {
  const int lane = threadIdx.x & 63, rl = lane & 15, g = lane >> 4;
  const int p = min(p0 + rl, a.P - 1);
  const bool live = p0 + rl < a.P;
  const float* __restrict__ M = a.master;
  const bool featg = (a.flags & PSL_FEAT_GRAD) != 0;
  const bool has = live && a.ws.cnt[p] >= a.min_nn;
  float docc;
  if (tf.on) {     // the tracker's ray stage inside this kernel: the ray of this lane's sample under the launch-wide threshold
    docc = live ? track_cotangent(a, tf, p, thr_track, false).w : 0.f;
  } else {
    docc = live ? a.ws.d_raw[(size_t)p * 4 + 3] : 0.f;
  }
  // G = d_occ * w_out (output_linear.weight [1][32]), channel 16 nt + 4 g + r
  f32x4 G[2], dcg[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) G[nt][r] = docc * M[MO(PI_G_OUT) + nt * 16 + 4 * g + r];
  f32x4 dE[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) dE[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 4; i >= 0; --i) {
    constexpr int BLs[5] = {BL_G0, BL_G1, BL_G2, BL_G3, BL_G4};
    constexpr int BLf[5] = {BL_GF0, BL_GF1, BL_GF2, BL_GF3, BL_GF4};
    sched_fence_b();
    const int ff = bfirst(BLf[i]), fb = bfirst(BLs[i]);
    f32x4 wc[4], wh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wc[j] = ldfragb(WB, ff + j, lane);            // fragment (it, q) at ff + 2 it + q
    if (i > 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) wh[j] = ldfragb(WB, fb + j, lane);          // hidden tiles come first
    }
    f32x4 dz[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const f32x4 y = *reinterpret_cast<const f32x4*>(a.ws.g_y + ((size_t)i * a.ws.Ppad + p0 + rl) * HG + nt * 16 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) dz[nt][r] = (y[r] > 0.f) ? G[nt][r] : 0.f;      // ReLU
    }
    sched_fence_b();
    // dL/dc += Wc_i^T G   (fc_c.i.weight [32][32]); same accumulation order as geo_tile_bwd: q outer, it inner
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int it = 0; it < 2; ++it) mma4b(dcg[it], wc[it * 2 + q], G[q]);
    // dL/d(input of layer i) = W_i^T dz
    if (i > 0) {
      f32x4 Gn[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int it = 0; it < 2; ++it) mma4b(Gn[it], wh[it * 2 + q], dz[q]);
      G[0] = Gn[0]; G[1] = Gn[1];
    }
    if (i == 3 || i == 0) {
      // embedding part of the skip layer (input tiles 2..7) / of layer 0 (tiles 0..5): dE[it] += W^T[tile it] dz, q = 0 then 1
      const int e0 = fb + (i == 3 ? 4 : 0);
      f32x4 wa = ldfragb(WB, e0, lane), wb2 = ldfragb(WB, e0 + 1, lane);
#pragma unroll
      for (int it = 0; it < 6; ++it) {
        sched_fence_b();
        const f32x4 c0 = wa, c1 = wb2;
        if (it < 5) { wa = ldfragb(WB, e0 + (it + 1) * 2, lane); wb2 = ldfragb(WB, e0 + (it + 1) * 2 + 1, lane); }
        mma4b(dE[it], c0, dz[0]);
        mma4b(dE[it], c1, dz[1]);
      }
    }
  }
  sched_fence_b();
  // lane-derived indices again from an opaque copy of the lane id (left to CSE, hipcc carries the sign-extended sample
  // index and 4 g across the chain and spills them)
  int l2 = threadIdx.x;
  asm volatile("" : "+v"(l2));
  const int rl2 = l2 & 15, g2 = (l2 >> 4) & 3;
  const int p2 = min(p0 + rl2, a.P - 1);
  const SampleGeom sg = sample_geom(a, p2);
  // ---- (2) Fourier embedding sin(2 pi p . B) (93 frequencies), this lane's 24 channels: consumes dE
  float px, py, pz;
  {
    const float* __restrict__ Bg = M + MO(PI_G_B);
    float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      sched_fence_b();      // four channels (12 loads of B, four cosines) at a time: unfenced, all 72 loads are hoisted
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 16 * q + 4 * g2 + r;
        if (f < EG) {
          const float dy2 = TWO_PI * dE[q][r] * fast_cosf(fourier_phase(sg.x, sg.y, sg.z, Bg, EG, f));
          ax += dy2 * Bg[f]; ay += dy2 * Bg[EG + f]; az += dy2 * Bg[2 * EG + f];
        }
      }
    }
    ax += __shfl_xor(ax, 16); ax += __shfl_xor(ax, 32);
    ay += __shfl_xor(ay, 16); ay += __shfl_xor(ay, 32);
    az += __shfl_xor(az, 16); az += __shfl_xor(az, 32);
    px = ax; py = ay; pz = az;
  }
  sched_fence_b();
  // ---- neighbour lists and weights (saved by the forward)
  int nb[K];
  {
    const int4 i0 = *reinterpret_cast<const int4*>(a.ws.I + (size_t)p2 * K);
    const int4 i1 = *reinterpret_cast<const int4*>(a.ws.I + (size_t)p2 * K + 4);
    nb[0] = i0.x; nb[1] = i0.y; nb[2] = i0.z; nb[3] = i0.w; nb[4] = i1.x; nb[5] = i1.y; nb[6] = i1.z; nb[7] = i1.w;
  }
  float w[K];
  {
    const float4 w0 = *reinterpret_cast<const float4*>(a.ws.w + (size_t)p2 * K), w1 = *reinterpret_cast<const float4*>(a.ws.w + (size_t)p2 * K + 4);
    w[0] = w0.x; w[1] = w0.y; w[2] = w0.z; w[3] = w0.w; w[4] = w1.x; w[5] = w1.y; w[6] = w1.z; w[7] = w1.w;
  }
  if (featg) {
    int dst[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int i = nb[k];
      dst[k] = (i >= 0 && has && w[k] != 0.f) ? (o.row_map ? o.row_map[i] : i) : -1;
    }
    scatter_interp_rows(sl, o.g_geo, o.t_geo, dcg, w, dst);
  }
  // ---- (1) interpolation weights: w = a/S, a = [D<=r2]/(D+1e-10), D = |x_k - p|^2   (decoder.py:143-160)
  float gw[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    gw[k] = 0.f;
    const int i = nb[k];
    if (i >= 0 && has) {
      const float* frow = a.geo_feats + (size_t)i * C + 4 * g2;
      const f32x4 f0 = *reinterpret_cast<const f32x4*>(frow), f1 = *reinterpret_cast<const f32x4*>(frow + 16);
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) v += f0[r] * dcg[0][r] + f1[r] * dcg[1][r];
      gw[k] = v;
    }
  }
  sched_fence_b();
  {
    float qx = 0.f, qy = 0.f, qz = 0.f;
    float av[K], rx[K], ry[K], rz[K];
    float S1 = 0.f, dot = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      gw[k] += __shfl_xor(gw[k], 16); gw[k] += __shfl_xor(gw[k], 32);       // over the four channel groups
      const float4 q = a.pos[max(nb[k], 0)];
      rx[k] = (nb[k] >= 0) ? __fsub_rn(q.x, sg.x) : 0.f; ry[k] = (nb[k] >= 0) ? __fsub_rn(q.y, sg.y) : 0.f;
      rz[k] = (nb[k] >= 0) ? __fsub_rn(q.z, sg.z) : 0.f;
      const float D = (nb[k] >= 0) ? __fadd_rn(__fadd_rn(__fmul_rn(rx[k], rx[k]), __fmul_rn(ry[k], ry[k])), __fmul_rn(rz[k], rz[k]))
                                   : __int_as_float(0x7F800000);
      av[k] = (D > sg.r2) ? 0.f : 1.0f / (D + 1e-10f);
      S1 += av[k];
      gw[k] = has ? gw[k] : 0.f;
      dot += gw[k] * w[k];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float da = (gw[k] - dot) / fmaxf(S1, 1e-12f);
      const float dD = -da * av[k] * av[k];
      qx += -2.f * dD * rx[k]; qy += -2.f * dD * ry[k]; qz += -2.f * dD * rz[k];
    }
    if (g2 == 0 && live) reinterpret_cast<float4*>(a.ws.dp2)[p2] = make_float4(qx + px, qy + py, qz + pz, 0.f);
  }
}

// ------------------------------------------------------------------------------------------------ colour role
template <bool PTSG>
__device__ __forceinline__ void color_tile_bwd(const DecodeArgs& a, const Bwd2Out& o, const float* __restrict__ WB, float* smem, int p0,
                                               const RayFuse& rf, const TrackFuse& tf, float thr_track) {
  using L = Bwd2Lds;
  int* sI = (int*)(smem + L::oI);           // [16][8]
  float* sW = smem + L::oW;                 // [16][8] normalised weights
  float* sRel = smem + L::oRel;             // [16][8][3]
  float* sPts = smem + L::oPts;             // [16][4]
  int* sHas = (int*)(smem + L::oHas);       // [16]
  float* sDO = smem + L::oDO;               // [16][4] dL/d colour logits (pre-affine)
  float* sDB = smem + L::oDB;               // [32]    dL/dB_rel (30 used)
  float* sAffP = smem + L::oAffP;           // [16][12] per-sample dL/d affine
  float* sGW = smem + L::oGW;               // [16][8]  dL/dw            (PTSG)
  float* sDP = smem + L::oDP;               // [16][4]  dL/dp            (PTSG)
  float* sDcc = smem + L::oDcc;             // [2][64][4] dL/dc_col, fragment order
  float* sDccP = smem + L::oDccP;           // [8][2][64][4] per-wave partial tiles
  float* sDZ = smem + L::oDZ;               // [2][8][64][4] dz tile, fragment order, double buffered
  float* sDE = smem + L::oDE;               // [16][44] dL/d colour embedding (PTSG)
  float* sXe = smem + L::oXe;               // [8][16][22] rel-pos part of F_theta's dX1, per wave
  const int t = threadIdx.x, lane = t & 63, rl = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool relpos = (a.flags & 0x10000) != 0;
  const bool featg = (a.flags & PSL_FEAT_GRAD) != 0;
  const bool parg = (a.flags & PSL_PARAM_GRAD) != 0;
  const float* __restrict__ M = a.master;

  PSL_STAMP(0);
  const float* sWn = smem + L::oWn;
  if (relpos) nbr_stage_dma_b(WB, smem + L::oWn, wave, lane);      // F_theta's backward weights -> LDS; the loads fly during phase 0
  // ---------------------------------------------------------------- phase 0: per-sample state, d(logits)
  if (t < TILE * K) {
    const int s = t >> 3, k = t & 7;
    const int p = min(p0 + s, a.P - 1);
    // list entry, saved weight and count are requested before the sample geometry (see sample_geom) -- in the mapper's
    // instantiation; the pose-gradient one keeps the old order: hoisted, it needs four registers more than its 128
    int i, cnt_p;
    float w_saved;
    SampleGeom sg;
    if constexpr (PTSG) {
      sg = sample_geom(a, p);
      i = a.ws.I[(size_t)p * K + k]; w_saved = a.ws.w[(size_t)p * K + k]; cnt_p = a.ws.cnt[p];
    } else {
      i = a.ws.I[(size_t)p * K + k]; w_saved = a.ws.w[(size_t)p * K + k]; cnt_p = a.ws.cnt[p];
      sg = sample_geom(a, p);
    }
    const float4 q = a.pos[max(i, 0)];
    sI[t] = i;
    sW[t] = w_saved;
    sRel[t * 3 + 0] = (i >= 0) ? __fsub_rn(q.x, sg.x) : 0.f;
    sRel[t * 3 + 1] = (i >= 0) ? __fsub_rn(q.y, sg.y) : 0.f;
    sRel[t * 3 + 2] = (i >= 0) ? __fsub_rn(q.z, sg.z) : 0.f;
    if constexpr (PTSG) sGW[t] = 0.f;
    if (k == 0) {
      sPts[s * 4 + 0] = sg.x; sPts[s * 4 + 1] = sg.y; sPts[s * 4 + 2] = sg.z; sPts[s * 4 + 3] = sg.r2;
      // samples past the end of the batch behave as "no neighbours, zero gradient"
      sHas[s] = (p0 + s < a.P && cnt_p >= a.min_nn) ? 1 : 0;
    }
  } else if (t < TILE * K + TILE) {
    // ---- d(logits): sigmoid and exposure-affine backward (decoder.py:432-448), one thread per sample
    const int s = t - TILE * K;
    const int p = p0 + s;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
    float ag[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) ag[j] = 0.f;
    double lg = 0.0, lc = 0.0, lcnt = 0.0;
    if (p < a.P) {
      // psl_map_iters without per-frame exposure: the ray stage (compositing, loss, compositing backward) of this sample's ray
      // runs here instead of in a launch of its own between the two decode kernels
      float4 dr;
      if (PTSG && tf.on) dr = track_cotangent(a, tf, p, thr_track, true);
      else dr = rf.on ? ray_cotangent(a, rf, p, true, lg, lc, lcnt) : reinterpret_cast<const float4*>(a.ws.d_raw)[p];
      const float4 rw = reinterpret_cast<const float4*>(a.ws.raw)[p];
      d0 = dr.x; d1 = dr.y; d2 = dr.z;
      if (!(a.flags & PSL_NO_SIGMOID)) { d0 *= rw.x * (1.f - rw.x); d1 *= rw.y * (1.f - rw.y); d2 *= rw.z * (1.f - rw.z); }
      if (a.flags & PSL_HAS_AFFINE) {
        const float* A = a.affine;
        const float o0 = a.ws.out3[(size_t)p * 4], o1 = a.ws.out3[(size_t)p * 4 + 1], o2 = a.ws.out3[(size_t)p * 4 + 2];
        // out' = out @ A + t : dA[i][j] = out_i d_j ; dt_j = d_j ; d out_i = sum_j A[i][j] d_j
        ag[0] = o0 * d0; ag[1] = o0 * d1; ag[2] = o0 * d2; ag[3] = o1 * d0; ag[4] = o1 * d1; ag[5] = o1 * d2;
        ag[6] = o2 * d0; ag[7] = o2 * d1; ag[8] = o2 * d2; ag[9] = d0; ag[10] = d1; ag[11] = d2;
        const float e0 = A[0] * d0 + A[1] * d1 + A[2] * d2;
        const float e1 = A[3] * d0 + A[4] * d1 + A[5] * d2;
        const float e2 = A[6] * d0 + A[7] * d1 + A[8] * d2;
        d0 = e0; d1 = e1; d2 = e2;
      }
      if (a.ws.d_out3) reinterpret_cast<float4*>(a.ws.d_out3)[p] = make_float4(d0, d1, d2, 0.f);
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) sAffP[s * 12 + j] = ag[j];
    sDO[s * 4] = d0; sDO[s * 4 + 1] = d1; sDO[s * 4 + 2] = d2; sDO[s * 4 + 3] = 0.f;
    if (rf.on) {     // the tile's loss terms (owners of at most four rays among these 16 lanes) -> one slot of the iteration
#pragma unroll
      for (int ofs = 8; ofs > 0; ofs >>= 1) { lg += __shfl_xor(lg, ofs); lc += __shfl_xor(lc, ofs); lcnt += __shfl_xor(lcnt, ofs); }
      if (s == 0 && lcnt != 0.0) {
        double* acc = rf.loss_acc + 4 * (blockIdx.x & (kLossSlots - 1));
        atomicAdd(acc + 0, lg); atomicAdd(acc + 1, lc); atomicAdd(acc + 2, lcnt);
      }
    }
  } else if (t < TILE * K + TILE + 32) {
    sDB[t - TILE * K - TILE] = 0.f;
  } else if (PTSG && t < TILE * K + TILE + 32 + 64) {
    sDP[t - TILE * K - TILE - 32] = 0.f;
  }
  if (relpos && wave != 2 && wave != 7) wait_dma();     // phases of waves 0, 1, 3..6 issue no global store before this point
  // Weight fragments of the trunk's first layer (i = 4), requested in front of the barrier; during the layers a slot is refilled
  // with layer i - 1's fragment as soon as its MFMAs have issued (mapper instantiation; the pose-gradient one has no registers
  // to hold them across a layer boundary and keeps loading at the layer's start: "pre" 2-4 k cycles per layer in its stamps)
  f32x4 wq[8];
  if constexpr (!PTSG) {
#pragma unroll
    for (int q = 0; q < 8; ++q) wq[q] = ldfragb(WB, bfirst(BL_C4) + wave * 8 + q, lane);
  }
  lds_barrier();
  PSL_STAMP(1);

  // ---------------------------------------------------------------- colour trunk, wave w = hidden channel tile w
  {
    const int nt = wave;
    // G = d_out3 * W_out  (output_linear.weight [3][128]), channel 16 nt + 4 g + r
    f32x4 G;
    {
      const float d0 = sDO[rl * 4], d1 = sDO[rl * 4 + 1], d2 = sDO[rl * 4 + 2];
      const float* wo = M + MO(PI_C_OUT) + nt * 16 + 4 * g;
#pragma unroll
      for (int r = 0; r < 4; ++r) G[r] = d0 * wo[r] + d1 * wo[HC + r] + d2 * wo[2 * HC + r];
    }
    f32x4 dccp[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 dEc = {0.f, 0.f, 0.f, 0.f};       // waves 0..2: dL/d(colour embedding) tile (PTSG)
    auto ld_y = [&](int i) {
      return *reinterpret_cast<const f32x4*>(a.ws.c_y + ((size_t)i * a.ws.Ppad + p0 + rl) * HC + nt * 16 + 4 * g);
    };
    f32x4 ynext = ld_y(4);
    // fc_c fragments of the NEXT layer in flight during the current one: dL/dc is the first product of a layer and its
    // weights were the one load whose L2 latency stood exposed at every layer start (phase stamps: "pre" 3-5 k cycles)
    f32x4 wcn[2] = {ldfragb(WB, bfirst(BL_CF4) + 0 * 8 + nt, lane), ldfragb(WB, bfirst(BL_CF4) + 1 * 8 + nt, lane)};
    auto layer = [&](auto I_) {
      constexpr int i = decltype(I_)::value;
      constexpr int BLs[5] = {BL_C0, BL_C1, BL_C2, BL_C3, BL_C4};
      constexpr int BLf[5] = {BL_CF0, BL_CF1, BL_CF2, BL_CF3, BL_CF4};
      sched_fence_b();
      // this layer's weight fragments: hidden-part tile nt of W_i^T (8 groups), the two fc_c tiles for group nt
      f32x4 wc[2];
      const int fb = bfirst(BLs[i]);
      if (PTSG && i > 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) wq[q] = ldfragb(WB, fb + nt * 8 + q, lane);
      }
      wc[0] = wcn[0]; wc[1] = wcn[1];
      if (i > 0) {
        wcn[0] = ldfragb(WB, bfirst(BLf[i > 0 ? i - 1 : 0]) + 0 * 8 + nt, lane);
        wcn[1] = ldfragb(WB, bfirst(BLf[i > 0 ? i - 1 : 0]) + 1 * 8 + nt, lane);
      }
      const f32x4 y = ynext;
      if (i > 0) ynext = ld_y(i > 0 ? i - 1 : 0);
      sched_fence_b();
      // step A: dz = G * act'(y)
      f32x4 dz;
#pragma unroll
      for (int r = 0; r < 4; ++r) dz[r] = G[r] * softplus100_grad_from_out(y[r]);
      if (parg) {
        const size_t o_ = ((size_t)i * a.ws.Ppad + p0 + rl) * HC + nt * 16 + 4 * g;
        *reinterpret_cast<f32x4*>(a.ws.c_dz + o_) = dz;
        *reinterpret_cast<f32x4*>(a.ws.c_g + o_) = G;
      }
      float* buf = sDZ + (i & 1) * 8 * FRAG;
      *reinterpret_cast<f32x4*>(buf + nt * FRAG + lane * 4) = dz;
      // step B (K-split): dL/dc partial += Wc_i^T[:, own 16 channels] G
      mma4b(dccp[0], wc[0], G);
      mma4b(dccp[1], wc[1], G);
      PSL_STAMP(2 + 3 * (4 - i));
      lds_barrier();
      PSL_STAMP(3 + 3 * (4 - i));
      // step C: dL/d(input of layer i) = W_i^T dz, hidden part (this wave's 16 channels), all 128 dz channels
      if (i > 0) {
        f32x4 ga = {0.f, 0.f, 0.f, 0.f}, gb = {0.f, 0.f, 0.f, 0.f};
        f32x4 z0 = *reinterpret_cast<const f32x4*>(buf + lane * 4), z1 = *reinterpret_cast<const f32x4*>(buf + FRAG + lane * 4);
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
          sched_fence_b();
          const f32x4 c0 = z0, c1 = z1;
          if (q < 6) {
            z0 = *reinterpret_cast<const f32x4*>(buf + (q + 2) * FRAG + lane * 4);
            z1 = *reinterpret_cast<const f32x4*>(buf + (q + 3) * FRAG + lane * 4);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) { ga = mfma16(wq[q][r], c0[r], ga); gb = mfma16(wq[q + 1][r], c1[r], gb); }
          if (!PTSG && i > 1) {
            sched_fence_b();
            const int fn = bfirst(BLs[i > 1 ? i - 1 : 1]);
            wq[q] = ldfragb(WB, fn + nt * 8 + q, lane); wq[q + 1] = ldfragb(WB, fn + nt * 8 + q + 1, lane);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) G[r] = ga[r] + gb[r];
      }
      PSL_STAMP(4 + 3 * (4 - i));
      if (PTSG && (i == 3 || i == 0) && wave < 3) {   // embedding part: input tiles 8..10 of the skip layer, 0..2 of layer 0
        const int tile0 = (i == 3) ? 8 : 0;
#pragma unroll
        for (int q = 0; q < 8; ++q)
          mma4b(dEc, ldfragb(WB, fb + (tile0 + wave) * 8 + q, lane), *reinterpret_cast<const f32x4*>(buf + q * FRAG + lane * 4));
      }
    };
    layer(std::integral_constant<int, 4>{});
    layer(std::integral_constant<int, 3>{});
    layer(std::integral_constant<int, 2>{});
    layer(std::integral_constant<int, 1>{});
    layer(std::integral_constant<int, 0>{});
    // the eight K-split partial tiles of dL/dc meet in LDS (in the dz buffers: every wave is past the last layer's barrier,
    // and only the pose-gradient instantiation still reads dz after it -- that one synchronises first)
    if constexpr (PTSG) lds_barrier();
    *reinterpret_cast<f32x4*>(sDccP + (wave * 2 + 0) * FRAG + lane * 4) = dccp[0];
    *reinterpret_cast<f32x4*>(sDccP + (wave * 2 + 1) * FRAG + lane * 4) = dccp[1];
    if (PTSG && wave < 3) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int e = 16 * wave + 4 * g + r; if (e < EC) sDE[rl * LD_E2 + e] = dEc[r]; }
    }
  }
  PSL_STAMP(17);
  // F_theta's saved hidden activations of this wave's rows (HBM: the forward wrote them): requested in front of the two barriers
  // of the dL/dc reduction instead of behind them (the activation step waited 6-10 k cycles for them, phase stamps r06s)
  f32x4 h1v[8];
  if (relpos) {
    const size_t grow0 = (size_t)p0 * K + 16 * wave + rl;
#pragma unroll
    for (int it = 0; it < 8; ++it) h1v[it] = *reinterpret_cast<const f32x4*>(a.ws.n_h1 + grow0 * HC + it * 16 + 4 * g);
  }
  lds_barrier();
  {   // 512 threads, 512 elements [it][lane][r]: sum over the waves, mask samples without neighbours
    const int e = t;
    float v = 0.f;
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) v += sDccP[w8 * 2 * FRAG + e];
    const int s = (e >> 2) & 15;
    sDcc[e] = sHas[s] ? v : 0.f;
  }
  lds_barrier();

  PSL_STAMP(18);
  // ---------------------------------------------------------------- colour features: scatter / F_theta backward
  {
    const int row = 16 * wave + rl;            // (sample, neighbour) pair of this lane
    const int s = row >> 3;
    const int i = sI[row];
    const float wgt = sW[row];
    const bool has = sHas[s] != 0;
    const size_t grow = (size_t)p0 * K + row;
    const bool live = (p0 + s) < a.P;
    f32x4 dc[2];     // dL/dc_col of this row's sample, channels 16 jt + 4 g + r
    dc[0] = *reinterpret_cast<const f32x4*>(sDcc + (g * 16 + s) * 4);
    dc[1] = *reinterpret_cast<const f32x4*>(sDcc + FRAG + (g * 16 + s) * 4);
    int dst = -1;
    if (i >= 0 && has && wgt != 0.f) dst = o.row_map ? o.row_map[i] : i;
    if (featg && dst >= 0 && o.t_col && g == 0) o.t_col[dst] = 1;
    if (!relpos) {
      // ---- plain interpolation: scatter w_k * dC into the colour feature rows, collect dL/dw_k
      // coalesced as in the F_theta path (psl_decode2.h): the wave's 16 pair rows go through its slice of the dead dz
      // buffers and leave as two whole 128-byte rows per atomic instruction.  Rounds 2-3 issued eight scalar atomics per lane
      // here, each instruction touching 16 rows with four scattered dwords -- the TUM / ScanNet mapper backward (no
      // per-neighbour MLP) ran at 18 % of peak at 50 000 samples where the F_theta variant reaches 28 %.
      if constexpr (!PTSG) {
        if (featg) {
          f32x4 dxf[2];
#pragma unroll
          for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dxf[jt][r] = wgt * dc[jt][r];
          scatter_pair_rows(sXe + wave * TILE * C, o.g_col, dxf, dst);
        }
      } else if (featg && dst >= 0) {   // pose AND feature gradients in one call (drop-in callers only; the tracker trains no
#pragma unroll                         // features): the scalar form keeps this instantiation inside its 128 registers
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
          for (int r = 0; r < 4; ++r) atomic_add_f32(&o.g_col[(size_t)dst * C + jt * 16 + 4 * g + r], wgt * dc[jt][r]);
      }
      if constexpr (PTSG) {
        float v = 0.f;
        if (i >= 0 && has) {
          const float* frow = a.col_feats + (size_t)i * C + 4 * g;
          const f32x4 f0 = *reinterpret_cast<const f32x4*>(frow), f1 = *reinterpret_cast<const f32x4*>(frow + 16);
#pragma unroll
          for (int r = 0; r < 4; ++r) v += f0[r] * dc[0][r] + f1[r] * dc[1][r];
        }
        v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        if (g == 0) sGW[row] = v;
      }
    } else {
      // ---- F_theta backward, rows private to this wave.  d_nf[row][ch] = w[row] * dC[s][ch]
      f32x4 dnf[2];
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dnf[jt][r] = wgt * dc[jt][r];
        if (parg) *reinterpret_cast<f32x4*>(a.ws.n_dnf + grow * C + jt * 16 + 4 * g) = dnf[jt];
      }
      if constexpr (PTSG) {   // dL/dw[s][k] = sum_ch nf[row][ch] dC[s][ch]
        float v = 0.f;
        if (live) {
          const f32x4 n0 = *reinterpret_cast<const f32x4*>(a.ws.n_out + grow * C + 4 * g);
          const f32x4 n1 = *reinterpret_cast<const f32x4*>(a.ws.n_out + grow * C + 16 + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) v += n0[r] * dc[0][r] + n1[r] * dc[1][r];
        }
        v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        if (g == 0) sGW[row] = v;
      }
      // dH1^T[hid][row] = W2^T d_nf^T (linear2.weight [32][128]): 8 hidden tiles x 2 k-groups, walked as four steps of
      // four tiles; the fragments of step + 1 and -- from the start -- the saved hidden activations h1 of this wave's
      // rows (HBM: written by the forward kernel) are in flight while the MFMAs of a step issue.
      f32x4 dh[8];
      constexpr int b2 = bfirst(BL_N2);
      constexpr int b1 = bfirst(BL_N1);
      f32x4 wn[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) wn[j] = ldsfragb(sWn, b2 + j * 2 + 0, lane);
#pragma unroll
      for (int it = 0; it < 8; ++it) dh[it] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < 4; ++st) {       // st = 2 * half + q
        sched_fence_b();
        const int half = st >> 1, q = st & 1;
        f32x4 wc4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wc4[j] = wn[j];
        if (st < 3) {
          const int h2 = (st + 1) >> 1, q2 = (st + 1) & 1;
#pragma unroll
          for (int j = 0; j < 4; ++j) wn[j] = ldsfragb(sWn, b2 + (4 * h2 + j) * 2 + q2, lane);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) wn[j] = ldsfragb(sWn, b1 + j * 8 + 0, lane);       // first step of the next product
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int j = 0; j < 4; ++j) dh[4 * half + j] = mfma16(wc4[j][r], dnf[q][r], dh[4 * half + j]);
      }
      PSL_STAMP(19);
      // dz1 = dH1 * softplus'(h1)
#pragma unroll
      for (int it = 0; it < 8; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dh[it][r] = live ? dh[it][r] * softplus100_grad_from_out(h1v[it][r]) : 0.f;
        if (parg) *reinterpret_cast<f32x4*>(a.ws.n_dz1 + grow * HC + it * 16 + 4 * g) = dh[it];
      }
      PSL_STAMP(20);
      // dX1^T[x][row] = W1^T dz1^T (linear1.weight [128][52]): input tiles (feat 0..15, feat 16..31, rel 0..15, rel 16..19),
      // eight k-groups; the four fragments of group q + 1 are in flight during group q
      f32x4 dx[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) dx[it] = f32x4{0.f, 0.f, 0.f, 0.f};
      const bool need_rel = parg || PTSG;
      // sin / cos of the wave's 160 (pair, frequency) entries as the forward saved them: requested before the dX product
      // (three dependent global loads used to sit between the MFMAs and the rel-pos contraction)
      float psn[3], pcs[3];
      if (need_rel && a.ws.n_x) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int e = lane + 64 * u, r2 = min(e, 16 * ERF - 1) / ERF, f = min(e, 16 * ERF - 1) - r2 * ERF;
          const float* xr = a.ws.n_x + ((size_t)p0 * K + 16 * wave + r2) * NX;
          psn[u] = xr[f]; pcs[u] = xr[ERF + f];
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        sched_fence_b();
        f32x4 wf4[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) wf4[it] = wn[it];
        if (q < 7) {
#pragma unroll
          for (int it = 0; it < 4; ++it) wn[it] = ldsfragb(sWn, b1 + it * 8 + q + 1, lane);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dx[0] = mfma16(wf4[0][r], dh[q][r], dx[0]);
          dx[1] = mfma16(wf4[1][r], dh[q][r], dx[1]);
          if (need_rel) { dx[2] = mfma16(wf4[2][r], dh[q][r], dx[2]); dx[3] = mfma16(wf4[3][r], dh[q][r], dx[3]); }
        }
      }
      PSL_STAMP(21);
      // feature part -> the colour feature rows, two whole rows per atomic instruction (psl_decode2.h); the wave's 2 KiB slice
      // of the (dead) dz buffers is the transpose tile, the rel-pos tile below reuses it
      if (featg) { const f32x4 dxf[2] = {dx[0], dx[1]}; scatter_pair_rows(sXe + wave * TILE * C, o.g_col, dxf, dst); }
      // rel-pos embedding part: y_f = 2pi rel . B[:,f]; e = [sin y, cos y].  Step 1: dL/dy for the 160 (row, frequency)
      // pairs of this wave, in place in its LDS tile.  Step 2: the two small contractions over them --
      // dB_rel[a][f] = sum_rows dy[row][f] rel[row][a] (30 lanes) and dp[s][a] -= sum_{rows of s, f} dy[row][f] B[a][f]
      // (6 lanes) -- so that a wave ends in 30 (+6) LDS atomics instead of 480 (+480).
      if (need_rel) {
        float* xw = sXe + wave * TILE * C;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          xw[rl * LD_X2 + 4 * g + r] = dx[2][r];
          if (g == 0) xw[rl * LD_X2 + 16 + r] = dx[3][r];
        }
        wave_lds_sync();
        const float* Brel = M + MO(PI_C_BREL);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int e = lane + 64 * u;
          if (e >= 16 * ERF) break;
          const int r2 = e / ERF, f = e - r2 * ERF;
          const int row2 = 16 * wave + r2, s2 = row2 >> 3;
          const bool lv = (p0 + s2) < a.P && sI[row2] >= 0;
          float sn, cs;
          if (a.ws.n_x) {     // the forward pass saved [sin | cos] in the first 20 columns of F_theta's input
            sn = psn[u]; cs = pcs[u];
          } else {
            fast_sincosf(fourier_phase(sRel[row2 * 3], sRel[row2 * 3 + 1], sRel[row2 * 3 + 2], Brel, ERF, f), sn, cs);
          }
          const float dy2 = TWO_PI * (xw[r2 * LD_X2 + f] * cs - xw[r2 * LD_X2 + ERF + f] * sn);
          xw[r2 * LD_X2 + f] = lv ? dy2 : 0.f;        // each pair is read and rewritten by its own lane only
        }
        wave_lds_sync();
        if (parg && lane < 3 * ERF) {
          const int ax = lane / ERF, f = lane - ax * ERF;
          float v = 0.f;
#pragma unroll
          for (int r2 = 0; r2 < 16; ++r2) v += xw[r2 * LD_X2 + f] * sRel[(16 * wave + r2) * 3 + ax];
          atomic_add_f32(&sDB[ax * ERF + f], v);
        }
        if constexpr (PTSG) {   // rel = x_k - p  =>  dp -= d_rel
          if (lane >= 32 && lane < 38) {
            const int sl = (lane - 32) / 3, ax = (lane - 32) - 3 * sl;
            float v = 0.f;
            for (int r2 = 8 * sl; r2 < 8 * sl + 8; ++r2)
#pragma unroll
              for (int f = 0; f < ERF; ++f) v += xw[r2 * LD_X2 + f] * Brel[ax * ERF + f];
            atomic_add_f32(&sDP[(2 * wave + sl) * 4 + ax], -v);
          }
        }
      }
    }
  }
  PSL_STAMP(22);
  lds_barrier();
  PSL_STAMP(23);

  // ---------------------------------------------------------------- position gradient of the colour branch (tracker)
  if constexpr (PTSG) {
    // (1) interpolation weights: w = a/S, a = [D<=r2]/(D+1e-10), D = |x_k - p|^2   (decoder.py:143-160)
    if (t < TILE * K) {
      const int s = t >> 3;
      const float rx = sRel[t * 3], ry = sRel[t * 3 + 1], rz = sRel[t * 3 + 2];
      const float D = (sI[t] >= 0) ? __fadd_rn(__fadd_rn(__fmul_rn(rx, rx), __fmul_rn(ry, ry)), __fmul_rn(rz, rz))
                                   : __int_as_float(0x7F800000);
      const float av = (D > sPts[s * 4 + 3]) ? 0.f : 1.0f / (D + 1e-10f);
      float S1 = av;
      S1 += __shfl_xor(S1, 1); S1 += __shfl_xor(S1, 2); S1 += __shfl_xor(S1, 4);
      const float gw = sHas[s] ? sGW[t] : 0.f;
      float dot = gw * sW[t];
      dot += __shfl_xor(dot, 1); dot += __shfl_xor(dot, 2); dot += __shfl_xor(dot, 4);
      const float da = (gw - dot) / fmaxf(S1, 1e-12f);
      const float dD = -da * av * av;                 // a = 1/(D+eps) -> da/dD = -a^2 ; masked slots: a = 0
      float px = -2.f * dD * rx, py = -2.f * dD * ry, pz = -2.f * dD * rz;      // dD/dp = -2 (x_k - p)
      px += __shfl_xor(px, 1); px += __shfl_xor(px, 2); px += __shfl_xor(px, 4);
      py += __shfl_xor(py, 1); py += __shfl_xor(py, 2); py += __shfl_xor(py, 4);
      pz += __shfl_xor(pz, 1); pz += __shfl_xor(pz, 2); pz += __shfl_xor(pz, 4);
      if ((t & 7) == 0) { atomic_add_f32(&sDP[s * 4], px); atomic_add_f32(&sDP[s * 4 + 1], py); atomic_add_f32(&sDP[s * 4 + 2], pz); }
    } else if (t < TILE * K + TILE * ECF) {
      // (2) colour Fourier embedding [sin, cos] (20 + 20): one (sample, frequency) pair per thread
      const int e = t - TILE * K;
      const int s = e / ECF, f = e - s * ECF;
      float sn, cs;
      fast_sincosf(fourier_phase(sPts[s * 4], sPts[s * 4 + 1], sPts[s * 4 + 2], a.Bcol, ECF, f), sn, cs);
      const float dy2 = TWO_PI * (sDE[s * LD_E2 + f] * cs - sDE[s * LD_E2 + ECF + f] * sn);
      atomic_add_f32(&sDP[s * 4], dy2 * a.Bcol[f]); atomic_add_f32(&sDP[s * 4 + 1], dy2 * a.Bcol[ECF + f]);
      atomic_add_f32(&sDP[s * 4 + 2], dy2 * a.Bcol[2 * ECF + f]);
    }
    lds_barrier();
    if (t < TILE && p0 + t < a.P)
      reinterpret_cast<float4*>(a.ws.dp)[p0 + t] = make_float4(sDP[t * 4], sDP[t * 4 + 1], sDP[t * 4 + 2], 0.f);
  }
  // tile-level reductions that go out with a handful of global atomics
  if (parg && relpos && t < 3 * ERF && o.g_brel) atomic_add_f32(&o.g_brel[t], sDB[t]);
  if ((a.flags & PSL_HAS_AFFINE) && t < 12 && o.g_affine) {
    float v = 0.f;
#pragma unroll
    for (int s = 0; s < TILE; ++s) v += sAffP[s * 12 + t];
    atomic_add_f32(&o.g_affine[t], v);
  }
}

// ================================================================================================ split colour stage (round 6)
// See psl_decode_fwd2.hip (k_nbr_fwd / k_trunk_fwd) for the why.  Backward of the split:
//   k_trunk_bwd -- per 16-sample tile: d(logits) incl. the ray stage (RayFuse / TrackFuse), the five trunk layers, the K-split
//                  dL/dc reduction; leaves dL/dc rows in `dcc` (unmasked) and, when the ray stage ran here, d_raw for the geometry role.
//   k_nbr_bwd   -- F_theta's backward and the feature scatter in units of one wavefront (16 pairs), four per workgroup, with
//                  the one-wave geometry tiles and the lazy Adam's work-list blocks in the same grid.
template <int MT> struct TrunkBLdsT {
  static constexpr int oDO = 0, oAffP = MT * 64, oDP = oAffP + MT * 16 * 12, oDZ = MT * 320, oDccP = oDZ, oDE = oDZ + 2 * MT * 8 * FRAG,
                       total = oDE + MT * 16 * LD_E2;      // 19 KB per 16-sample sub-tile
};
using TrunkBLds = TrunkBLdsT<1>;

// one tile of MT x 16 samples (see trunk_tile_fwd for the double tile)
template <bool PTSG, int MT>
__device__ __forceinline__ void trunk_tile_bwd(const DecodeArgs& a, const Bwd2Out& o, const float* __restrict__ WB, float* smem, int p0,
                                               const RayFuse& rf, const TrackFuse& tf, float thr_track) {
  using L = TrunkBLdsT<MT>;
  float* sDO = smem + L::oDO;               // [MT 16][4] dL/d colour logits (pre-affine)
  float* sAffP = smem + L::oAffP;           // [MT 16][12] per-sample dL/d affine
  float* sDP = smem + L::oDP;               // [MT 16][4]  dL/dp            (PTSG)
  float* sDccP = smem + L::oDccP;           // [8][MT][2][64][4] per-wave partial tiles
  float* sDZ = smem + L::oDZ;               // [2][MT][8][64][4] dz tiles, fragment order, double buffered
  float* sDE = smem + L::oDE;               // [MT 16][44] dL/d colour embedding (PTSG)
  const int t = threadIdx.x, lane = t & 63, rl = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nt = wave;
  const bool parg = (a.flags & PSL_PARAM_GRAD) != 0;
  const float* __restrict__ M = a.master;
  PSL_STAMP(0);
  // weights of the first layer (i = 4) and the first saved activations: in flight during the d(logits) set-up
  f32x4 wq[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) wq[q] = ldfragb(WB, bfirst(BL_C4) + nt * 8 + q, lane);
  auto ld_y = [&](int i, int m) {
    return *reinterpret_cast<const f32x4*>(a.ws.c_y + ((size_t)i * a.ws.Ppad + p0 + m * TILE + rl) * HC + nt * 16 + 4 * g);
  };
  f32x4 ynext[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) ynext[m] = ld_y(4, m);
  f32x4 wcn[2] = {ldfragb(WB, bfirst(BL_CF4) + 0 * 8 + nt, lane), ldfragb(WB, bfirst(BL_CF4) + 1 * 8 + nt, lane)};
  // ---------------------------------------------------------------- d(logits): sigmoid and exposure-affine backward (decoder.py:432-448)
  if (t < MT * TILE) {
    const int s = t;
    const int p = p0 + s;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
    float ag[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) ag[j] = 0.f;
    double lg = 0.0, lc = 0.0, lcnt = 0.0;
    if (p < a.P) {
      float4 dr;
      if (PTSG && tf.on) dr = track_cotangent(a, tf, p, thr_track, true);
      else dr = rf.on ? ray_cotangent(a, rf, p, true, lg, lc, lcnt) : reinterpret_cast<const float4*>(a.ws.d_raw)[p];
      if ((PTSG && tf.on) || rf.on) reinterpret_cast<float4*>(a.ws.d_raw)[p] = dr;     // .w: the geometry role's cotangent (k_nbr_bwd)
      const float4 rw = reinterpret_cast<const float4*>(a.ws.raw)[p];
      d0 = dr.x; d1 = dr.y; d2 = dr.z;
      if (!(a.flags & PSL_NO_SIGMOID)) { d0 *= rw.x * (1.f - rw.x); d1 *= rw.y * (1.f - rw.y); d2 *= rw.z * (1.f - rw.z); }
      if (a.flags & PSL_HAS_AFFINE) {
        const float* A = a.affine;
        const float o0 = a.ws.out3[(size_t)p * 4], o1 = a.ws.out3[(size_t)p * 4 + 1], o2 = a.ws.out3[(size_t)p * 4 + 2];
        ag[0] = o0 * d0; ag[1] = o0 * d1; ag[2] = o0 * d2; ag[3] = o1 * d0; ag[4] = o1 * d1; ag[5] = o1 * d2;
        ag[6] = o2 * d0; ag[7] = o2 * d1; ag[8] = o2 * d2; ag[9] = d0; ag[10] = d1; ag[11] = d2;
        const float e0 = A[0] * d0 + A[1] * d1 + A[2] * d2;
        const float e1 = A[3] * d0 + A[4] * d1 + A[5] * d2;
        const float e2 = A[6] * d0 + A[7] * d1 + A[8] * d2;
        d0 = e0; d1 = e1; d2 = e2;
      }
      if (a.ws.d_out3) reinterpret_cast<float4*>(a.ws.d_out3)[p] = make_float4(d0, d1, d2, 0.f);
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) sAffP[s * 12 + j] = ag[j];
    sDO[s * 4] = d0; sDO[s * 4 + 1] = d1; sDO[s * 4 + 2] = d2; sDO[s * 4 + 3] = 0.f;
    if (rf.on) {     // the loss terms of 16 samples (owners of at most four rays) -> one slot of the iteration
#pragma unroll
      for (int ofs = 8; ofs > 0; ofs >>= 1) { lg += __shfl_xor(lg, ofs); lc += __shfl_xor(lc, ofs); lcnt += __shfl_xor(lcnt, ofs); }
      if ((s & 15) == 0 && lcnt != 0.0) {
        double* acc = rf.loss_acc + 4 * ((blockIdx.x + (s >> 4)) & (kLossSlots - 1));
        atomicAdd(acc + 0, lg); atomicAdd(acc + 1, lc); atomicAdd(acc + 2, lcnt);
      }
    }
  } else if (PTSG && t >= 64 && t < 64 + MT * 64) {
    sDP[t - 64] = 0.f;
  }
  sched_fence_b();
  f32x4 wo0, wo1, wo2;       // output_linear.weight [3][128], this lane's four channels: requested in front of the barrier
  {                          // (behind the ray stage, whose registers are dead by now)
    const float* wo = M + MO(PI_C_OUT) + nt * 16 + 4 * g;
    wo0 = *reinterpret_cast<const f32x4*>(wo); wo1 = *reinterpret_cast<const f32x4*>(wo + HC); wo2 = *reinterpret_cast<const f32x4*>(wo + 2 * HC);
  }
  sched_fence_b();
  lds_barrier();
  PSL_STAMP(1);
  // ---------------------------------------------------------------- colour trunk, wave w = hidden channel tile w
  f32x4 G[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const float d0 = sDO[(m * TILE + rl) * 4], d1 = sDO[(m * TILE + rl) * 4 + 1], d2 = sDO[(m * TILE + rl) * 4 + 2];
#pragma unroll
    for (int r = 0; r < 4; ++r) G[m][r] = d0 * wo0[r] + d1 * wo1[r] + d2 * wo2[r];
  }
  f32x4 dccp[MT][2], dEc[MT];       // dEc: waves 0..2: dL/d(colour embedding) tile (PTSG)
#pragma unroll
  for (int m = 0; m < MT; ++m) { dccp[m][0] = dccp[m][1] = dEc[m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  auto layer = [&](auto I_) {
    constexpr int i = decltype(I_)::value;
    constexpr int BLs[5] = {BL_C0, BL_C1, BL_C2, BL_C3, BL_C4};
    constexpr int BLf[5] = {BL_CF0, BL_CF1, BL_CF2, BL_CF3, BL_CF4};
    sched_fence_b();
    const int fb = bfirst(BLs[i]);
    f32x4 wc[2];
    wc[0] = wcn[0]; wc[1] = wcn[1];
    if (i > 0) {
      wcn[0] = ldfragb(WB, bfirst(BLf[i > 0 ? i - 1 : 0]) + 0 * 8 + nt, lane);
      wcn[1] = ldfragb(WB, bfirst(BLf[i > 0 ? i - 1 : 0]) + 1 * 8 + nt, lane);
    }
    f32x4 y[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) { y[m] = ynext[m]; if (i > 0) ynext[m] = ld_y(i > 0 ? i - 1 : 0, m); }
    sched_fence_b();
    float* buf = sDZ + (i & 1) * MT * 8 * FRAG;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      // step A: dz = G * act'(y)
      f32x4 dz;
#pragma unroll
      for (int r = 0; r < 4; ++r) dz[r] = G[m][r] * softplus100_grad_from_out(y[m][r]);
      if (parg) {
        const size_t o_ = ((size_t)i * a.ws.Ppad + p0 + m * TILE + rl) * HC + nt * 16 + 4 * g;
        *reinterpret_cast<f32x4*>(a.ws.c_dz + o_) = dz;
        *reinterpret_cast<f32x4*>(a.ws.c_g + o_) = G[m];
      }
      *reinterpret_cast<f32x4*>(buf + m * 8 * FRAG + nt * FRAG + lane * 4) = dz;
      // step B (K-split): dL/dc partial += Wc_i^T[:, own 16 channels] G
      mma4b(dccp[m][0], wc[0], G[m]);
      mma4b(dccp[m][1], wc[1], G[m]);
    }
    PSL_STAMP(2 + 3 * (4 - i));
    lds_barrier();
    PSL_STAMP(3 + 3 * (4 - i));
    // step C: dL/d(input of layer i) = W_i^T dz, hidden part (this wave's 16 channels), all 128 dz channels; a weight slot is
    // refilled with layer i - 1's fragment as soon as its MFMAs have issued (the loads used to stand at the next layer's start)
    if (i > 0) {
      f32x4 ga[MT], gb[MT], z0[MT], z1[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        ga[m] = gb[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        z0[m] = *reinterpret_cast<const f32x4*>(buf + m * 8 * FRAG + lane * 4);
        z1[m] = *reinterpret_cast<const f32x4*>(buf + m * 8 * FRAG + FRAG + lane * 4);
      }
#pragma unroll
      for (int q = 0; q < 8; q += 2) {
        sched_fence_b();
        f32x4 c0[MT], c1[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          c0[m] = z0[m]; c1[m] = z1[m];
          if (q < 6) {
            z0[m] = *reinterpret_cast<const f32x4*>(buf + m * 8 * FRAG + (q + 2) * FRAG + lane * 4);
            z1[m] = *reinterpret_cast<const f32x4*>(buf + m * 8 * FRAG + (q + 3) * FRAG + lane * 4);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int m = 0; m < MT; ++m) { ga[m] = mfma16(wq[q][r], c0[m][r], ga[m]); gb[m] = mfma16(wq[q + 1][r], c1[m][r], gb[m]); }
        if (i > 1) {
          sched_fence_b();
          const int fn = bfirst(BLs[i > 1 ? i - 1 : 1]);
          wq[q] = ldfragb(WB, fn + nt * 8 + q, lane); wq[q + 1] = ldfragb(WB, fn + nt * 8 + q + 1, lane);
        }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) G[m][r] = ga[m][r] + gb[m][r];
    }
    PSL_STAMP(4 + 3 * (4 - i));
    if (PTSG && (i == 3 || i == 0) && wave < 3) {   // embedding part: input tiles 8..10 of the skip layer, 0..2 of layer 0
      const int tile0 = (i == 3) ? 8 : 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const f32x4 we = ldfragb(WB, fb + (tile0 + wave) * 8 + q, lane);
#pragma unroll
        for (int m = 0; m < MT; ++m) mma4b(dEc[m], we, *reinterpret_cast<const f32x4*>(buf + m * 8 * FRAG + q * FRAG + lane * 4));
      }
    }
  };
  layer(std::integral_constant<int, 4>{});
  layer(std::integral_constant<int, 3>{});
  layer(std::integral_constant<int, 2>{});
  layer(std::integral_constant<int, 1>{});
  layer(std::integral_constant<int, 0>{});
  // the eight K-split partial tiles of dL/dc meet in LDS (in the dz buffers: every wave is past the last layer's barrier,
  // and only the pose-gradient instantiation still reads dz after it -- that one synchronises first)
  if constexpr (PTSG) lds_barrier();
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    *reinterpret_cast<f32x4*>(sDccP + ((wave * MT + m) * 2 + 0) * FRAG + lane * 4) = dccp[m][0];
    *reinterpret_cast<f32x4*>(sDccP + ((wave * MT + m) * 2 + 1) * FRAG + lane * 4) = dccp[m][1];
    if (PTSG && wave < 3) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int e = 16 * wave + 4 * g + r; if (e < EC) sDE[(m * TILE + rl) * LD_E2 + e] = dEc[m][r]; }
    }
  }
  PSL_STAMP(17);
  lds_barrier();
#pragma unroll
  for (int m = 0; m < MT; ++m) {   // 512 threads, 512 elements [it][lane][r] per sub-tile: sum over the waves -> dcc[row][16 it + 4 g + r]
    const int e = t;               // (k_nbr_bwd masks samples without neighbours)
    float v = 0.f;
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) v += sDccP[(w8 * MT + m) * 2 * FRAG + e];
    const int it = e >> 8, ln = (e >> 2) & 63, r = e & 3;
    a.ws.dcc[(size_t)(p0 + m * TILE + (ln & 15)) * C + it * 16 + 4 * (ln >> 4) + r] = v;
  }
  PSL_STAMP(18);
  if constexpr (PTSG) {
    // colour Fourier embedding [sin, cos] (20 + 20): one (sample, frequency) pair per thread -> dL/dp of the tile's samples
    // (k_nbr_bwd adds the interpolation-weight and rel-pos shares)
    for (int e = t; e < MT * TILE * ECF; e += WG) {
      const int s = e / ECF, f = e - s * ECF;
      const SampleGeom sg = sample_geom(a, min(p0 + s, a.P - 1));
      float sn, cs;
      fast_sincosf(fourier_phase(sg.x, sg.y, sg.z, a.Bcol, ECF, f), sn, cs);
      const float dy2 = TWO_PI * (sDE[s * LD_E2 + f] * cs - sDE[s * LD_E2 + ECF + f] * sn);
      atomic_add_f32(&sDP[s * 4], dy2 * a.Bcol[f]); atomic_add_f32(&sDP[s * 4 + 1], dy2 * a.Bcol[ECF + f]);
      atomic_add_f32(&sDP[s * 4 + 2], dy2 * a.Bcol[2 * ECF + f]);
    }
    lds_barrier();
    if (t < MT * TILE && p0 + t < a.P)
      reinterpret_cast<float4*>(a.ws.dp)[p0 + t] = make_float4(sDP[t * 4], sDP[t * 4 + 1], sDP[t * 4 + 2], 0.f);
  }
  if ((a.flags & PSL_HAS_AFFINE) && t < 12 && o.g_affine) {
    float v = 0.f;
#pragma unroll
    for (int s = 0; s < MT * TILE; ++s) v += sAffP[s * 12 + t];
    atomic_add_f32(&o.g_affine[t], v);
  }
}

// first n2 workgroups: double tiles, the rest single ones (trunk_plan, psl_decode_fwd2.hip)
template <bool PTSG>
__global__ __launch_bounds__(WG, 2) void k_trunk_bwd(DecodeArgs a, Bwd2Out o, const float* __restrict__ WB, int n2, RayFuse rf, TrackFuse tf) {
  __builtin_amdgcn_s_setprio(1);      // above the side-stream k-NN prefetch (see k_decode_fwd2)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  BlkTrace bt(a);
  const int b = (int)blockIdx.x;
  float thr_track = 0.f;
  if constexpr (PTSG) {
    if (tf.on) thr_track = track_threshold_block(a, tf, smem + TrunkBLdsT<1>::oDZ, b == 0);
  }
  if (b < n2) trunk_tile_bwd<PTSG, 2>(a, o, WB, smem, b * 2 * TILE, rf, tf, thr_track);
  else trunk_tile_bwd<PTSG, 1>(a, o, WB, smem, (2 * n2 + (b - n2)) * TILE, rf, tf, thr_track);
  bt.done(a);
}

// ---- F_theta backward / feature scatter of one wavefront's 16 pairs (samples 2 u, 2 u + 1)
struct NbrBLds {     // floats; the per-wave transpose tiles reuse the weight region once every wave is past its last MFMA
  static constexpr int oWn = 0, oDB = kNbrFragsB * FRAG, total = oDB + 32, tile = 512,
                       tRel = 16 * LD_X2, tDp = tRel + 48, tOk = tDp + 8;
};
static_assert(NbrBLds::tOk + 16 <= NbrBLds::tile, "per-wave tile of k_nbr_bwd");

template <bool PTSG, bool RELPOS>
__device__ __forceinline__ void nbr_unit_bwd(const DecodeArgs& a, const Bwd2Out& o, const float* __restrict__ WB, float* smem, int u_in,
                                             int n_units) {
  using L = NbrBLds;
  const int t = threadIdx.x, lane = t & 63, rl = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool act = u_in < n_units;
  const int u = min(u_in, n_units - 1);
  const bool featg = (a.flags & PSL_FEAT_GRAD) != 0;
  const bool parg = (a.flags & PSL_PARAM_GRAD) != 0;
  const float* __restrict__ M = a.master;
  const float* sWn = smem + L::oWn;
  float* sDB = smem + L::oDB;                 // [32] dL/dB_rel of the workgroup (30 used)
  float* tile = smem + wave * L::tile;        // valid after the barrier that follows the last MFMA
  // ---- set-up of this lane's pair (the list entry first, the weight-fragment DMA behind it: see nbr_unit_fwd)
  const int s = rl >> 3, k = rl & 7;
  const int ps = 2 * u + s;
  const int p = min(ps, a.P - 1);
  const bool live = act && ps < a.P;
  const int i = a.ws.I[(size_t)p * K + k];
  const float wgt = a.ws.w[(size_t)p * K + k];
  const int cnt_p = a.ws.cnt[p];
  const SampleGeom sg = sample_geom(a, p);
  sched_fence_b();
  if (RELPOS) {
#pragma unroll
    for (int j = 0; j < kNbrFragsB / 4; ++j) glds16(WB + ((size_t)(j * 4 + wave) * 64 + lane) * 4, smem + L::oWn + (j * 4 + wave) * FRAG);
    if (t < 32) sDB[t] = 0.f;
  }
  sched_fence_b();
  const float4 q = a.pos[max(i, 0)];
  const bool has = live && cnt_p >= a.min_nn;
  const size_t grow = (size_t)u * 16 + rl;
  f32x4 dc[2];     // dL/dc_col of this row's sample, channels 16 jt + 4 g + r; zero where the sample has too few neighbours
  {
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(a.ws.dcc + (size_t)ps * C + 4 * g);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(a.ws.dcc + (size_t)ps * C + 16 + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) { dc[0][r] = has ? v0[r] : 0.f; dc[1][r] = has ? v1[r] : 0.f; }
  }
  int dst = -1;
  if (i >= 0 && has && wgt != 0.f) dst = o.row_map ? o.row_map[i] : i;
  const float rx = (i >= 0) ? __fsub_rn(q.x, sg.x) : 0.f, ry = (i >= 0) ? __fsub_rn(q.y, sg.y) : 0.f,
              rz = (i >= 0) ? __fsub_rn(q.z, sg.z) : 0.f;
  float gwv = 0.f;       // dL/dw of this pair (PTSG)
  if constexpr (RELPOS) {
    // saved hidden activations of this wave's rows (HBM: written by k_nbr_fwd): requested before the barrier
    f32x4 h1v[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) h1v[it] = *reinterpret_cast<const f32x4*>(a.ws.n_h1 + grow * HC + it * 16 + 4 * g);
    f32x4 n0, n1;
    if constexpr (PTSG) {
      n0 = *reinterpret_cast<const f32x4*>(a.ws.n_out + grow * C + 4 * g);
      n1 = *reinterpret_cast<const f32x4*>(a.ws.n_out + grow * C + 16 + 4 * g);
    }
    const bool need_rel = parg || PTSG;
    float psn[3], pcs[3];
    if (need_rel && a.ws.n_x) {
#pragma unroll
      for (int uu = 0; uu < 3; ++uu) {
        const int e = lane + 64 * uu, r2 = min(e, 16 * ERF - 1) / ERF, f = min(e, 16 * ERF - 1) - r2 * ERF;
        const float* xr = a.ws.n_x + ((size_t)u * 16 + r2) * NX;
        psn[uu] = xr[f]; pcs[uu] = xr[ERF + f];
      }
    }
    lds_barrier_dma();        // weight fragments published; no global store / atomic above this line
    if (featg && dst >= 0 && o.t_col && g == 0) o.t_col[dst] = 1;
    // d_nf[row][ch] = w[row] * dC[s][ch]
    f32x4 dnf[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) dnf[jt][r] = wgt * dc[jt][r];
      if (parg && act) *reinterpret_cast<f32x4*>(a.ws.n_dnf + grow * C + jt * 16 + 4 * g) = dnf[jt];
    }
    if constexpr (PTSG) {   // dL/dw[s][k] = sum_ch nf[row][ch] dC[s][ch]
      float v = 0.f;
      if (live) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v += n0[r] * dc[0][r] + n1[r] * dc[1][r];
      }
      v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
      gwv = v;
    }
    // dH1^T[hid][row] = W2^T d_nf^T (linear2.weight [32][128]): 8 hidden tiles x 2 k-groups, four steps of four tiles
    f32x4 dh[8];
    constexpr int b2 = bfirst(BL_N2);
    constexpr int b1 = bfirst(BL_N1);
    f32x4 wn[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wn[j] = ldsfragb(sWn, b2 + j * 2 + 0, lane);
#pragma unroll
    for (int it = 0; it < 8; ++it) dh[it] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < 4; ++st) {       // st = 2 * half + q
      sched_fence_b();
      const int half = st >> 1, qq = st & 1;
      f32x4 wc4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) wc4[j] = wn[j];
      if (st < 3) {
        const int h2 = (st + 1) >> 1, q2 = (st + 1) & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) wn[j] = ldsfragb(sWn, b2 + (4 * h2 + j) * 2 + q2, lane);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) wn[j] = ldsfragb(sWn, b1 + j * 8 + 0, lane);       // first step of the next product
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) dh[4 * half + j] = mfma16(wc4[j][r], dnf[qq][r], dh[4 * half + j]);
    }
    // dz1 = dH1 * softplus'(h1)
#pragma unroll
    for (int it = 0; it < 8; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r) dh[it][r] = live ? dh[it][r] * softplus100_grad_from_out(h1v[it][r]) : 0.f;
      if (parg && act) *reinterpret_cast<f32x4*>(a.ws.n_dz1 + grow * HC + it * 16 + 4 * g) = dh[it];
    }
    // dX1^T[x][row] = W1^T dz1^T (linear1.weight [128][52]): input tiles (feat 0..15, feat 16..31, rel 0..15, rel 16..19)
    f32x4 dx[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) dx[it] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int qq = 0; qq < 8; ++qq) {
      sched_fence_b();
      f32x4 wf4[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) wf4[it] = wn[it];
      if (qq < 7) {
#pragma unroll
        for (int it = 0; it < 4; ++it) wn[it] = ldsfragb(sWn, b1 + it * 8 + qq + 1, lane);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dx[0] = mfma16(wf4[0][r], dh[qq][r], dx[0]);
        dx[1] = mfma16(wf4[1][r], dh[qq][r], dx[1]);
        if (need_rel) { dx[2] = mfma16(wf4[2][r], dh[qq][r], dx[2]); dx[3] = mfma16(wf4[3][r], dh[qq][r], dx[3]); }
      }
    }
    lds_barrier();            // every wave of the workgroup is done with the weight fragments: their LDS becomes the tiles
    if (featg) { const f32x4 dxf[2] = {dx[0], dx[1]}; scatter_pair_rows(tile, o.g_col, dxf, dst); }
    // rel-pos embedding part: y_f = 2pi rel . B[:,f]; e = [sin y, cos y] (see color_tile_bwd)
    if (need_rel) {
      float* xw = tile;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xw[rl * LD_X2 + 4 * g + r] = dx[2][r];
        if (g == 0) xw[rl * LD_X2 + 16 + r] = dx[3][r];
      }
      if (g == 0) {
        xw[L::tRel + rl * 3] = rx; xw[L::tRel + rl * 3 + 1] = ry; xw[L::tRel + rl * 3 + 2] = rz;
        xw[L::tOk + rl] = (live && i >= 0) ? 1.f : 0.f;
      }
      wave_lds_sync();
      const float* Brel = M + MO(PI_C_BREL);
#pragma unroll
      for (int uu = 0; uu < 3; ++uu) {
        const int e = lane + 64 * uu;
        if (e >= 16 * ERF) break;
        const int r2 = e / ERF, f = e - r2 * ERF;
        const bool lv = xw[L::tOk + r2] != 0.f;
        float sn, cs;
        if (a.ws.n_x) {
          sn = psn[uu]; cs = pcs[uu];
        } else {
          fast_sincosf(fourier_phase(xw[L::tRel + r2 * 3], xw[L::tRel + r2 * 3 + 1], xw[L::tRel + r2 * 3 + 2], Brel, ERF, f), sn, cs);
        }
        const float dy2 = TWO_PI * (xw[r2 * LD_X2 + f] * cs - xw[r2 * LD_X2 + ERF + f] * sn);
        xw[r2 * LD_X2 + f] = lv ? dy2 : 0.f;        // each pair is read and rewritten by its own lane only
      }
      wave_lds_sync();
      if (parg && lane < 3 * ERF) {
        const int ax = lane / ERF, f = lane - ax * ERF;
        float v = 0.f;
#pragma unroll
        for (int r2 = 0; r2 < 16; ++r2) v += xw[r2 * LD_X2 + f] * xw[L::tRel + r2 * 3 + ax];
        atomic_add_f32(&sDB[ax * ERF + f], v);
      }
      if constexpr (PTSG) {   // rel = x_k - p  =>  dp -= d_rel
        if (lane >= 32 && lane < 38) {
          const int sl = (lane - 32) / 3, ax = (lane - 32) - 3 * sl;
          float v = 0.f;
          for (int r2 = 8 * sl; r2 < 8 * sl + 8; ++r2)
#pragma unroll
            for (int f = 0; f < ERF; ++f) v += xw[r2 * LD_X2 + f] * Brel[ax * ERF + f];
          xw[L::tDp + sl * 3 + ax] = -v;
        }
        wave_lds_sync();
      }
    }
    lds_barrier();
    if (parg && t < 3 * ERF && o.g_brel) atomic_add_f32(&o.g_brel[t], sDB[t]);
  } else {
    if (featg && dst >= 0 && o.t_col && g == 0) o.t_col[dst] = 1;
    // ---- plain interpolation: scatter w_k * dC into the colour feature rows, collect dL/dw_k
    if (featg) {
      f32x4 dxf[2];
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dxf[jt][r] = wgt * dc[jt][r];
      scatter_pair_rows(tile, o.g_col, dxf, dst);
    }
    if constexpr (PTSG) {
      float v = 0.f;
      if (i >= 0 && has) {
        const float* frow = a.col_feats + (size_t)i * C + 4 * g;
        const f32x4 f0 = *reinterpret_cast<const f32x4*>(frow), f1 = *reinterpret_cast<const f32x4*>(frow + 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) v += f0[r] * dc[0][r] + f1[r] * dc[1][r];
      }
      v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
      gwv = v;
    }
  }
  if constexpr (PTSG) {
    // interpolation weights: w = a/S, a = [D<=r2]/(D+1e-10), D = |x_k - p|^2   (decoder.py:143-160)
    const float D = (i >= 0) ? __fadd_rn(__fadd_rn(__fmul_rn(rx, rx), __fmul_rn(ry, ry)), __fmul_rn(rz, rz)) : __int_as_float(0x7F800000);
    const float av = (D > sg.r2) ? 0.f : 1.0f / (D + 1e-10f);
    const float S1 = group8_sum(av);
    const float gw = has ? gwv : 0.f;
    const float dot = group8_sum(gw * wgt);
    const float da = (gw - dot) / fmaxf(S1, 1e-12f);
    const float dD = -da * av * av;                 // a = 1/(D+eps) -> da/dD = -a^2 ; masked slots: a = 0
    float px = group8_sum(-2.f * dD * rx), py = group8_sum(-2.f * dD * ry), pz = group8_sum(-2.f * dD * rz);      // dD/dp = -2 (x_k - p)
    if (RELPOS) { px += tile[NbrBLds::tDp + s * 3]; py += tile[NbrBLds::tDp + s * 3 + 1]; pz += tile[NbrBLds::tDp + s * 3 + 2]; }
    if (k == 0 && g == 0 && live) {     // k_trunk_bwd left the colour-embedding share in dp
      float4 d = reinterpret_cast<float4*>(a.ws.dp)[ps];
      d.x += px; d.y += py; d.z += pz;
      reinterpret_cast<float4*>(a.ws.dp)[ps] = d;
    }
  }
}

// grid: [0, geo_blocks) four one-wave geometry tiles each, [geo_blocks, wl_block0) four F_theta units each, then the work-list blocks
template <bool PTSG, bool RELPOS>
__global__ __launch_bounds__(NBR_WG_B, 3) void k_nbr_bwd(DecodeArgs a, Bwd2Out o, const float* __restrict__ WB, int geo_blocks, int n_units,
                                                        AdamWorklist wl, int wl_block0) {
  __builtin_amdgcn_s_setprio(1);      // above the side-stream k-NN prefetch (see k_decode_fwd2)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = (int)blockIdx.x;
  if (b >= wl_block0) {
    worklist_role_wave(wl, (b - wl_block0) * (int)blockDim.x + (int)threadIdx.x);
    return;
  }
  BlkTrace bt(a);
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  if (b < geo_blocks) {
    const int p0 = (b * 4 + wave) * TILE;
    __builtin_amdgcn_s_setprio(3);       // the launch's critical path (see k_nbr_fwd)
    if (p0 < a.P) {
      ScatterLds& sl = *reinterpret_cast<ScatterLds*>(smem + wave * (sizeof(ScatterLds) / sizeof(float)));
      TrackFuse tf0{};
      if constexpr (PTSG) geo_tile_bwd_ptsg(a, o, WB, p0, sl, tf0, 0.f);
      else geo_tile_bwd<false>(a, o, WB, p0, sl, nullptr);
    }
  } else {
    nbr_unit_bwd<PTSG, RELPOS>(a, o, WB, smem, (b - geo_blocks) * 4 + wave, n_units);
  }
  bt.done(a);
}

// grid as in the forward: [0, color_tiles) colour role, then one geometry-role WAVEFRONT per tile in a workgroup of its own
// (psl_map_iters, colour stage without exposure: rf.on -- the ray stage runs inside this kernel, and the workgroups from
//  wl_block0 on build the work list of the iteration's lazy Adam, which the ray kernel used to carry)
template <bool PTSG, bool COLOR>
__global__ __launch_bounds__(COLOR ? WG : 64, COLOR ? 4 : 2) void k_decode_bwd2(DecodeArgs a, Bwd2Out o, const float* __restrict__ WB, int color_tiles,
                                                                                            RayFuse rf, AdamWorklist wl, int wl_block0, TrackFuse tf) {
  // above the mapper's side-stream k-NN prefetch (priority 0), whose waves share the SIMDs of this launch for 2 of every 7 ms of a mapped frame
  __builtin_amdgcn_s_setprio(1);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (COLOR && (int)blockIdx.x >= wl_block0) {
    worklist_role_wave(wl, ((int)blockIdx.x - wl_block0) * (int)blockDim.x + (int)threadIdx.x);
    return;
  }
  BlkTrace bt(a);
  const int b = (int)blockIdx.x;      // workgroup -> (role, tile) as in the forward kernel
  const bool is_color = COLOR && b < color_tiles;
  const int tile = is_color ? b : b - color_tiles;
  // the tracker's ray stage inside this kernel (TrackFuse): the launch-wide mask threshold first, by every thread of the
  // workgroup -- in the geometry role too, whose other seven wavefronts then leave
  float thr_track = 0.f;
  if constexpr (PTSG && COLOR) {
    if (tf.on) thr_track = track_threshold_block(a, tf, smem + Bwd2Lds::oDZ, is_color && b == 0);
  }
  if (is_color) {
    color_tile_bwd<PTSG>(a, o, WB, smem, tile * TILE, rf, tf, thr_track);
  } else {
    if (threadIdx.x >= 64) return;
    if constexpr (PTSG) geo_tile_bwd_ptsg(a, o, WB, tile * TILE, *reinterpret_cast<ScatterLds*>(smem), tf, thr_track);
    else geo_tile_bwd<false>(a, o, WB, tile * TILE, *reinterpret_cast<ScatterLds*>(smem), &rf);
  }
  bt.done(a);
}

int launch_decode_bwd2(psl_ctx* ctx, const DecodeArgs& a_in, const psl_render_grads& g, float* small, hipStream_t s) {
  DecodeArgs a = a_in;
  Bwd2Out o;
  o.g_geo = g.g_geo_feats; o.g_col = g.g_col_feats; o.row_map = g.feat_row_map;
  o.g_brel = small;
  o.g_affine = small + 32;
  o.t_geo = ctx->touched_geo; o.t_col = ctx->touched_col;
  const int tiles = (a.P + TILE - 1) / TILE;
  const bool color = a.flags & PSL_STAGE_COLOR;
  const bool ptsg = a.flags & PSL_PTS_GRAD;
  const size_t lds = sizeof(float) * ((a.flags & 0x10000) ? Bwd2Lds::total_nbr : Bwd2Lds::total);
  const size_t lds_max = sizeof(float) * Bwd2Lds::total_nbr;
  static bool attr_set = false;
  if (!attr_set) {
    PSL_HIP(hipFuncSetAttribute((const void*)k_decode_bwd2<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    PSL_HIP(hipFuncSetAttribute((const void*)k_decode_bwd2<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    attr_set = true;
  }
  const float* WB = ctx->wb;
  static int dbg_on = -1;
  if (dbg_on < 0) { const char* e = getenv("PSL_DEBUG_PHASES"); dbg_on = (e && e[0] == '1') ? 1 : 0; }
  if (dbg_on && a.dbg) PSL_HIP(hipMemsetAsync(a.dbg, 0, 64 * sizeof(unsigned long long), s));
  RayFuse rf{};
  AdamWorklist wl{};
  int wl_blocks = 0;
  if (ctx->ray_fuse && color && !ptsg) {
    rf = *(const RayFuse*)ctx->ray_fuse;
    if (ctx->ray_wl) wl = *(const AdamWorklist*)ctx->ray_wl;
    if (wl.I_a && wl.n4 > 0) wl_blocks = ((wl.I_b ? 2 : 1) * wl.n4 + WG - 1) / WG;
  }
  TrackFuse tf{};
  if (ctx->track_fuse && color && ptsg) tf = *(const TrackFuse*)ctx->track_fuse;
  if (color && color_split_on(tiles)) {
    const bool relpos = (a.flags & 0x10000) != 0;
    const int n_units = tiles * (TILE / 2), f_blocks = (n_units + 3) / 4, geo_blocks = (tiles + 3) / 4;
    const TrunkPlan tp = trunk_plan(tiles);
    const size_t lds3 = sizeof(float) * (tp.n2 ? TrunkBLdsT<2>::total : TrunkBLdsT<1>::total);
    { int rc = blk_trace_begin(a, tp.n2 + tp.n1, s); if (rc) return rc; }
    if (ptsg) PSL_KLAUNCH2((k_trunk_bwd<true>), true, false, dim3(tp.n2 + tp.n1), dim3(WG), lds3, s, a, o, WB, tp.n2, rf, tf);
    else PSL_KLAUNCH2((k_trunk_bwd<false>), true, false, dim3(tp.n2 + tp.n1), dim3(WG), lds3, s, a, o, WB, tp.n2, rf, tf);
    PSL_LAUNCH_CHECK();
    { int rc = blk_trace_end(a, ptsg ? "trunk_bwd_ptsg" : "trunk_bwd", tp.n2 + tp.n1, tp.n2, WG); if (rc) return rc; }
    const int wlb = (wl.I_a && wl.n4 > 0) ? ((wl.I_b ? 2 : 1) * wl.n4 + NBR_WG_B - 1) / NBR_WG_B : 0;
    const int grid4 = geo_blocks + f_blocks + wlb;
    const size_t lds4 = sizeof(float) * (relpos ? (size_t)NbrBLds::total : (size_t)4 * (sizeof(ScatterLds) / sizeof(float)));
    { int rc = blk_trace_begin(a, grid4, s); if (rc) return rc; }
    if (ptsg) {
      if (relpos) PSL_KLAUNCH2((k_nbr_bwd<true, true>), false, true, dim3(grid4), dim3(NBR_WG_B), lds4, s, a, o, WB, geo_blocks, n_units, wl, geo_blocks + f_blocks);
      else PSL_KLAUNCH2((k_nbr_bwd<true, false>), false, true, dim3(grid4), dim3(NBR_WG_B), lds4, s, a, o, WB, geo_blocks, n_units, wl, geo_blocks + f_blocks);
    } else {
      if (relpos) PSL_KLAUNCH2((k_nbr_bwd<false, true>), false, true, dim3(grid4), dim3(NBR_WG_B), lds4, s, a, o, WB, geo_blocks, n_units, wl, geo_blocks + f_blocks);
      else PSL_KLAUNCH2((k_nbr_bwd<false, false>), false, true, dim3(grid4), dim3(NBR_WG_B), lds4, s, a, o, WB, geo_blocks, n_units, wl, geo_blocks + f_blocks);
    }
    PSL_LAUNCH_CHECK();
    { int rc = blk_trace_end(a, ptsg ? "nbr_bwd_ptsg" : "nbr_bwd", geo_blocks + f_blocks, -geo_blocks, NBR_WG_B); if (rc) return rc; }
    if (dbg_on && a.dbg) {
      unsigned long long h[64];
      PSL_HIP(hipMemcpy(h, a.dbg, sizeof(h), hipMemcpyDeviceToHost));
      fprintf(stderr, "[psl trunk_bwd P=%d ptsg=%d] set-up %llu |", a.P, (int)ptsg, h[1] - h[0]);
      unsigned long long prev = h[1];
      for (int L = 0; L < 5; ++L) {
        fprintf(stderr, " L%d: pre %llu bar %llu mfma %llu |", 4 - L, h[2 + 3 * L] - prev, h[3 + 3 * L] - h[2 + 3 * L], h[4 + 3 * L] - h[3 + 3 * L]);
        prev = h[4 + 3 * L];
      }
      fprintf(stderr, " dcc-bar %llu reduce %llu | total %llu\n", h[17] - prev, h[18] - h[17], h[18] - h[0]);
    }
    return PSL_OK;
  }
  const int grid_c = 2 * tiles + wl_blocks;
  { int rc = blk_trace_begin(a, color ? grid_c : tiles, s); if (rc) return rc; }
  if (color) {
    if (ptsg) PSL_KLAUNCH((k_decode_bwd2<true, true>), dim3(2 * tiles), dim3(WG), lds, s, a, o, WB, tiles, rf, wl, 2 * tiles, tf);
    else PSL_KLAUNCH((k_decode_bwd2<false, true>), dim3(grid_c), dim3(WG), lds, s, a, o, WB, tiles, rf, wl, 2 * tiles, tf);
  } else {
    if (ptsg) PSL_KLAUNCH((k_decode_bwd2<true, false>), dim3(tiles), dim3(64), sizeof(ScatterLds), s, a, o, WB, 0, rf, wl, tiles, tf);
    else PSL_KLAUNCH((k_decode_bwd2<false, false>), dim3(tiles), dim3(64), sizeof(ScatterLds), s, a, o, WB, 0, rf, wl, tiles, tf);
  }
  PSL_LAUNCH_CHECK();
  { int rc = blk_trace_end(a, ptsg ? "bwd2_ptsg" : "bwd2", color ? grid_c : tiles, color ? tiles : 0, color ? WG : 64); if (rc) return rc; }
  if (dbg_on && a.dbg && color) {
    unsigned long long h[64];
    PSL_HIP(hipMemcpy(h, a.dbg, sizeof(h), hipMemcpyDeviceToHost));
    fprintf(stderr, "[psl bwd2 colour P=%d ptsg=%d] phase0 %llu |", a.P, (int)ptsg, h[1] - h[0]);
    unsigned long long prev = h[1];
    for (int L = 0; L < 5; ++L) {
      fprintf(stderr, " L%d: pre %llu bar %llu mfma %llu |", 4 - L, h[2 + 3 * L] - prev, h[3 + 3 * L] - h[2 + 3 * L], h[4 + 3 * L] - h[3 + 3 * L]);
      prev = h[4 + 3 * L];
    }
    fprintf(stderr, " dcc-bar %llu reduce %llu | F: dH %llu act %llu dX %llu scatter+rel %llu bar %llu | total %llu\n",
            h[17] - prev, h[18] - h[17], h[19] - h[18], h[20] - h[19], h[21] - h[20], h[22] - h[21], h[23] - h[22], h[23] - h[0]);
  }
  return PSL_OK;
}

int launch_dw(psl_ctx* ctx, const DecodeArgs& a, float* g_params, const float* g_brel, hipStream_t s);

// render backward of the decode stage: the register-chained kernels above, then the parameter-gradient GEMM
int launch_decode_bwd(psl_ctx* ctx, const DecodeArgs& a, const psl_render_grads& g, hipStream_t s) {
  const bool color = a.flags & PSL_STAGE_COLOR;
  if ((a.flags & PSL_FEAT_GRAD) && (!g.g_geo_feats || (color && !g.g_col_feats))) {
    set_error("psl_render_bwd: PSL_FEAT_GRAD needs g_geo_feats/g_col_feats"); return PSL_ERR_ARG;
  }
  if ((a.flags & PSL_PARAM_GRAD) && !g.g_params) { set_error("psl_render_bwd: PSL_PARAM_GRAD needs g_params"); return PSL_ERR_ARG; }
  if ((a.flags & PSL_HAS_AFFINE) && !g.g_exposure_affine) { set_error("psl_render_bwd: affine gradient buffer missing"); return PSL_ERR_ARG; }
  // small accumulators: [0..31] dB_rel, [32..47] affine  (ctx->d_small)
  float* small = ctx->d_small;     // cleared by the compositing-backward kernel that always runs just before
  static unsigned long long* dbg = nullptr;
  static int dbg_on = -1;
  if (dbg_on < 0) { const char* e = getenv("PSL_DEBUG_PHASES"); dbg_on = (e && e[0] == '1') ? 1 : 0; }
  DecodeArgs a2 = a;
  if (dbg_on) { if (!dbg) PSL_HIP(hipMalloc(&dbg, 64 * sizeof(unsigned long long))); a2.dbg = dbg; }
  {
    ProfScope ps(ctx, prof_decode_slot(a.flags, true), s, bwd_flops_per_sample(a.flags) * a.P, true);
    int rc = launch_decode_bwd2(ctx, a2, g, small, s);
    if (rc) return rc;
  }
  if ((a.flags & PSL_HAS_AFFINE) && color)
    PSL_HIP(hipMemcpyAsync(g.g_exposure_affine, small + 32, sizeof(float) * 12, hipMemcpyDeviceToDevice, s));
  if (a.flags & PSL_PARAM_GRAD) {
    if (color) {
      ProfScope ps(ctx, PROF_DW, s, dw_flops_per_sample(a.flags) * a.P, ctx->dw_defer_reduce != 0);
      int rc = launch_dw(ctx, a, g.g_params, small, s);
      if (rc) return rc;
    } else {
      // geometry stage: the colour decoder is not evaluated; the geometry decoder is frozen
      // (mapping.fix_geo_decoder, configs/point_slam.yaml:47) -> all-zero parameter gradient
      PSL_HIP(hipMemsetAsync(g.g_params, 0, sizeof(float) * kMasterFloats, s));
    }
  }
  return PSL_OK;
}


}  // namespace psl
