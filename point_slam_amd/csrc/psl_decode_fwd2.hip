// Fused per-sample decode, forward, register-chained form (see psl_frag.h for the operand algebra).
//
// Reference: MLP_geometry / MLP_color .get_feature_at_pos + .forward and POINT.forward
// (src/conv_onet/models/decoder.py:130-222, 341-449, 476-518).
//
// Two roles share one launch (no barrier couples them; they meet in `raw`, consumed by the next kernel):
//  * colour role  -- one 512-thread workgroup per 16-sample tile.  Phase F: wavefront w evaluates F_theta for the 16
//    (sample, neighbour) rows 16 w .. 16 w + 15 entirely in registers (gathered features and rel-pos sin/cos are B
//    operands straight from the loads; hidden activations feed the second layer from the accumulators), reduces over
//    the 8 neighbours of a sample with three DPP-style shuffles.  Phase T: the colour trunk, wavefront w owns output
//    columns 16 w .. 16 w + 15 of every layer; the only shared state is the 8 KiB hidden tile, exchanged through LDS in
//    fragment order (one ds_write_b128 + eight ds_read_b128 per lane per layer, ONE barrier per layer, two buffers).
//  * geometry role -- one WAVEFRONT per 16-sample tile, 8 tiles per workgroup, no LDS and no barrier at all: inverse-
//    distance weights, feature interpolation, 93 Fourier features and the five 32-wide layers stay in registers.
//    In stage 'geometry' only this role is launched.
// Weights are read as 1 KiB fragments (64 lanes x 16 B, fully coalesced) from the L2-resident fragment buffer.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <type_traits>
#include "psl_decode.h"
#include "psl_frag.h"
#include "psl_decode2.h"

namespace psl {

int launch_trunk_fwd_w(psl_ctx* ctx, const DecodeArgs& a, int tiles, bool last, hipStream_t s);   // psl_trunk_wave.hip
constexpr int kNbrFrags = 48;     // fragments of F_theta's two layers (see NbrStage below)
struct Fwd2Lds {
  static constexpr int oI = 0, oW = 128, oRel = 256, oPts = 640, oHas = 704, oCc = 720, oH = oCc + 2 * FRAG,
                       oOut = oH + 2 * 8 * FRAG, oFb = oOut + 8 * TILE * 4,   // K-split partial colour logits; the fallback feature vector
                       total = oFb + 32,              // 5.9 K floats = 23 KB
                       oWn = total, total_nbr = oWn + kNbrFrags * FRAG;   // + F_theta's weight fragments: 71 KB
};

// F_theta's weights in LDS.  A wavefront uses every fragment of F_theta once per 16 pairs, eight (four) wavefronts of a
// workgroup use the same ones, and fetched per wavefront from L2 one step ahead the fragment latency (~1.3 k cycles), not
// the MFMA pipe, set the pace of both products (phase stamps: linear1 10.9 k cycles for 3.3 k cycles of MFMA issue).  The
// 48 fragments of linear1 + linear2 (48 KiB, contiguous at the start of the fragment buffer) are therefore copied into LDS
// once per workgroup -- all loads of the copy in flight together, ONE L2 round trip -- and read from there (ds_read_b128,
// ~100 cycles) by every step.
static_assert(ffirst(FL_N1) == 0 && ffirst(FL_N2) == 32 && ffirst(FL_C0) == kNbrFrags, "F_theta fragments lead the forward buffer");
// The copy is LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, lane l -> base + 16 l, which IS the fragment
// layout): no staging registers (round 3 bounced 24 VGPRs per thread through NbrStage::load/store and spilled 19), no
// ds_write pass.  hipcc does not count these loads: the issuing wave waits vmcnt(0) itself before the barrier that
// publishes the fragments (lds_barrier_dma), so no global STORE may be issued before that barrier.
template <int NFRAG>
__device__ __forceinline__ void nbr_stage_dma(const float* __restrict__ W, float* sW, int slot, int nslots, int lane) {
#pragma unroll
  for (int j = 0; j < NFRAG / nslots; ++j) glds16(W + ((size_t)(j * nslots + slot) * 64 + lane) * 4, sW + (j * nslots + slot) * FRAG);
}
__device__ __forceinline__ f32x4 ldsfrag(const float* sW, int frag, int lane) {
  return *reinterpret_cast<const f32x4*>(sW + frag * FRAG + lane * 4);
}

// ------------------------------------------------------------------------------------------------ geometry role
// One wavefront, 16 samples: lane (rl = sample, g).  Writes raw[p].w (and xyz = 0 when `full_raw`), g_y, w (when asked).
// GEO_AHEAD: prefetch distance in steps (4 in the stage-'geometry' kernel; 3 inside the colour-stage kernel, whose
// 128-register budget is set by the colour role's two workgroups per CU -- 122 registers, no spill; at distance 4 three
// registers go to scratch).
template <int GEO_AHEAD>
__device__ __forceinline__ void geo_tile(const DecodeArgs& a, const float* __restrict__ WF, int p0, bool full_raw, bool save_w) {
  const int lane = threadIdx.x & 63, rl = lane & 15, g = lane >> 4;
  const int p = min(p0 + rl, a.P - 1);
  const float* __restrict__ M = a.master;
  f32x4 W0[kGeo.n], W1[kGeo.n];
  PSL_STAMP(32);
  // ---- neighbours, inverse-distance weights (decoder.py:152-160), interpolation (:162-171); no control flow:
  // absent neighbours (index -1) read point 0 and get weight 0.  The lists are requested before the sample geometry: the
  // positions they name are the next dependent round trip
  int nb[K];
  {
    const int4 i0 = *reinterpret_cast<const int4*>(a.ws.I + (size_t)p * K);
    const int4 i1 = *reinterpret_cast<const int4*>(a.ws.I + (size_t)p * K + 4);
    nb[0] = i0.x; nb[1] = i0.y; nb[2] = i0.z; nb[3] = i0.w; nb[4] = i1.x; nb[5] = i1.y; nb[6] = i1.z; nb[7] = i1.w;
  }
  const int cnt_p = a.ws.cnt[p];
  const SampleGeom sg = sample_geom(a, p);
  const bool has = cnt_p >= a.min_nn;     // has_neighbors (decoder.py:150); loaded with the lists, kept as a lane mask
  float w[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float4 q = a.pos[max(nb[k], 0)];
    const float D = (nb[k] >= 0) ? dist2(q.x, q.y, q.z, sg.x, sg.y, sg.z) : __int_as_float(0x7F800000);
    w[k] = nn_weight(D, sg.r2, (a.flags & kFlagExpoW) != 0);
  }
  // the reference sums the 8 weights inside F.normalize(p=1); any order is within 1 ulp
  const float wsum = ((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]));
  const float inv = fmaxf(wsum, 1e-12f);
#pragma unroll
  for (int k = 0; k < K; ++k) w[k] = w[k] / inv;
  if (save_w && g == 0) {     // rows up to Ppad exist in every workspace buffer: no bounds test on stores
    *reinterpret_cast<float4*>(a.ws.w + (size_t)(p0 + rl) * K) = make_float4(w[0], w[1], w[2], w[3]);
    *reinterpret_cast<float4*>(a.ws.w + (size_t)(p0 + rl) * K + 4) = make_float4(w[4], w[5], w[6], w[7]);
  }
  PSL_STAMP(33);
  f32x4 cg[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if ((k & 3) == 0) sched_fence();      // four neighbour rows (8 vector loads) in flight at a time
    const float* row = a.geo_feats + (size_t)max(nb[k], 0) * C + 4 * g;
    const f32x4 f0 = *reinterpret_cast<const f32x4*>(row), f1 = *reinterpret_cast<const f32x4*>(row + 16);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      cg[0][r] = __fadd_rn(cg[0][r], __fmul_rn(w[k], f0[r]));
      cg[1][r] = __fadd_rn(cg[1][r], __fmul_rn(w[k], f1[r]));
    }
  }
  {
    const f32x4 fb0 = *reinterpret_cast<const f32x4*>(a.fb_geo + 4 * g), fb1 = *reinterpret_cast<const f32x4*>(a.fb_geo + 16 + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) { cg[0][r] = has ? cg[0][r] : fb0[r]; cg[1][r] = has ? cg[1][r] : fb1[r]; }
  }
  pin(cg[0]); pin(cg[1]);
  sched_fence();
  PSL_STAMP(34);
  // weight fragments of the first steps: their L2 latency elapses behind the 24 sine evaluations
#pragma unroll
  for (int st = 0; st < GEO_AHEAD; ++st) {
    W0[st] = ldfrag(WF, kGeo.s[st].f0, lane);
    if (kGeo.s[st].f1 >= 0) W1[st] = ldfrag(WF, kGeo.s[st].f1, lane);
  }
  sched_fence();
  // ---- Fourier features sin(2 pi p . B) (decoder.py:8-37), channel 16 q + 4 g + r
  const float* __restrict__ Bg = M + MO(PI_G_B);
  f32x4 eg[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    // under the colour-stage kernel's 128-register budget: two groups of four channels (their 24 entries of B) at a time;
    // the stage-'geometry' kernel leaves all 72 loads and 24 polynomials to the scheduler
    if (GEO_AHEAD < 4 && (q & 1) == 0) sched_fence();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = min(16 * q + 4 * g + r, EG - 1);
      const float v = fast_sinf(fourier_phase(sg.x, sg.y, sg.z, Bg, EG, f));
      eg[q][r] = (16 * q + 4 * g + r < EG) ? v : 0.f;
    }
  }
  PSL_STAMP(35);
  // ---- five blocks: h = relu(W_i h + b_i) + (Wc_i c + bc_i); the embedding is re-attached after block 2
  f32x4 h[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  f32x4 acc[2], u[2], oo[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  acc[0] = ldbias(WF, fbias(FL_G0), 0, g); acc[1] = ldbias(WF, fbias(FL_G0), 1, g);
  u[0] = ldbias(WF, fbias(FL_GF0), 0, g); u[1] = ldbias(WF, fbias(FL_GF0), 1, g);
#pragma unroll
  for (int st = 0; st < kGeo.n; ++st) {
    sched_fence();
    if (st + GEO_AHEAD < kGeo.n) {
      W0[st + GEO_AHEAD] = ldfrag(WF, kGeo.s[st + GEO_AHEAD].f0, lane);
      if (kGeo.s[st + GEO_AHEAD].f1 >= 0) W1[st + GEO_AHEAD] = ldfrag(WF, kGeo.s[st + GEO_AHEAD].f1, lane);
    }
    const int bs = kGeo.s[st].bsel;
    const f32x4 b = bs < 6 ? eg[bs < 6 ? bs : 0] : (bs < 8 ? h[bs < 8 ? (bs >= 6 ? bs - 6 : 0) : 0] : cg[bs >= 8 ? bs - 8 : 0]);
    if (kGeo.s[st].dst == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { acc[0] = mfma16(W0[st][r], b[r], acc[0]); acc[1] = mfma16(W1[st][r], b[r], acc[1]); }
    } else if (kGeo.s[st].dst == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { u[0] = mfma16(W0[st][r], b[r], u[0]); u[1] = mfma16(W1[st][r], b[r], u[1]); }
    } else {
      mma4(oo[st & 1], W0[st], b);
    }
    if (kGeo.s[st].layer_end) {
      const int i = kGeo.s[st].layer_end - 1;
      PSL_STAMP(36 + i);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        f32x4 y;
#pragma unroll
        for (int r = 0; r < 4; ++r) { y[r] = fmaxf(acc[nt][r], 0.f); h[nt][r] = y[r] + u[nt][r]; }
        if (a.ws.g_y) *reinterpret_cast<f32x4*>(a.ws.g_y + ((size_t)i * a.ws.Ppad + p0 + rl) * HG + nt * 16 + 4 * g) = y;
      }
      if (i < 4) {
        constexpr int FLs[5] = {FL_G0, FL_G1, FL_G2, FL_G3, FL_G4};
        constexpr int FLf[5] = {FL_GF0, FL_GF1, FL_GF2, FL_GF3, FL_GF4};
        acc[0] = ldbias(WF, fbias(FLs[i + 1]), 0, g); acc[1] = ldbias(WF, fbias(FLs[i + 1]), 1, g);
        u[0] = ldbias(WF, fbias(FLf[i + 1]), 0, g); u[1] = ldbias(WF, fbias(FLf[i + 1]), 1, g);
      }
    }
  }
  PSL_STAMP(41);
  // ---- output_linear 32 -> 1 as one padded tile: row 0 of the accumulator = occupancy logit of sample rl
  {
    // the sample index again from an opaque copy of the lane id: left to CSE, hipcc carries the sign-extended index of the
    // first lines across the whole tile (and spills it under the colour-stage kernel's 128-register budget)
    int l2 = threadIdx.x;
    asm volatile("" : "+v"(l2));
    const int pe = p0 + (l2 & 15);
    if ((l2 & 48) == 0 && pe < a.P) {
      // raw[~point_mask, -1] = -100 (Renderer.py:189-190)
      const float occ = has ? (oo[0][0] + oo[1][0]) + M[MO(PI_G_OUT + 1)] : -100.0f;
      if (full_raw) reinterpret_cast<float4*>(a.ws.raw)[pe] = make_float4(0.f, 0.f, 0.f, occ);
      else a.ws.raw[(size_t)pe * 4 + 3] = occ;
    }
  }
}

// ------------------------------------------------------------------------------------------------ colour role
constexpr int kTrunkL[5] = {FL_C0, FL_C1, FL_C2, FL_C3, FL_C4};
constexpr int kTrunkF[5] = {FL_CF0, FL_CF1, FL_CF2, FL_CF3, FL_CF4};
// weight slots of a trunk wavefront (output tile nt): eight fragment slots, the two fc_c fragments, the two bias tiles
struct TrunkRegs { f32x4 w[8], c[2], bias, cbias; };
// layer 0's operands: embedding fragments in slots 0..3, fc_c.0, biases (requested as early as the caller can afford) ...
__device__ __forceinline__ void trunk_prologue_a(TrunkRegs& R, const float* __restrict__ WF, int nt, int lane, int g) {
  constexpr int b0 = ffirst(FL_C0);
#pragma unroll
  for (int q = 0; q < 4; ++q) R.w[q] = ldfrag(WF, b0 + nt * 4 + q, lane);
  R.c[0] = ldfrag(WF, ffirst(FL_CF0) + nt * 2 + 0, lane); R.c[1] = ldfrag(WF, ffirst(FL_CF0) + nt * 2 + 1, lane);
  R.bias = ldbias(WF, fbias(FL_C0), nt, g); R.cbias = ldbias(WF, fbias(FL_CF0), nt, g);
}
// ... and layer 1's fragments 4..7 into the slots layer 0 leaves empty
__device__ __forceinline__ void trunk_prologue_b(TrunkRegs& R, const float* __restrict__ WF, int nt, int lane) {
  constexpr int b1 = ffirst(FL_C1);
#pragma unroll
  for (int q = 4; q < 8; ++q) R.w[q] = ldfrag(WF, b1 + nt * 8 + q, lane);
}

template <int MT>
__device__ __forceinline__ void trunk_layers_fwd(const DecodeArgs& a, const float* __restrict__ WF, float* sH, float* sOut, int p0, TrunkRegs& R,
                                                 const f32x4 (&ccb0)[MT], const f32x4 (&ccb1)[MT], const f32x4 (&esn)[MT], const f32x4 (&ecs)[MT],
                                                 const float (&sn4)[MT], const float (&cs4)[MT], float ob0, float ob1, float ob2);

__device__ __forceinline__ void color_tile(const DecodeArgs& a, const float* __restrict__ WF, float* smem, int p0) {
  using L = Fwd2Lds;
  int* sI = (int*)(smem + L::oI);           // [16][8]
  float* sW = smem + L::oW;                 // [16][8]
  float* sRel = smem + L::oRel;             // [16][8][3]
  float* sPts = smem + L::oPts;             // [16][4]
  int* sHas = (int*)(smem + L::oHas);       // [16]
  float* sCc = smem + L::oCc;               // [2][64][4]   interpolated colour features, fragment order
  float* sH = smem + L::oH;                 // [2][8][64][4] hidden tile, fragment order, double buffered
  const int t = threadIdx.x, lane = t & 63, rl = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool relpos = (a.flags & 0x10000) != 0;
  const float* __restrict__ M = a.master;
  // Every workspace buffer holds Ppad (a multiple of the tile) rows, so saves need no bounds test; slots past the end
  // of the batch compute on a clamped sample and their rows are never read.

  PSL_STAMP(0);
  // F_theta's weight fragments do not depend on anything: their copy into LDS is requested before the neighbour set-up
  constexpr int f1 = ffirst(FL_N1);
  const float* sWn = smem + L::oWn;
  if (relpos) nbr_stage_dma<kNbrFrags>(WF, smem + L::oWn, wave, WG / 64, lane);

  // ---------------------------------------------------------------- phase 0: neighbours, weights (one thread per pair)
  if (t < TILE * K) {
    const int s = t >> 3, k = t & 7;
    const int p = min(p0 + s, a.P - 1);
    const int i = a.ws.I[(size_t)p * K + k];          // requested before the sample geometry (see sample_geom)
    const int cnt_p = a.ws.cnt[p];
    const SampleGeom sg = sample_geom(a, p);
    const float4 q = a.pos[max(i, 0)];
    const float D = (i >= 0) ? dist2(q.x, q.y, q.z, sg.x, sg.y, sg.z) : __int_as_float(0x7F800000);
    float w = nn_weight(D, sg.r2, (a.flags & kFlagExpoW) != 0);
    float sum = w;
    sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
    w = w / fmaxf(sum, 1e-12f);
    sI[t] = i; sW[t] = w;
    sRel[t * 3 + 0] = (i >= 0) ? __fsub_rn(q.x, sg.x) : 0.f;
    sRel[t * 3 + 1] = (i >= 0) ? __fsub_rn(q.y, sg.y) : 0.f;
    sRel[t * 3 + 2] = (i >= 0) ? __fsub_rn(q.z, sg.z) : 0.f;
    if (k == 0) {
      sPts[s * 4 + 0] = sg.x; sPts[s * 4 + 1] = sg.y; sPts[s * 4 + 2] = sg.z; sPts[s * 4 + 3] = sg.r2;
      sHas[s] = (cnt_p >= a.min_nn) ? 1 : 0;
    }
  } else if (t < TILE * K + C) {
    smem[L::oFb + t - TILE * K] = a.fb_col[t - TILE * K];    // the fallback vector (decoder.py:386-388): a late global load at the reduction otherwise
  }
  lds_barrier_dma();
  PSL_STAMP(1);
  if (t < TILE * K) a.ws.w[(size_t)p0 * K + t] = sW[t];     // saved for the backward; after the barrier (see nbr_stage_dma)
  f32x4 afn[4];
  if (relpos) {
#pragma unroll
    for (int j = 0; j < 4; ++j) afn[j] = ldsfrag(sWn, f1 + j * 4 + 0, lane);
  }

  TrunkRegs tw;
  // ---------------------------------------------------------------- phase F: colour features of the tile
  {
    const int row = 16 * wave + rl;            // (sample, neighbour) pair of this lane; 4 lanes (g) share a pair
    const int s = row >> 3;
    const int i = sI[row];
    const float wgt = sW[row];
    const size_t grow = (size_t)p0 * K + row;  // row of the per-pair save buffers
    f32x4 xf[2];
    {
      const float* frow = a.col_feats + (size_t)max(i, 0) * C + 4 * g;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(frow), v1 = *reinterpret_cast<const f32x4*>(frow + 16);
#pragma unroll
      for (int r = 0; r < 4; ++r) { xf[0][r] = (i >= 0) ? v0[r] : 0.f; xf[1][r] = (i >= 0) ? v1[r] : 0.f; }
    }
    f32x4 cc[2];
    if (relpos) {
      // F_theta input [sin(10) cos(10) | feat(32)] (decoder.py:371-378); this lane holds sin or cos of f = 2 s + (g >> 1)
      const float* __restrict__ Brel = M + MO(PI_C_BREL);
      const float rx = sRel[row * 3], ry = sRel[row * 3 + 1], rz = sRel[row * 3 + 2];
      f32x4 xe; float xe4 = 0.f;
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) {
        const int f = 2 * ks + (g >> 1);
        float sn, cs;
        fast_sincosf(fourier_phase(rx, ry, rz, Brel, ERF, f), sn, cs);
        const float v = (g & 1) ? cs : sn;
        if (ks < 4) xe[ks] = v; else xe4 = v;
        if (a.ws.n_x) a.ws.n_x[grow * NX + (g & 1) * ERF + f] = v;
      }
      if (a.ws.n_x) {
        *reinterpret_cast<f32x4*>(a.ws.n_x + grow * NX + ER + 4 * g) = xf[0];
        *reinterpret_cast<f32x4*>(a.ws.n_x + grow * NX + ER + 16 + 4 * g) = xf[1];
      }
      PSL_STAMP(2);
      // linear1 52 -> 128 in two halves of four output tiles; the fragments of step + 1 are in flight during step.
      // The softplus of a finished tile (~70 VALU instructions) is written next to the 16 MFMAs of a later step: a
      // wavefront issues about five VALU instructions in the shadow of every MFMA, so the activation costs no time.
      f32x4 hid[8];
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) hid[nt] = ldbias(WF, fbias(FL_N1), nt, g);
      auto activate = [&](int nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) hid[nt][r] = softplus100_nb(hid[nt][r]);
        if (a.ws.n_h1) *reinterpret_cast<f32x4*>(a.ws.n_h1 + grow * HC + nt * 16 + 4 * g) = hid[nt];
      };
      constexpr int f2 = ffirst(FL_N2);
      f32x4 a0n, a1n;
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        sched_fence();
        const int half = st >> 2, q = st & 3;
        f32x4 af[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) af[j] = afn[j];
        if (st < 7) {
          const int h2 = (st + 1) >> 2, q2 = (st + 1) & 3;
#pragma unroll
          for (int j = 0; j < 4; ++j) afn[j] = ldsfrag(sWn, f1 + (4 * h2 + j) * 4 + q2, lane);
        } else {
          a0n = ldsfrag(sWn, f2 + 0, lane); a1n = ldsfrag(sWn, f2 + 8 + 0, lane);     // linear2's first pair
        }
        if (q < 3) {
          const f32x4 b = (q == 0) ? xf[0] : (q == 1 ? xf[1] : xe);
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) hid[4 * half + j] = mfma16(af[j][r], b[r], hid[4 * half + j]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) hid[4 * half + j] = mfma16(af[j][0], xe4, hid[4 * half + j]);
        }
        if (half == 1) activate(q);            // tile q of the first half: complete since step 3
      }
      PSL_STAMP(3);
      PSL_STAMP(4);
      // linear2 128 -> 32; two fragments per hidden group, the next pair in flight
      f32x4 nf[2];
      nf[0] = ldbias(WF, fbias(FL_N2), 0, g); nf[1] = ldbias(WF, fbias(FL_N2), 1, g);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        sched_fence();
        const f32x4 a0 = a0n, a1 = a1n;
        if (q < 7) { a0n = ldsfrag(sWn, f2 + q + 1, lane); a1n = ldsfrag(sWn, f2 + 8 + q + 1, lane); }
        if (q < 4) activate(4 + q);            // second half: needed from q = 4 on
#pragma unroll
        for (int r = 0; r < 4; ++r) { nf[0] = mfma16(a0[r], hid[q][r], nf[0]); nf[1] = mfma16(a1[r], hid[q][r], nf[1]); }
      }
      sched_fence();
      PSL_STAMP(5);
      trunk_prologue_a(tw, WF, wave, lane, g);  // the trunk's first layer: in flight across the reduction and the barrier
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        if (a.ws.n_out) *reinterpret_cast<f32x4*>(a.ws.n_out + grow * C + nt * 16 + 4 * g) = nf[nt];
#pragma unroll
        for (int r = 0; r < 4; ++r) cc[nt][r] = group8_sum(__fmul_rn(wgt, nf[nt][r]));   // sum_k w_k F_theta(.)  (decoder.py:380-385)
      }
    } else {
      trunk_prologue_a(tw, WF, wave, lane, g);
      // plain interpolation sum_k w_k f[I_k] (decoder.py:380-385 without the neighbour MLP)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) cc[nt][r] = group8_sum(__fmul_rn(wgt, xf[nt][r]));
    }
    if ((rl & 7) == 0) {     // one lane per (sample, g) publishes the tile's colour features in fragment order
      const bool has = sHas[s] != 0;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x4 fb = *reinterpret_cast<const f32x4*>(smem + L::oFb + nt * 16 + 4 * g);      // decoder.py:386-388
        f32x4 c;
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = has ? cc[nt][r] : fb[r];
        *reinterpret_cast<f32x4*>(sCc + nt * FRAG + (g * 16 + s) * 4) = c;
        *reinterpret_cast<f32x4*>(a.ws.cc + (size_t)(p0 + s) * C + nt * 16 + 4 * g) = c;
      }
    }
  }
  PSL_STAMP(6);
  // In front of the barrier that closes phase F -- a wavefront that finished its F_theta rows early waits there for the slowest
  // one (stamps: ~6 k cycles for wave 0 of a tile that shares its CU) --: layer 1's fragments 4..7 are requested and the Fourier
  // features of the sample positions (decoder.py:8-37,411: frequencies f = 4 ks + g, sin and cos; they need phase 0 only) are
  // evaluated.  Both stood at the head of layer 0 behind the barrier (its "mfma" stamp: 7 k cycles against 2.5 k for a layer).
  trunk_prologue_b(tw, WF, wave, lane);
  float sn[5], cs[5];
  {
    const float x = sPts[rl * 4], y = sPts[rl * 4 + 1], z = sPts[rl * 4 + 2];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      const int f = 4 * ks + g;
      fast_sincosf(fourier_phase(x, y, z, a.Bcol, ECF, f), sn[ks], cs[ks]);
      if (wave == 0 && a.ws.c_emb) {
        a.ws.c_emb[(size_t)(p0 + rl) * EC + f] = sn[ks];
        a.ws.c_emb[(size_t)(p0 + rl) * EC + ECF + f] = cs[ks];
      }
    }
  }
  const float ob0 = M[MO(PI_C_OUT + 1) + 0], ob1 = M[MO(PI_C_OUT + 1) + 1], ob2 = M[MO(PI_C_OUT + 1) + 2];
  lds_barrier();
  PSL_STAMP(7);

  // ---------------------------------------------------------------- phase T: colour trunk, wave w = output tile w
  // (the layers themselves are trunk_layers_fwd, shared with k_trunk_fwd: weight slots refilled under the MFMAs, K-split head)
  {
    const f32x4 ccb0[1] = {*reinterpret_cast<const f32x4*>(sCc + lane * 4)}, ccb1[1] = {*reinterpret_cast<const f32x4*>(sCc + FRAG + lane * 4)};
    const f32x4 esn[1] = {f32x4{sn[0], sn[1], sn[2], sn[3]}}, ecs[1] = {f32x4{cs[0], cs[1], cs[2], cs[3]}};
    const float sn4[1] = {sn[4]}, cs4[1] = {cs[4]};
    trunk_layers_fwd<1>(a, WF, sH, smem + L::oOut, p0, tw, ccb0, ccb1, esn, ecs, sn4, cs4, ob0, ob1, ob2);
  }
}

// ================================================================================================ split colour stage (round 6)
// The fused colour tile above puts F_theta (43 % of a sample's FLOPs, wave-private: 16 (sample, neighbour) pairs per
// wavefront) and the trunk (barrier-coupled: 16 samples per 8 wavefronts) into one 512-thread workgroup, so the unit of
// load balance is the 16-sample tile: 5 000 samples = 313 tiles on 256 CUs, 57 CUs carry two and the launch takes their
// 45 us while 199 CUs idle from 30 us on (profiles/r05_block_trace.txt).  Split:
//   k_nbr_fwd   -- F_theta and the interpolation weights in units of ONE WAVEFRONT (16 pairs = 2 samples), four per workgroup,
//                  three workgroups per CU (48 KiB of weight fragments in LDS each): 2 500 units spread over 1 024 SIMDs;
//                  the one-wave geometry tiles ride in the same launch (four per workgroup, first in the grid).
//   k_trunk_fwd -- the colour trunk per 16-sample tile, no neighbour phase: the tile's features come from `cc`.
// Same products, same order of every sum as color_tile() except the 128 -> 3 output layer, which is K-split over the eight
// wavefronts (each contracts the 16 hidden channels it holds in registers).
constexpr int NBR_WG = 256;

// one wavefront: pairs 16 u .. 16 u + 15 = samples 2 u, 2 u + 1.  Lane (pair rl, g); the four lanes of a pair repeat the
// pair's scalar set-up (no LDS, no barrier except the one that publishes the weight fragments).
template <bool RELPOS>
__device__ __forceinline__ void nbr_unit_fwd(const DecodeArgs& a, const float* __restrict__ WF, float* smem, int u_in, int n_units, int sb) {
  const int lane = threadIdx.x & 63, rl = lane & 15, g = lane >> 4;
  PSL_STAMPB(50, sb);
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const bool act = u_in < n_units;
  const int u = min(u_in, n_units - 1);
  const float* __restrict__ M = a.master;
  const float* sWn = smem;
  constexpr int f1 = ffirst(FL_N1);
  // ---- set-up of this lane's pair: list entry, neighbour position, inverse-distance weight (decoder.py:143-160).  The list
  // entry heads a chain of two dependent memory trips (I -> position / feature row): it is requested FIRST, the 12 KiB of
  // weight-fragment DMA of this wave behind it (issued first, the DMA stood in the CU's one vector-memory queue in front of
  // the latency-critical loads of all twelve resident wavefronts)
  const int s = rl >> 3, k = rl & 7;
  const int ps = 2 * u + s;                       // sample slot (< Ppad)
  const int p = min(ps, a.P - 1);
  const int i = a.ws.I[(size_t)p * K + k];
  const int cnt_p = a.ws.cnt[p];
  const SampleGeom sg = sample_geom(a, p);
  sched_fence();
  if (RELPOS) nbr_stage_dma<kNbrFrags>(WF, smem, wave, NBR_WG / 64, lane);
  sched_fence();
  const float4 q = a.pos[max(i, 0)];
  f32x4 xf[2];
  {
    const float* frow = a.col_feats + (size_t)max(i, 0) * C + 4 * g;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(frow), v1 = *reinterpret_cast<const f32x4*>(frow + 16);
#pragma unroll
    for (int r = 0; r < 4; ++r) { xf[0][r] = (i >= 0) ? v0[r] : 0.f; xf[1][r] = (i >= 0) ? v1[r] : 0.f; }
  }
  const float D = (i >= 0) ? dist2(q.x, q.y, q.z, sg.x, sg.y, sg.z) : __int_as_float(0x7F800000);
  float wgt = nn_weight(D, sg.r2, (a.flags & kFlagExpoW) != 0);
  wgt = wgt / fmaxf(group8_sum(wgt), 1e-12f);
  const float rx = (i >= 0) ? __fsub_rn(q.x, sg.x) : 0.f, ry = (i >= 0) ? __fsub_rn(q.y, sg.y) : 0.f,
              rz = (i >= 0) ? __fsub_rn(q.z, sg.z) : 0.f;
  const bool has = cnt_p >= a.min_nn;
  const size_t grow = (size_t)u * 16 + rl;        // row of the per-pair save buffers = sample slot * 8 + k
  // Fourier features of the two sample positions for the trunk (decoder.py:8-37,411): 2 x 20 (sin, cos) pairs on the first 20
  // of a sample's 32 lanes; stored after the barrier
  float esn = 0.f, ecs = 0.f;
  const int ef = k * 4 + g;
  if (ef < ECF) fast_sincosf(fourier_phase(sg.x, sg.y, sg.z, a.Bcol, ECF, ef), esn, ecs);
  PSL_STAMPB(51, sb);
  if (RELPOS) lds_barrier_dma();                  // no global store in front of this barrier (see nbr_stage_dma)
  PSL_STAMPB(52, sb);
  if (act && g == 0) a.ws.w[grow] = wgt;
  if (act && ef < ECF) {
    // lane order of k_trunk_fwd: lane (sample, g) reads the ten values of frequencies 4 ks + g as one 40-byte run
    float* e2 = a.ws.c_emb2 + (size_t)ps * EC + (ef & 3) * 10 + (ef >> 2);
    e2[0] = esn; e2[5] = ecs;
    if (a.ws.c_emb) { a.ws.c_emb[(size_t)ps * EC + ef] = esn; a.ws.c_emb[(size_t)ps * EC + ECF + ef] = ecs; }
  }
  f32x4 cc[2];
  if constexpr (RELPOS) {
    f32x4 afn[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) afn[j] = ldsfrag(sWn, f1 + j * 4 + 0, lane);
    // F_theta input [sin(10) cos(10) | feat(32)] (decoder.py:371-378); this lane holds sin or cos of f = 2 ks + (g >> 1)
    const float* __restrict__ Brel = M + MO(PI_C_BREL);
    f32x4 xe; float xe4 = 0.f;
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      const int f = 2 * ks + (g >> 1);
      float sn, cs;
      fast_sincosf(fourier_phase(rx, ry, rz, Brel, ERF, f), sn, cs);
      const float v = (g & 1) ? cs : sn;
      if (ks < 4) xe[ks] = v; else xe4 = v;
      if (a.ws.n_x && act) a.ws.n_x[grow * NX + (g & 1) * ERF + f] = v;
    }
    if (a.ws.n_x && act) {
      *reinterpret_cast<f32x4*>(a.ws.n_x + grow * NX + ER + 4 * g) = xf[0];
      *reinterpret_cast<f32x4*>(a.ws.n_x + grow * NX + ER + 16 + 4 * g) = xf[1];
    }
    PSL_STAMPB(53, sb);
    f32x4 hid[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) hid[nt] = ldbias(WF, fbias(FL_N1), nt, g);
    auto activate = [&](int nt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) hid[nt][r] = softplus100_nb(hid[nt][r]);
      if (a.ws.n_h1 && act) *reinterpret_cast<f32x4*>(a.ws.n_h1 + grow * HC + nt * 16 + 4 * g) = hid[nt];
    };
    constexpr int f2 = ffirst(FL_N2);
    f32x4 a0n, a1n;
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      sched_fence();
      const int half = st >> 2, qq = st & 3;
      f32x4 af[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) af[j] = afn[j];
      if (st < 7) {
        const int h2 = (st + 1) >> 2, q2 = (st + 1) & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) afn[j] = ldsfrag(sWn, f1 + (4 * h2 + j) * 4 + q2, lane);
      } else {
        a0n = ldsfrag(sWn, f2 + 0, lane); a1n = ldsfrag(sWn, f2 + 8 + 0, lane);
      }
      if (qq < 3) {
        const f32x4 b = (qq == 0) ? xf[0] : (qq == 1 ? xf[1] : xe);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int j = 0; j < 4; ++j) hid[4 * half + j] = mfma16(af[j][r], b[r], hid[4 * half + j]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) hid[4 * half + j] = mfma16(af[j][0], xe4, hid[4 * half + j]);
      }
      if (half == 1) activate(qq);
    }
    PSL_STAMPB(54, sb);
    f32x4 nf[2];
    nf[0] = ldbias(WF, fbias(FL_N2), 0, g); nf[1] = ldbias(WF, fbias(FL_N2), 1, g);
#pragma unroll
    for (int qq = 0; qq < 8; ++qq) {
      sched_fence();
      const f32x4 a0 = a0n, a1 = a1n;
      if (qq < 7) { a0n = ldsfrag(sWn, f2 + qq + 1, lane); a1n = ldsfrag(sWn, f2 + 8 + qq + 1, lane); }
      if (qq < 4) activate(4 + qq);
#pragma unroll
      for (int r = 0; r < 4; ++r) { nf[0] = mfma16(a0[r], hid[qq][r], nf[0]); nf[1] = mfma16(a1[r], hid[qq][r], nf[1]); }
    }
    sched_fence();
    PSL_STAMPB(55, sb);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      if (a.ws.n_out && act) *reinterpret_cast<f32x4*>(a.ws.n_out + grow * C + nt * 16 + 4 * g) = nf[nt];
#pragma unroll
      for (int r = 0; r < 4; ++r) cc[nt][r] = group8_sum(__fmul_rn(wgt, nf[nt][r]));   // sum_k w_k F_theta(.)  (decoder.py:380-385)
    }
  } else {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) cc[nt][r] = group8_sum(__fmul_rn(wgt, xf[nt][r]));
  }
  if (act && k == 0) {       // one lane per (sample, g): the sample's colour features, fallback where it has too few neighbours
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const f32x4 fb = *reinterpret_cast<const f32x4*>(a.fb_col + nt * 16 + 4 * g);      // decoder.py:386-388
      f32x4 c;
#pragma unroll
      for (int r = 0; r < 4; ++r) c[r] = has ? cc[nt][r] : fb[r];
      *reinterpret_cast<f32x4*>(a.ws.cc + (size_t)ps * C + nt * 16 + 4 * g) = c;
    }
  }
  PSL_STAMPB(56, sb);
}

// grid: [0, geo_blocks) four one-wave geometry tiles each (the longest units start first), then four F_theta units each
template <bool RELPOS>
__global__ __launch_bounds__(NBR_WG, 3) void k_nbr_fwd(DecodeArgs a, const float* __restrict__ WF, int geo_blocks, int n_units) {
  __builtin_amdgcn_s_setprio(1);      // above the side-stream k-NN prefetch (see k_decode_fwd2)
  if (a.zero64 && blockIdx.x == 0 && threadIdx.x < 64) a.zero64[threadIdx.x] = 0.f;   // accumulators of the backward that follows
  extern __shared__ __attribute__((aligned(16))) float smem[];
  BlkTrace bt(a);
  const int b = (int)blockIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  if (b < geo_blocks) {
    const int p0 = (b * 4 + wave) * TILE;
    // a geometry tile is one long dependent chain (its wave is the launch's critical path): it wins the issue arbitration
    // against the F_theta wavefronts it shares a SIMD with
    __builtin_amdgcn_s_setprio(3);
    if (p0 < a.P) geo_tile<4>(a, WF, p0, false, false);
  } else {
    nbr_unit_fwd<RELPOS>(a, WF, smem, (b - geo_blocks) * 4 + wave, n_units, geo_blocks);
  }
  bt.done(a);
}

// ---- trunk: one 512-thread workgroup per tile of MT x 16 samples, wavefront w owns output columns 16 w .. 16 w + 15 of every
// layer for all MT sub-tiles (one A fragment feeds MT MFMAs).
//  * Weight fragments live in eight register slots; a slot is refilled with the NEXT layer's fragment right after the MFMAs
//    that consumed it, so that the 80 KB of weights a tile needs per layer stream in under its MFMAs instead of in a burst in
//    the epilogue (phase stamps of the fused tile: "epi" 2.4 k cycles per layer = eight wavefronts x 12 KB through the CU's one
//    vector-memory path, next to 2.2 k cycles of MFMA).
//  * MT = 2 where a CU would otherwise hold two 16-sample tiles (5 000 samples = 313 sub-tiles on 256 CUs: 57 double tiles +
//    199 single ones, ONE workgroup per CU): two co-resident tiles took 17.5-20.8 us against 11.9 us for one
//    (gpurun r06g block trace) -- they fetch every weight twice and meet at twice the barriers; the double tile fetches once.
template <int MT> struct TrunkLdsT { static constexpr int oH = 0, oOut = 2 * MT * 8 * FRAG, total = oOut + 8 * MT * TILE * 4; };
template <int MT>
__device__ __forceinline__ void trunk_tile_fwd(const DecodeArgs& a, const float* __restrict__ WF, float* smem, int p0) {
  using L = TrunkLdsT<MT>;
  float* sH = smem + L::oH;            // [2][MT][8][64][4] hidden tiles, fragment order, double buffered
  float* sOut = smem + L::oOut;        // [8][MT * 16][4] K-split partial colour logits
  const int t = threadIdx.x, lane = t & 63, rl = lane & 15, g = lane >> 4;
  const int nt = __builtin_amdgcn_readfirstlane(t >> 6);
  const float* __restrict__ M = a.master;
  PSL_STAMP(0);
  TrunkRegs R;
  trunk_prologue_a(R, WF, nt, lane, g);
  trunk_prologue_b(R, WF, nt, lane);
  // the tile's interpolated colour features are the B operands of the fc_c products as they lie in `cc`; the Fourier features
  // of the sample positions come from k_nbr_fwd in this lane's order (no dependent position chain in front of layer 0)
  f32x4 ccb0[MT], ccb1[MT], esn[MT], ecs[MT];
  float sn4[MT], cs4[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const size_t row = (size_t)(p0 + m * TILE + rl);
    ccb0[m] = *reinterpret_cast<const f32x4*>(a.ws.cc + row * C + 4 * g);
    ccb1[m] = *reinterpret_cast<const f32x4*>(a.ws.cc + row * C + 16 + 4 * g);
    const float2* e2 = reinterpret_cast<const float2*>(a.ws.c_emb2 + row * EC + g * 10);
    const float2 v0 = e2[0], v1 = e2[1], v2 = e2[2], v3 = e2[3], v4 = e2[4];
    esn[m] = f32x4{v0.x, v0.y, v1.x, v1.y}; sn4[m] = v2.x;
    ecs[m] = f32x4{v2.y, v3.x, v3.y, v4.x}; cs4[m] = v4.y;
  }
  const float ob0 = M[MO(PI_C_OUT + 1) + 0], ob1 = M[MO(PI_C_OUT + 1) + 1], ob2 = M[MO(PI_C_OUT + 1) + 2];
  PSL_STAMP(7);
  trunk_layers_fwd<MT>(a, WF, sH, sOut, p0, R, ccb0, ccb1, esn, ecs, sn4, cs4, ob0, ob1, ob2);
}

// the five trunk layers and the colour head of a tile of MT x 16 samples: shared by k_trunk_fwd and the fused colour tile
template <int MT>
__device__ __forceinline__ void trunk_layers_fwd(const DecodeArgs& a, const float* __restrict__ WF, float* sH, float* sOut, int p0, TrunkRegs& R,
                                                 const f32x4 (&ccb0)[MT], const f32x4 (&ccb1)[MT], const f32x4 (&esn)[MT], const f32x4 (&ecs)[MT],
                                                 const float (&sn4)[MT], const float (&cs4)[MT], float ob0, float ob1, float ob2) {
  const int t = threadIdx.x, lane = t & 63, rl = lane & 15, g = lane >> 4;
  const int nt = __builtin_amdgcn_readfirstlane(t >> 6);
  f32x4 (&w)[8] = R.w;
  f32x4 (&c)[2] = R.c;
  f32x4& bias = R.bias;
  f32x4& cbias = R.cbias;
  f32x4 hh[MT];
  auto layer = [&](auto I_) {
    constexpr int i = decltype(I_)::value;
    constexpr int nxt = i < 4 ? kTrunkL[i < 4 ? i + 1 : 4] : FL_COUT;
    constexpr int nq_n = kFLayers[nxt].ngroups;                         // fragments per output tile of the next layer
    const int nbase = ffirst(nxt) + (i < 4 ? nt * nq_n : 0);            // slot s <- fragment nbase + s
    const float* bufp = sH + ((i + 1) & 1) * MT * 8 * FRAG;             // the previous layer's hidden tiles
    f32x4 acc_a[MT], acc_b[MT], u[MT];
    f32x4 hq0[MT], hq1[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      acc_a[m] = bias; acc_b[m] = f32x4{0.f, 0.f, 0.f, 0.f}; u[m] = cbias;
      if (i != 0) {
        hq0[m] = *reinterpret_cast<const f32x4*>(bufp + m * 8 * FRAG + lane * 4);
        hq1[m] = *reinterpret_cast<const f32x4*>(bufp + m * 8 * FRAG + FRAG + lane * 4);
      }
    }
    sched_fence();
#pragma unroll
    for (int m = 0; m < MT; ++m) mma4(u[m], c[0], ccb0[m]);
    if (i == 0 || i == 3) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        mma4(acc_a[m], w[0], esn[m]);
        mma4(acc_b[m], w[2], ecs[m]);
        acc_a[m] = mfma16(w[1][0], sn4[m], acc_a[m]);
        acc_b[m] = mfma16(w[3][0], cs4[m], acc_b[m]);
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) mma4(u[m], c[1], ccb1[m]);
    sched_fence();
    if constexpr (i < 4) {
      c[0] = ldfrag(WF, ffirst(kTrunkF[i + 1]) + nt * 2 + 0, lane); c[1] = ldfrag(WF, ffirst(kTrunkF[i + 1]) + nt * 2 + 1, lane);
      bias = ldbias(WF, fbias(kTrunkL[i + 1]), nt, g); cbias = ldbias(WF, fbias(kTrunkF[i + 1]), nt, g);
    }
    if constexpr (i == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) w[q] = ldfrag(WF, nbase + q, lane);
    }
    if constexpr (i == 3) {       // hidden groups 4..7 of the skip layer go where its embedding fragments were
#pragma unroll
      for (int q = 0; q < 4; ++q) w[q] = ldfrag(WF, ffirst(FL_C3) + nt * 12 + 8 + q, lane);
    }
    if (i != 0) {
#pragma unroll
      for (int q = 0; q < 8; q += 2) {
        sched_fence();
        f32x4 h0[MT], h1[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          h0[m] = hq0[m]; h1[m] = hq1[m];
          if (q < 6) {
            hq0[m] = *reinterpret_cast<const f32x4*>(bufp + m * 8 * FRAG + (q + 2) * FRAG + lane * 4);
            hq1[m] = *reinterpret_cast<const f32x4*>(bufp + m * 8 * FRAG + (q + 3) * FRAG + lane * 4);
          }
        }
        const int s0 = (i == 3) ? (q < 4 ? 4 + q : q - 4) : q;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int m = 0; m < MT; ++m) { acc_a[m] = mfma16(w[s0][r], h0[m][r], acc_a[m]); acc_b[m] = mfma16(w[s0 + 1][r], h1[m][r], acc_b[m]); }
        sched_fence();
        if (i < 4) { w[s0] = ldfrag(WF, nbase + s0, lane); w[s0 + 1] = ldfrag(WF, nbase + s0 + 1, lane); }
        else if (s0 == 0) w[0] = ldfrag(WF, nbase + nt, lane);     // output layer, K-split: this wave's one fragment
      }
    }
    sched_fence();
    PSL_STAMP(10 + 3 * i);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      f32x4 y;
#pragma unroll
      for (int r = 0; r < 4; ++r) { y[r] = softplus100_nb(acc_a[m][r] + acc_b[m][r]); hh[m][r] = y[r] + u[m][r]; }
      if (a.ws.c_y) {
        const size_t o = ((size_t)i * a.ws.Ppad + p0 + m * TILE + rl) * HC + nt * 16 + 4 * g;
        *reinterpret_cast<f32x4*>(a.ws.c_y + o) = y;
        if (a.ws.c_hin) *reinterpret_cast<f32x4*>(a.ws.c_hin + o) = hh[m];
      }
      if (i < 4) *reinterpret_cast<f32x4*>(sH + (i & 1) * MT * 8 * FRAG + m * 8 * FRAG + nt * FRAG + lane * 4) = hh[m];
    }
    if (i < 4) {
      PSL_STAMP(10 + 3 * i + 1);
      lds_barrier();
      PSL_STAMP(10 + 3 * i + 2);
    }
  };
  layer(std::integral_constant<int, 0>{});
  layer(std::integral_constant<int, 1>{});
  layer(std::integral_constant<int, 2>{});
  layer(std::integral_constant<int, 3>{});
  layer(std::integral_constant<int, 4>{});
  // ---- output_linear 128 -> 3, K-split: this wave contracts the 16 hidden channels it holds (decoder.py:430-448)
  {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      f32x4 oa = {0.f, 0.f, 0.f, 0.f};
      mma4(oa, w[0], hh[m]);
      if (g == 0) *reinterpret_cast<f32x4*>(sOut + (nt * MT * TILE + m * TILE + rl) * 4) = oa;     // rows 0..2 of the padded tile
    }
    lds_barrier();
    if (t < MT * TILE && p0 + t < a.P) {
      const int pp = p0 + t;
      float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) {
        const float* po = sOut + (w8 * MT * TILE + t) * 4;
        r0 += po[0]; r1 += po[1]; r2 += po[2];
      }
      r0 += ob0; r1 += ob1; r2 += ob2;
      a.ws.out3[(size_t)pp * 4 + 0] = r0; a.ws.out3[(size_t)pp * 4 + 1] = r1; a.ws.out3[(size_t)pp * 4 + 2] = r2;
      if (a.flags & PSL_HAS_AFFINE) {  // out @ rot + trans (decoder.py:433-436)
        const float* A = a.affine;
        const float q0 = r0 * A[0] + r1 * A[3] + r2 * A[6] + A[9];
        const float q1 = r0 * A[1] + r1 * A[4] + r2 * A[7] + A[10];
        const float q2 = r0 * A[2] + r1 * A[5] + r2 * A[8] + A[11];
        r0 = q0; r1 = q1; r2 = q2;
      }
      if (!(a.flags & PSL_NO_SIGMOID)) { r0 = sigmoidf(r0); r1 = sigmoidf(r1); r2 = sigmoidf(r2); }
      a.ws.raw[(size_t)pp * 4 + 0] = r0; a.ws.raw[(size_t)pp * 4 + 1] = r1; a.ws.raw[(size_t)pp * 4 + 2] = r2;
    }
    PSL_STAMP(26);
  }
}

// workgroup b -> its tile: the first n2 workgroups take double tiles (32 samples), the rest single ones (see trunk_plan)
struct TrunkPlan { int n2, n1; };
TrunkPlan trunk_plan(int tiles) {      // tiles = 16-sample sub-tiles of the launch
  constexpr int kCUs = 256;
  if (tiles <= kCUs) return TrunkPlan{0, tiles};                      // one single tile per CU
  if (tiles <= 2 * kCUs) return TrunkPlan{tiles - kCUs, 2 * kCUs - tiles};   // one workgroup per CU, as few double tiles as that takes
  return TrunkPlan{tiles / 2, tiles & 1};                             // throughput regime: double tiles throughout
}
__global__ __launch_bounds__(WG, 2) void k_trunk_fwd(DecodeArgs a, const float* __restrict__ WF, int n2) {
  __builtin_amdgcn_s_setprio(1);      // above the side-stream k-NN prefetch (see k_decode_fwd2)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  BlkTrace bt(a);
  const int b = (int)blockIdx.x;
  if (b < n2) trunk_tile_fwd<2>(a, WF, smem, b * 2 * TILE);
  else trunk_tile_fwd<1>(a, WF, smem, (2 * n2 + (b - n2)) * TILE);
  bt.done(a);
}

// grid: [0, color_tiles) colour role (one tile per workgroup), then the geometry role: ONE WAVEFRONT per tile, each in a
// workgroup of its own so that the tiles spread over all CUs (eight tiles in one workgroup would sit on one CU and share
// its L1 port and its four SIMDs).  In the colour-stage launch the workgroup size is the colour role's 512: the seven
// idle wavefronts of a geometry workgroup exit at once.  COLOR = false is the stage-'geometry' launch (64-thread
// workgroups, geometry role only); two instantiations so that profiles tell them apart.
template <bool COLOR>
__global__ __launch_bounds__(COLOR ? WG : 64, COLOR ? 4 : 2) void k_decode_fwd2(DecodeArgs a, const float* __restrict__ WF, int color_tiles) {
  // above the mapper's side-stream k-NN prefetch (priority 0), whose waves share the SIMDs of this launch for 2 of every 7 ms of a mapped frame
  __builtin_amdgcn_s_setprio(1);
  if (a.zero64 && blockIdx.x == 0 && threadIdx.x < 64) a.zero64[threadIdx.x] = 0.f;   // accumulators of the backward that follows
  extern __shared__ __attribute__((aligned(16))) float smem[];
  BlkTrace bt(a);
  // workgroup -> (role, tile): the colour tiles first, then the geometry tiles.  (Alternating the two roles in the grid, so
  // that the one-wave geometry workgroups run next to colour tiles instead of after them, was measured in round 4: slower at
  // every launch size -- base mix 93.2 -> 79.5 frames/s, TUM yaml 8.85 -> 7.13, ScanNet 13.7 -> 10.9, Replica 35.4 -> 30.4: a
  // resident geometry wave holds 128 registers of one SIMD and keeps a second colour tile off the CU.)
  const int b = (int)blockIdx.x;
  const bool is_color = COLOR && b < color_tiles;
  const int tile = is_color ? b : b - color_tiles;
  if (is_color) {
    color_tile(a, WF, smem, tile * TILE);
  } else {
    if (threadIdx.x >= 64) return;
    const int p0 = tile * TILE;
    geo_tile<COLOR ? 3 : 4>(a, WF, p0, !COLOR, !COLOR);
  }
  bt.done(a);
}

// ------------------------------------------------------------------------------------------------ fragment buffers
// element e of the forward fragment buffer <- master blob (one thread per element; also used to build the inverse
// index that lets the Adam kernel keep the fragments in step with the master copy)
__device__ __forceinline__ int ffrag_source(int e) {     // master offset feeding element e, or -1 (zero padding)
  if (e >= kFFrags * FRAG) {   // bias area
    int li = 0;
#pragma unroll
    for (int j = 1; j < FL_COUNT; ++j) if (e >= fbias(j)) li = j;
    const int n = e - fbias(li);
    return n < kFLayers[li].N ? poff(kFLayers[li].pi + 1) + n : -1;
  }
  const int frag = e / FRAG, lane = (e % FRAG) >> 2, r = e & 3;
  int li = 0;
#pragma unroll
  for (int j = 1; j < FL_COUNT; ++j) if (frag >= ffirst(j)) li = j;
  const FLayer Ld = kFLayers[li];
  const int loc = frag - ffirst(li);
  const int nt = loc / Ld.ngroups, q = loc - nt * Ld.ngroups;
  const int out = nt * 16 + (lane & 15);
  const int ch = frag_chan(kFGroups[Ld.g0 + q], lane >> 4, r);
  if (out >= Ld.N || ch < 0 || ch >= Ld.K) return -1;
  return poff(Ld.pi) + out * Ld.K + ch;
}
__device__ __forceinline__ int bfrag_source(int e) {
  const int frag = e / FRAG, lane = (e % FRAG) >> 2, r = e & 3;
  int li = 0;
#pragma unroll
  for (int j = 1; j < BL_COUNT; ++j) if (frag >= bfirst(j)) li = j;
  const BLayer Ld = kBLayers[li];
  const int loc = frag - bfirst(li);
  const int it = loc / Ld.ngroups, q = loc - it * Ld.ngroups;
  const BTile bt = kBTiles[Ld.t0 + it];
  const int in = bt.in0 + (lane & 15);
  const int out = 16 * q + 4 * (lane >> 4) + r;
  if (in >= bt.lim || out >= Ld.N) return -1;
  return poff(Ld.pi) + out * Ld.K + in;
}

__global__ __launch_bounds__(256) void k_frag_repack(const float* __restrict__ master, float* __restrict__ wf, float* __restrict__ wb) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < kFFloats) { const int s = ffrag_source(e); wf[e] = s >= 0 ? master[s] : 0.f; }
  if (e < kBFloats) { const int s = bfrag_source(e); wb[e] = s >= 0 ? master[s] : 0.f; }
}
// inverse maps for the colour group: master element -> its forward / backward fragment element (or -1)
__global__ __launch_bounds__(256) void k_frag_index(int* __restrict__ wf_index, int* __restrict__ wb_index) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < kFFloats) { const int s = ffrag_source(e); if (s >= 0 && s < kColorFloats) wf_index[s] = e; }
  if (e < kBFloats) { const int s = bfrag_source(e); if (s >= 0 && s < kColorFloats) wb_index[s] = e; }
}

int build_frag_index(psl_ctx* ctx, hipStream_t s) {
  PSL_HIP(hipMemsetAsync(ctx->wf_index, 0xFF, sizeof(int) * kColorFloats, s));
  PSL_HIP(hipMemsetAsync(ctx->wb_index, 0xFF, sizeof(int) * kColorFloats, s));
  const int n = std::max(kFFloats, kBFloats);
  hipLaunchKernelGGL(k_frag_index, dim3((n + 255) / 256), dim3(256), 0, s, ctx->wf_index, ctx->wb_index);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}

int repack_frags(psl_ctx* ctx, const float* master, hipStream_t s) {
  const int n = std::max(kFFloats, kBFloats);
  hipLaunchKernelGGL(k_frag_repack, dim3((n + 255) / 256), dim3(256), 0, s, master, ctx->wf, ctx->wb);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}

int launch_decode_fwd2(psl_ctx* ctx, const DecodeArgs& a_in, hipStream_t s) {
  if (a_in.P <= 0) return PSL_OK;
  static unsigned long long* dbg = nullptr;
  static int dbg_on = -1;
  if (dbg_on < 0) { const char* e = getenv("PSL_DEBUG_PHASES"); dbg_on = (e && e[0] == '1') ? 1 : 0; }
  DecodeArgs a = a_in;
  if (dbg_on) {
    if (!dbg) PSL_HIP(hipMalloc(&dbg, 64 * sizeof(unsigned long long)));
    PSL_HIP(hipMemsetAsync(dbg, 0, 64 * sizeof(unsigned long long), s));
    a.dbg = dbg;
  }
  const size_t lds = sizeof(float) * ((a.flags & 0x10000) ? Fwd2Lds::total_nbr : Fwd2Lds::total);
  const int tiles = (a.P + TILE - 1) / TILE;
  static bool attr_set = false;
  if (!attr_set) {
    PSL_HIP(hipFuncSetAttribute((const void*)k_decode_fwd2<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(sizeof(float) * Fwd2Lds::total_nbr)));
    attr_set = true;
  }
  const bool color = a.flags & PSL_STAGE_COLOR;
  if (color && color_split_on(tiles)) {
    // split colour stage: F_theta + geometry role in wave-sized units, then the trunk per tile
    const bool relpos = (a.flags & 0x10000) != 0;
    const int n_units = tiles * (TILE / 2), f_blocks = (n_units + 3) / 4, geo_blocks = (tiles + 3) / 4;
    const size_t lds1 = relpos ? sizeof(float) * kNbrFrags * FRAG : 0;
    { int rc = blk_trace_begin(a, geo_blocks + f_blocks, s); if (rc) return rc; }
    if (relpos) PSL_KLAUNCH2(k_nbr_fwd<true>, true, false, dim3(geo_blocks + f_blocks), dim3(NBR_WG), lds1, s, a, (const float*)ctx->wf, geo_blocks, n_units);
    else PSL_KLAUNCH2(k_nbr_fwd<false>, true, false, dim3(geo_blocks + f_blocks), dim3(NBR_WG), lds1, s, a, (const float*)ctx->wf, geo_blocks, n_units);
    PSL_LAUNCH_CHECK();
    { int rc = blk_trace_end(a, "nbr_fwd", geo_blocks + f_blocks, -geo_blocks, NBR_WG); if (rc) return rc; }
    if (wave_trunk_on(tiles)) {      // throughput regime: one wavefront per tile (psl_trunk_wave.hip)
      { int rc = blk_trace_begin(a, (tiles + 3) / 4, s); if (rc) return rc; }
      int rc = launch_trunk_fwd_w(ctx, a, tiles, true, s);
      if (rc) return rc;
      { int rc2 = blk_trace_end(a, "trunk_fwd_w", (tiles + 3) / 4, (tiles + 3) / 4, 256); if (rc2) return rc2; }
      return PSL_OK;
    }
    const TrunkPlan tp = trunk_plan(tiles);
    const size_t lds2 = sizeof(float) * (tp.n2 ? TrunkLdsT<2>::total : TrunkLdsT<1>::total);
    { int rc = blk_trace_begin(a, tp.n2 + tp.n1, s); if (rc) return rc; }
    PSL_KLAUNCH2(k_trunk_fwd, false, true, dim3(tp.n2 + tp.n1), dim3(WG), lds2, s, a, (const float*)ctx->wf, tp.n2);
    PSL_LAUNCH_CHECK();
    { int rc = blk_trace_end(a, "trunk_fwd", tp.n2 + tp.n1, tp.n2, WG); if (rc) return rc; }
    if (dbg_on) {
      unsigned long long h[64];
      PSL_HIP(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
      fprintf(stderr, "[psl nbr_fwd P=%d] F unit: issue %llu wait+barrier %llu sincos %llu lin1 %llu lin2 %llu reduce+store %llu | total %llu\n", a.P, h[51] - h[50],
              h[52] - h[51], h[53] - h[52], h[54] - h[53], h[55] - h[54], h[56] - h[55], h[56] - h[50]);
      fprintf(stderr, "[psl trunk_fwd P=%d] set-up %llu |", a.P, h[7] - h[0]);
      unsigned long long prev = h[7];
      for (int i = 0; i < 5; ++i) {
        if (i < 4) { fprintf(stderr, " L%d: mfma %llu epi %llu bar %llu |", i, h[10 + 3 * i] - prev, h[11 + 3 * i] - h[10 + 3 * i], h[12 + 3 * i] - h[11 + 3 * i]); prev = h[12 + 3 * i]; }
        else { fprintf(stderr, " L4: mfma %llu |", h[22] - prev); prev = h[22]; }
      }
      fprintf(stderr, " epi+out %llu | total %llu\n", h[26] - prev, h[26] - h[0]);
    }
    return PSL_OK;
  }
  { int rc = blk_trace_begin(a, color ? 2 * tiles : tiles, s); if (rc) return rc; }
  if (color)
    PSL_KLAUNCH(k_decode_fwd2<true>, dim3(2 * tiles), dim3(WG), lds, s, a, (const float*)ctx->wf, tiles);
  else
    PSL_KLAUNCH(k_decode_fwd2<false>, dim3(tiles), dim3(64), 0, s, a, (const float*)ctx->wf, 0);
  PSL_LAUNCH_CHECK();
  { int rc = blk_trace_end(a, "fwd2", color ? 2 * tiles : tiles, color ? tiles : 0, color ? WG : 64); if (rc) return rc; }
  if (dbg_on) {
    unsigned long long h[64];
    PSL_HIP(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
    if (a.flags & PSL_STAGE_COLOR) {
      fprintf(stderr, "[psl fwd2 colour P=%d] phase0 %llu | F: gather+sincos %llu lin1 %llu softplus %llu lin2 %llu reduce %llu barrier %llu |",
              a.P, h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[7] - h[6]);
      unsigned long long prev = h[7];
      for (int i = 0; i < 5; ++i) {
        fprintf(stderr, " L%d: mfma %llu epi %llu bar %llu |", i, h[10 + 3 * i] - prev, h[11 + 3 * i] - h[10 + 3 * i], h[12 + 3 * i] - h[11 + 3 * i]);
        prev = h[12 + 3 * i];
      }
      fprintf(stderr, " out %llu | total %llu\n", h[26] - prev, h[26] - h[0]);
    } else {
      fprintf(stderr, "[psl fwd2 geo P=%d] nbr+weights %llu gather %llu sin %llu | L0 %llu L1 %llu L2 %llu L3 %llu L4 %llu | total %llu\n", a.P,
              h[33] - h[32], h[34] - h[33], h[35] - h[34], h[36] - h[35], h[37] - h[36], h[38] - h[37], h[39] - h[38], h[40] - h[39],
              h[41] - h[32]);
    }
  }
  return PSL_OK;
}

}  // namespace psl
