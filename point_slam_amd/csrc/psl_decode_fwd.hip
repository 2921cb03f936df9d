// Fused per-sample decode, forward: inverse-distance feature interpolation over the 8 neighbours,
// per-neighbour colour MLP F_theta, Fourier embeddings and the two small MLP decoders.
//
// Reference: MLP_geometry / MLP_color .get_feature_at_pos + .forward and POINT.forward
// (src/conv_onet/models/decoder.py:130-222, 341-449, 476-518), which the reference runs as ~900 ATen
// launches.  Here one 512-thread workgroup (8 wavefronts) owns a tile of 16 samples:
//   * gathers are coalesced 128 B feature rows; candidate neighbours / weights live in LDS;
//   * every linear layer is an exact-fp32 MFMA (v_mfma_f32_16x16x4_f32): the tile's activations
//     X[16][K] sit in LDS (A operand), the weights stream from L2 in a [K][N] layout (B operand);
//     the 8 waves split the 128 output columns (colour trunk) or the 8 neighbour row-tiles (F_theta);
//   * bias, activation, the `+ fc_c(c)` skip term and the activation save for the backward pass are
//     fused into the MFMA epilogue.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include "psl_decode.h"

namespace psl {

// MT = number of 16-row MFMA M-tiles per workgroup tile (TM = 16*MT sample slots, of which the first a.spt are
// real).  MT=2 halves the weight traffic and the exposed L2 latencies per sample; the host picks the smallest
// MT that puts the whole batch on the chip in ONE round of workgroups (256 CUs, 1 workgroup per CU).
template <int MT>
struct FwdLds {
  static constexpr int TM = 16 * MT;
  // The F_theta tiles (sXn, per-wave half-hidden sHn) and the decoder input tiles (sXg, sXc) live in the SAME
  // region: F_theta runs first, the Fourier embeddings are written afterwards.  69 KB (MT=1) / 77 KB (MT=2):
  // two workgroups per CU.
  static constexpr int LD_HH = 66;   // one 64-column half of F_theta's hidden layer, per wave [16][64]
  static constexpr int oI = 0, oW = oI + TM * K, oRel = oW + TM * K, oPts = oRel + TM * K * 3, oHas = oPts + TM * 4,
                       oCg = oHas + TM, oCc = oCg + TM * LD_CF, oOcc = oCc + TM * LD_CF, oOut = oOcc + TM,
                       oU = oOut + TM * 4;
  static constexpr int oXn = oU, oHn = oXn + 128 * LD_XN, nbr = 128 * LD_XN + 8 * 16 * LD_HH;
  static constexpr int oXg = oU, oXc = oXg + TM * LD_G, dec = TM * LD_G + TM * LD_C;
  static constexpr int total = oU + (nbr > dec ? nbr : dec);
};

// acc[mt](16 x 16 slice at column n0) = X[mt*16 .. +16][K] * W[K][N]; the B fragments are fetched once and shared
// by the MT row tiles; all of them are requested before the first MFMA (one exposed L2 latency per product).
template <int KDIM, int MT>
__device__ __forceinline__ void gemm16m(const float* Xs, int ldx, const float* __restrict__ W, int ldw, int n0,
                                        f32x4 (&acc)[MT]) {
  const int lane = threadIdx.x & 63;
  const float* xp = Xs + (lane & 15) * ldx + (lane >> 4);
  const float* wp = W + (size_t)(lane >> 4) * ldw + n0 + (lane & 15);
  constexpr int NK = KDIM / 4;
  float wv[NK];
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) wv[ks] = wp[(size_t)(4 * ks) * ldw];
  if constexpr (MT == 1) {
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};   // two chains hide the 40-cycle dependent latency
#pragma unroll
    for (int ks = 0; ks + 1 < NK; ks += 2) { a0 = mfma16(xp[4 * ks], wv[ks], a0); a1 = mfma16(xp[4 * ks + 4], wv[ks + 1], a1); }
    if (NK & 1) a0 = mfma16(xp[4 * (NK - 1)], wv[NK - 1], a0);
    acc[0] = a0 + a1;
  } else {
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = mfma16(xp[m * 16 * ldx + 4 * ks], wv[ks], acc[m]);
    }
  }
}

// The same product in two halves, so that the B fragments of the NEXT layer can be requested before the barriers
// and the activation of the current one (the weights do not depend on the data): fetch_b issues the loads,
// mma16m consumes them.
template <int KDIM>
__device__ __forceinline__ void fetch_b(const float* __restrict__ W, int ldw, int n0, float (&wv)[KDIM / 4]) {
  const int lane = threadIdx.x & 63;
  const float* wp = W + (size_t)(lane >> 4) * ldw + n0 + (lane & 15);
#pragma unroll
  for (int ks = 0; ks < KDIM / 4; ++ks) wv[ks] = wp[(size_t)(4 * ks) * ldw];
}
template <int KDIM, int MT>
__device__ __forceinline__ void mma16m(const float* Xs, int ldx, const float (&wv)[KDIM / 4], f32x4 (&acc)[MT]) {
  const int lane = threadIdx.x & 63;
  const float* xp = Xs + (lane & 15) * ldx + (lane >> 4);
  constexpr int NK = KDIM / 4;
  if constexpr (MT == 1) {
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks + 1 < NK; ks += 2) { a0 = mfma16(xp[4 * ks], wv[ks], a0); a1 = mfma16(xp[4 * ks + 4], wv[ks + 1], a1); }
    if (NK & 1) a0 = mfma16(xp[4 * (NK - 1)], wv[NK - 1], a0);
    acc[0] = a0 + a1;
  } else {
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = mfma16(xp[m * 16 * ldx + 4 * ks], wv[ks], acc[m]);
    }
  }
}

template <int MT>
__global__ __launch_bounds__(WG, 4) void k_decode_fwd(DecodeArgs a) {
  using L = FwdLds<MT>;
  constexpr int TM = L::TM;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* sI = (int*)(smem + L::oI);           // [TM][8]
  float* sW = smem + L::oW;                 // [TM][8]
  float* sRel = smem + L::oRel;             // [TM][8][3]
  float* sPts = smem + L::oPts;             // [TM][4]
  int* sHas = (int*)(smem + L::oHas);       // [TM]
  float* sCg = smem + L::oCg;               // [TM][34]
  float* sCc = smem + L::oCc;               // [TM][34]
  float* sXg = smem + L::oXg;               // [TM][130]
  float* sXc = smem + L::oXc;               // [TM][170]
  float* sOcc = smem + L::oOcc;             // [TM]
  float* sOut = smem + L::oOut;             // [TM][4]
  float* sXn = smem + L::oXn;               // [128][54]   (one 16-sample sub-tile at a time)
  float* sHn = smem + L::oHn;               // [8][16][66]  (aliases sXg/sXc together with sXn)

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int spt = a.spt;
  const int p0 = blockIdx.x * spt;
  const bool color = (a.flags & PSL_STAGE_COLOR) != 0;
  const bool relpos = color && (a.flags & 0x10000) != 0;  // internal bit: encode_rel_pos
  const float* __restrict__ M = a.master;
  const float* __restrict__ WT = a.wt;
  auto live = [&](int s) { return s < spt && p0 + s < a.P; };
  auto pidx = [&](int s) { return min(p0 + min(s, spt - 1), a.P - 1); };

  PSL_STAMP(0);
  // ---------------------------------------------------------------- phase 0: neighbours, weights
  if (t < TM * K) {
    const int s = t >> 3, k = t & 7;
    const int p = pidx(s);
    SampleGeom g = sample_geom(a, p);
    int i = a.ws.I[p * K + k];
    float nx = 0.f, ny = 0.f, nz = 0.f, D = __int_as_float(0x7F800000);
    if (i >= 0) {
      float4 q = a.pos[i];
      nx = q.x; ny = q.y; nz = q.z;
      D = dist2(nx, ny, nz, g.x, g.y, g.z);
    }
    // weights = 1/(D+1e-10); weights[D > r2] = 0; L1-normalise (decoder.py:152-160)
    float w = (D > g.r2) ? 0.f : 1.0f / (D + 1e-10f);
    float sum = w;
    sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
    w = w / fmaxf(sum, 1e-12f);
    sI[s * K + k] = i;
    sW[s * K + k] = w;
    sRel[(s * K + k) * 3 + 0] = (i >= 0) ? __fsub_rn(nx, g.x) : 0.f;
    sRel[(s * K + k) * 3 + 1] = (i >= 0) ? __fsub_rn(ny, g.y) : 0.f;
    sRel[(s * K + k) * 3 + 2] = (i >= 0) ? __fsub_rn(nz, g.z) : 0.f;
    if (live(s)) a.ws.w[p * K + k] = w;
    if (k == 0) {
      sPts[s * 4 + 0] = g.x; sPts[s * 4 + 1] = g.y; sPts[s * 4 + 2] = g.z; sPts[s * 4 + 3] = g.r2;
      sHas[s] = (a.ws.cnt[p] >= a.min_nn) ? 1 : 0;   // has_neighbors (decoder.py:150)
    }
  }
  lds_barrier();

  PSL_STAMP(1);
  // ---------------------------------------------------------------- phase 1: gathers (plain interpolation)
  for (int e = t; e < TM * C; e += WG) {
    const int s = e >> 5, ch = e & 31;
    // geometry feature: c = sum_k w_k f[I_k]; no-neighbour samples get the fallback vector (decoder.py:162-171)
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      int i = sI[s * K + k];
      if (i >= 0) acc = __fadd_rn(acc, __fmul_rn(sW[s * K + k], a.geo_feats[(size_t)i * C + ch]));
    }
    if (!sHas[s]) acc = a.fb_geo[ch];
    sCg[s * LD_CF + ch] = acc;
    if (live(s)) a.ws.cg[(size_t)(p0 + s) * C + ch] = acc;
    if (color && !relpos) {
      float ac = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        int i = sI[s * K + k];
        if (i >= 0) ac = __fadd_rn(ac, __fmul_rn(sW[s * K + k], a.col_feats[(size_t)i * C + ch]));
      }
      if (!sHas[s]) ac = a.fb_col[ch];
      sCc[s * LD_CF + ch] = ac;
      if (live(s)) a.ws.cc[(size_t)(p0 + s) * C + ch] = ac;
    }
  }
  PSL_STAMPF(8);
  lds_barrier();
  PSL_STAMPF(9);
  // ---------------------------------------------------------------- F_theta per neighbour, 16 samples at a time
  if (color) {
  if (relpos) {
    const float* Brel = M + MO(PI_C_BREL);
    for (int sub = 0; sub < MT; ++sub) {
      if (16 * sub >= spt) break;
      const int sb = 16 * sub;                  // first sample slot of this sub-tile
      // launder the weight pointers: they are loop-invariant, and LICM would hoist ALL weight loads of both
      // products out of the sub-tile loop (170 live registers -> scratch spills)
      // (an opaque zero OFFSET, not an opaque pointer: a laundered pointer loses its address space and every
      // weight load becomes a flat_load that the LDS waits -- lgkmcnt -- then also wait for)
      int opaque0 = 0;
      asm volatile("" : "+s"(opaque0));
      const float* WTs = WT + opaque0; const float* Ms = M + opaque0;
      // F_theta input rows [sin(10) cos(10) | feat(32)] for the 128 (sample, neighbour) pairs (decoder.py:371-378)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int e = t + WG * j;
        int row = e >> 5, ch = e & 31;
        int i = sI[sb * K + row];
        float v = (i >= 0) ? a.col_feats[(size_t)i * C + ch] : 0.f;
        sXn[row * LD_XN + ER + ch] = v;
      }
      for (int e = t; e < 128 * ERF; e += WG) {
        int row = e / ERF, f = e - row * ERF;
        const float* rl = sRel + (sb * K + row) * 3;
        float sn, cs;
        fast_sincosf(fourier_phase(rl[0], rl[1], rl[2], Brel, ERF, f), sn, cs);
        sXn[row * LD_XN + f] = sn;
        sXn[row * LD_XN + ERF + f] = cs;
      }
      PSL_STAMPF(10);
      lds_barrier();
      PSL_STAMPF(11);
      if (a.ws.n_x) {   // each wave saves its own 16 rows, one 208 B row per store
        float* nx = a.ws.n_x + ((size_t)(p0 + sb) * K + 16 * wave) * NX;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (lane < NX && live(sb + ((16 * wave + r) >> 3))) nx[r * NX + lane] = sXn[(16 * wave + r) * LD_XN + lane];
      }
      PSL_STAMPF(12);
      float* Hw = sHn + wave * 16 * L::LD_HH;
      const float* Xw = sXn + wave * 16 * LD_XN;
      const int g = lane >> 4, colw = lane & 15;
      // hidden layer in two 64-column halves: linear1 -> softplus -> (wave-private LDS tile) -> partial linear2
      f32x4 acc2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        f32x4 acc[4];
        gemm16_multi<NX, 4>(Xw, LD_XN, WTs + wtoff(WT_C_N1) + 64 * half, HC, acc);
        PSL_STAMPF(13 + 3 * half);
        if (half) wave_lds_sync();
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const int col = 64 * half + 16 * nt + colw;
          float b = Ms[MO(PI_C_N1 + 1) + col];
          f32x4 hv;
#pragma unroll
          for (int r = 0; r < 4; ++r) hv[r] = softplus100(acc[nt][r] + b);
          frag_store(Hw, L::LD_HH, 16 * nt, hv);
          if (a.ws.n_h1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              int row = 16 * wave + 4 * g + r;
              if (live(sb + (row >> 3))) a.ws.n_h1[((size_t)(p0 + sb) * K + row) * HC + col] = hv[r];
            }
          }
        }
        PSL_STAMPF(14 + 3 * half);
        wave_lds_sync();
        f32x4 part[2];
        gemm16_multi<64, 2>(Hw, L::LD_HH, WTs + wtoff(WT_C_N2) + 64 * half * C, C, part);
        acc2[0] += part[0]; acc2[1] += part[1];
        PSL_STAMPF(15 + 3 * half);
      }
      {
        const int s = sb + 2 * wave + (g >> 1);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          float b = Ms[MO(PI_C_N2 + 1) + 16 * nt + colw];
          float part = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float nf = acc2[nt][r] + b;
            int row = 16 * wave + 4 * g + r;
            if (a.ws.n_out && live(sb + (row >> 3))) a.ws.n_out[((size_t)(p0 + sb) * K + row) * C + 16 * nt + colw] = nf;
            part = __fadd_rn(part, __fmul_rn(sW[s * K + 4 * (g & 1) + r], nf));
          }
          float tot = part + __shfl_xor(part, 16);
          if ((g & 1) == 0) {
            float c = sHas[s] ? tot : a.fb_col[16 * nt + colw];
            sCc[s * LD_CF + 16 * nt + colw] = c;
            if (live(s)) a.ws.cc[(size_t)(p0 + s) * C + 16 * nt + colw] = c;
          }
        }
      }
      PSL_STAMPF(19);
      lds_barrier();        // sXn / sHn are reused by the next sub-tile
    }
    if (MT * 16 > spt) {    // slots of sub-tiles that were skipped: defined (zero) colour features
      for (int e = t; e < TM * C; e += WG) { int s = e >> 5; if (s >= ((spt + 15) & ~15)) sCc[s * LD_CF + (e & 31)] = 0.f; }
    }
  }
  }
  lds_barrier();      // sXn/sHn are dead from here on: their region now receives the decoder inputs
  PSL_STAMP(2);
  // ---------------------------------------------------------------- phase 2: Fourier embeddings of p
  {
    const float* Bg = M + MO(PI_G_B);
    for (int e = t; e < TM * EGP; e += WG) {
      int s = e / EGP, f = e - s * EGP;
      float v = 0.f;
      if (f < EG) v = fast_sinf(fourier_phase(sPts[s * 4], sPts[s * 4 + 1], sPts[s * 4 + 2], Bg, EG, f));
      sXg[s * LD_G + f] = v;
    }
    if (color) {
      for (int e = t; e < TM * ECF; e += WG) {
        int s = e / ECF, f = e - s * ECF;
        float sn, cs;
        fast_sincosf(fourier_phase(sPts[s * 4], sPts[s * 4 + 1], sPts[s * 4 + 2], a.Bcol, ECF, f), sn, cs);
        sXc[s * LD_C + f] = sn;
        sXc[s * LD_C + ECF + f] = cs;
        if (live(s) && a.ws.c_emb) {
          a.ws.c_emb[(size_t)(p0 + s) * EC + f] = sn; a.ws.c_emb[(size_t)(p0 + s) * EC + ECF + f] = cs;
        }
      }
    }
  }
  lds_barrier();

  PSL_STAMP(3);
  // ---------------------------------------------------------------- phase 3: geometry MLP, K-split over 8 waves
  // h = relu(W_i h + b_i) + (Wc_i c + bc_i); after block 2 the embedding is re-attached (decoder.py:207-219).
  // The layer is only 32 columns wide (2 MFMA column tiles): wave w takes column tile w&1 and the K-quarter w>>1,
  // the four partial tiles meet in LDS and one thread per element finishes the layer (bias, ReLU, skip term, save).
  // In the colour stage these two steps ride inside the colour trunk's layer loop and share its two barriers; the
  // partial tiles live in the part of the F_theta region that the decoder inputs leave free.
  float* sPm = sXc + TM * LD_C;             // [4][TM][34] partial sums of W_i h
  float* sPf = sPm + 4 * TM * LD_CF;        // [TM][34]    Wc_i c
  auto geo_partial = [&](int i) {
    const int n0g = 16 * (wave & 1), kq = wave >> 1;
    f32x4 pm[MT];
    if (i == 0) gemm16m<EGP / 4, MT>(sXg + (EGP / 4) * kq, LD_G, WT + wtoff(WT_G_L + 0) + (EGP / 4) * kq * HG, HG, n0g, pm);
    else if (i == 3) gemm16m<(EGP + HG) / 4, MT>(sXg + ((EGP + HG) / 4) * kq, LD_G,
                                                 WT + wtoff(WT_G_L + 3) + ((EGP + HG) / 4) * kq * HG, HG, n0g, pm);
    else gemm16m<HG / 4, MT>(sXg + EGP + (HG / 4) * kq, LD_G, WT + wtoff(WT_G_L + i) + (HG / 4) * kq * HG, HG, n0g, pm);
#pragma unroll
    for (int m = 0; m < MT; ++m) frag_store(sPm + (kq * TM + 16 * m) * LD_CF, LD_CF, n0g, pm[m]);
    if (kq == 0) {
      f32x4 u[MT];
      gemm16m<C, MT>(sCg, LD_CF, WT + wtoff(WT_G_FCC + i), HG, n0g, u);
#pragma unroll
      for (int m = 0; m < MT; ++m) frag_store(sPf + 16 * m * LD_CF, LD_CF, n0g, u[m]);
    }
  };
  auto geo_finish = [&](int i) {
    for (int e = t; e < TM * HG; e += WG) {
      const int sm = e >> 5, c = e & 31;
      const float z = (sPm[sm * LD_CF + c] + sPm[(TM + sm) * LD_CF + c]) +
                      (sPm[(2 * TM + sm) * LD_CF + c] + sPm[(3 * TM + sm) * LD_CF + c]);
      const float y = fmaxf(z + M[MO(PI_G_L + 2 * i + 1) + c], 0.f);
      const float h = y + (sPf[sm * LD_CF + c] + M[MO(PI_G_FCC + 2 * i + 1) + c]);
      sXg[sm * LD_G + EGP + c] = h;
      if (a.ws.g_y && live(sm)) a.ws.g_y[((size_t)i * a.ws.Ppad + p0 + sm) * HG + c] = y;
    }
  };
  if (!color) {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      geo_partial(i);
      lds_barrier();
      geo_finish(i);
      lds_barrier();
    }
  }

  PSL_STAMP(4);
  if (color) {
    PSL_STAMP(5);
    // -------------------------------------------------------------- phase 5: colour trunk, 8 waves x 16 columns
    {
      const int n0 = 16 * wave, g4 = 4 * (lane >> 4), colw = lane & 15;
      // MT == 1: the weight fragments of layer i+1 are requested right after the MFMAs of layer i were issued, i.e.
      // their L2 latency elapses behind the activation, the two barriers and the activation saves (the 32-row tile
      // has no registers to spare for this).
      constexpr bool kPrefetch = (MT == 1);
      float wA[(EC + HC) / 4];
      if constexpr (kPrefetch) {
        float (&w0)[EC / 4] = reinterpret_cast<float (&)[EC / 4]>(wA);
        fetch_b<EC>(WT + wtoff(WT_C_L + 0), HC, n0, w0);
      }
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        f32x4 acc[MT], u[MT];
        if constexpr (kPrefetch) {
          if (i == 0) mma16m<EC, MT>(sXc, LD_C, reinterpret_cast<float (&)[EC / 4]>(wA), acc);
          else if (i == 3) mma16m<EC + HC, MT>(sXc, LD_C, wA, acc);
          else mma16m<HC, MT>(sXc + EC, LD_C, reinterpret_cast<float (&)[HC / 4]>(wA), acc);
          gemm16m<C, MT>(sCc, LD_CF, WT + wtoff(WT_C_FCC + i), HC, n0, u);
          if (i + 1 < 5) {
            if (i + 1 == 3) fetch_b<EC + HC>(WT + wtoff(WT_C_L + 3), HC, n0, wA);
            else fetch_b<HC>(WT + wtoff(WT_C_L + i + 1), HC, n0, reinterpret_cast<float (&)[HC / 4]>(wA));
          }
        } else {
          if (i == 0) gemm16m<EC, MT>(sXc, LD_C, WT + wtoff(WT_C_L + 0), HC, n0, acc);
          else if (i == 3) gemm16m<EC + HC, MT>(sXc, LD_C, WT + wtoff(WT_C_L + 3), HC, n0, acc);
          else gemm16m<HC, MT>(sXc + EC, LD_C, WT + wtoff(WT_C_L + i), HC, n0, acc);
          gemm16m<C, MT>(sCc, LD_CF, WT + wtoff(WT_C_FCC + i), HC, n0, u);
        }
        geo_partial(i);
        float b = M[MO(PI_C_L + 2 * i + 1) + n0 + colw];
        float bc = M[MO(PI_C_FCC + 2 * i + 1) + n0 + colw];
        f32x4 y[MT], h[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            y[m][r] = softplus100(acc[m][r] + b);
            h[m][r] = y[m][r] + (u[m][r] + bc);
          }
        lds_barrier();
        geo_finish(i);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          frag_store(sXc + m * 16 * LD_C + EC, LD_C, n0, h[m]);
          if (a.ws.c_y) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              int s = 16 * m + g4 + r;
              if (live(s)) {
                a.ws.c_y[((size_t)i * a.ws.Ppad + p0 + s) * HC + n0 + colw] = y[m][r];
                if (a.ws.c_hin) a.ws.c_hin[((size_t)i * a.ws.Ppad + p0 + s) * HC + n0 + colw] = h[m][r];
              }
            }
          }
        }
        lds_barrier();
      }
      for (int e = t; e < TM * 3 * 8; e += WG) {  // output_linear 128 -> 3: 8 lanes per output, 16 inputs each
        const int o3 = e >> 3, part = e & 7;
        const int s = o3 / 3, j = o3 - 3 * s;
        const float* wo = M + MO(PI_C_OUT) + j * HC + 16 * part;
        const float* xs = sXc + s * LD_C + EC + 16 * part;
        float o = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) o = fmaf(xs[k], wo[k], o);
        o += __shfl_xor(o, 1); o += __shfl_xor(o, 2); o += __shfl_xor(o, 4);
        if (part == 0) sOut[s * 4 + j] = o + M[MO(PI_C_OUT + 1) + j];
      }
    }
  }
  for (int e = t; e < TM * 8; e += WG) {  // geometry output_linear 32 -> 1: 8 lanes per sample, 4 inputs each
    const int sm = e >> 3, part = e & 7;
    const float* wo = M + MO(PI_G_OUT) + 4 * part;
    const float* xs = sXg + sm * LD_G + EGP + 4 * part;
    float o = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) o = fmaf(xs[k], wo[k], o);
    o += __shfl_xor(o, 1); o += __shfl_xor(o, 2); o += __shfl_xor(o, 4);
    if (part == 0) sOcc[sm] = o + M[MO(PI_G_OUT + 1)];
  }
  lds_barrier();
  PSL_STAMP(6);
  // ------------------------------------------------------------------ raw = [rgb, occ]
  if (t < TM && live(t)) {
    int p = p0 + t;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
    if (color) {
      float r0 = sOut[t * 4], r1 = sOut[t * 4 + 1], r2 = sOut[t * 4 + 2];
      a.ws.out3[(size_t)p * 4 + 0] = r0; a.ws.out3[(size_t)p * 4 + 1] = r1; a.ws.out3[(size_t)p * 4 + 2] = r2;
      if (a.flags & PSL_HAS_AFFINE) {  // out @ rot + trans (decoder.py:433-436)
        const float* A = a.affine;
        float q0 = r0 * A[0] + r1 * A[3] + r2 * A[6] + A[9];
        float q1 = r0 * A[1] + r1 * A[4] + r2 * A[7] + A[10];
        float q2 = r0 * A[2] + r1 * A[5] + r2 * A[8] + A[11];
        r0 = q0; r1 = q1; r2 = q2;
      }
      if (!(a.flags & PSL_NO_SIGMOID)) { r0 = sigmoidf(r0); r1 = sigmoidf(r1); r2 = sigmoidf(r2); }
      o0 = r0; o1 = r1; o2 = r2;
    }
    // raw[~point_mask, -1] = -100 (Renderer.py:189-190)
    float occ = sHas[t] ? sOcc[t] : -100.0f;
    reinterpret_cast<float4*>(a.ws.raw)[p] = make_float4(o0, o1, o2, occ);
  }
}

// forward-layout weights from the master blob (one thread per padded element)
__global__ __launch_bounds__(256) void k_repack(const float* __restrict__ master, float* __restrict__ wt) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= kWtFloats) return;
  int li = 0;
#pragma unroll
  for (int j = 1; j < kNumWt; ++j) if (e >= wtoff(j)) li = j;
  const WtDesc d = kWt[li];
  int loc = e - wtoff(li);
  int kp = loc / d.N, n = loc - kp * d.N;
  int k = -1;
  if (kp < d.split) k = kp;
  else if (kp >= d.split + d.gap) k = kp - d.gap;
  float v = 0.f;
  if (k >= 0 && k < d.Kin) v = master[poff(d.pi) + n * d.Kin + k];
  wt[e] = v;
}

// inverse of k_repack for the colour decoder: wt_index[master element] = forward-layout element
__global__ __launch_bounds__(256) void k_wt_index(int* __restrict__ wt_index) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= wtoff(WT_G_FCC)) return;          // colour entries come first in kWt
  int li = 0;
#pragma unroll
  for (int j = 1; j < kNumWt; ++j) if (e >= wtoff(j)) li = j;
  const WtDesc d = kWt[li];
  int loc = e - wtoff(li);
  int kp = loc / d.N, n = loc - kp * d.N;
  int k = -1;
  if (kp < d.split) k = kp;
  else if (kp >= d.split + d.gap) k = kp - d.gap;
  if (k >= 0 && k < d.Kin) wt_index[poff(d.pi) + n * d.Kin + k] = e;
}

int build_wt_index(psl_ctx* ctx, hipStream_t s) {
  PSL_HIP(hipMemsetAsync(ctx->wt_index, 0xFF, sizeof(int) * kColorFloats, s));     // -1
  hipLaunchKernelGGL(k_wt_index, dim3((wtoff(WT_G_FCC) + 255) / 256), dim3(256), 0, s, ctx->wt_index);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}

int repack_weights(psl_ctx* ctx, const float* master, hipStream_t s) {
  hipLaunchKernelGGL(k_repack, dim3((kWtFloats + 255) / 256), dim3(256), 0, s, master, ctx->wt);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}

// Tile geometry: the smallest samples-per-tile (<= 32) that keeps the whole batch resident in ONE round of
// workgroups: 256 CUs x 2 workgroups per CU (both instantiations need < 80 KB of LDS and <= 128 VGPRs).
void choose_tile(int P, int& mt, int& spt) {
  const int kSlots = 512;
  if ((P + 15) / 16 <= kSlots) { mt = 1; spt = 16; return; }
  mt = 2;
  spt = std::min(32, std::max(17, (P + kSlots - 1) / kSlots));
}

template <int MT>
static int launch_fwd_t(const DecodeArgs& a, hipStream_t s) {
  static bool attr_set = false;
  const size_t lds = sizeof(float) * FwdLds<MT>::total;
  if (!attr_set) {
    PSL_HIP(hipFuncSetAttribute((const void*)k_decode_fwd<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  int tiles = (a.P + a.spt - 1) / a.spt;
  hipLaunchKernelGGL(k_decode_fwd<MT>, dim3(tiles), dim3(WG), lds, s, a);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}

int launch_decode_fwd(const DecodeArgs& a, hipStream_t s) {
  if (a.P <= 0) return PSL_OK;
  static unsigned long long* dbg = nullptr;
  static int dbg_on = -1;
  if (dbg_on < 0) { const char* e = getenv("PSL_DEBUG_PHASES"); dbg_on = (e && e[0] == '1') ? 1 : 0; }
  DecodeArgs a2 = a;
  int mt;
  choose_tile(a.P, mt, a2.spt);
  if (dbg_on) {
    if (!dbg) PSL_HIP(hipMalloc(&dbg, 64 * sizeof(unsigned long long)));
    a2.dbg = dbg;
  }
  static int nosave = -1;     // timing experiments only (results of the backward pass are garbage): PSL_DEBUG_NOSAVE
  if (nosave < 0) { const char* e = getenv("PSL_DEBUG_NOSAVE"); nosave = e ? atoi(e) : 0; }
  if (nosave & 1) a2.ws.n_h1 = nullptr;
  if (nosave & 2) { a2.ws.n_x = nullptr; a2.ws.n_out = nullptr; }
  if (nosave & 4) { a2.ws.c_y = nullptr; a2.ws.c_emb = nullptr; }
  if (nosave & 8) a2.ws.g_y = nullptr;
  int rc = (mt == 1) ? launch_fwd_t<1>(a2, s) : launch_fwd_t<2>(a2, s);
  if (rc) return rc;
  if (dbg_on) {
    unsigned long long h[32];
    PSL_HIP(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
    if (a.flags & PSL_STAGE_COLOR) {
      fprintf(stderr, "[psl fwd fine] p1: gather %llu bar %llu nbr-in %llu bar %llu nx-store %llu | g1a %llu act %llu g2a %llu g1b %llu act %llu g2b %llu epi %llu\n",
              h[8] - h[1], h[9] - h[8], h[10] - h[9], h[11] - h[10], h[12] - h[11], h[13] - h[12], h[14] - h[13],
              h[15] - h[14], h[16] - h[15], h[17] - h[16], h[18] - h[17], h[19] - h[18]);
      fprintf(stderr, "[psl fwd fine] trunk layer1: gemm %llu fcc %llu act %llu bar %llu store %llu bar %llu | out-linear %llu\n",
              h[21] - h[20], h[22] - h[21], h[23] - h[22], h[24] - h[23], h[25] - h[24], h[26] - h[25], h[6] - h[27]);
    }
    fprintf(stderr, "[psl fwd P=%d flags=%x spt=%d] cycles: p0 %llu p1 %llu p2 %llu geo %llu nbr %llu trunk %llu | total %llu\n",
            a.P, a.flags, a2.spt, h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[6] - h[0]);
  }
  return PSL_OK;
}

}  // namespace psl
