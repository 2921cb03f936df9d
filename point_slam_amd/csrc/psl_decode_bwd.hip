// Fused per-sample decode, backward: gradients w.r.t. the interpolated features (scatter-added to the
// neural-point feature rows), w.r.t. the sample positions (-> camera pose) and the per-layer dZ / G
// tiles that the parameter-gradient GEMM (psl_dw.hip) contracts over all samples.
//
// Mirrors autograd through MLP_color / MLP_geometry / get_feature_at_pos
// (src/conv_onet/models/decoder.py:130-222,341-449), which the reference executes as ~1000 ATen
// backward launches incl. a dense [N,32] zero-fill + index_add per gather.
// Same tiling as the forward: 16 samples per 512-thread workgroup, every dX = dZ * W product is an
// exact-fp32 MFMA whose B operand is the torch-layout weight itself ([out][in] row-major).
#include <cstdio>
#include <cstdlib>
#include "psl_decode.h"

namespace psl {

constexpr int LD_DE = 98;   // geo d_emb tile [16][96]  (98/2 odd)
constexpr int LD_DEC = 50;  // colour d_emb tile [16][40..48]
constexpr int LD_Z1 = 66;   // half of F_theta's dz1 tile, per wave [16][64]
constexpr int LD_XE = 22;   // rel-pos part of F_theta's dX1 tile, per wave [16][20]
constexpr int PW = 16 * LD_CF + 16 * LD_Z1 + 16 * LD_XE;   // per-wave F_theta scratch

// LDS plan.  The per-wave F_theta scratch ALIASES the trunk tiles sG/sDZ (dead between the colour-trunk and the
// geometry backward), F_theta's dz1 is processed in two 64-column halves and its feature gradients are scattered
// straight from the MFMA registers, and the position-gradient tiles exist only in the PTSG instantiation: the
// mapper instantiation needs 70 KB, so TWO workgroups share a CU and a 5 000-sample batch (313 tiles) is resident
// in one round instead of two.
template <bool PTSG>
struct BwdLds {
  static constexpr int oI = 0, oW = 128, oRel = 256, oPts = 640, oHas = 704, oDB = 720, oAff = 752, oDO = 768,
                       oDCg = 832, oDCc = oDCg + 16 * LD_CF, oGg = oDCc + 16 * LD_CF, oDZg = oGg + 16 * LD_CF,
                       oP = oDZg + 16 * LD_CF;
  static constexpr int oGW = oP, oDP = oGW + (PTSG ? 128 : 0), oDEg = oDP + (PTSG ? 64 : 0),
                       oDEc = oDEg + (PTSG ? 16 * LD_DE : 0), oU = oDEc + (PTSG ? 16 * LD_DEC : 0);
  static constexpr int oG = oU, oDZ = oG + 16 * LD_HN, trunk = 2 * 16 * LD_HN, waves = 8 * PW;
  static constexpr int total = oU + (trunk > waves ? trunk : waves);
};

struct BwdOut {
  float* g_geo; float* g_col; const int* row_map;
  float* g_brel;     // [30] accumulated with atomics (pre-zeroed)
  float* g_affine;   // [12] accumulated with atomics (pre-zeroed)
};

// read a C/D-layout fragment from an LDS tile
__device__ __forceinline__ f32x4 frag_load(const float* src, int ld, int n0) {
  const int lane = threadIdx.x & 63;
  const float* p = src + (4 * (lane >> 4)) * ld + n0 + (lane & 15);
  f32x4 v;
  v[0] = p[0]; v[1] = p[ld]; v[2] = p[2 * ld]; v[3] = p[3 * ld];
  return v;
}

template <bool PTSG>
__global__ __launch_bounds__(WG, PTSG ? 2 : 4) void k_decode_bwd(DecodeArgs a, BwdOut o) {
  using L = BwdLds<PTSG>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* sI = (int*)(smem + L::oI);           // [16][8]
  float* sW = smem + L::oW;                 // [16][8] normalised weights
  float* sRel = smem + L::oRel;             // [16][8][3]
  float* sPts = smem + L::oPts;             // [16][4]
  int* sHas = (int*)(smem + L::oHas);       // [16]
  float* sDB = smem + L::oDB;               // [32]    dL/dB_rel (30 used)
  float* sAff = smem + L::oAff;             // [16]    dL/d affine (12 used)
  float* sDO = smem + L::oDO;               // [16][4] dL/d colour logits (pre-affine)
  float* sDCg = smem + L::oDCg;             // [16][34] dL/d c_geo
  float* sDCc = smem + L::oDCc;             // [16][34] dL/d c_col
  float* sGg = smem + L::oGg;               // [16][34] geometry dL/dh tile (fused geometry backward, mapper)
  float* sDZg = smem + L::oDZg;             // [16][34] geometry dz tile
  float* sGW = smem + L::oGW;               // [16][8]  dL/dw            (PTSG only)
  float* sDP = smem + L::oDP;               // [16][4]  dL/dp            (PTSG only)
  float* sDEg = smem + L::oDEg;             // [16][98] dL/d geo emb     (PTSG only)
  float* sDEc = smem + L::oDEc;             // [16][50] dL/d colour emb  (PTSG only)
  float* sG = smem + L::oG;                 // [16][130] dL/dh tile (colour: cols 0..127, geo 0..31)
  float* sDZ = smem + L::oDZ;               // [16][130]
  float* sWave = smem + L::oU;              // per-wave F_theta scratch, aliases sG/sDZ

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g = lane >> 4, colw = lane & 15, g4 = 4 * g;
  const int p0 = blockIdx.x * TILE;
  const bool color = (a.flags & PSL_STAGE_COLOR) != 0;
  const bool relpos = color && (a.flags & 0x10000) != 0;
  constexpr bool ptsg = PTSG;
  const bool featg = (a.flags & PSL_FEAT_GRAD) != 0;
  const bool parg = (a.flags & PSL_PARAM_GRAD) != 0 && color;
  const float* __restrict__ M = a.master;
  // Mapper, colour stage: the geometry decoder's backward chain (2 column tiles wide) rides along inside the colour
  // trunk's layer loop on waves 2..6, which have slack there (waves 0,1 carry the extra dL/dc product), instead of
  // running as a 15-barrier phase of its own with six waves idle.  The tracker instantiation needs the 93/125-wide
  // embedding gradients (all 8 waves) and keeps the separate phase.
  const bool geo_fused = !PTSG && color;

  PSL_STAMP(0);
  // ---------------------------------------------------------------- phase 0: reload per-sample state, zero accumulators
  if (t < 128) {
    const int s = t >> 3, k = t & 7;
    const int p = min(p0 + s, a.P - 1);
    SampleGeom sg = sample_geom(a, p);
    int i = a.ws.I[p * K + k];
    float rx = 0.f, ry = 0.f, rz = 0.f;
    if (i >= 0) {
      float4 q = a.pos[i];
      rx = __fsub_rn(q.x, sg.x); ry = __fsub_rn(q.y, sg.y); rz = __fsub_rn(q.z, sg.z);
    }
    sI[s * K + k] = i;
    sW[s * K + k] = a.ws.w[p * K + k];
    sRel[(s * K + k) * 3 + 0] = rx; sRel[(s * K + k) * 3 + 1] = ry; sRel[(s * K + k) * 3 + 2] = rz;
    if constexpr (PTSG) sGW[s * K + k] = 0.f;
    if (k == 0) {
      sPts[s * 4 + 0] = sg.x; sPts[s * 4 + 1] = sg.y; sPts[s * 4 + 2] = sg.z; sPts[s * 4 + 3] = sg.r2;
      // samples past the end of the batch behave as "no neighbours, zero gradient"
      sHas[s] = (p0 + s < a.P && a.ws.cnt[p] >= a.min_nn) ? 1 : 0;
    }
  }
  if (t < 32) sDB[t] = 0.f;
  if (t < 16) sAff[t] = 0.f;
  if constexpr (PTSG) {
    if (t < 64) sDP[t] = 0.f;
    for (int e = t; e < 16 * LD_DE; e += WG) sDEg[e] = 0.f;
    for (int e = t; e < 16 * LD_DEC; e += WG) sDEc[e] = 0.f;
  }
  lds_barrier();

  PSL_STAMP(1);
  // ================================================================== colour decoder
  if (color) {
    // ---- d(logits): sigmoid and exposure-affine backward (decoder.py:432-448)
    if (t < TILE) {
      int p = p0 + t;
      float d0 = 0.f, d1 = 0.f, d2 = 0.f;
      if (p < a.P) {
        float4 dr = reinterpret_cast<const float4*>(a.ws.d_raw)[p];
        float4 rw = reinterpret_cast<const float4*>(a.ws.raw)[p];
        d0 = dr.x; d1 = dr.y; d2 = dr.z;
        if (!(a.flags & PSL_NO_SIGMOID)) { d0 *= rw.x * (1.f - rw.x); d1 *= rw.y * (1.f - rw.y); d2 *= rw.z * (1.f - rw.z); }
        if (a.flags & PSL_HAS_AFFINE) {
          const float* A = a.affine;
          float o0 = a.ws.out3[(size_t)p * 4], o1 = a.ws.out3[(size_t)p * 4 + 1], o2 = a.ws.out3[(size_t)p * 4 + 2];
          // out' = out @ A + t : dA[i][j] = out_i d_j ; dt_j = d_j ; d out_i = sum_j A[i][j] d_j
          atomic_add_f32(&sAff[0], o0 * d0); atomic_add_f32(&sAff[1], o0 * d1); atomic_add_f32(&sAff[2], o0 * d2);
          atomic_add_f32(&sAff[3], o1 * d0); atomic_add_f32(&sAff[4], o1 * d1); atomic_add_f32(&sAff[5], o1 * d2);
          atomic_add_f32(&sAff[6], o2 * d0); atomic_add_f32(&sAff[7], o2 * d1); atomic_add_f32(&sAff[8], o2 * d2);
          atomic_add_f32(&sAff[9], d0); atomic_add_f32(&sAff[10], d1); atomic_add_f32(&sAff[11], d2);
          float e0 = A[0] * d0 + A[1] * d1 + A[2] * d2;
          float e1 = A[3] * d0 + A[4] * d1 + A[5] * d2;
          float e2 = A[6] * d0 + A[7] * d1 + A[8] * d2;
          d0 = e0; d1 = e1; d2 = e2;
        }
        if (a.ws.d_out3) reinterpret_cast<float4*>(a.ws.d_out3)[p] = make_float4(d0, d1, d2, 0.f);
      }
      sDO[t * 4] = d0; sDO[t * 4 + 1] = d1; sDO[t * 4 + 2] = d2; sDO[t * 4 + 3] = 0.f;
    }
    lds_barrier();
    // ---- G = d_out3 * W_out  (output_linear.weight [3][128])
    {
      const float* wo = M + MO(PI_C_OUT);
      for (int e = t; e < TILE * HC; e += WG) {
        int s = e >> 7, k = e & 127;
        sG[s * LD_HN + k] = sDO[s * 4] * wo[k] + sDO[s * 4 + 1] * wo[HC + k] + sDO[s * 4 + 2] * wo[2 * HC + k];
      }
      if (geo_fused && t < TILE * HG) {   // geometry: G = d_occ * w_out (see the geometry phase below)
        int s = t >> 5, k = t & 31;
        int p = p0 + s;
        float docc = (p < a.P) ? a.ws.d_raw[(size_t)p * 4 + 3] : 0.f;
        sGg[s * LD_CF + k] = docc * M[MO(PI_G_OUT) + k];
      }
    }
    lds_barrier();
    f32x4 dcacc = {0.f, 0.f, 0.f, 0.f};   // waves 0,1: dL/dc_col column slice
    f32x4 dcg = {0.f, 0.f, 0.f, 0.f};     // waves 2,3: dL/dc_geo column slice (fused geometry backward)
    const int n0 = 16 * wave;
    // saved activations of the layer are requested one layer ahead (they depend on the forward pass only)
    auto load_y = [&](int i, float (&yv)[4]) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int p = p0 + g4 + r;
        yv[r] = (p < a.P) ? a.ws.c_y[((size_t)i * a.ws.Ppad + p) * HC + n0 + colw] : 0.f;
      }
    };
    float ycur[4], ynxt[4];
    load_y(4, ycur);
#pragma unroll
    for (int i = 4; i >= 0; --i) {
      // the weight fragments of step C depend on nothing computed here: request them first, so that their L2
      // latency elapses behind step A/B and the barrier
      float wC[HC / 4];
      if (i > 0) load_y(i - 1, ynxt);
      const float* Wi = M + MO(PI_C_L + 2 * i);
      if (i == 3) fetch_b16<HC>(Wi, EC + HC, EC + n0, wC);
      else if (i > 0) fetch_b16<HC>(Wi, HC, n0, wC);
      // step A: dz = G * act'(y)
      f32x4 gv = frag_load(sG, LD_HN, n0), dz;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int p = p0 + g4 + r;
        float y = ycur[r];
        dz[r] = (p < a.P) ? gv[r] * softplus100_grad_from_out(y) : 0.f;
        if (parg && p < a.P) {
          a.ws.c_dz[((size_t)i * a.ws.Ppad + p) * HC + n0 + colw] = dz[r];
          a.ws.c_g[((size_t)i * a.ws.Ppad + p) * HC + n0 + colw] = gv[r];
        }
      }
      frag_store(sDZ, LD_HN, n0, dz);
      // step B: dL/dc += G * Wc_i   (fc_c.i.weight [128][32])
      if (wave < 2) dcacc += gemm16<HC>(sG, LD_HN, M + MO(PI_C_FCC + 2 * i), C, n0);
      if (geo_fused && (wave == 2 || wave == 3)) {     // geometry steps A, B
        const int ng = 16 * (wave - 2);
        f32x4 gg = frag_load(sGg, LD_CF, ng), dzg;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int p = p0 + g4 + r;
          float y = (p < a.P) ? a.ws.g_y[((size_t)i * a.ws.Ppad + p) * HG + ng + colw] : 0.f;
          dzg[r] = (p < a.P && y > 0.f) ? gg[r] : 0.f;       // ReLU
        }
        frag_store(sDZg, LD_CF, ng, dzg);
        dcg += gemm16<HG>(sGg, LD_CF, M + MO(PI_G_FCC + 2 * i), C, ng);
      }
      lds_barrier();
      // step C: dL/d(input of layer i) = dz * W_i   (pts_linears.i.weight [128][Kin])
      f32x4 gn = {0.f, 0.f, 0.f, 0.f}, ge = {0.f, 0.f, 0.f, 0.f};
      if (i == 3) {
        gn = mma16<HC>(sDZ, LD_HN, wC);                                     // h part: input cols 40..167
        if (ptsg && wave < 3) ge = gemm16<HC>(sDZ, LD_HN, Wi, EC + HC, n0); // embedding part: cols 0..39(47)
      } else if (i == 0) {
        if (ptsg && wave < 3) ge = gemm16<HC>(sDZ, LD_HN, Wi, EC, n0);
      } else {
        gn = mma16<HC>(sDZ, LD_HN, wC);
      }
      // geometry step C on waves 4..6: dz * W_i; layer 3 only feeds columns 93..124 (the h part of [emb | h])
      f32x4 gxg = {0.f, 0.f, 0.f, 0.f};
      const bool geoC = geo_fused && i > 0 && wave >= 4 && wave < (i == 3 ? 7 : 6);
      const int ngc = (i == 3 ? 80 : 0) + 16 * (wave - 4);
      if (geoC) gxg = gemm16<HG>(sDZg, LD_CF, M + MO(PI_G_L + 2 * i), i == 3 ? EG + HG : HG, ngc);
      lds_barrier();
      if (geoC) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int col = ngc + colw;
          if (i == 3) { if (col >= EG && col < EG + HG) sGg[(g4 + r) * LD_CF + col - EG] = gxg[r]; }
          else sGg[(g4 + r) * LD_CF + col] = gxg[r];
        }
      }
      if (i > 0) frag_store(sG, LD_HN, n0, gn);
      if ((i == 3 || i == 0) && ptsg && wave < 3) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n0 + colw < EC) sDEc[(g4 + r) * LD_DEC + n0 + colw] += ge[r];
      }
      lds_barrier();
#pragma unroll
      for (int r = 0; r < 4; ++r) ycur[r] = ynxt[r];
    }
    if (wave < 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) sDCc[(g4 + r) * LD_CF + n0 + colw] = sHas[g4 + r] ? dcacc[r] : 0.f;
    }
    if (geo_fused && (wave == 2 || wave == 3)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) sDCg[(g4 + r) * LD_CF + 16 * (wave - 2) + colw] = sHas[g4 + r] ? dcg[r] : 0.f;
    }
    lds_barrier();
    PSL_STAMP(2);

    if (!relpos) {
      // ---- plain interpolation: scatter w_k * dC into the colour feature rows, collect dL/dw_k
      const int s = t >> 5, ch = t & 31;
      float dc = sDCc[s * LD_CF + ch];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        int i = sI[s * K + k];
        float w = sW[s * K + k];
        float gwk = 0.f;
        if (i >= 0 && sHas[s]) {
          if (featg && w != 0.f) {
            int row = o.row_map ? o.row_map[i] : i;
            if (row >= 0) atomic_add_f32(&o.g_col[(size_t)row * C + ch], w * dc);
          }
          if (ptsg) gwk = a.col_feats[(size_t)i * C + ch] * dc;
        }
        if (ptsg) {
#pragma unroll
          for (int off = 16; off > 0; off >>= 1) gwk += __shfl_xor(gwk, off);
          if (ch == 0) sGW[s * K + k] += gwk;
        }
      }
    } else {
      // ---- F_theta backward, one wave per 16 (sample, neighbour) rows.  Everything below is wave-private (its own
      // scratch, its own rows), so the waves only order their own LDS traffic (s_waitcnt) and never meet at a barrier.
      float* sDnf = sWave + wave * PW;          // [16][34]  A operand of dH1 = d_nf * W2
      float* sDz1 = sDnf + 16 * LD_CF;          // [16][66]  one 64-column half of dz1
      float* sDxe = sDz1 + 16 * LD_Z1;          // [16][22]  rel-pos part of dX1
      // d_nf[row][ch] = w[s][k] * dC[s][ch];  dL/dw[s][k] = sum_ch nf[row][ch] dC[s][ch]
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int e = lane + 64 * j;            // 16 rows x 32 channels
        int rl = e >> 5, ch = e & 31;
        int row = 16 * wave + rl, s = row >> 3, k = row & 7;
        float dc = sDCc[s * LD_CF + ch];
        float dnf = sW[s * K + k] * dc;
        sDnf[rl * LD_CF + ch] = dnf;
        bool live = (p0 + s) < a.P;
        if (parg && live) a.ws.n_dnf[((size_t)p0 * K + row) * C + ch] = dnf;
        if constexpr (PTSG) {
          float v = live ? a.ws.n_out[((size_t)p0 * K + row) * C + ch] * dc : 0.f;
#pragma unroll
          for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);
          if (ch == 0) sGW[s * K + k] += v;     // one writer per (s,k): this wave owns rows 16w..16w+15
        }
      }
      wave_lds_sync();
      // dH1 = d_nf * W2 (linear2.weight [32][128]); dz1 = dH1 * softplus'(h1); dX1 = dz1 * W1 (linear1.weight
      // [128][52], columns [sin 10 | cos 10 | feat 32]) -- in two 64-column halves of the hidden layer.  The saved
      // h1 values and the weight fragments of a half are requested up front (one exposed latency per product).
      f32x4 dxa[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) dxa[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        float h1v[4][4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int row = 16 * wave + g4 + r;
            bool live = (p0 + (row >> 3)) < a.P;
            h1v[nt][r] = live ? a.ws.n_h1[((size_t)p0 * K + row) * HC + 64 * half + 16 * nt + colw] : 0.f;
          }
        f32x4 dh[4];
        gemm16_multi<C, 4>(sDnf, LD_CF, M + MO(PI_C_N2) + 64 * half, HC, dh);
        if (half) wave_lds_sync();      // the first half's dz1 tile has been consumed
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          f32x4 dz;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int row = 16 * wave + g4 + r;
            bool live = (p0 + (row >> 3)) < a.P;
            dz[r] = live ? dh[nt][r] * softplus100_grad_from_out(h1v[nt][r]) : 0.f;
            if (parg && live) a.ws.n_dz1[((size_t)p0 * K + row) * HC + 64 * half + 16 * nt + colw] = dz[r];
          }
          frag_store(sDz1, LD_Z1, 16 * nt, dz);
        }
        wave_lds_sync();
        f32x4 dx[4];
        gemm16_multi<64, 4>(sDz1, LD_Z1, M + MO(PI_C_N1) + 64 * half * NX, NX, dx);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) dxa[kt] += dx[kt];
      }
      // feature part (columns 20..51) -> scattered into the colour feature rows straight from the MFMA registers;
      // rel-pos part (columns 0..19) -> small LDS tile
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const int col = 16 * kt + colw;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rl = g4 + r, row = 16 * wave + rl;
          if (col < ER) sDxe[rl * LD_XE + col] = dxa[kt][r];
          else if (featg && col < NX) {
            int i = sI[row];
            if (i >= 0 && sW[row] != 0.f && sHas[row >> 3]) {
              int dst = o.row_map ? o.row_map[i] : i;
              if (dst >= 0) atomic_add_f32(&o.g_col[(size_t)dst * C + col - ER], dxa[kt][r]);
            }
          }
        }
      }
      wave_lds_sync();
      // rel-pos embedding part: y_f = 2pi rel . B[:,f]; e = [sin y, cos y]; one (row, frequency) pair per lane
      if (parg || ptsg) {
        const float* Brel = M + MO(PI_C_BREL);
        for (int e = lane; e < 16 * ERF; e += 64) {
          int rl = e / ERF, f = e - rl * ERF;
          int row = 16 * wave + rl, s = row >> 3;
          bool live = (p0 + s) < a.P && sI[row] >= 0;
          if (!live) continue;
          float rx = sRel[row * 3], ry = sRel[row * 3 + 1], rz = sRel[row * 3 + 2];
          float sn, cs;
          if (a.ws.n_x) {     // the forward pass saved [sin | cos] in the first 20 columns of F_theta's input
            const float* xr = a.ws.n_x + ((size_t)p0 * K + row) * NX;
            sn = xr[f]; cs = xr[ERF + f];
          } else {
            fast_sincosf(fourier_phase(rx, ry, rz, Brel, ERF, f), sn, cs);
          }
          float dy2 = TWO_PI * (sDxe[rl * LD_XE + f] * cs - sDxe[rl * LD_XE + ERF + f] * sn);
          if (parg) {
            atomic_add_f32(&sDB[f], dy2 * rx); atomic_add_f32(&sDB[ERF + f], dy2 * ry);
            atomic_add_f32(&sDB[2 * ERF + f], dy2 * rz);
          }
          if constexpr (PTSG) {   // rel = x_k - p  =>  dp -= d_rel
            atomic_add_f32(&sDP[s * 4], -dy2 * Brel[f]); atomic_add_f32(&sDP[s * 4 + 1], -dy2 * Brel[ERF + f]);
            atomic_add_f32(&sDP[s * 4 + 2], -dy2 * Brel[2 * ERF + f]);
          }
        }
      }
    }
    lds_barrier();
  }

  PSL_STAMP(3);
  // ================================================================== geometry decoder
  {
    if (!geo_fused) {
    // G = d_occ * w_out (output_linear.weight [1][32]); d_occ flows for masked samples too (straight-through)
    if (t < TILE * HG) {
      int s = t >> 5, k = t & 31;
      int p = p0 + s;
      float docc = (p < a.P) ? a.ws.d_raw[(size_t)p * 4 + 3] : 0.f;
      sG[s * LD_HN + k] = docc * M[MO(PI_G_OUT) + k];
    }
    lds_barrier();
    f32x4 dcacc = {0.f, 0.f, 0.f, 0.f};
    const int n0 = 16 * wave;
#pragma unroll
    for (int i = 4; i >= 0; --i) {
      if (wave < 2) {
        f32x4 gv = frag_load(sG, LD_HN, n0), dz;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int p = p0 + g4 + r;
          float y = (p < a.P) ? a.ws.g_y[((size_t)i * a.ws.Ppad + p) * HG + n0 + colw] : 0.f;
          dz[r] = (p < a.P && y > 0.f) ? gv[r] : 0.f;       // ReLU
        }
        frag_store(sDZ, LD_HN, n0, dz);
        dcacc += gemm16<HG>(sG, LD_HN, M + MO(PI_G_FCC + 2 * i), C, n0);   // fc_c.i.weight [32][32]
      }
      lds_barrier();
      const float* Wi = M + MO(PI_G_L + 2 * i);
      f32x4 gx = {0.f, 0.f, 0.f, 0.f};
      const int Kin = (i == 0) ? EG : (i == 3 ? EG + HG : HG);
      bool act = false;
      if (i == 3) { act = ptsg || n0 + 15 >= EG; if (act) gx = gemm16<HG>(sDZ, LD_HN, Wi, EG + HG, n0); }   // 8 slices of [32][125]
      else if (i == 0) { act = ptsg && wave < 6; if (act) gx = gemm16<HG>(sDZ, LD_HN, Wi, EG, n0); }
      else { act = wave < 2; if (act) gx = gemm16<HG>(sDZ, LD_HN, Wi, HG, n0); }
      lds_barrier();
      if (act) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int col = n0 + colw;
          if (i == 3) {
            if (col < EG) { if (ptsg) sDEg[(g4 + r) * LD_DE + col] += gx[r]; }
            else if (col < Kin) sG[(g4 + r) * LD_HN + col - EG] = gx[r];
          } else if (i == 0) {
            if (col < EG) sDEg[(g4 + r) * LD_DE + col] += gx[r];
          } else {
            sG[(g4 + r) * LD_HN + col] = gx[r];
          }
        }
      }
      lds_barrier();
    }
    if (wave < 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) sDCg[(g4 + r) * LD_CF + n0 + colw] = sHas[g4 + r] ? dcacc[r] : 0.f;
    }
    lds_barrier();
    }  // !geo_fused
    // scatter into the geometry feature rows, collect dL/dw
    {
      const int s = t >> 5, ch = t & 31;
      float dc = sDCg[s * LD_CF + ch];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        int i = sI[s * K + k];
        float w = sW[s * K + k];
        float gwk = 0.f;
        if (i >= 0 && sHas[s]) {
          if (featg && w != 0.f) {
            int row = o.row_map ? o.row_map[i] : i;
            if (row >= 0) atomic_add_f32(&o.g_geo[(size_t)row * C + ch], w * dc);
          }
          if (ptsg) gwk = a.geo_feats[(size_t)i * C + ch] * dc;
        }
        if (ptsg) {
#pragma unroll
          for (int off = 16; off > 0; off >>= 1) gwk += __shfl_xor(gwk, off);
          if (ch == 0) atomic_add_f32(&sGW[s * K + k], gwk);
        }
      }
    }
    lds_barrier();
  }

  PSL_STAMP(4);
  // ================================================================== position gradient
  if (ptsg) {
    // (1) interpolation weights: w = a/S, a = [D<=r2]/(D+1e-10), D = |x_k - p|^2   (decoder.py:143-160)
    if (t < 128) {
      const int s = t >> 3, k = t & 7;
      float rx = sRel[(s * K + k) * 3], ry = sRel[(s * K + k) * 3 + 1], rz = sRel[(s * K + k) * 3 + 2];
      float D = (sI[s * K + k] >= 0)
                    ? __fadd_rn(__fadd_rn(__fmul_rn(rx, rx), __fmul_rn(ry, ry)), __fmul_rn(rz, rz))
                    : __int_as_float(0x7F800000);
      float av = (D > sPts[s * 4 + 3]) ? 0.f : 1.0f / (D + 1e-10f);
      float S1 = av;
      S1 += __shfl_xor(S1, 1); S1 += __shfl_xor(S1, 2); S1 += __shfl_xor(S1, 4);
      float gw = sHas[s] ? sGW[s * K + k] : 0.f;
      float dot = gw * sW[s * K + k];
      dot += __shfl_xor(dot, 1); dot += __shfl_xor(dot, 2); dot += __shfl_xor(dot, 4);
      float da = (gw - dot) / fmaxf(S1, 1e-12f);
      float dD = -da * av * av;                 // a = 1/(D+eps) -> da/dD = -a^2 ; masked slots: a = 0
      // dD/dp = -2 (x_k - p)
      float px = -2.f * dD * rx, py = -2.f * dD * ry, pz = -2.f * dD * rz;
      px += __shfl_xor(px, 1); px += __shfl_xor(px, 2); px += __shfl_xor(px, 4);
      py += __shfl_xor(py, 1); py += __shfl_xor(py, 2); py += __shfl_xor(py, 4);
      pz += __shfl_xor(pz, 1); pz += __shfl_xor(pz, 2); pz += __shfl_xor(pz, 4);
      if (k == 0) {
        atomic_add_f32(&sDP[s * 4], px); atomic_add_f32(&sDP[s * 4 + 1], py); atomic_add_f32(&sDP[s * 4 + 2], pz);
      }
    }
    // (2) Fourier embeddings of p: geometry sin(2pi p.B) (93), colour [sin,cos] (20+20)
    {
      const int s = t >> 5, l32 = t & 31;
      const float x = sPts[s * 4], y = sPts[s * 4 + 1], z = sPts[s * 4 + 2];
      const float* Bg = M + MO(PI_G_B);
      float ax = 0.f, ay = 0.f, az = 0.f;
      for (int f = l32; f < EG; f += 32) {
        float dy2 = TWO_PI * sDEg[s * LD_DE + f] * fast_cosf(fourier_phase(x, y, z, Bg, EG, f));
        ax += dy2 * Bg[f]; ay += dy2 * Bg[EG + f]; az += dy2 * Bg[2 * EG + f];
      }
      if (color && l32 < ECF) {
        float sn, cs;
        fast_sincosf(fourier_phase(x, y, z, a.Bcol, ECF, l32), sn, cs);
        float dy2 = TWO_PI * (sDEc[s * LD_DEC + l32] * cs - sDEc[s * LD_DEC + ECF + l32] * sn);
        ax += dy2 * a.Bcol[l32]; ay += dy2 * a.Bcol[ECF + l32]; az += dy2 * a.Bcol[2 * ECF + l32];
      }
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) {
        ax += __shfl_xor(ax, off); ay += __shfl_xor(ay, off); az += __shfl_xor(az, off);
      }
      if (l32 == 0) { atomic_add_f32(&sDP[s * 4], ax); atomic_add_f32(&sDP[s * 4 + 1], ay); atomic_add_f32(&sDP[s * 4 + 2], az); }
    }
    lds_barrier();
    if (t < TILE && p0 + t < a.P)
      reinterpret_cast<float4*>(a.ws.dp)[p0 + t] = make_float4(sDP[t * 4], sDP[t * 4 + 1], sDP[t * 4 + 2], 0.f);
  }
  PSL_STAMP(5);
  // tile-level reductions that go out with a handful of global atomics
  if (parg && relpos && t < 3 * ERF && o.g_brel) atomic_add_f32(&o.g_brel[t], sDB[t]);
  if ((a.flags & PSL_HAS_AFFINE) && color && t < 12 && o.g_affine) atomic_add_f32(&o.g_affine[t], sAff[t]);
}

int launch_dw(psl_ctx* ctx, const DecodeArgs& a, float* g_params, const float* g_brel, hipStream_t s);
int launch_decode_bwd2(psl_ctx* ctx, const DecodeArgs& a, const psl_render_grads& g, float* small, hipStream_t s);

template <bool PTSG>
static int launch_bwd_t(const DecodeArgs& a, const BwdOut& o, int tiles, hipStream_t s) {
  static bool attr_set = false;
  const size_t lds = sizeof(float) * BwdLds<PTSG>::total;
  if (!attr_set) {
    PSL_HIP(hipFuncSetAttribute((const void*)k_decode_bwd<PTSG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(k_decode_bwd<PTSG>, dim3(tiles), dim3(WG), lds, s, a, o);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}

int launch_decode_bwd(psl_ctx* ctx, const DecodeArgs& a, const psl_render_grads& g, hipStream_t s) {
  const bool color = a.flags & PSL_STAGE_COLOR;
  if ((a.flags & PSL_FEAT_GRAD) && (!g.g_geo_feats || (color && !g.g_col_feats))) {
    set_error("psl_render_bwd: PSL_FEAT_GRAD needs g_geo_feats/g_col_feats"); return PSL_ERR_ARG;
  }
  if ((a.flags & PSL_PARAM_GRAD) && !g.g_params) { set_error("psl_render_bwd: PSL_PARAM_GRAD needs g_params"); return PSL_ERR_ARG; }
  if ((a.flags & PSL_HAS_AFFINE) && !g.g_exposure_affine) { set_error("psl_render_bwd: affine gradient buffer missing"); return PSL_ERR_ARG; }
  BwdOut o;
  o.g_geo = g.g_geo_feats; o.g_col = g.g_col_feats; o.row_map = g.feat_row_map;
  // small accumulators: [0..31] dB_rel, [32..47] affine  (ctx->d_small)
  float* small = ctx->d_small;     // cleared by the compositing-backward kernel that always runs just before
  o.g_brel = small;
  o.g_affine = small + 32;
  int tiles = (a.P + TILE - 1) / TILE;
  static unsigned long long* dbg = nullptr;
  static int dbg_on = -1;
  if (dbg_on < 0) { const char* e = getenv("PSL_DEBUG_PHASES"); dbg_on = (e && e[0] == '1') ? 1 : 0; }
  DecodeArgs a2 = a;
  if (dbg_on) { if (!dbg) PSL_HIP(hipMalloc(&dbg, 64 * sizeof(unsigned long long))); a2.dbg = dbg; }
  {
    ProfScope ps(ctx, prof_decode_slot(a.flags, true), s, bwd_flops_per_sample(a.flags) * a.P);
    int rc;
    if (ctx->decode_bwd_version >= 2) rc = launch_decode_bwd2(ctx, a2, g, small, s);
    else rc = (a.flags & PSL_PTS_GRAD) ? launch_bwd_t<true>(a2, o, tiles, s) : launch_bwd_t<false>(a2, o, tiles, s);
    if (rc) return rc;
  }
  if (dbg_on) {
    unsigned long long h[8];
    PSL_HIP(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
    fprintf(stderr, "[psl bwd P=%d flags=%x] cycles: p0 %llu col_trunk %llu col_nbr %llu geo %llu dp %llu | total %llu\n", a.P,
            a.flags, h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[5] - h[0]);
  }
  if ((a.flags & PSL_HAS_AFFINE) && color)
    PSL_HIP(hipMemcpyAsync(g.g_exposure_affine, small + 32, sizeof(float) * 12, hipMemcpyDeviceToDevice, s));
  if (a.flags & PSL_PARAM_GRAD) {
    if (color) {
      ProfScope ps(ctx, PROF_DW, s, dw_flops_per_sample(a.flags) * a.P);
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
