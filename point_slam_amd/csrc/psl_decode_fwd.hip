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
#include "psl_decode.h"

namespace psl {

constexpr int FWD_LDS_FLOATS = 128 + 128 + 384 + 64 + 16 + 16 * LD_CF * 2 + 16 * LD_G + 16 * LD_C + 16 + 64 +
                               128 * LD_XN + 8 * 16 * LD_HN;

__global__ __launch_bounds__(WG) void k_decode_fwd(DecodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* sI = (int*)smem;                     // [16][8]
  float* sW = smem + 128;                   // [16][8]
  float* sRel = sW + 128;                   // [16][8][3]
  float* sPts = sRel + 384;                 // [16][4]
  int* sHas = (int*)(sPts + 64);            // [16]
  float* sCg = (float*)(sHas + 16);         // [16][34]
  float* sCc = sCg + 16 * LD_CF;            // [16][34]
  float* sXg = sCc + 16 * LD_CF;            // [16][130]
  float* sXc = sXg + 16 * LD_G;             // [16][170]
  float* sOcc = sXc + 16 * LD_C;            // [16]
  float* sOut = sOcc + 16;                  // [16][4]
  float* sXn = sOut + 64;                   // [128][54]
  float* sHn = sXn + 128 * LD_XN;           // [8][16][130]

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int tile = blockIdx.x;
  const int p0 = tile * TILE;
  const bool color = (a.flags & PSL_STAGE_COLOR) != 0;
  const bool relpos = color && (a.flags & 0x10000) != 0;  // internal bit: encode_rel_pos
  const float* __restrict__ M = a.master;
  const float* __restrict__ WT = a.wt;

  // ---------------------------------------------------------------- phase 0: neighbours, weights
  if (t < 128) {
    const int s = t >> 3, k = t & 7;
    const int p = min(p0 + s, a.P - 1);
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
    if (p0 + s < a.P) a.ws.w[p * K + k] = w;
    if (k == 0) {
      sPts[s * 4 + 0] = g.x; sPts[s * 4 + 1] = g.y; sPts[s * 4 + 2] = g.z; sPts[s * 4 + 3] = g.r2;
      sHas[s] = (a.ws.cnt[p] >= a.min_nn) ? 1 : 0;   // has_neighbors (decoder.py:150)
    }
  }
  __syncthreads();

  // ---------------------------------------------------------------- phase 1: gathers
  {
    const int s = t >> 5, ch = t & 31;
    const int p = p0 + s;
    // geometry feature: c = sum_k w_k f[I_k]; no-neighbour samples get the fallback vector (decoder.py:162-171)
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      int i = sI[s * K + k];
      if (i >= 0) acc = __fadd_rn(acc, __fmul_rn(sW[s * K + k], a.geo_feats[(size_t)i * C + ch]));
    }
    if (!sHas[s]) acc = a.fb_geo[ch];
    sCg[s * LD_CF + ch] = acc;
    if (p < a.P) a.ws.cg[(size_t)p * C + ch] = acc;
    if (color && !relpos) {
      float ac = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        int i = sI[s * K + k];
        if (i >= 0) ac = __fadd_rn(ac, __fmul_rn(sW[s * K + k], a.col_feats[(size_t)i * C + ch]));
      }
      if (!sHas[s]) ac = a.fb_col[ch];
      sCc[s * LD_CF + ch] = ac;
      if (p < a.P) a.ws.cc[(size_t)p * C + ch] = ac;
    }
  }
  if (relpos) {
    // F_theta input rows [sin(10) cos(10) | feat(32)] for the 128 (sample, neighbour) pairs (decoder.py:371-378)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int e = t + WG * j;
      int row = e >> 5, ch = e & 31;
      int i = sI[row];
      float v = (i >= 0) ? a.col_feats[(size_t)i * C + ch] : 0.f;
      sXn[row * LD_XN + ER + ch] = v;
    }
    const float* Brel = M + MO(PI_C_BREL);
    for (int e = t; e < 128 * ERF; e += WG) {
      int row = e / ERF, f = e - row * ERF;
      float ph = fourier_phase(sRel[row * 3], sRel[row * 3 + 1], sRel[row * 3 + 2], Brel, ERF, f);
      float sn, cs;
      sincosf(ph, &sn, &cs);
      sXn[row * LD_XN + f] = sn;
      sXn[row * LD_XN + ERF + f] = cs;
    }
  }
  // ---------------------------------------------------------------- phase 2: Fourier embeddings of p
  {
    const float* Bg = M + MO(PI_G_B);
    for (int e = t; e < TILE * EGP; e += WG) {
      int s = e / EGP, f = e - s * EGP;
      float v = 0.f;
      if (f < EG) v = sinf(fourier_phase(sPts[s * 4], sPts[s * 4 + 1], sPts[s * 4 + 2], Bg, EG, f));
      sXg[s * LD_G + f] = v;
    }
    if (color && t < TILE * ECF) {
      int s = t / ECF, f = t - s * ECF;
      float sn, cs;
      sincosf(fourier_phase(sPts[s * 4], sPts[s * 4 + 1], sPts[s * 4 + 2], a.Bcol, ECF, f), &sn, &cs);
      sXc[s * LD_C + f] = sn;
      sXc[s * LD_C + ECF + f] = cs;
      int p = p0 + s;
      if (p < a.P && a.ws.c_emb) { a.ws.c_emb[(size_t)p * EC + f] = sn; a.ws.c_emb[(size_t)p * EC + ECF + f] = cs; }
    }
  }
  __syncthreads();
  if (relpos && a.ws.n_x) {
    for (int e = t; e < 128 * NX; e += WG) {
      int row = e / NX, c = e - row * NX;
      int p = p0 + (row >> 3);
      if (p < a.P) a.ws.n_x[((size_t)p0 * K + row) * NX + c] = sXn[row * LD_XN + c];
    }
  }

  // ---------------------------------------------------------------- phase 3: geometry MLP (waves 0,1)
  // h = relu(W_i h + b_i) + (Wc_i c + bc_i); after block 2 the embedding is re-attached (decoder.py:207-219)
  {
    const int g4 = 4 * (lane >> 4), colw = lane & 15;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      f32x4 y = {0.f, 0.f, 0.f, 0.f}, h = {0.f, 0.f, 0.f, 0.f};
      const int n0 = 16 * wave;
      if (wave < 2) {
        f32x4 acc;
        if (i == 0) acc = gemm16<EGP>(sXg, LD_G, WT + wtoff(WT_G_L + 0), HG, n0);
        else if (i == 3) acc = gemm16<EGP + HG>(sXg, LD_G, WT + wtoff(WT_G_L + 3), HG, n0);
        else acc = gemm16<HG>(sXg + EGP, LD_G, WT + wtoff(WT_G_L + i), HG, n0);
        f32x4 u = gemm16<C>(sCg, LD_CF, WT + wtoff(WT_G_FCC + i), HG, n0);
        float b = M[MO(PI_G_L + 2 * i + 1) + n0 + colw];
        float bc = M[MO(PI_G_FCC + 2 * i + 1) + n0 + colw];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          y[r] = fmaxf(acc[r] + b, 0.f);
          h[r] = y[r] + (u[r] + bc);
        }
      }
      __syncthreads();
      if (wave < 2) {
        frag_store(sXg + EGP, LD_G, n0, h);
        if (a.ws.g_y) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int p = p0 + g4 + r;
            if (p < a.P) a.ws.g_y[((size_t)p * 5 + i) * HG + n0 + colw] = y[r];
          }
        }
      }
      __syncthreads();
    }
    if (t < TILE) {  // output_linear 32 -> 1
      const float* wo = M + MO(PI_G_OUT);
      float o = 0.f;
#pragma unroll
      for (int k = 0; k < HG; ++k) o = fmaf(sXg[t * LD_G + EGP + k], wo[k], o);
      sOcc[t] = o + M[MO(PI_G_OUT + 1)];
    }
  }

  if (color) {
    // -------------------------------------------------------------- phase 4: F_theta per neighbour
    if (relpos) {
      float* Hw = sHn + wave * 16 * LD_HN;
      const float* Xw = sXn + wave * 16 * LD_XN;
      const int g = lane >> 4, colw = lane & 15;
      {
        f32x4 acc[8];
        gemm16_multi<NX, 8>(Xw, LD_XN, WT + wtoff(WT_C_N1), HC, acc);
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
          float b = M[MO(PI_C_N1 + 1) + 16 * nt + colw];
          f32x4 hv;
#pragma unroll
          for (int r = 0; r < 4; ++r) hv[r] = softplus100(acc[nt][r] + b);
          frag_store(Hw, LD_HN, 16 * nt, hv);
          if (a.ws.n_h1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              int row = 16 * wave + 4 * g + r;
              if (p0 + (row >> 3) < a.P) a.ws.n_h1[((size_t)p0 * K + row) * HC + 16 * nt + colw] = hv[r];
            }
          }
        }
      }
      __syncthreads();
      {
        f32x4 acc[2];
        gemm16_multi<HC, 2>(Hw, LD_HN, WT + wtoff(WT_C_N2), C, acc);
        const int s = 2 * wave + (g >> 1);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          float b = M[MO(PI_C_N2 + 1) + 16 * nt + colw];
          float part = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float nf = acc[nt][r] + b;
            int row = 16 * wave + 4 * g + r;
            if (a.ws.n_out && p0 + (row >> 3) < a.P) a.ws.n_out[((size_t)p0 * K + row) * C + 16 * nt + colw] = nf;
            part = __fadd_rn(part, __fmul_rn(sW[s * K + 4 * (g & 1) + r], nf));
          }
          float tot = part + __shfl_xor(part, 16);
          if ((g & 1) == 0) {
            float c = sHas[s] ? tot : a.fb_col[16 * nt + colw];
            sCc[s * LD_CF + 16 * nt + colw] = c;
            if (p0 + s < a.P) a.ws.cc[(size_t)(p0 + s) * C + 16 * nt + colw] = c;
          }
        }
      }
    }
    __syncthreads();
    // -------------------------------------------------------------- phase 5: colour trunk, 8 waves x 16 columns
    {
      const int n0 = 16 * wave, g4 = 4 * (lane >> 4), colw = lane & 15;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        f32x4 acc;
        if (i == 0) acc = gemm16<EC>(sXc, LD_C, WT + wtoff(WT_C_L + 0), HC, n0);
        else if (i == 3) acc = gemm16<EC + HC>(sXc, LD_C, WT + wtoff(WT_C_L + 3), HC, n0);
        else acc = gemm16<HC>(sXc + EC, LD_C, WT + wtoff(WT_C_L + i), HC, n0);
        f32x4 u = gemm16<C>(sCc, LD_CF, WT + wtoff(WT_C_FCC + i), HC, n0);
        float b = M[MO(PI_C_L + 2 * i + 1) + n0 + colw];
        float bc = M[MO(PI_C_FCC + 2 * i + 1) + n0 + colw];
        f32x4 y, h;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          y[r] = softplus100(acc[r] + b);
          h[r] = y[r] + (u[r] + bc);
        }
        __syncthreads();
        frag_store(sXc + EC, LD_C, n0, h);
        if (a.ws.c_y) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int p = p0 + g4 + r;
            if (p < a.P) {
              a.ws.c_y[((size_t)p * 5 + i) * HC + n0 + colw] = y[r];
              a.ws.c_hin[((size_t)p * 5 + i) * HC + n0 + colw] = h[r];
            }
          }
        }
        __syncthreads();
      }
      if (t < TILE * 3) {  // output_linear 128 -> 3
        int s = t / 3, j = t - 3 * s;
        const float* wo = M + MO(PI_C_OUT) + j * HC;
        float o = 0.f;
#pragma unroll 8
        for (int k = 0; k < HC; ++k) o = fmaf(sXc[s * LD_C + EC + k], wo[k], o);
        sOut[s * 4 + j] = o + M[MO(PI_C_OUT + 1) + j];
      }
    }
  }
  __syncthreads();
  // ------------------------------------------------------------------ raw = [rgb, occ]
  if (t < TILE) {
    int p = p0 + t;
    if (p < a.P) {
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

int repack_weights(psl_ctx* ctx, const float* master, hipStream_t s) {
  hipLaunchKernelGGL(k_repack, dim3((kWtFloats + 255) / 256), dim3(256), 0, s, master, ctx->wt);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}

int launch_decode_fwd(const DecodeArgs& a, hipStream_t s) {
  static bool attr_set = false;
  const size_t lds = sizeof(float) * FWD_LDS_FLOATS;
  if (!attr_set) {
    PSL_HIP(hipFuncSetAttribute((const void*)k_decode_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  int tiles = (a.P + TILE - 1) / TILE;
  if (tiles == 0) return PSL_OK;
  hipLaunchKernelGGL(k_decode_fwd, dim3(tiles), dim3(WG), lds, s, a);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}

}  // namespace psl
