// Parameter gradients of the colour decoder: dW[n][k] = sum_p dZ[p][n] * X[p][k] for every linear layer,
// contracted over ALL samples of the batch with exact-fp32 MFMAs (one wavefront per 16x16 output tile per
// row-chunk), written to per-chunk slabs and reduced in a fixed order -> deterministic, no float atomics.
// (The reference gets these from ~500 ATen mm/sum launches in loss.backward(), Mapper.py:555.)
#include <cstring>
#include <algorithm>
#include "psl_decode.h"

namespace psl {

struct DwJob {
  const float* A; int lda; int n_valid;
  const float* B0; int ldb0; int k0_cols;
  const float* B1; int ldb1; int k1_cols;
  long long rows;
  int out_off, ld_out, bias_off;
  int n_tiles, k_tiles;
  int item_base, items;      // items per chunk for this job = n_tiles*k_tiles + (bias ? n_tiles : 0)
  int rows_per_chunk;
};
constexpr int MAX_JOBS = 16;
struct DwArgs { DwJob job[MAX_JOBS]; int n_jobs; int items_per_chunk; int n_chunks; float* slabs; };

__global__ __launch_bounds__(256) void k_dw(DwArgs d) {
  const int wid = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  if (wid >= d.items_per_chunk * d.n_chunks) return;
  const int chunk = wid / d.items_per_chunk;
  int item = wid - chunk * d.items_per_chunk;
  int ji = 0;
  for (int j = 1; j < d.n_jobs; ++j) if (item >= d.job[j].item_base) ji = j;
  const DwJob& J = d.job[ji];
  item -= J.item_base;
  const int lane = threadIdx.x & 63, g = lane >> 4, colw = lane & 15;
  const long long r0 = (long long)chunk * J.rows_per_chunk;
  const long long r1 = min(J.rows, r0 + J.rows_per_chunk);
  float* slab = d.slabs + (size_t)chunk * kColorFloats;
  const int ntk = J.n_tiles * J.k_tiles;
  if (item < ntk) {
    const int nt = item / J.k_tiles, kt = item - nt * J.k_tiles;
    const int ncol = 16 * nt + colw, kcol = 16 * kt + colw;
    const bool nok = ncol < J.n_valid;
    const int ktot = J.k0_cols + J.k1_cols;
    const bool kok = kcol < ktot;
    const float* bp; int ldb;
    if (kcol < J.k0_cols) { bp = J.B0 + kcol; ldb = J.ldb0; }
    else { bp = J.B1 + (kcol - J.k0_cols); ldb = J.ldb1; }
    const float* ap = J.A + ncol;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    long long r = r0;
    for (; r + 8 <= r1; r += 8) {
      long long ra = r + g, rb = r + 4 + g;
      float a0 = nok ? ap[ra * J.lda] : 0.f, b0 = kok ? bp[ra * ldb] : 0.f;
      float a1 = nok ? ap[rb * J.lda] : 0.f, b1 = kok ? bp[rb * ldb] : 0.f;
      acc0 = mfma16(a0, b0, acc0);
      acc1 = mfma16(a1, b1, acc1);
    }
    for (; r < r1; r += 4) {
      long long ra = r + g;
      bool v = ra < r1;
      float a0 = (v && nok) ? ap[ra * J.lda] : 0.f, b0 = (v && kok) ? bp[ra * ldb] : 0.f;
      acc0 = mfma16(a0, b0, acc0);
    }
    acc0 += acc1;
    if (kok) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int n = 16 * nt + 4 * g + q;
        if (n < J.n_valid) slab[J.out_off + n * J.ld_out + kcol] = acc0[q];
      }
    }
  } else {
    // bias: column sums of A
    const int nt = item - ntk;
    const int ncol = 16 * nt + colw;
    float s = 0.f;
    if (ncol < J.n_valid)
      for (long long r = r0 + g; r < r1; r += 4) s += J.A[r * J.lda + ncol];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (g == 0 && ncol < J.n_valid) slab[J.bias_off + ncol] = s;
  }
}

__global__ __launch_bounds__(256) void k_dw_reduce(const float* __restrict__ slabs, int n_chunks,
                                                   const float* __restrict__ g_brel, float* __restrict__ g_params) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= kMasterFloats) return;
  float v = 0.f;
  if (e < kColorFloats) {
    constexpr int b0 = poff(PI_C_BREL);
    if (e >= b0 && e < b0 + 3 * ERF) v = g_brel[e - b0];
    else for (int c = 0; c < n_chunks; ++c) v += slabs[(size_t)c * kColorFloats + e];
  }
  g_params[e] = v;   // geometry-decoder group: 0 (fix_geo_decoder, configs/point_slam.yaml:47)
}

int launch_dw(psl_ctx* ctx, const DecodeArgs& a, float* g_params, const float* g_brel, hipStream_t s) {
  const bool relpos = a.flags & 0x10000;
  const long long P = a.P;
  int n_chunks = (int)std::min<long long>(std::max<long long>((P + 511) / 512, 1), 64);
  if (ctx->dw_slab_cap < n_chunks) {
    if (ctx->dw_slabs) PSL_HIP(hipFree(ctx->dw_slabs));
    PSL_HIP(hipMalloc(&ctx->dw_slabs, sizeof(float) * (size_t)kColorFloats * 64));
    ctx->dw_slab_cap = 64;
  }
  PSL_HIP(hipMemsetAsync(ctx->dw_slabs, 0, sizeof(float) * (size_t)kColorFloats * n_chunks, s));
  DwArgs d;
  memset(&d, 0, sizeof(d));
  int nj = 0, base = 0;
  auto add = [&](const float* A, int lda, int nv, const float* B0, int ldb0, int k0, const float* B1, int ldb1, int k1,
                 long long rows, int out_pi) {
    DwJob& J = d.job[nj++];
    J.A = A; J.lda = lda; J.n_valid = nv; J.B0 = B0; J.ldb0 = ldb0; J.k0_cols = k0; J.B1 = B1; J.ldb1 = ldb1;
    J.k1_cols = k1; J.rows = rows; J.out_off = poff(out_pi); J.ld_out = k0 + k1; J.bias_off = poff(out_pi + 1);
    J.n_tiles = (nv + 15) / 16; J.k_tiles = (k0 + k1 + 15) / 16;
    J.items = J.n_tiles * J.k_tiles + J.n_tiles;
    J.item_base = base; base += J.items;
    long long rpc = (rows + n_chunks - 1) / n_chunks;
    J.rows_per_chunk = (int)((rpc + 3) / 4 * 4);
  };
  const RenderWs& w = a.ws;
  for (int i = 0; i < 5; ++i) {
    const float* dz = w.c_dz + i * HC;
    if (i == 0) add(dz, 5 * HC, HC, w.c_emb, EC, EC, nullptr, 0, 0, P, PI_C_L);
    else if (i == 3) add(dz, 5 * HC, HC, w.c_emb, EC, EC, w.c_hin + 2 * HC, 5 * HC, HC, P, PI_C_L + 6);
    else add(dz, 5 * HC, HC, w.c_hin + (i - 1) * HC, 5 * HC, HC, nullptr, 0, 0, P, PI_C_L + 2 * i);
    add(w.c_g + i * HC, 5 * HC, HC, w.cc, C, C, nullptr, 0, 0, P, PI_C_FCC + 2 * i);
  }
  add(w.d_out3, 4, 3, w.c_hin + 4 * HC, 5 * HC, HC, nullptr, 0, 0, P, PI_C_OUT);
  if (relpos) {
    add(w.n_dnf, C, C, w.n_h1, HC, HC, nullptr, 0, 0, P * K, PI_C_N2);
    add(w.n_dz1, HC, HC, w.n_x, NX, NX, nullptr, 0, 0, P * K, PI_C_N1);
  }
  d.n_jobs = nj; d.items_per_chunk = base; d.n_chunks = n_chunks; d.slabs = ctx->dw_slabs;
  long long waves = (long long)base * n_chunks;
  hipLaunchKernelGGL(k_dw, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, d);
  PSL_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_dw_reduce, dim3((kMasterFloats + 255) / 256), dim3(256), 0, s, ctx->dw_slabs, n_chunks, g_brel,
                     g_params);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}

}  // namespace psl
