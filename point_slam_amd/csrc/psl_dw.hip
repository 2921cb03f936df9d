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
  int item_base, items;      // items of this job = n_tiles*k_tiles*n_chunks
  int rows_per_chunk, n_chunks;
};
constexpr int MAX_JOBS = 16;
constexpr int MAX_CHUNKS = 256;
struct DwArgs { DwJob job[MAX_JOBS]; int n_jobs; int n_items; float* slabs; };
struct DwReduceArgs { int chunks_of_entry[kNumColorParams]; };

// One wavefront = one 32x32 output tile (2x2 MFMA tiles: every A/B fragment is used twice) of one layer for one
// chunk of rows.  8 row-steps are issued per loop trip (32 independent loads, then 32 MFMAs on 4 independent
// accumulators) so that L2 latency is covered by the loads already in flight.
__global__ __launch_bounds__(256) void k_dw(DwArgs d) {
  const int wid = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  if (wid >= d.n_items) return;
  int item = wid;
  int ji = 0;
  for (int j = 1; j < d.n_jobs; ++j) if (item >= d.job[j].item_base) ji = j;
  const DwJob& J = d.job[ji];
  item -= J.item_base;
  const int tiles = J.n_tiles * J.k_tiles;
  const int chunk = item / tiles;
  item -= chunk * tiles;
  const int lane = threadIdx.x & 63, g = lane >> 4, colw = lane & 15;
  const long long r0 = (long long)chunk * J.rows_per_chunk;
  const long long r1 = min(J.rows, r0 + J.rows_per_chunk);
  float* slab = d.slabs + (size_t)chunk * kColorFloats;
  const int nt = item / J.k_tiles, kt = item - nt * J.k_tiles;
  const int ktot = J.k0_cols + J.k1_cols;
  const float* ap[2]; const float* bp[2]; int ldb[2]; bool nok[2], kok[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    int ncol = 32 * nt + 16 * h + colw, kcol = 32 * kt + 16 * h + colw;
    nok[h] = ncol < J.n_valid; kok[h] = kcol < ktot;
    ap[h] = J.A + (nok[h] ? ncol : 0);
    if (kcol < J.k0_cols || !kok[h]) { bp[h] = J.B0 + (kok[h] ? kcol : 0); ldb[h] = J.ldb0; }
    else { bp[h] = J.B1 + (kcol - J.k0_cols); ldb[h] = J.ldb1; }
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum0 = 0.f, bsum1 = 0.f;
  constexpr int U = 8;
  for (long long r = r0; r < r1; r += 4 * U) {
    float av[U][2], bv[U][2];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long long row = r + 4 * u + g;
      bool v = row < r1;
      long long rr = v ? row : r0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float a = ap[h][rr * J.lda], b = bp[h][rr * ldb[h]];
        av[u][h] = (v && nok[h]) ? a : 0.f;
        bv[u][h] = (v && kok[h]) ? b : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc[0][0] = mfma16(av[u][0], bv[u][0], acc[0][0]);
      acc[0][1] = mfma16(av[u][0], bv[u][1], acc[0][1]);
      acc[1][0] = mfma16(av[u][1], bv[u][0], acc[1][0]);
      acc[1][1] = mfma16(av[u][1], bv[u][1], acc[1][1]);
      bsum0 += av[u][0]; bsum1 += av[u][1];
    }
  }
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      int kcol = 32 * kt + 16 * y + colw;
      if (kcol < ktot) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int n = 32 * nt + 16 * x + 4 * g + q;
          if (n < J.n_valid) slab[J.out_off + n * J.ld_out + kcol] = acc[x][y][q];
        }
      }
    }
  if (kt == 0) {   // bias gradient = column sums of dZ (this lane saw rows g, g+4, ... of column ncol)
    bsum0 += __shfl_xor(bsum0, 16); bsum0 += __shfl_xor(bsum0, 32);
    bsum1 += __shfl_xor(bsum1, 16); bsum1 += __shfl_xor(bsum1, 32);
    if (g == 0) {
      int n0c = 32 * nt + colw;
      if (n0c < J.n_valid) slab[J.bias_off + n0c] = bsum0;
      if (n0c + 16 < J.n_valid) slab[J.bias_off + n0c + 16] = bsum1;
    }
  }
}

__global__ __launch_bounds__(256) void k_dw_reduce(const float* __restrict__ slabs, DwReduceArgs ra,
                                                   const float* __restrict__ g_brel, float* __restrict__ g_params) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= kMasterFloats) return;
  float v = 0.f;
  if (e < kColorFloats) {
    constexpr int b0 = poff(PI_C_BREL);
    if (e >= b0 && e < b0 + 3 * ERF) v = g_brel[e - b0];
    else {
      int ent = 0;
#pragma unroll
      for (int j = 1; j < kNumColorParams; ++j) if (e >= poff(j)) ent = j;
      const int n_chunks = ra.chunks_of_entry[ent];      // fixed summation order -> deterministic
      for (int c = 0; c < n_chunks; ++c) v += slabs[(size_t)c * kColorFloats + e];
    }
  }
  g_params[e] = v;   // geometry-decoder group: 0 (fix_geo_decoder, configs/point_slam.yaml:47)
}

int launch_dw(psl_ctx* ctx, const DecodeArgs& a, float* g_params, const float* g_brel, hipStream_t s) {
  const bool relpos = a.flags & 0x10000;
  const long long P = a.P;
  if (ctx->dw_slab_cap < MAX_CHUNKS) {
    if (ctx->dw_slabs) PSL_HIP(hipFree(ctx->dw_slabs));
    PSL_HIP(hipMalloc(&ctx->dw_slabs, sizeof(float) * (size_t)kColorFloats * MAX_CHUNKS));
    ctx->dw_slab_cap = MAX_CHUNKS;
  }
  DwArgs d;
  DwReduceArgs ra;
  memset(&d, 0, sizeof(d));
  memset(&ra, 0, sizeof(ra));
  int nj = 0, base = 0;
  // every (chunk, tile) item writes its whole tile, so a slab entry is defined for exactly the chunks of its job:
  // no memset; the reduction reads chunks_of_entry[] slabs per parameter tensor
  auto add = [&](const float* A, int lda, int nv, const float* B0, int ldb0, int k0, const float* B1, int ldb1, int k1,
                 long long rows, int out_pi) {
    DwJob& J = d.job[nj++];
    J.A = A; J.lda = lda; J.n_valid = nv; J.B0 = B0; J.ldb0 = ldb0; J.k0_cols = k0; J.B1 = B1; J.ldb1 = ldb1;
    J.k1_cols = k1; J.rows = rows; J.out_off = poff(out_pi); J.ld_out = k0 + k1; J.bias_off = poff(out_pi + 1);
    J.n_tiles = (nv + 31) / 32; J.k_tiles = (k0 + k1 + 31) / 32;
    J.n_chunks = (int)std::min<long long>(std::max<long long>((rows + 255) / 256, 1), MAX_CHUNKS);
    long long rpc = (rows + J.n_chunks - 1) / J.n_chunks;
    J.rows_per_chunk = (int)((rpc + 31) / 32 * 32);
    J.n_chunks = (int)((rows + J.rows_per_chunk - 1) / J.rows_per_chunk);
    J.items = J.n_tiles * J.k_tiles * J.n_chunks;
    J.item_base = base; base += J.items;
    ra.chunks_of_entry[out_pi] = J.n_chunks;
    ra.chunks_of_entry[out_pi + 1] = J.n_chunks;
  };
  const RenderWs& w = a.ws;
  for (int i = 0; i < 5; ++i) {
    // saved activations are layer-major [5][Ppad][128]: the rows of one layer are one contiguous stream
    const size_t LS = (size_t)w.Ppad * HC;
    const float* dz = w.c_dz + i * LS;
    if (i == 0) add(dz, HC, HC, w.c_emb, EC, EC, nullptr, 0, 0, P, PI_C_L);
    else if (i == 3) add(dz, HC, HC, w.c_emb, EC, EC, w.c_hin + 2 * LS, HC, HC, P, PI_C_L + 6);
    else add(dz, HC, HC, w.c_hin + (i - 1) * LS, HC, HC, nullptr, 0, 0, P, PI_C_L + 2 * i);
    add(w.c_g + i * LS, HC, HC, w.cc, C, C, nullptr, 0, 0, P, PI_C_FCC + 2 * i);
  }
  add(w.d_out3, 4, 3, w.c_hin + 4 * (size_t)w.Ppad * HC, HC, HC, nullptr, 0, 0, P, PI_C_OUT);
  if (relpos) {
    add(w.n_dnf, C, C, w.n_h1, HC, HC, nullptr, 0, 0, P * K, PI_C_N2);
    add(w.n_dz1, HC, HC, w.n_x, NX, NX, nullptr, 0, 0, P * K, PI_C_N1);
  }
  d.n_jobs = nj; d.n_items = base; d.slabs = ctx->dw_slabs;
  hipLaunchKernelGGL(k_dw, dim3((unsigned)((base + 3) / 4)), dim3(256), 0, s, d);
  PSL_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_dw_reduce, dim3((kMasterFloats + 255) / 256), dim3(256), 0, s, ctx->dw_slabs, ra, g_brel,
                     g_params);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}

}  // namespace psl
