// Parameter gradients of the colour decoder: dW[n][k] = sum_p dZ[p][n] * X[p][k] for every linear layer,
// contracted over ALL samples of the batch with exact-fp32 MFMAs (one wavefront per 16x16 output tile per
// row-chunk), written to per-chunk slabs and reduced in a fixed order -> deterministic, no float atomics.
// (The reference gets these from ~500 ATen mm/sum launches in loss.backward(), Mapper.py:555.)
#include <cstring>
#include <cstdlib>
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
struct DwReduceArgs { int chunks_of_entry[kNumColorParams]; int slab_off[kNumColorParams]; };
// slab layout = master layout with every tensor start rounded up to 4 floats, so that a lane's 4 consecutive k
// (one float4) is 16-byte aligned in every layer (all row lengths are multiples of 4)
constexpr int SLAB_STRIDE = kColorFloats + 4 * kNumColorParams;
static int slab_off_of(int pi) { int o = 0; for (int j = 0; j < pi; ++j) o += (kParams[j].rows * kParams[j].cols + 3) / 4 * 4; return o; }

// One wavefront = one 64x64 output tile of one layer for one chunk of rows.  Each lane fetches ONE float4 of dZ
// (4 consecutive n-columns) and ONE float4 of X (4 consecutive k-columns) per 4-row step and feeds 16 MFMAs with
// them (element jn of the A quad x element jk of the B quad): 8x fewer load instructions per MFMA than a
// scalar-fragment tile and half the L2 traffic of 32x32 tiles.  MFMA tile (jn,jk) therefore owns the interleaved
// columns n = n0 + 4*i + jn, k = k0 + 4*j + jk; the epilogue writes 4 consecutive k per lane.
// 4 row-steps are issued per loop trip (8 independent 16-B loads, then 64 MFMAs on 16 independent accumulators).
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
  float* slab = d.slabs + (size_t)chunk * SLAB_STRIDE;
  const int nt = item / J.k_tiles, kt = item - nt * J.k_tiles;
  const int ktot = J.k0_cols + J.k1_cols;
  const int nq = 64 * nt + 4 * colw, kq = 64 * kt + 4 * colw;     // first column of this lane's quads
  // the quad never straddles the B0|B1 seam (k0_cols is a multiple of 4); columns past the end are masked and
  // their loads redirected to column 0 so that nothing is read outside the scratch buffer
  const bool nin = nq < J.n_valid, kin = kq < ktot;
  const float* ap = J.A + (nin ? nq : 0);
  const float* bp; int ldb;
  if (!kin) { bp = J.B0; ldb = J.ldb0; }
  else if (kq < J.k0_cols) { bp = J.B0 + kq; ldb = J.ldb0; }
  else { bp = J.B1 + (kq - J.k0_cols); ldb = J.ldb1; }
  bool am[4], bm[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { am[e] = nq + e < J.n_valid; bm[e] = kq + e < ktot; }
  f32x4 acc[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
  constexpr int U = 4;
  // software pipeline: the loads of trip t+1 are in flight while the 64 MFMAs of trip t issue
  float4 av[U], bv[U], an[U], bn[U];
  auto fetch = [&](long long r, float4 (&a_out)[U], float4 (&b_out)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long long row = r + 4 * u + g;
      bool v = row < r1;
      long long rr = v ? row : r0;
      float4 a4 = *reinterpret_cast<const float4*>(ap + rr * J.lda);
      float4 b4 = *reinterpret_cast<const float4*>(bp + rr * ldb);
      a_out[u] = make_float4((v && am[0]) ? a4.x : 0.f, (v && am[1]) ? a4.y : 0.f, (v && am[2]) ? a4.z : 0.f,
                             (v && am[3]) ? a4.w : 0.f);
      b_out[u] = make_float4(bm[0] ? b4.x : 0.f, bm[1] ? b4.y : 0.f, bm[2] ? b4.z : 0.f, bm[3] ? b4.w : 0.f);
    }
  };
  fetch(r0, av, bv);
  for (long long r = r0; r < r1; r += 4 * U) {
    const bool more = r + 4 * U < r1;
    if (more) fetch(r + 4 * U, an, bn);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float a_[4] = {av[u].x, av[u].y, av[u].z, av[u].w};
      const float b_[4] = {bv[u].x, bv[u].y, bv[u].z, bv[u].w};
#pragma unroll
      for (int x = 0; x < 4; ++x) {
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = mfma16(a_[x], b_[y], acc[x][y]);
        bs[x] += a_[x];
      }
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < U; ++u) { av[u] = an[u]; bv[u] = bn[u]; }
    }
  }
  // acc[x][y][q] (lane g,colw) = dW[n = 64nt + 4(4g+q) + x][k = 64kt + 4colw + y]
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int n = 64 * nt + 4 * (4 * g + q) + x;
      if (n < J.n_valid && kin)     // ktot is a multiple of 4: the quad is entirely inside or outside
        *reinterpret_cast<float4*>(slab + J.out_off + n * J.ld_out + kq) =
            make_float4(acc[x][0][q], acc[x][1][q], acc[x][2][q], acc[x][3][q]);
    }
  if (kt == 0) {   // bias gradient = column sums of dZ: this lane saw rows g, g+4, ... of columns nq..nq+3
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      float v = bs[x];
      v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
      if (g == 0 && nq + x < J.n_valid) slab[J.bias_off + nq + x] = v;
    }
  }
}

// g_params[e] = sum over the chunks of e's tensor, in a fixed order (deterministic): 32 consecutive elements x 8
// chunk lanes per workgroup (lane c sums chunks c, c+8, ...; the 8 partials are added in order through LDS), so a
// tensor reduced over 100+ chunks does not serialise 100+ dependent loads in one thread.
__global__ __launch_bounds__(256) void k_dw_reduce(const float* __restrict__ slabs, DwReduceArgs ra,
                                                   const float* __restrict__ g_brel, float* __restrict__ g_params) {
  __shared__ float part[8][32];
  const int el = threadIdx.x & 31, cl = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;
  float v = 0.f;
  bool brel = false;
  if (e < kColorFloats) {
    constexpr int b0 = poff(PI_C_BREL);
    if (e >= b0 && e < b0 + 3 * ERF) { brel = true; if (cl == 0) v = g_brel[e - b0]; }
    else {
      int ent = 0;
#pragma unroll
      for (int j = 1; j < kNumColorParams; ++j) if (e >= poff(j)) ent = j;
      const int n_chunks = ra.chunks_of_entry[ent];
      const int se = ra.slab_off[ent] + (e - poff(ent));
      for (int c = cl; c < n_chunks; c += 8) v += slabs[(size_t)c * SLAB_STRIDE + se];
    }
  }
  (void)brel;
  part[cl][el] = v;
  __syncthreads();
  if (cl == 0 && e < kMasterFloats) {
    float t = part[0][el];
#pragma unroll
    for (int c = 1; c < 8; ++c) t += part[c][el];
    g_params[e] = t;   // geometry-decoder group: 0 (fix_geo_decoder, configs/point_slam.yaml:47)
  }
}

int launch_dw(psl_ctx* ctx, const DecodeArgs& a, float* g_params, const float* g_brel, hipStream_t s) {
  const bool relpos = a.flags & 0x10000;
  const long long P = a.P;
  if (ctx->dw_slab_cap < MAX_CHUNKS) {
    if (ctx->dw_slabs) PSL_HIP(hipFree(ctx->dw_slabs));
    PSL_HIP(hipMalloc(&ctx->dw_slabs, sizeof(float) * (size_t)SLAB_STRIDE * MAX_CHUNKS));
    ctx->dw_slab_cap = MAX_CHUNKS;
  }
  static int chunk_rows = 0;
  if (!chunk_rows) { const char* e = getenv("PSL_DW_CHUNK"); chunk_rows = e ? atoi(e) : 256; if (chunk_rows < 16) chunk_rows = 256; }
  DwArgs d;
  DwReduceArgs ra;
  memset(&d, 0, sizeof(d));
  memset(&ra, 0, sizeof(ra));
  for (int j = 0; j < kNumColorParams; ++j) ra.slab_off[j] = slab_off_of(j);
  int nj = 0, base = 0;
  // every (chunk, tile) item writes its whole tile, so a slab entry is defined for exactly the chunks of its job:
  // no memset; the reduction reads chunks_of_entry[] slabs per parameter tensor
  auto add = [&](const float* A, int lda, int nv, const float* B0, int ldb0, int k0, const float* B1, int ldb1, int k1,
                 long long rows, int out_pi) {
    DwJob& J = d.job[nj++];
    J.A = A; J.lda = lda; J.n_valid = nv; J.B0 = B0; J.ldb0 = ldb0; J.k0_cols = k0; J.B1 = B1; J.ldb1 = ldb1;
    J.k1_cols = k1; J.rows = rows; J.out_off = slab_off_of(out_pi); J.ld_out = k0 + k1; J.bias_off = slab_off_of(out_pi + 1);
    J.n_tiles = (nv + 63) / 64; J.k_tiles = (k0 + k1 + 63) / 64;
    J.n_chunks = (int)std::min<long long>(std::max<long long>((rows + chunk_rows - 1) / chunk_rows, 1), MAX_CHUNKS);
    long long rpc = (rows + J.n_chunks - 1) / J.n_chunks;
    J.rows_per_chunk = (int)((rpc + 15) / 16 * 16);
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
  hipLaunchKernelGGL(k_dw_reduce, dim3((kMasterFloats + 31) / 32), dim3(256), 0, s, ctx->dw_slabs, ra, g_brel,
                     g_params);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}

}  // namespace psl
