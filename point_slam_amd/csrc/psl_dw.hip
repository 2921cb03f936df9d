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
  int aq, bq;                // columns per lane vector on the dZ / X side: 4 (64-wide tile) or 2 (32-wide)
  int item_base, items;      // items of this job = n_tiles*k_tiles*n_chunks
  int rows_per_chunk, n_chunks;
};
constexpr int MAX_JOBS = 16;
constexpr int MAX_CHUNKS = 256;
struct DwArgs { DwJob job[MAX_JOBS]; int n_jobs; int n_items; float* slabs; };
// slab layout = master layout with every tensor start rounded up to 4 floats, so that a lane's 4 consecutive k
// (one float4) is 16-byte aligned in every layer (all row lengths are multiples of 4)
constexpr int SLAB_STRIDE = kDwSlabStride;
static int slab_off_of(int pi) { int o = 0; for (int j = 0; j < pi; ++j) o += (kParams[j].rows * kParams[j].cols + 3) / 4 * 4; return o; }

// One wavefront = one (16*AQ) x (16*BQ) output tile of one layer for one chunk of rows (AQ, BQ in {4, 2}: 64 or 32
// columns; the 32-wide forms serve fc_c (32 inputs) and F_theta's second layer (32 outputs) without computing
// padding).  Each lane fetches ONE AQ-vector of dZ (consecutive n-columns) and ONE BQ-vector of X (consecutive
// k-columns) per 4-row step and feeds AQ*BQ MFMAs with them: 8x fewer load instructions per MFMA than a
// scalar-fragment tile and half the L2 traffic of 32x32 tiles.  MFMA tile (jn,jk) therefore owns the interleaved
// columns n = n0 + AQ*i + jn, k = k0 + BQ*j + jk; the epilogue writes BQ consecutive k per lane.
// 4 row-steps are issued per loop trip (8 independent loads, then up to 64 MFMAs on independent accumulators).
template <int Q> struct VecOf;
template <> struct VecOf<4> { using type = float4; };
template <> struct VecOf<2> { using type = float2; };
template <int Q> __device__ __forceinline__ void unpack(const typename VecOf<Q>::type& v, float (&o)[Q]);
template <> __device__ __forceinline__ void unpack<4>(const float4& v, float (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
template <> __device__ __forceinline__ void unpack<2>(const float2& v, float (&o)[2]) { o[0] = v.x; o[1] = v.y; }

template <int AQ, int BQ>
__device__ __forceinline__ void dw_tile(const DwJob& J, int item, int chunk, float* slab) {
  using AV = typename VecOf<AQ>::type;
  using BV = typename VecOf<BQ>::type;
  const int lane = threadIdx.x & 63, g = lane >> 4, colw = lane & 15;
  const long long r0 = (long long)chunk * J.rows_per_chunk;
  const long long r1 = min(J.rows, r0 + J.rows_per_chunk);
  const int nt = item / J.k_tiles, kt = item - nt * J.k_tiles;
  const int ktot = J.k0_cols + J.k1_cols;
  const int nq = 16 * AQ * nt + AQ * colw, kq = 16 * BQ * kt + BQ * colw;   // first column of this lane's vectors
  // a vector never straddles the B0|B1 seam (k0_cols is a multiple of 4); lanes whose columns lie past the end read
  // column 0 instead (finite, in-buffer) and their results are never stored
  const bool nin = nq < J.n_valid, kin = kq < ktot;
  const float* ap = J.A + (nin ? nq : 0);
  const float* bp; int ldb;
  if (!kin) { bp = J.B0; ldb = J.ldb0; }
  else if (kq < J.k0_cols) { bp = J.B0 + kq; ldb = J.ldb0; }
  else { bp = J.B1 + (kq - J.k0_cols); ldb = J.ldb1; }
  f32x4 acc[AQ][BQ];
#pragma unroll
  for (int x = 0; x < AQ; ++x)
#pragma unroll
    for (int y = 0; y < BQ; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bs[AQ];
#pragma unroll
  for (int x = 0; x < AQ; ++x) bs[x] = 0.f;
  constexpr int U = 4;
  // No element masks: columns past n_valid / ktot read finite in-buffer values whose products land in output
  // elements that are never stored (and in bias sums that are never stored).  Rows are masked in the tail trip only.
  // Software pipeline: the RAW loads of trip t+1 are issued before the MFMAs of trip t and not touched until
  // trip t+1 (any arithmetic on them here -- e.g. masking -- would make the compiler wait for them first).
  const float* a_ptr = ap + (r0 + g) * (long long)J.lda;
  const float* b_ptr = bp + (r0 + g) * (long long)ldb;
  const long long a_step = 4LL * J.lda, b_step = 4LL * ldb;
  const int nfull = (int)((r1 - r0) / (4 * U));
  AV av[U], an[U];
  BV bv[U], bn[U];
  auto fetch = [&](AV (&a_out)[U], BV (&b_out)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a_out[u] = *reinterpret_cast<const AV*>(a_ptr + u * a_step);
      b_out[u] = *reinterpret_cast<const BV*>(b_ptr + u * b_step);
    }
    a_ptr += U * a_step; b_ptr += U * b_step;
  };
  auto trip = [&](const AV (&a4)[U], const BV (&b4)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float a_[AQ], b_[BQ];
      unpack<AQ>(a4[u], a_);
      unpack<BQ>(b4[u], b_);
#pragma unroll
      for (int x = 0; x < AQ; ++x) {
#pragma unroll
        for (int y = 0; y < BQ; ++y) acc[x][y] = mfma16(a_[x], b_[y], acc[x][y]);
        bs[x] += a_[x];
      }
    }
  };
  if (nfull > 0) fetch(an, bn);
  for (int t = 0; t < nfull; ++t) {
#pragma unroll
    for (int u = 0; u < U; ++u) { av[u] = an[u]; bv[u] = bn[u]; }
    if (t + 1 < nfull) fetch(an, bn);
    trip(av, bv);
  }
  {  // tail: < 16 rows left; rows past r1 contribute zeros (A masked, B read from a clamped row)
    const long long rt = r0 + (long long)nfull * (4 * U);
    if (rt < r1) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long row = rt + 4 * u + g;
        const bool v = row < r1;
        const long long rr = v ? row : r0;
        AV a4 = *reinterpret_cast<const AV*>(ap + rr * J.lda);
        bv[u] = *reinterpret_cast<const BV*>(bp + rr * ldb);
        if (!v) {
          if constexpr (AQ == 4) a4 = make_float4(0.f, 0.f, 0.f, 0.f); else a4 = make_float2(0.f, 0.f);
        }
        av[u] = a4;
      }
      trip(av, bv);
    }
  }
  // acc[x][y][q] (lane g,colw) = dW[n = 16AQ*nt + AQ(4g+q) + x][k = 16BQ*kt + BQ*colw + y]
#pragma unroll
  for (int x = 0; x < AQ; ++x)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int n = 16 * AQ * nt + AQ * (4 * g + q) + x;
      if (n < J.n_valid && kin) {    // ktot is a multiple of 4: the vector is entirely inside or outside
        float* dst = slab + J.out_off + n * J.ld_out + kq;
        if constexpr (BQ == 4) *reinterpret_cast<float4*>(dst) = make_float4(acc[x][0][q], acc[x][1][q], acc[x][2][q], acc[x][3][q]);
        else *reinterpret_cast<float2*>(dst) = make_float2(acc[x][0][q], acc[x][1][q]);
      }
    }
  if (kt == 0) {   // bias gradient = column sums of dZ: this lane saw rows g, g+4, ... of columns nq..nq+AQ-1
#pragma unroll
    for (int x = 0; x < AQ; ++x) {
      float v = bs[x];
      v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
      if (g == 0 && nq + x < J.n_valid) slab[J.bias_off + nq + x] = v;
    }
  }
}

__global__ __launch_bounds__(256) void k_dw(DwArgs d) {
  __builtin_amdgcn_s_setprio(1);      // above the side-stream k-NN prefetch (see k_decode_fwd2)
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
  float* slab = d.slabs + (size_t)chunk * SLAB_STRIDE;
  if (J.aq == 4 && J.bq == 4) dw_tile<4, 4>(J, item, chunk, slab);
  else if (J.aq == 4) dw_tile<4, 2>(J, item, chunk, slab);
  else dw_tile<2, 4>(J, item, chunk, slab);
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
    PSL_HIP(hipMalloc(&ctx->dw_slabs, sizeof(float) * (size_t)SLAB_STRIDE * MAX_CHUNKS)); psl::poison(ctx->dw_slabs, sizeof(float) * (size_t)SLAB_STRIDE * MAX_CHUNKS);
    ctx->dw_slab_cap = MAX_CHUNKS;
    dbg_range("dw_slabs", ctx->dw_slabs, sizeof(float) * (size_t)SLAB_STRIDE * MAX_CHUNKS);
  }
  static int chunk_rows = -1;    // > 0: fixed rows per chunk (debug); default: balanced sizing, see finish()
  if (chunk_rows < 0) { const char* e = getenv("PSL_DW_CHUNK"); chunk_rows = e ? atoi(e) : 0; if (chunk_rows < 16) chunk_rows = 0; }
  DwArgs d;
  DwReduceArgs ra;
  memset(&d, 0, sizeof(d));
  memset(&ra, 0, sizeof(ra));
  for (int j = 0; j < kNumColorParams; ++j) ra.slab_off[j] = slab_off_of(j);
  int nj = 0;
  // every (chunk, tile) item writes its whole tile, so a slab entry is defined for exactly the chunks of its job:
  // no memset; the reduction reads chunks_of_entry[] slabs per parameter tensor
  int out_pi_of[MAX_JOBS];
  auto add = [&](const float* A, int lda, int nv, const float* B0, int ldb0, int k0, const float* B1, int ldb1, int k1,
                 long long rows, int out_pi) {
    DwJob& J = d.job[nj];
    out_pi_of[nj++] = out_pi;
    J.A = A; J.lda = lda; J.n_valid = nv; J.B0 = B0; J.ldb0 = ldb0; J.k0_cols = k0; J.B1 = B1; J.ldb1 = ldb1;
    J.k1_cols = k1; J.rows = rows; J.out_off = slab_off_of(out_pi); J.ld_out = k0 + k1; J.bias_off = slab_off_of(out_pi + 1);
    // 32-wide vectors where the operand has exactly 32 columns (fc_c: 32 inputs; F_theta linear2: 32 outputs)
    J.aq = (nv == 32) ? 2 : 4; J.bq = (k0 + k1 == 32 && J.aq == 4) ? 2 : 4;
    J.n_tiles = (nv + 16 * J.aq - 1) / (16 * J.aq); J.k_tiles = (k0 + k1 + 16 * J.bq - 1) / (16 * J.bq);
  };
  // Chunking: one item = one wavefront = one tile x one row chunk, and a wavefront is a serial chain of
  // rows*aq*bq/4 MFMAs.  Items are sized to EQUAL MFMA cost with about one item per SIMD (1024) in total: a batch of
  // 5 000 samples then finishes in one balanced round instead of 0.6 or 1.2 waves per SIMD of unequal length.
  auto finish = [&]() {
    double total = 0.0;
    for (int j = 0; j < nj; ++j) {
      const DwJob& J = d.job[j];
      total += (double)J.rows * J.n_tiles * J.k_tiles * (J.aq * J.bq) / 4.0;
    }
    static int n_items_target = -1;   // PSL_DW_ITEMS: wavefronts the work is cut into (default 1 000 ~ one per SIMD)
    if (n_items_target < 0) { const char* e = getenv("PSL_DW_ITEMS"); n_items_target = e ? atoi(e) : 1000; if (n_items_target < 64) n_items_target = 1000; }
    double per_item = std::max(total / (double)n_items_target, 64.0 * 16 / 4);      // MFMAs per item (>= 64 rows of a full tile)
    if (chunk_rows > 0) per_item = chunk_rows * 16 / 4.0;           // PSL_DW_CHUNK: rows per chunk of a FULL tile
    int base = 0;
    for (int j = 0; j < nj; ++j) {
      DwJob& J = d.job[j];
      long long want = (long long)(per_item * 4.0 / (J.aq * J.bq));  // rows per chunk for this job's tile size
      want = std::max<long long>((want + 15) / 16 * 16, 64);
      J.n_chunks = (int)std::min<long long>(std::max<long long>((J.rows + want - 1) / want, 1), MAX_CHUNKS);
      long long rpc = (J.rows + J.n_chunks - 1) / J.n_chunks;
      J.rows_per_chunk = (int)((rpc + 15) / 16 * 16);
      J.n_chunks = (int)((J.rows + J.rows_per_chunk - 1) / J.rows_per_chunk);
      J.items = J.n_tiles * J.k_tiles * J.n_chunks;
      J.item_base = base; base += J.items;
      ra.chunks_of_entry[out_pi_of[j]] = J.n_chunks;
      ra.chunks_of_entry[out_pi_of[j] + 1] = J.n_chunks;
    }
    return base;
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
  const int base = finish();
  d.n_jobs = nj; d.n_items = base; d.slabs = ctx->dw_slabs;
  PSL_KLAUNCH(k_dw, dim3((unsigned)((base + 3) / 4)), dim3(256), 0, s, d);
  PSL_LAUNCH_CHECK();
  ctx->dw_ra = ra;
  if (ctx->dw_defer_reduce) return PSL_OK;   // psl_map_iters: the Adam launch sums the chunk partials itself
  hipLaunchKernelGGL(k_dw_reduce, dim3((kMasterFloats + 31) / 32), dim3(256), 0, s, ctx->dw_slabs, ra, g_brel,
                     g_params);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}

}  // namespace psl
