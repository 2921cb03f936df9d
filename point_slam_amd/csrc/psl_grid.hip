// Uniform-grid spatial index + exact radius-bounded 8-NN for gfx950.
//
// Replaces FAISS-GPU IndexIVFFlat as used at src/neural_point.py:37-41,161-164,193
// (reference tree).  Design (MI355X-first, not a FAISS translation):
//   * points are counting-sorted by grid cell (cell >= max query radius) into `spos`
//     (float4: xyz + original index) so the candidates of an x-run of cells are ONE
//     contiguous, coalesced 16 B/lane stream;
//   * one 64-lane wavefront serves one ray (its 5 samples share most candidate cells) or one
//     free query; lanes evaluate 64 candidates per step, survivors are found with a
//     wave ballot and inserted into a wave-uniform sorted top-8 of 64-bit keys
//     (distance bits << 32 | index): a single integer compare gives the oracle's
//     (distance, lower-index-first) order, so results do not depend on scan order;
//   * only candidates with d2 <= r2 are ever admitted: slots beyond the query radius are
//     reported as (inf, -1) -- they carry weight 0 everywhere in the reference
//     (decoder.py:157,367; neural_point.py:210-213).
#include <cstdlib>
#include "psl_common.h"
#include "psl_pose.h"
#include "psl_device.h"

namespace psl {

// ------------------------------------------------------------------ grid build
__device__ __forceinline__ int f2ord(float f) {
  int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

__global__ void k_bounds_init(int* b) {
  if (threadIdx.x < 3) b[threadIdx.x] = 0x7FFFFFFF;
  else if (threadIdx.x < 6) b[threadIdx.x] = (int)0x80000000;
}

__global__ __launch_bounds__(256) void k_bounds(const float4* __restrict__ pos, int n, int* b) {
  int mn[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF};
  int mx[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float4 p = pos[i];
    int ox = f2ord(p.x), oy = f2ord(p.y), oz = f2ord(p.z);
    mn[0] = min(mn[0], ox); mn[1] = min(mn[1], oy); mn[2] = min(mn[2], oz);
    mx[0] = max(mx[0], ox); mx[1] = max(mx[1], oy); mx[2] = max(mx[2], oz);
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn[a] = min(mn[a], __shfl_xor(mn[a], o));
      mx[a] = max(mx[a], __shfl_xor(mx[a], o));
    }
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { atomicMin(&b[a], mn[a]); atomicMax(&b[3 + a], mx[a]); }
  }
}

__global__ void k_grid_meta(const int* b, int n, float min_cell, GridMeta* m) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float lo[3], hi[3];
  for (int a = 0; a < 3; ++a) { lo[a] = ord2f(b[a]); hi[a] = ord2f(b[3 + a]); }
  if (n == 0) { for (int a = 0; a < 3; ++a) { lo[a] = 0.f; hi[a] = 0.f; } }
  // a non-finite extent (an inf position in the cloud) would never fit the cell budget below: collapse that axis, the
  // offending points are clamped into its only cell by cell_coord
  for (int a = 0; a < 3; ++a) if (!(fabsf(hi[a] - lo[a]) < 3.0e38f)) { lo[a] = 0.f; hi[a] = 0.f; }
  float cell = min_cell * 1.001f;  // margin: a point within r <= min_cell never lands two cells away
  int nx, ny, nz;
  for (;;) {
    // (extents clamped as floats: a NaN / inf position in the cloud must not reach the float->int conversion)
    nx = (int)fminf(fmaxf(floorf((hi[0] - lo[0]) / cell), 0.f), 1.0e9f) + 1;
    ny = (int)fminf(fmaxf(floorf((hi[1] - lo[1]) / cell), 0.f), 1.0e9f) + 1;
    nz = (int)fminf(fmaxf(floorf((hi[2] - lo[2]) / cell), 0.f), 1.0e9f) + 1;
    if ((long long)nx * ny * nz <= (long long)kMaxCells) break;
    cell *= 1.26f;
  }
  m->ox = lo[0]; m->oy = lo[1]; m->oz = lo[2];
  m->cell = cell; m->inv_cell = 1.0f / cell;
  m->nx = nx; m->ny = ny; m->nz = nz; m->ncells = nx * ny * nz; m->npts = n;
  m->cnx = (nx + 3) / 4; m->cny = (ny + 3) / 4; m->cnz = (nz + 3) / 4;
}

__device__ __forceinline__ int cell_coord(float x, float o, float inv, int n) {
  // clamped as a float BEFORE the conversion: (int) of NaN / inf / |x| > 2^31 is undefined behaviour
  const float f = fminf(fmaxf(floorf((x - o) * inv), 0.f), (float)(n - 1));   // fmaxf(NaN, 0) = 0
  return (int)f;
}

__global__ __launch_bounds__(256) void k_cell_count(const float4* __restrict__ pos, const GridMeta* __restrict__ m,
                                                    int* cell_of, int* cell_fill, int* coarse) {
  int n = m->npts;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float4 p = pos[i];
    int cx = cell_coord(p.x, m->ox, m->inv_cell, m->nx);
    int cy = cell_coord(p.y, m->oy, m->inv_cell, m->ny);
    int cz = cell_coord(p.z, m->oz, m->inv_cell, m->nz);
    int c = (cz * m->ny + cy) * m->nx + cx;
    cell_of[i] = c;
    atomicAdd(&cell_fill[c], 1);
    // occupancy FLAG of the 4x4x4-cell block (wave_box_empty tests it against zero only): a plain store -- as a counter it was 10^6
    // atomics on ~10^4 addresses, the hot part of this kernel
    coarse[((cz >> 2) * m->cny + (cy >> 2)) * m->cnx + (cx >> 2)] = 1;
  }
}

// exclusive scan of cell_fill[0..ncells) -> cell_start[0..ncells], in 3 launches of 1024-wide blocks
constexpr int SCAN_B = 1024;  // elements per block (256 threads x 4)

__device__ __forceinline__ int block_excl_scan_256(int v, int* lds, int& total) {
  // exclusive scan of one int per thread across 256 threads
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { int y = __shfl_up(x, o); if (lane >= o) x += y; }
  if (lane == 63) lds[w] = x;
  __syncthreads();
  int woff = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) { int s = lds[i]; if (i < w) woff += s; tot += s; }
  __syncthreads();
  total = tot;
  return woff + x - v;
}

__global__ __launch_bounds__(256) void k_scan_block_sums(const int* __restrict__ in, const GridMeta* __restrict__ m,
                                                         int* block_sums) {
  __shared__ int lds[4];
  int n = m->ncells;
  int base = blockIdx.x * SCAN_B;
  if (base >= n) { if (threadIdx.x == 0) block_sums[blockIdx.x] = 0; return; }
  int s = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) { int i = base + threadIdx.x * 4 + j; if (i < n) s += in[i]; }
  int tot;
  block_excl_scan_256(s, lds, tot);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_scan_top(int* block_sums, int nblocks) {
  // nblocks <= 4096: each thread owns 16 consecutive entries
  __shared__ int lds[4];
  int v[16]; int s = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) { int i = threadIdx.x * 16 + j; v[j] = (i < nblocks) ? block_sums[i] : 0; s += v[j]; }
  int tot;
  int off = block_excl_scan_256(s, lds, tot);
#pragma unroll
  for (int j = 0; j < 16; ++j) { int i = threadIdx.x * 16 + j; if (i < nblocks) block_sums[i] = off; off += v[j]; }
}

__global__ __launch_bounds__(256) void k_scan_final(int* cell_fill, const GridMeta* __restrict__ m,
                                                    const int* __restrict__ block_sums, int* cell_start) {
  __shared__ int lds[4];
  int n = m->ncells;
  int base = blockIdx.x * SCAN_B;
  if (base >= n) return;
  int v[4]; int s = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) { int i = base + threadIdx.x * 4 + j; v[j] = (i < n) ? cell_fill[i] : 0; s += v[j]; }
  int tot;
  int off = block_excl_scan_256(s, lds, tot) + block_sums[blockIdx.x];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int i = base + threadIdx.x * 4 + j;
    if (i < n) { cell_start[i] = off; cell_fill[i] = 0; }
    off += v[j];
  }
  if (blockIdx.x == (n - 1) / SCAN_B && threadIdx.x == 0) cell_start[n] = m->npts;
}

__global__ __launch_bounds__(256) void k_scatter(const float4* __restrict__ pos, const GridMeta* __restrict__ m,
                                                 const int* __restrict__ cell_of, const int* __restrict__ cell_start,
                                                 int* cell_fill, float4* spos) {
  int n = m->npts;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    int c = cell_of[i];
    int slot = cell_start[c] + atomicAdd(&cell_fill[c], 1);
    float4 p = pos[i];
    p.w = __int_as_float(i);
    spos[slot] = p;
  }
}

int grid_build(psl_ctx* ctx, hipStream_t s) {
  int n = ctx->n_points;
  hipLaunchKernelGGL(k_bounds_init, dim3(1), dim3(64), 0, s, ctx->bounds);
  if (n > 0) {
    int nb = min((n + 255) / 256, 128);   // few blocks: every wave ends in 6 same-address atomics
    hipLaunchKernelGGL(k_bounds, dim3(nb), dim3(256), 0, s, ctx->pos, n, ctx->bounds);
  }
  // cell = a quarter of the largest query radius: ~10-20 points per occupied cell at 10^4 points/m^2, so that the first
  // (one-cell) ring of the expanding search already holds the 8 nearest neighbours
  hipLaunchKernelGGL(k_grid_meta, dim3(1), dim3(1), 0, s, ctx->bounds, n, 0.25f * ctx->cfg.max_query_radius, ctx->meta);
  PSL_HIP(hipMemsetAsync(ctx->cell_fill, 0, sizeof(int) * kMaxCells, s));
  PSL_HIP(hipMemsetAsync(ctx->coarse, 0, sizeof(int) * kMaxCoarse, s));
  const int nblk = kMaxCells / SCAN_B;
  if (n > 0) {
    int nb = min((n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_cell_count, dim3(nb), dim3(256), 0, s, ctx->pos, ctx->meta, ctx->cell_of, ctx->cell_fill, ctx->coarse);
  }
  hipLaunchKernelGGL(k_scan_block_sums, dim3(nblk), dim3(256), 0, s, ctx->cell_fill, ctx->meta, ctx->scan_tmp);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(256), 0, s, ctx->scan_tmp, nblk);
  hipLaunchKernelGGL(k_scan_final, dim3(nblk), dim3(256), 0, s, ctx->cell_fill, ctx->meta, ctx->scan_tmp,
                     ctx->cell_start);
  if (n > 0) {
    int nb = min((n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_scatter, dim3(nb), dim3(256), 0, s, ctx->pos, ctx->meta, ctx->cell_of, ctx->cell_start,
                       ctx->cell_fill, ctx->spos);
  }
  PSL_LAUNCH_CHECK();
  ctx->index_points = n;
  return PSL_OK;
}

// ------------------------------------------------------------------------ k-NN
typedef unsigned long long u64;

struct CellBox { int lo[3], hi[3]; };

__device__ __forceinline__ void box_of(const GridMeta& m, float x, float y, float z, float r, CellBox& bx) {
  float rr = r * 1.0001f + 1e-6f;
  bx.lo[0] = cell_coord(x - rr, m.ox, m.inv_cell, m.nx); bx.hi[0] = cell_coord(x + rr, m.ox, m.inv_cell, m.nx);
  bx.lo[1] = cell_coord(y - rr, m.oy, m.inv_cell, m.ny); bx.hi[1] = cell_coord(y + rr, m.oy, m.inv_cell, m.ny);
  bx.lo[2] = cell_coord(z - rr, m.oz, m.inv_cell, m.nz); bx.hi[2] = cell_coord(z + rr, m.oz, m.inv_cell, m.nz);
}

// Coarse occupancy test, one wavefront: true when NO point lies in the 4x4x4-cell blocks that overlap the box
// [lo - r, hi + r] -- then no point lies within r of anything inside [lo, hi] and the search is over before it began
// (queries in still-unmapped space otherwise walk the whole r-cube, 9^3 cells, to learn the same).
__device__ __forceinline__ bool wave_box_empty(const GridMeta& m, const int* __restrict__ coarse, float lox, float loy,
                                               float loz, float hix, float hiy, float hiz, float r) {
  const int lane = threadIdx.x & 63;
  const float rr = r * 1.0001f + 1e-6f;
  const int x0 = cell_coord(lox - rr, m.ox, m.inv_cell, m.nx) >> 2, x1 = max(x0, cell_coord(hix + rr, m.ox, m.inv_cell, m.nx) >> 2);
  const int y0 = cell_coord(loy - rr, m.oy, m.inv_cell, m.ny) >> 2, y1 = max(y0, cell_coord(hiy + rr, m.oy, m.inv_cell, m.ny) >> 2);
  const int z0 = cell_coord(loz - rr, m.oz, m.inv_cell, m.nz) >> 2, z1 = max(z0, cell_coord(hiz + rr, m.oz, m.inv_cell, m.nz) >> 2);
  const int nxb = x1 - x0 + 1, nyb = y1 - y0 + 1, nb = nxb * nyb * (z1 - z0 + 1);
  int any = 0;
  for (int e = lane; e < nb; e += 64) {
    const int ex = e % nxb, ey = (e / nxb) % nyb, ez = e / (nxb * nyb);
    any |= coarse[((z0 + ez) * m.cny + (y0 + ey)) * m.cnx + (x0 + ex)];
  }
  return __ballot(any != 0) == 0ull;
}

// One wavefront answers one query with an EXPANDING search: scan the cells overlapping the cube [q-rho, q+rho]
// (rho starts at one cell), keep the 8 smallest (d2, index) keys with d2 <= rho^2; if 8 were found, every
// unscanned point is farther than rho (some coordinate differs by more than rho) and the result is final;
// otherwise double rho, until rho reaches the query radius r (final pass admits d2 <= r2 exactly).  At 1 M points
// (~10^4 points/m^2) the first pass -- <= 27 cells, ~10^2 candidates -- almost always suffices, where a fixed
// r-sized neighbourhood holds several thousand candidates.
// The (cz,cy) rows of the cube are x-runs of cells = contiguous ranges of `spos`; their [begin,end) pairs are
// fetched 64 rows at a time, one row per lane, then walked with coalesced 16 B/lane candidate loads.
__device__ __forceinline__ unsigned dpp_shr1(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);   // row_shr:1 (lane 0 of a row keeps 0)
}
__device__ __forceinline__ u64 readlane64(u64 v, int l) {
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(v >> 32), l);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(v & 0xFFFFFFFFull), l);
  return ((u64)hi << 32) | lo;
}

// insertion of a wave-uniform key kk < thr into the lane-distributed sorted list (entry j in lane j < 8)
__device__ __forceinline__ void lane_list_insert(u64& mine, u64& thr, u64 kk) {
  const u64 prev = ((u64)dpp_shr1((unsigned)(mine >> 32)) << 32) | dpp_shr1((unsigned)(mine & 0xFFFFFFFFull));
  mine = (kk < prev) ? prev : ((kk < mine) ? kk : mine);
  thr = readlane64(mine, K - 1);
}

// One query, sorted top-8 ACROSS lanes (entry j in lane j < 8).  A wave-uniform list costs an 8-deep compare/select
// chain (~300 cycles) per accepted candidate, ~45 of them per query on the tracker's launches (58.5 k cycles per query in
// the trace; 49.9 k with this list and the second chunk in flight): here an insertion is one DPP shift + two 64-bit
// selects, the threshold one 64-bit readlane of lane 7.
// Same keys, same order: bit-identical answers to knn_scan_rows / wave_knn.
__device__ __forceinline__ void knn_scan_rows_lane(const GridMeta& m, const float4* __restrict__ spos,
                                                   const int* __restrict__ cell_start, float qx, float qy, float qz,
                                                   float re, u64& mine, u64& thr, unsigned long long& cand) {
  const int lane = threadIdx.x & 63, grp = lane >> 4, l16 = lane & 15;
  CellBox bx;
  box_of(m, qx, qy, qz, re, bx);
  const int ny_b = bx.hi[1] - bx.lo[1] + 1;
  const int nrows = (bx.hi[2] - bx.lo[2] + 1) * ny_b;
  auto row_range = [&](int row, int& beg, int& end) {
    beg = 0; end = 0;
    if (row < nrows) {
      const int cz = bx.lo[2] + row / ny_b, cy = bx.lo[1] + row % ny_b;
      const int rowbase = (cz * m.ny + cy) * m.nx;
      beg = cell_start[rowbase + bx.lo[0]];
      end = cell_start[rowbase + bx.hi[0] + 1];
    }
  };
  int beg, end, nbeg, nend;
  row_range(grp, beg, end);
  for (int rb = 0; rb < nrows; rb += 4) {
    row_range(rb + 4 + grp, nbeg, nend);
    int j = beg + l16;
    // two chunks of every row in flight: rows hold ~45 candidates, i.e. three dependent loads with one-deep prefetch
    float4 c = (j < end) ? spos[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 c1 = (j + 16 < end) ? spos[j + 16] : make_float4(0.f, 0.f, 0.f, 0.f);
    while (__ballot(j < end)) {
      const bool valid = j < end;
      const int jn = j + 16;
      const float4 c2 = (jn + 16 < end) ? spos[jn + 16] : make_float4(0.f, 0.f, 0.f, 0.f);
      cand += (unsigned long long)__popcll(__ballot(valid));
      const unsigned idx = __float_as_uint(c.w);
      const float d2 = dist2(c.x, c.y, c.z, qx, qy, qz);
      const u64 key = ((u64)__float_as_uint(d2) << 32) | idx;
      u64 mask = __ballot(valid && key < thr);
      while (mask) {
        lane_list_insert(mine, thr, readlane64(key, __builtin_ctzll(mask)));
        mask &= mask - 1;
        if (mask) mask &= __ballot(valid && key < thr);
      }
      j = jn; c = c1; c1 = c2;
    }
    beg = nbeg; end = nend;
  }
}

// expanding search with the lane-distributed list; on return lane j < 8 holds entry j in `mine`
__device__ __forceinline__ void wave_knn_lane(const GridMeta& m, const float4* __restrict__ spos,
                                              const int* __restrict__ cell_start, float qx, float qy, float qz, float r,
                                              float r2, u64& mine, unsigned long long& cand, const int* __restrict__ coarse,
                                              int& passes) {
  float rho = m.cell;
  passes = 0;
  if (coarse && wave_box_empty(m, coarse, qx, qy, qz, qx, qy, qz, r)) {
    mine = ((u64)__float_as_uint(r2) << 32) | 0xFFFFFFFFull;
    return;
  }
  for (;;) {
    const bool last = rho >= r;
    const float re = last ? r : rho;
    const float t2 = last ? r2 : __fmul_rn(rho, rho);
    const u64 sentinel = ((u64)__float_as_uint(t2) << 32) | 0xFFFFFFFFull;
    mine = sentinel;
    u64 thr = sentinel;
    knn_scan_rows_lane(m, spos, cell_start, qx, qy, qz, re, mine, thr, cand);
    ++passes;
    if (last || thr != sentinel) break;
    rho *= 2.0f;
  }
}

// PSL_KNN_TRACE=1: per-query cost distribution of the one-wavefront-per-sample kernel (shader cycles, candidates, passes),
// dumped by psl_debug_option("knn_trace_dump", 1)
struct KnnTrace { unsigned long long n, sum_cyc, max_cyc, sum_cand, max_cand, sum_pass, hist_cyc[24], hist_cand[24], hist_pass[4]; };
__device__ KnnTrace g_knn_trace;
__device__ __forceinline__ int log2_bucket(unsigned long long v) { int b = 0; while (v > 1 && b < 23) { v >>= 1; ++b; } return b; }

// ---------------------------------------------------------------------------------------------------------------
// Flat candidate enumeration for SMALL launches (the tracker: 1 000 queries, about one wavefront per SIMD).  The sorted
// positions (16 MB at 1 M points) and the cell table do not fit an XCD's 4 MB L2, so every dependent step of a query is a
// ~2 k-cycle trip to the memory side, and a lone wavefront hides none of it: the row walk above costs a pass over the r-cube
// (81 rows) ~60 k cycles, a three-pass query 130-240 k (PSL_KNN_TRACE: 2 % of the queries, and the launch lasts as long as
// its slowest query).  Here a pass is TWO dependent trips however large its box:
//   1. the [begin, end) ranges of ALL rows of the box, one or two per lane, requested together; a wave scan of their
//      lengths gives every candidate a position in the concatenated list (empty rows drop out);
//   2. lane l owns the contiguous slice [l q, (l + 1) q) of that list, q = ceil(T / 64), and requests its candidates eight
//      at a time -- up to 512 records in flight per wavefront instead of 48.
// A box with fewer than 8 candidates cannot close the search and is not scanned at all.  Same keys, same order relation,
// bit-identical answers to knn_scan_rows_lane.
int g_knn_start_hint = 3;                      // psl_debug_option("knn_start_hint", bits): see knn_rays (bits 1 / 2 of the kernels' `trace` argument switch the two mechanisms off)
constexpr int kFlatRows = 128;                  // >= rows of all passes of a query together (9 + 25 + 81 at cell = r_max / 4)
constexpr int kFlatPasses = 3;                 // cell = r_max / 4: radii cell, 2 cell, r
struct FlatLds { int rbeg[kFlatRows]; int rcnt[kFlatRows];            // [begin, length) of every row of every pass
                 float rd2[kFlatRows];                                  // lower bound of the squared (y, z) distance from the query to the row
                 int beg[kFlatRows]; int cnt[kFlatRows]; int off[kFlatRows + 1]; };   // non-empty rows of the current pass

// [begin, length) of row (cz, cy) of the box of radius `re` around q, TRIMMED to the sphere: a row whose (y, z) interval
// lies farther than re from q holds no candidate, and along x only the chord of the sphere matters -- a cube of the
// query radius holds ~2x the candidates of its inscribed sphere, and the slowest query of a launch sets its duration.
// Conservative: the box inflation of box_of plus a margin on the cell intervals; boundary cells are unbounded (points
// beyond the grid's extent are clamped into them).
__device__ __forceinline__ void flat_row_range(const GridMeta& m, const int* __restrict__ cell_start, const CellBox& bx, int row,
                                               int ny_b, float qx, float qy, float qz, float re, int& beg, int& cnt, float& d2yz) {
  const int cz = bx.lo[2] + row / ny_b, cy = bx.lo[1] + row % ny_b;
  const float rr = re * 1.0001f + 1e-6f, eps = 1e-4f * m.cell;
  float dy = 0.f, dz = 0.f;
  {
    const float lo = m.oy + (float)cy * m.cell, hi = lo + m.cell;
    if (qy < lo && cy > 0) dy = lo - qy; else if (qy > hi && cy < m.ny - 1) dy = qy - hi;
    const float lz = m.oz + (float)cz * m.cell, hz = lz + m.cell;
    if (qz < lz && cz > 0) dz = lz - qz; else if (qz > hz && cz < m.nz - 1) dz = qz - hz;
    dy = fmaxf(dy - eps, 0.f); dz = fmaxf(dz - eps, 0.f);
  }
  d2yz = dy * dy + dz * dz;
  const float w2 = rr * rr - d2yz;
  beg = 0; cnt = 0;
  if (!(w2 >= 0.f)) return;
  const float w = sqrtf(w2) * 1.0001f + eps;
  const int x0 = max(bx.lo[0], cell_coord(qx - w, m.ox, m.inv_cell, m.nx)), x1 = min(bx.hi[0], cell_coord(qx + w, m.ox, m.inv_cell, m.nx));
  if (x1 < x0) return;
  const int rowbase = (cz * m.ny + cy) * m.nx;
  beg = cell_start[rowbase + x0];
  cnt = cell_start[rowbase + x1 + 1] - beg;
}

// one pass over the rows [row0, row0 + nrows) of the table that wave_knn_flat has filled
template <int U>
__device__ __forceinline__ void knn_scan_flat(const float4* __restrict__ spos, float qx, float qy, float qz, u64& mine, u64& thr,
                                              unsigned long long& cand, FlatLds& L, int row0, int nrows, bool need_full, float dmax2) {
  const int lane = threadIdx.x & 63;
  // compact the non-empty rows into the pass table: position = number of non-empty rows before this one
  int n_tab = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = lane + 64 * h;
    const int b = row < nrows ? L.rbeg[row0 + row] : 0;
    const int c = (row < nrows && !(L.rd2[row0 + row] > dmax2)) ? L.rcnt[row0 + row] : 0;   // rows beyond the bound a smaller pass left hold nothing
    const u64 ne = __ballot(c > 0);
    if (c > 0) { const int k = n_tab + __popcll(ne & ((1ull << lane) - 1ull)); L.beg[k] = b; L.cnt[k] = c; }
    n_tab += __popcll(ne);
  }
  wave_lds_sync();
  // exclusive prefix sums of the table's lengths (<= 128 entries: two per lane, wave scan)
  int c0 = lane < n_tab ? L.cnt[lane] : 0, c1 = lane + 64 < n_tab ? L.cnt[lane + 64] : 0;
  int x0 = c0, x1 = c1;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int y0 = __shfl_up(x0, o), y1 = __shfl_up(x1, o); if (lane >= o) { x0 += y0; x1 += y1; } }
  const int tot0 = __shfl(x0, 63);
  const int T = tot0 + __shfl(x1, 63);
  if (lane < n_tab) L.off[lane] = x0 - c0;
  if (lane + 64 < n_tab) L.off[lane + 64] = tot0 + x1 - c1;
  if (lane == 0) L.off[n_tab] = T;
  wave_lds_sync();
  cand += (unsigned long long)T;
  if (T == 0 || (!need_full && T < K)) return;           // fewer than 8 candidates cannot close the search at this radius
  // ---- this lane's slice of the concatenated list
  const int q = (T + 63) >> 6;
  const int j0 = lane * q;
  int tr = 0;                                    // table row that holds candidate j0: last tr with off[tr] <= j0
  {
    int lo = 0, hi = n_tab - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (L.off[mid] <= j0) lo = mid; else hi = mid - 1; }
    tr = lo;
  }
  int cur = 0, row_end = 0;
  if (j0 < T) { cur = L.beg[tr] + (j0 - L.off[tr]); row_end = L.beg[tr] + L.cnt[tr]; }
  int left = min(q, max(T - j0, 0));
  // ---- U records per lane in flight (repeated for passes of more than 64 U candidates)
  for (int it = 0; it < q; it += U) {
    float4 c[U];
    bool v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[u] = left > 0;
      c[u] = v[u] ? spos[cur] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (v[u]) {
        --left; ++cur;
        if (cur == row_end && left > 0) { ++tr; cur = L.beg[tr]; row_end = cur + L.cnt[tr]; }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned idx = __float_as_uint(c[u].w);
      const float d2 = dist2(c[u].x, c[u].y, c[u].z, qx, qy, qz);
      const u64 key = ((u64)__float_as_uint(d2) << 32) | idx;
      u64 mask = __ballot(v[u] && key < thr);
      while (mask) {
        lane_list_insert(mine, thr, readlane64(key, __builtin_ctzll(mask)));
        mask &= mask - 1;
        if (mask) mask &= __ballot(v[u] && key < thr);
      }
    }
  }
}

template <int U>
__device__ __forceinline__ void wave_knn_flat(const GridMeta& m, const float4* __restrict__ spos,
                                              const int* __restrict__ cell_start, float qx, float qy, float qz, float r,
                                              float r2, u64& mine, unsigned long long& cand, const int* __restrict__ coarse,
                                              int& passes, FlatLds& L, bool start_hint, bool carry_bound) {
  const int lane = threadIdx.x & 63;
  passes = 0;
  // the radii of the expanding search (one cell, doubling, the last one = r) and their boxes: known before any load.
  // (Fully unrolled with compile-time indices: the per-pass records stay in registers.)
  float re[kFlatPasses];
  CellBox bx[kFlatPasses];
  int nyb[kFlatPasses], base[kFlatPasses + 1];
  int np = 0;
  base[0] = 0;
  {
    float rho = m.cell;
    bool done = false;
#pragma unroll
    for (int k = 0; k < kFlatPasses; ++k) {
      const bool last = rho >= r;
      re[k] = last ? r : rho;
      box_of(m, qx, qy, qz, re[k], bx[k]);
      nyb[k] = bx[k].hi[1] - bx[k].lo[1] + 1;
      const int rows = done ? 0 : (bx[k].hi[2] - bx[k].lo[2] + 1) * nyb[k];
      base[k + 1] = base[k] + rows;
      if (!done) np = k + 1;
      done = done || last;
      rho *= 2.0f;
    }
    if (!done || base[kFlatPasses] > kFlatRows) {   // (never with cell = r_max / 4) -- the row walk of k_knn_rays
      wave_knn_lane(m, spos, cell_start, qx, qy, qz, r, r2, mine, cand, coarse, passes);
      return;
    }
  }
  // ONE trip for the row ranges of ALL passes (two rows per lane) and, concurrently, the coarse occupancy test
  int rb[2], rc[2];
  float rd[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int gr = lane + 64 * h;
    rb[h] = 0; rc[h] = 0; rd[h] = 0.f;
#pragma unroll
    for (int k = 0; k < kFlatPasses; ++k)
      if (gr >= base[k] && gr < base[k + 1]) flat_row_range(m, cell_start, bx[k], gr - base[k], nyb[k], qx, qy, qz, re[k], rb[h], rc[h], rd[h]);
  }
  const bool empty = coarse && wave_box_empty(m, coarse, qx, qy, qz, qx, qy, qz, r);
  if (empty) { mine = ((u64)__float_as_uint(r2) << 32) | 0xFFFFFFFFull; return; }
#pragma unroll
  for (int h = 0; h < 2; ++h) { const int gr = lane + 64 * h; if (gr < base[kFlatPasses]) { L.rbeg[gr] = rb[h]; L.rcnt[gr] = rc[h]; L.rd2[gr] = rd[h]; } }
  wave_lds_sync();
  // Where to start (round 6).  A pass whose whole candidate list fits ONE trip of U records per lane costs the same whatever its
  // radius (two dependent memory trips + the table), and a larger radius closes the search wherever a smaller one would have, and more
  // often: the search starts at the LARGEST pass of at most 64 U candidates -- known from the row lengths, before anything is scanned.
  // Any starting pass gives the same answer: a pass closes only with eight points inside its own radius, all of which it has scanned.
  int start = 0;
  if (start_hint) {
    unsigned pk = 0;                          // bits 0..15: candidates of pass 1, bits 16..31: of pass 2 (rows clamped to 511: 128 rows < 2^16)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int gr = lane + 64 * h;
      const unsigned c = (unsigned)min(rc[h], 511);
      if (gr >= base[1] && gr < base[2]) pk += c;
      if (gr >= base[2] && gr < base[3]) pk += c << 16;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pk += (unsigned)__shfl_xor((int)pk, o);
    if (np >= 2 && (pk & 0xFFFFu) <= 64u * U) start = 1;
    if (np >= 3 && (pk >> 16) <= 64u * U) start = 2;
  }
  // Every pass admits candidates up to the QUERY radius, not only up to its own (round 6): a pass that does not close -- fewer than
  // eight points inside its radius -- still leaves the eighth-best distance of its box, an upper bound of the answer's, and the next pass
  // starts from it: its list admits nothing farther, and rows of its box whose (y, z) distance exceeds the bound are not read at all.
  // (Before, the last pass of a sample 10 cm off a surface read the 1 000-3 600 points of the whole 16-cm sphere to find eight at 10-11 cm;
  //  the slowest query of a launch sets its duration.)  Same answers: a pass closes when its eighth entry lies inside its own radius.
  const u64 sentinel_r = ((u64)__float_as_uint(r2) << 32) | 0xFFFFFFFFull;
  u64 carry = 0ull;
  bool closed = false;
#pragma unroll
  for (int k = 0; k < kFlatPasses; ++k) {
    if (k < np && k >= start && !closed) {
      const bool last = k == np - 1;
      const float t2 = last ? r2 : __fmul_rn(re[k], re[k]);
      const u64 sentinel_k = ((u64)__float_as_uint(t2) << 32) | 0xFFFFFFFFull;
      // the bound is only usable where this pass reads every point inside it: its own radius must reach the bound (always true for the
      // last pass, whose radius is the query radius)
      const bool bounded = carry_bound && carry != 0ull && (last || (unsigned)(carry >> 32) <= __float_as_uint(t2));
      const u64 init = bounded ? carry : (carry_bound ? sentinel_r : sentinel_k);
      mine = init;
      u64 thr = init;
      const float dmax2 = bounded ? __uint_as_float((unsigned)(carry >> 32)) : __int_as_float(0x7F800000);
      knn_scan_flat<U>(spos, qx, qy, qz, mine, thr, cand, L, base[k], base[k + 1] - base[k], last || bounded, dmax2);
      ++passes;
      const bool full = thr != init;                       // the eighth slot holds a point
      closed = last || (full && thr <= sentinel_k);
      // next float above the eighth-best distance, "no point" index: admits the eighth-best itself when the next pass meets it again, and a
      // slot that stayed empty would decode as "no neighbour" (cannot happen: the eight points of this pass lie inside the next one's reach)
      const u64 up = ((u64)((unsigned)(thr >> 32) + 1u) << 32) | 0xFFFFFFFFull;
      carry = (carry_bound && full) ? (up < sentinel_r ? up : sentinel_r) : 0ull;
    }
  }
  // (slots that never received a point keep `init`, whose index part is the "no neighbour" marker in every case)
}

// ray mode, small launches: one wave per SAMPLE with the flat enumeration; outputs as k_knn_rays
// <U, MINW>: records in flight per lane and the wavefronts per SIMD the register budget is cut for.  <8, 5> (81 VGPRs, six resident) is the
// latency shape of the small launches; <4, 8> (64 VGPRs: a third more queries resident per CU) is the default from 5 000
// queries on, where the chip is filled several times over (PSL_KNN_FLAT_LARGE, see knn_rays)
// The pose step of the tracker's previous iteration, by every workgroup of the k-NN launch (TrackPose, psl_pose.h).  The ray
// gradients are reduced exactly as k_track_pre does it with one ray per thread of 1 024: "virtual wavefront" vw = rays 64 vw .. 64 vw + 63
// is summed by a butterfly, the wavefront totals are added in order by one thread, which then steps the pose on private copies.
__device__ __forceinline__ void track_pose_prologue(const TrackPose& tp, float (*red)[12], float* s_pose) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (!tp.do_step) {
    if (tid < 7) s_pose[tid] = tp.pose_in[tid];
    __syncthreads();
    return;
  }
  const int nw = min((tp.n + 63) >> 6, 16);
  float pose0[7], mv0[14];                 // requested before the reduction (uniform addresses)
#pragma unroll
  for (int j = 0; j < 7; ++j) pose0[j] = tp.pose_in[j];
#pragma unroll
  for (int j = 0; j < 14; ++j) mv0[j] = tp.adam_in[j];
  for (int vw = w; vw < nw; vw += 4) {
    float acc[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) acc[j] = 0.f;
    const int r = 64 * vw + lane;
    if (r < tp.n) {
      // g_rays_o = sum_s dp_s ; g_rays_d = sum_s z_s dp_s  (k_ray_grad)
      float go[3] = {0.f, 0.f, 0.f}, gd3[3] = {0.f, 0.f, 0.f};
      const float gt = tp.gd_prev[r];
#pragma unroll
      for (int s = 0; s < S; ++s) {
        float4 g = tp.dp[r * S + s];
        if (tp.dp2) { const float4 g2 = tp.dp2[r * S + s]; g.x += g2.x; g.y += g2.y; g.z += g2.z; }
        const float z = sample_z(gt, s, tp.near_s, tp.far_s);
        go[0] += g.x; go[1] += g.y; go[2] += g.z;
        gd3[0] += z * g.x; gd3[1] += z * g.y; gd3[2] += z * g.z;
      }
      const float d0 = tp.dirs_prev[r * 3], d1 = tp.dirs_prev[r * 3 + 1], d2 = tp.dirs_prev[r * 3 + 2];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        acc[a * 3 + 0] += d0 * gd3[a]; acc[a * 3 + 1] += d1 * gd3[a]; acc[a * 3 + 2] += d2 * gd3[a];   // dL/dR[a][k]
        acc[9 + a] += go[a];                                                                          // dL/dT[a]
      }
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc[j] += __shfl_xor(acc[j], o);
    }
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < 12; ++j) red[vw][j] = acc[j];
    }
  }
  __syncthreads();
  // wavefront totals added in order, one component per lane; then one thread steps the pose
  float tj = 0.f;
  if (tid < 12) for (int q = 0; q < nw; ++q) tj += red[q][tid];
  if (tid < 64) {
    float G[3][3], gT[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int k = 0; k < 3; ++k) G[a][k] = __shfl(tj, a * 3 + k);
      gT[a] = __shfl(tj, 9 + a);
    }
    if (tid == 0) {
      float pose[7], mv[14];
#pragma unroll
      for (int j = 0; j < 7; ++j) pose[j] = pose0[j];
#pragma unroll
      for (int j = 0; j < 14; ++j) mv[j] = mv0[j];
      pose_adam(G, gT, pose, mv, tp.step, tp.lr_T, tp.lr_q, &tp.bias);
#pragma unroll
      for (int j = 0; j < 7; ++j) s_pose[j] = pose[j];
      if (blockIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < 7; ++j) tp.pose_out[j] = pose[j];
#pragma unroll
        for (int j = 0; j < 14; ++j) tp.adam_out[j] = mv[j];
      }
    }
  }
  __syncthreads();
}

// POSE: 0 = rays from memory; 1 = the tracker's pose step in the prologue, rays turned from camera-frame directions (TrackPose);
// 2 = directions turned with the pose in memory, no step (tracker batches above 1 024 rays: their pose step stays a launch of its own)
template <int U, int MINW, int POSE>
__global__ __launch_bounds__(256, MINW) void k_knn_rays_flat(const GridMeta* __restrict__ meta, const float4* __restrict__ spos,
                                                       const int* __restrict__ cell_start,
                                                       const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                       const float* __restrict__ depth, const float* __restrict__ z_vals,
                                                       const float* __restrict__ r_query,
                                                       float r_fixed, float r2_fixed, float near_s, float far_s, int n_rays,
                                                       int* __restrict__ I_out, int* __restrict__ cnt_out,
                                                       unsigned long long* __restrict__ cand_counter, const int* __restrict__ coarse,
                                                       int trace, TrackPose tp) {
  __shared__ FlatLds lds[4];
  __shared__ float s_red[POSE == 1 ? 16 : 1][12];
  __shared__ float s_pose[8];
  const int p_raw = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  const int p = POSE == 1 ? min(p_raw, n_rays * S - 1) : p_raw;      // (POSE 1: every wavefront goes through the prologue's barriers)
  if (POSE != 1 && p >= n_rays * S) return;
  const unsigned long long t0 = (trace & 1) ? clock64() : 0ull;
  const int ray = p / S, si = p - ray * S;
  const int lane = threadIdx.x & 63;
  // everything that does not depend on the pose is requested in front of the prologue
  const GridMeta m = *meta;
  const float zq = z_vals ? z_vals[p] : sample_z(depth[ray], si, near_s, far_s);
  float r, r2;
  if (r_query) { r = r_query[ray]; r2 = __fmul_rn(r, r); } else { r = r_fixed; r2 = r2_fixed; }
  float dc0 = 0.f, dc1 = 0.f, dc2 = 0.f;
  if constexpr (POSE != 0) { dc0 = tp.dirs[ray * 3]; dc1 = tp.dirs[ray * 3 + 1]; dc2 = tp.dirs[ray * 3 + 2]; }
  float ps[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if constexpr (POSE == 1) {
    track_pose_prologue(tp, s_red, s_pose);
    if (p_raw >= n_rays * S) return;
#pragma unroll
    for (int j = 0; j < 7; ++j) ps[j] = s_pose[j];
  } else if constexpr (POSE == 2) {
#pragma unroll
    for (int j = 0; j < 7; ++j) ps[j] = tp.pose_in[j];
  }
  float qx, qy, qz;
  if constexpr (POSE != 0) {
    // get_rays_from_uv with the (stepped) pose (ray_setup_one, psl_slam.hip): the same expressions
    float q[4] = {ps[0], ps[1], ps[2], ps[3]}, R[3][3], o[3] = {ps[4], ps[5], ps[6]}, d[3];
    quat_to_rot(q, R);
    const float d0 = dc0, d1 = dc1, d2 = dc2;
#pragma unroll
    for (int a = 0; a < 3; ++a) d[a] = __fadd_rn(__fadd_rn(__fmul_rn(d0, R[a][0]), __fmul_rn(d1, R[a][1])), __fmul_rn(d2, R[a][2]));
    if (si == 0 && lane < 3) { tp.rays_o[ray * 3 + lane] = o[lane]; tp.rays_d[ray * 3 + lane] = d[lane]; }
    sample_point(o[0], o[1], o[2], d[0], d[1], d[2], zq, qx, qy, qz);
  } else {
    sample_point(rays_o[ray * 3], rays_o[ray * 3 + 1], rays_o[ray * 3 + 2], rays_d[ray * 3], rays_d[ray * 3 + 1],
                 rays_d[ray * 3 + 2], zq, qx, qy, qz);
  }
  u64 mine;
  unsigned long long n_cand = 0;
  int n_pass = 0;
  wave_knn_flat<U>(m, spos, cell_start, qx, qy, qz, r, r2, mine, n_cand, coarse, n_pass, lds[(threadIdx.x >> 6) & 3], (trace & 2) == 0, (trace & 4) == 0);
  const unsigned ib = (unsigned)(mine & 0xFFFFFFFFull), db = (unsigned)(mine >> 32);
  const int cnt = __popcll(__ballot(lane < K && ib != 0xFFFFFFFFu && db < __float_as_uint(r2)));
  if (lane < K) I_out[p * K + lane] = (ib == 0xFFFFFFFFu) ? -1 : (int)ib;
  if (lane == 0) { cnt_out[p] = cnt; if (cand_counter) atomicAdd(cand_counter + 8 * (blockIdx.x & (kKnnCandSlots - 1)), n_cand); }
  if ((trace & 1) && lane == 0) {
    const unsigned long long cyc = clock64() - t0;
    atomicAdd(&g_knn_trace.n, 1ull); atomicAdd(&g_knn_trace.sum_cyc, cyc); atomicMax(&g_knn_trace.max_cyc, cyc);
    atomicAdd(&g_knn_trace.sum_cand, n_cand); atomicMax(&g_knn_trace.max_cand, n_cand);
    atomicAdd(&g_knn_trace.sum_pass, (unsigned long long)n_pass);
    atomicAdd(&g_knn_trace.hist_cyc[log2_bucket(cyc)], 1ull); atomicAdd(&g_knn_trace.hist_cand[log2_bucket(n_cand + 1)], 1ull);
    atomicAdd(&g_knn_trace.hist_pass[min(n_pass, 3)], 1ull);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Ray mode, one wavefront per RAY.  The five samples of a ray lie within +-2 % (4 %) of the sensor depth of each
// other -- a few centimetres -- while the query radius is 4..16 cm, so their search cubes overlap almost entirely:
// the cells are read ONCE and every candidate is tested against all five samples.
//  * the rows of the (union) cell box are walked four at a time, 16 lanes per row: rows hold ~10..40 points, a
//    64-wide step per row would leave most lanes idle and serialise one memory latency per row;
//  * the five sorted top-8 lists live ACROSS lanes (list s in lanes 8 s .. 8 s + 7, one 64-bit key per lane): an
//    insertion is one DPP shift plus two 64-bit selects for all five lists at once, one candidate per list per step,
//    instead of an 8-deep compare/select chain on wave-uniform registers per candidate;
//  * results are the same keys (distance bits << 32 | index) as in the one-query kernel: bit-identical answers.
__device__ __forceinline__ void knn_ray2_one(int ray, const GridMeta& m, const float4* __restrict__ spos,
                                             const int* __restrict__ cell_start,
                                             const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                             const float* __restrict__ depth, const float* __restrict__ z_vals,
                                             const float* __restrict__ r_query,
                                             float r_fixed, float r2_fixed, float near_s, float far_s,
                                             int* __restrict__ I_out, int* __restrict__ cnt_out,
                                             unsigned long long& cand_total, const int* __restrict__ coarse) {
  const int lane = threadIdx.x & 63, grp = lane >> 4, l16 = lane & 15;
  float r, r2;
  if (r_query) { r = r_query[ray]; r2 = __fmul_rn(r, r); } else { r = r_fixed; r2 = r2_fixed; }
  float qx[S], qy[S], qz[S];
  float lox = 3.0e38f, loy = 3.0e38f, loz = 3.0e38f, hix = -3.0e38f, hiy = -3.0e38f, hiz = -3.0e38f;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const float zq = z_vals ? z_vals[ray * S + s] : sample_z(depth[ray], s, near_s, far_s);
    sample_point(rays_o[ray * 3], rays_o[ray * 3 + 1], rays_o[ray * 3 + 2], rays_d[ray * 3], rays_d[ray * 3 + 1],
                 rays_d[ray * 3 + 2], zq, qx[s], qy[s], qz[s]);
    lox = fminf(lox, qx[s]); loy = fminf(loy, qy[s]); loz = fminf(loz, qz[s]);
    hix = fmaxf(hix, qx[s]); hiy = fmaxf(hiy, qy[s]); hiz = fmaxf(hiz, qz[s]);
  }
  if (wave_box_empty(m, coarse, lox, loy, loz, hix, hiy, hiz, r)) {     // the whole ray segment lies in unmapped space
    if (lane < S * K) I_out[(size_t)ray * S * K + lane] = -1;
    if (lane < S) cnt_out[ray * S + lane] = 0;
    return;
  }
  // this lane's slot of the five lists: list = lane >> 3 (lanes >= 40 idle), entry = lane & 7
  const int my_list = lane >> 3;
  u64 mine = ~0ull;
  unsigned done_mask = 0;                      // bit s: list s is final
  unsigned long long n_cand = 0;
  float rho = m.cell;
  for (;;) {
    const bool last = rho >= r;
    const float re = last ? r : rho;
    const float t2 = last ? r2 : __fmul_rn(rho, rho);
    const u64 sentinel = ((u64)__float_as_uint(t2) << 32) | 0xFFFFFFFFull;
    // lists that are not final start over with the new threshold
    if (my_list < S && !((done_mask >> my_list) & 1u)) mine = sentinel;
    u64 thr[S];
#pragma unroll
    for (int s = 0; s < S; ++s) thr[s] = ((done_mask >> s) & 1u) ? 0ull : sentinel;     // a final list admits nothing
    // union of the five cubes [q_s - re, q_s + re]: a superset of every sample's own cube
    // (a NaN coordinate -- fminf/fmaxf skip it -- would leave lo = +3e38 > hi = -3e38, an inverted box with NEGATIVE
    //  extents whose "rows" index cell_start far outside the grid: the upper corner is clamped to the lower one)
    const float rr = re * 1.0001f + 1e-6f;
    const int bx0 = cell_coord(lox - rr, m.ox, m.inv_cell, m.nx), bx1 = max(bx0, cell_coord(hix + rr, m.ox, m.inv_cell, m.nx));
    const int by0 = cell_coord(loy - rr, m.oy, m.inv_cell, m.ny), by1 = max(by0, cell_coord(hiy + rr, m.oy, m.inv_cell, m.ny));
    const int bz0 = cell_coord(loz - rr, m.oz, m.inv_cell, m.nz), bz1 = max(bz0, cell_coord(hiz + rr, m.oz, m.inv_cell, m.nz));
    const int ny_b = by1 - by0 + 1;
    const int nrows = (bz1 - bz0 + 1) * ny_b;
    // [begin, end) of the first four rows
    auto row_range = [&](int row, int& beg, int& end) {
      beg = 0; end = 0;
      if (row < nrows) {
        const int cz = bz0 + row / ny_b, cy = by0 + row % ny_b;
        const int rowbase = (cz * m.ny + cy) * m.nx;
        beg = cell_start[rowbase + bx0];
        end = cell_start[rowbase + bx1 + 1];
      }
    };
    int beg, end, nbeg, nend;
    row_range(grp, beg, end);
    for (int rb = 0; rb < nrows; rb += 4) {
      row_range(rb + 4 + grp, nbeg, nend);      // next step's ranges are in flight while this step's rows are scanned
      int j = beg + l16;
      float4 c = (j < end) ? spos[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      while (__ballot(j < end)) {
        const bool valid = j < end;
        const int jn = j + 16;
        const float4 cn = (jn < end) ? spos[jn] : make_float4(0.f, 0.f, 0.f, 0.f);     // next chunk of this row
        n_cand += (unsigned long long)__popcll(__ballot(valid));
        const unsigned idx = __float_as_uint(c.w);
        u64 key[S], pend[S];
        u64 any = 0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
          key[s] = ((u64)__float_as_uint(dist2(c.x, c.y, c.z, qx[s], qy[s], qz[s])) << 32) | idx;
          pend[s] = __ballot(valid && key[s] < thr[s]);
          any |= pend[s];
        }
        while (any) {
          // one candidate per list: the lowest pending lane of each list broadcasts its key to that list's lanes
          u64 kk = ~0ull;
#pragma unroll
          for (int s = 0; s < S; ++s) {
            if (pend[s]) {
              const int l = __builtin_ctzll(pend[s]);
              const unsigned khi = (unsigned)__builtin_amdgcn_readlane((int)(key[s] >> 32), l);
              const unsigned klo = (unsigned)__builtin_amdgcn_readlane((int)(key[s] & 0xFFFFFFFFull), l);
              if (my_list == s) kk = ((u64)khi << 32) | klo;
              pend[s] &= pend[s] - 1;
            }
          }
          // sorted insertion across the 8 lanes of a list: b[j] <- kk < b[j-1] ? b[j-1] : (kk < b[j] ? kk : b[j])
          const unsigned plo = dpp_shr1((unsigned)(mine & 0xFFFFFFFFull)), phi = dpp_shr1((unsigned)(mine >> 32));
          const u64 prev = ((lane & 7) == 0) ? 0ull : (((u64)phi << 32) | plo);
          mine = (kk < prev) ? prev : ((kk < mine) ? kk : mine);
          // new thresholds = last entry of every list; candidates that no longer qualify are dropped
          any = 0;
#pragma unroll
          for (int s = 0; s < S; ++s) {
            if (pend[s]) {
              const unsigned thi = (unsigned)__builtin_amdgcn_readlane((int)(mine >> 32), 8 * s + 7);
              const unsigned tlo = (unsigned)__builtin_amdgcn_readlane((int)(mine & 0xFFFFFFFFull), 8 * s + 7);
              thr[s] = ((u64)thi << 32) | tlo;
              pend[s] &= __ballot(valid && key[s] < thr[s]);
              any |= pend[s];
            }
          }
        }
        // thresholds for the next chunk (lists that received entries above but drained their queue)
#pragma unroll
        for (int s = 0; s < S; ++s) {
          if (!((done_mask >> s) & 1u)) {
            const unsigned thi = (unsigned)__builtin_amdgcn_readlane((int)(mine >> 32), 8 * s + 7);
            const unsigned tlo = (unsigned)__builtin_amdgcn_readlane((int)(mine & 0xFFFFFFFFull), 8 * s + 7);
            thr[s] = ((u64)thi << 32) | tlo;
          }
        }
        j = jn; c = cn;
      }
      beg = nbeg; end = nend;
    }
    // a list is final when 8 entries were found inside rho (every unscanned point is farther), or after the r pass
    unsigned all_done = 1;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      if (!((done_mask >> s) & 1u)) {
        const unsigned thi = (unsigned)__builtin_amdgcn_readlane((int)(mine >> 32), 8 * s + 7);
        const unsigned tlo = (unsigned)__builtin_amdgcn_readlane((int)(mine & 0xFFFFFFFFull), 8 * s + 7);
        if (last || (((u64)thi << 32) | tlo) != sentinel) done_mask |= 1u << s;
        else all_done = 0;
      }
    }
    if (all_done) break;
    rho *= 2.0f;
  }
  // emit: lane 8 s + j holds entry j of sample s; count = #(index valid && d2 < r2)  (neural_point.py:207-213)
  if (lane < S * K) {
    const unsigned ib = (unsigned)(mine & 0xFFFFFFFFull);
    I_out[(size_t)ray * S * K + lane] = (ib == 0xFFFFFFFFu) ? -1 : (int)ib;
  }
  {
    const unsigned ib = (unsigned)(mine & 0xFFFFFFFFull), db = (unsigned)(mine >> 32);
    const u64 inr = __ballot(lane < S * K && ib != 0xFFFFFFFFu && db < __float_as_uint(r2));
    if (lane < S) cnt_out[ray * S + lane] = __popcll((inr >> (8 * lane)) & 0xFFull);
  }
  cand_total += n_cand;
}

// One wavefront per ray; with fewer wavefronts than rays (the mapper's block prefetch on the side stream is launched
// THROTTLED: two workgroups per CU, so that the decode kernels of the main stream keep finding free slots) every
// wavefront walks rays wave, wave + #waves, ...
__global__ __launch_bounds__(256) void k_knn_rays2(const GridMeta* __restrict__ meta, const float4* __restrict__ spos,
                                                   const int* __restrict__ cell_start,
                                                   const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                   const float* __restrict__ depth, const float* __restrict__ z_vals,
                                                   const float* __restrict__ r_query,
                                                   float r_fixed, float r2_fixed, float near_s, float far_s, int n_rays,
                                                   int* __restrict__ I_out, int* __restrict__ cnt_out,
                                                   unsigned long long* __restrict__ cand_counter, const int* __restrict__ coarse) {
  const int wave0 = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  const int n_waves = (int)(gridDim.x * (blockDim.x >> 6));
  const GridMeta m = *meta;
  unsigned long long cand = 0;
  for (int ray = wave0; ray < n_rays; ray += n_waves)
    knn_ray2_one(ray, m, spos, cell_start, rays_o, rays_d, depth, z_vals, r_query, r_fixed, r2_fixed, near_s, far_s, I_out,
                 cnt_out, cand, coarse);
  if (cand_counter && (threadIdx.x & 63) == 0 && cand) atomicAdd(cand_counter + 8 * (blockIdx.x & (kKnnCandSlots - 1)), cand);
}

// sample_near_pcl marching test (src/neural_point.py:232-249): one wave per (ray, step); a step "hits" when at least
// one neural point lies strictly inside the query radius (nearest of the 8-NN has D < r^2).  First hit ends the scan.
__global__ __launch_bounds__(256) void k_near_pcl_hits(const GridMeta* __restrict__ meta, const float4* __restrict__ spos,
                                                       const int* __restrict__ cell_start,
                                                       const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                       const float* __restrict__ z_steps, const int* __restrict__ step_row,
                                                       int n_rays, int n_steps, float r, float r2,
                                                       unsigned char* __restrict__ hits) {
  const int q = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  if (q >= n_rays * n_steps) return;
  const int ray = q / n_steps, st = q - ray * n_steps;
  const int lane = threadIdx.x & 63;
  const GridMeta m = *meta;
  float qx, qy, qz;
  sample_point(rays_o[ray * 3], rays_o[ray * 3 + 1], rays_o[ray * 3 + 2], rays_d[ray * 3], rays_d[ray * 3 + 1],
               rays_d[ray * 3 + 2], z_steps[(step_row ? step_row[ray] : 0) * n_steps + st], qx, qy, qz);
  CellBox bx;
  box_of(m, qx, qy, qz, r, bx);
  const int ny_b = bx.hi[1] - bx.lo[1] + 1;
  const int nrows = (bx.hi[2] - bx.lo[2] + 1) * ny_b;
  bool hit = false;
  for (int rb = 0; rb < nrows && !hit; rb += 64) {
    int beg = 0, end = 0;
    const int row = rb + lane;
    if (row < nrows) {
      const int cz = bx.lo[2] + row / ny_b, cy = bx.lo[1] + row % ny_b;
      const int rowbase = (cz * m.ny + cy) * m.nx;
      beg = cell_start[rowbase + bx.lo[0]];
      end = cell_start[rowbase + bx.hi[0] + 1];
    }
    const int nr = min(64, nrows - rb);
    for (int ri = 0; ri < nr && !hit; ++ri) {
      const int b0 = __builtin_amdgcn_readlane(beg, ri), e0 = __builtin_amdgcn_readlane(end, ri);
      for (int j0 = b0; j0 < e0 && !hit; j0 += 64) {
        const int j = j0 + lane;
        bool in = false;
        if (j < e0) { float4 c = spos[j]; in = dist2(c.x, c.y, c.z, qx, qy, qz) < r2; }
        hit = __ballot(in) != 0ull;
      }
    }
  }
  if (lane == 0) hits[q] = hit ? 1 : 0;
}

// free-query mode: wave per query; outputs follow find_neighbors_faiss (D f32, I int64, cnt int32)
__global__ __launch_bounds__(256) void k_knn_queries(const GridMeta* __restrict__ meta,
                                                     const float4* __restrict__ spos,
                                                     const int* __restrict__ cell_start, const float* __restrict__ q,
                                                     const float* __restrict__ r_per_query, float r_fixed, float r2_fixed,
                                                     int nq, float* __restrict__ D_out, long long* __restrict__ I_out,
                                                     int* __restrict__ cnt_out, const int* __restrict__ coarse) {
  const int qi = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  if (qi >= nq) return;
  const int lane = threadIdx.x & 63;
  const GridMeta m = *meta;
  float r, r2;
  if (r_per_query) { r = r_per_query[qi]; r2 = __fmul_rn(r, r); } else { r = r_fixed; r2 = r2_fixed; }
  u64 mine;
  unsigned long long n_cand = 0;
  int n_pass = 0;
  wave_knn_lane(m, spos, cell_start, q[qi * 3 + 0], q[qi * 3 + 1], q[qi * 3 + 2], r, r2, mine, n_cand, coarse, n_pass);
  // after an early exit the sentinel threshold was rho^2 <= r2: every kept entry has d2 <= rho^2 <= r2
  const unsigned ib = (unsigned)(mine & 0xFFFFFFFFull), db = (unsigned)(mine >> 32);
  const int cnt = __popcll(__ballot(lane < K && ib != 0xFFFFFFFFu && db < __float_as_uint(r2)));
  if (lane < K) {
    bool empty = ib == 0xFFFFFFFFu;
    if (I_out) I_out[(long long)qi * K + lane] = empty ? -1ll : (long long)ib;
    if (D_out) D_out[(long long)qi * K + lane] = empty ? __int_as_float(0x7F800000) : __uint_as_float(db);
  }
  if (lane == 0 && cnt_out) cnt_out[qi] = cnt;
}

// psl_dedupe_count: number of points with ORIGINAL index < idx_limit strictly inside the radius of each query -- the
// admission test of add_neural_points (neural_point.py:116-121: "no existing point within the radius"), restricted to a
// prefix of the cloud: the multi-GPU merge tests the other ranks' new locations against the BASE map while the index still
// covers this rank's own tail, so that no rebuild is needed before the test.  One wavefront per query, whole r-cube
// (an exact count, not a top-8), 64 rows of the cube per step.
__global__ __launch_bounds__(256) void k_dedupe_count(const GridMeta* __restrict__ meta, const float4* __restrict__ spos,
                                                      const int* __restrict__ cell_start, const int* __restrict__ coarse,
                                                      const float* __restrict__ q, const float* __restrict__ r_per_query,
                                                      float r_fixed, float r2_fixed, int nq, unsigned idx_limit,
                                                      int* __restrict__ cnt_out) {
  const int qi = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  if (qi >= nq) return;
  const int lane = threadIdx.x & 63;
  const GridMeta m = *meta;
  float r, r2;
  if (r_per_query) { r = r_per_query[qi]; r2 = __fmul_rn(r, r); } else { r = r_fixed; r2 = r2_fixed; }
  const float qx = q[qi * 3], qy = q[qi * 3 + 1], qz = q[qi * 3 + 2];
  int cnt = 0;
  if (!wave_box_empty(m, coarse, qx, qy, qz, qx, qy, qz, r)) {
    CellBox bx;
    box_of(m, qx, qy, qz, r, bx);
    const int ny_b = bx.hi[1] - bx.lo[1] + 1;
    const int nrows = (bx.hi[2] - bx.lo[2] + 1) * ny_b;
    for (int rb = 0; rb < nrows; rb += 64) {
      int beg = 0, end = 0;
      const int row = rb + lane;
      if (row < nrows) {
        const int cz = bx.lo[2] + row / ny_b, cy = bx.lo[1] + row % ny_b;
        const int rowbase = (cz * m.ny + cy) * m.nx;
        beg = cell_start[rowbase + bx.lo[0]];
        end = cell_start[rowbase + bx.hi[0] + 1];
      }
      const int nr = min(64, nrows - rb);
      for (int ri = 0; ri < nr; ++ri) {
        const int b0 = __builtin_amdgcn_readlane(beg, ri), e0 = __builtin_amdgcn_readlane(end, ri);
        for (int j = b0 + lane; j < e0; j += 64) {
          const float4 c = spos[j];
          cnt += (dist2(c.x, c.y, c.z, qx, qy, qz) < r2 && __float_as_uint(c.w) < idx_limit) ? 1 : 0;
        }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if (lane == 0) cnt_out[qi] = cnt;
}

static inline float r2_of(float r) { return (float)((double)r * (double)r); }   // python: radius**2 in double, then f32

int g_knn_version = -1;      // PSL_KNN / psl_debug_option("knn", v): 0 = by launch (default), 4 = per sample (flat), 2 = per ray

int knn_rays(psl_ctx* ctx, const float* rays_o, const float* rays_d, const float* depth, const float* z_vals,
             const float* r_query, int n_rays, int* I_out, int* cnt_out, hipStream_t s, int max_blocks) {
  if (n_rays <= 0) return PSL_OK;
  if (g_knn_version < 0) { const char* e = getenv("PSL_KNN"); g_knn_version = (e && (e[0] == '2' || e[0] == '4')) ? e[0] - '0' : 0; }
  // Two kernels, the same keys and therefore bit-identical answers:
  //  * k_knn_rays_flat (4): one wavefront per SAMPLE, flat candidate enumeration (two dependent memory trips per pass) --
  //    every unthrottled launch: tracker launch (200 rays) 25 us, 1 500 rays 39 us, 5 000 rays 64-80 us;
  //  * k_knn_rays2 (2): one wavefront per RAY, the five samples share one scan of the union box (173 candidates per query
  //    against 522) -- the THROTTLED side-stream prefetch of the mapper (max_blocks > 0: a persistent grid whose wavefronts
  //    walk several rays each, next to the decode kernels of the previous block).
  // Rounds 2-4 carried two more per-sample kernels (row walk; four wavefronts per sample, "measured equal"): deleted in
  // round 5, the flat enumeration replaced both everywhere (DESIGN_HISTORY.md).
  // (a launch that carries the tracker's pose step or turns its directions -- ctx->track_pose -- exists in the per-sample kernel only)
  const int ver = ctx->track_pose ? 4 : ((g_knn_version == 2 || g_knn_version == 4) ? g_knn_version : (max_blocks > 0 ? 2 : 4));
  if (ver == 4) {
    static int trace4 = -1;
    if (trace4 < 0) { const char* e = getenv("PSL_KNN_TRACE"); trace4 = (e && e[0] == '1') ? 1 : 0;
                      const char* h = getenv("PSL_KNN_START_HINT"); if (h) g_knn_start_hint = atoi(h); }
    // queries from which the high-occupancy instantiation <4 records in flight, 8 wavefronts per SIMD> is used (it is held to
    // 64 VGPRs and spills two registers to scratch, profiles/r05_kernel_resources.json; PSL_KNN_FLAT_LARGE, < 0 = never).
    // Measured [MI355X, round 4]: 25 000 queries (TUM / ScanNet tracker) 74.6 -> 64.1 us, TUM yaml +2.7 %, ScanNet +1 %,
    // Replica (7 500 queries) +1.5 %; at the base mix's 1 000 queries (4 wavefronts per CU: no occupancy to gain) -0.5 %
    static int large_from = -2;
    if (large_from == -2) { const char* e = getenv("PSL_KNN_FLAT_LARGE"); large_from = e ? atoi(e) : 5000; }
    unsigned long long* cand = (ctx->prof_on || trace4) ? ctx->knn_cand : nullptr;   // one atomic per query: only while measured
    const TrackPose* tpp = static_cast<const TrackPose*>(ctx->track_pose);
    const bool large = large_from >= 0 && n_rays * S >= large_from;
    // g_knn_start_hint: bit 0 = start at the pass the row lengths suggest, bit 1 = carry the eighth-best bound into the next pass; each for the
    // launches below 5 000 queries, bits 2 / 3 the same for the larger ones.  Default 3: measured [MI355X] 26.5 -> 22.2 us on the 1 000-query
    // tracker launch (the bound does it; the start pass alone 26.0); at 7 500 queries 35.5 -> 34.2 us on the 1 M-point cloud but 28.1 -> 28.4 on
    // the 50 k-point cloud of config 1, and 69.6 -> 73.7 us at 25 000 queries (more admitted candidates and insertions per query: the wrong
    // trade where the launch is bound by its total work), so the larger launches keep the plain expanding search
    const int hb = (n_rays * S >= 5000) ? (g_knn_start_hint >> 2) : g_knn_start_hint;
    const int knn_mode_bits = ((hb & 1) ? 0 : 2) | ((hb & 2) ? 0 : 4);
#define PSL_KNN_FLAT(UU, WW, PP, TP)                                                                                               \
    PSL_KLAUNCH((k_knn_rays_flat<UU, WW, PP>), dim3((n_rays * S + 3) / 4), dim3(256), 0, s, ctx->meta, ctx->spos, ctx->cell_start, \
                rays_o, rays_d, depth, z_vals, r_query, ctx->cfg.radius_query, r2_of(ctx->cfg.radius_query),                        \
                ctx->cfg.near_end_surface, ctx->cfg.far_end_surface, n_rays, I_out, cnt_out, cand, ctx->coarse, trace4 | knn_mode_bits, TP)
    if (tpp && !tpp->rotate_only) PSL_KNN_FLAT(8, 5, 1, *tpp);       // psl_track_iters, batches <= 1 024 rays: pose step in the prologue
    else if (tpp && large) PSL_KNN_FLAT(4, 8, 2, *tpp);              // larger tracker batches: directions turned here, no ray set-up launch
    else if (tpp) PSL_KNN_FLAT(8, 5, 2, *tpp);
    else if (large) PSL_KNN_FLAT(4, 8, 0, TrackPose{});
    else PSL_KNN_FLAT(8, 5, 0, TrackPose{});
#undef PSL_KNN_FLAT
    PSL_LAUNCH_CHECK();
    return PSL_OK;
  }
  int blocks = (n_rays + 3) / 4;
  if (max_blocks > 0) blocks = std::min(blocks, max_blocks);      // throttled: every wavefront walks several rays
  PSL_KLAUNCH(k_knn_rays2, dim3(blocks), dim3(256), 0, s, ctx->meta, ctx->spos, ctx->cell_start, rays_o,
                     rays_d, depth, z_vals, r_query, ctx->cfg.radius_query, r2_of(ctx->cfg.radius_query),
                     ctx->cfg.near_end_surface, ctx->cfg.far_end_surface, n_rays, I_out, cnt_out, ctx->knn_cand, ctx->coarse);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}

int knn_trace_dump() {
  KnnTrace t;
  if (hipMemcpyFromSymbol(&t, HIP_SYMBOL(g_knn_trace), sizeof(t)) != hipSuccess) return PSL_ERR_HIP;
  const double n = (double)std::max<unsigned long long>(t.n, 1);
  fprintf(stderr, "[psl knn trace] queries %llu | cycles mean %.0f max %llu | candidates mean %.1f max %llu | passes mean %.2f (1: %llu, 2: %llu, 3+: %llu)\n",
          t.n, t.sum_cyc / n, t.max_cyc, t.sum_cand / n, t.max_cand, t.sum_pass / n, t.hist_pass[1], t.hist_pass[2], t.hist_pass[3]);
  fprintf(stderr, "[psl knn trace] cycles histogram (log2 buckets):");
  for (int b = 0; b < 24; ++b) if (t.hist_cyc[b]) fprintf(stderr, " 2^%d:%llu", b, t.hist_cyc[b]);
  fprintf(stderr, "\n[psl knn trace] candidates histogram (log2 buckets):");
  for (int b = 0; b < 24; ++b) if (t.hist_cand[b]) fprintf(stderr, " 2^%d:%llu", b, t.hist_cand[b]);
  fprintf(stderr, "\n");
  t = KnnTrace{};
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_knn_trace), &t, sizeof(t)) != hipSuccess) return PSL_ERR_HIP;
  return PSL_OK;
}

int knn_queries(psl_ctx* ctx, const float* q, const float* r_per_query, float r_scalar, int nq, float* D_out,
                int64_t* I_out, int* cnt_out, hipStream_t s) {
  if (nq <= 0) return PSL_OK;
  int blocks = (nq + 3) / 4;
  hipLaunchKernelGGL(k_knn_queries, dim3(blocks), dim3(256), 0, s, ctx->meta, ctx->spos, ctx->cell_start, q,
                     r_per_query, r_scalar, r2_of(r_scalar), nq, D_out, (long long*)I_out, cnt_out, ctx->coarse);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}

}  // namespace psl

using namespace psl;

extern "C" int psl_index_build(psl_ctx* ctx, void* stream) {
  if (!ctx) return PSL_ERR_ARG;
  return grid_build(ctx, (hipStream_t)stream);
}

extern "C" int psl_knn(psl_ctx* ctx, const float* q, const float* r_per_query, float r_scalar, int nq,
                       float* D_out, int64_t* I_out, int32_t* cnt_out, void* stream) {
  if (!ctx || !q || nq < 0) { set_error("psl_knn: bad argument"); return PSL_ERR_ARG; }
  if (ctx->index_points != ctx->n_points) { set_error("psl_knn: index is stale, call psl_index_build"); return PSL_ERR_STATE; }
  return knn_queries(ctx, q, r_per_query, r_scalar, nq, D_out, I_out, cnt_out, (hipStream_t)stream);
}

extern "C" int psl_dedupe_count(psl_ctx* ctx, const float* q, const float* r_per_query, float r_scalar, int nq, int idx_limit,
                                int32_t* cnt_out, void* stream) {
  if (!ctx || !q || !cnt_out || nq < 0 || idx_limit < 0) { set_error("psl_dedupe_count: bad argument"); return PSL_ERR_ARG; }
  if (ctx->index_points != ctx->n_points) { set_error("psl_dedupe_count: index is stale, call psl_index_build"); return PSL_ERR_STATE; }
  if (nq == 0) return PSL_OK;
  if (ctx->n_points == 0 || idx_limit == 0) { PSL_HIP(hipMemsetAsync(cnt_out, 0, sizeof(int) * (size_t)nq, (hipStream_t)stream)); return PSL_OK; }
  hipLaunchKernelGGL(k_dedupe_count, dim3((nq + 3) / 4), dim3(256), 0, (hipStream_t)stream, ctx->meta, ctx->spos, ctx->cell_start,
                     ctx->coarse, q, r_per_query, r_scalar, r2_of(r_scalar), nq, (unsigned)idx_limit, cnt_out);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}

extern "C" int psl_near_pcl_hits(psl_ctx* ctx, const float* rays_o, const float* rays_d, int n_rays,
                                 const float* z_steps, const int32_t* step_row, int n_steps, float radius,
                                 uint8_t* hits, void* stream) {
  if (!ctx || !rays_o || !rays_d || !z_steps || !hits || n_rays < 0 || n_steps <= 0) {
    set_error("psl_near_pcl_hits: bad argument"); return PSL_ERR_ARG;
  }
  if (ctx->index_points != ctx->n_points) { set_error("psl_near_pcl_hits: index is stale, call psl_index_build"); return PSL_ERR_STATE; }
  if (n_rays == 0) return PSL_OK;
  if (ctx->n_points == 0) { PSL_HIP(hipMemsetAsync(hits, 0, (size_t)n_rays * n_steps, (hipStream_t)stream)); return PSL_OK; }
  long long nq = (long long)n_rays * n_steps;
  hipLaunchKernelGGL(k_near_pcl_hits, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, (hipStream_t)stream, ctx->meta,
                     ctx->spos, ctx->cell_start, rays_o, rays_d, z_steps, step_row, n_rays, n_steps, radius, r2_of(radius), hits);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}
