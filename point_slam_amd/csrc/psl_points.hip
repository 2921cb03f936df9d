// The point store: appending / truncating / downloading neural point positions, the admission test of
// NeuralPointCloud.add_neural_points (src/neural_point.py:91-167) and the cross-rank block dedupe of the multi-GPU merge.
// (Split from psl_grid.hip in round 5: that file keeps the grid index and the k-NN kernels.)
#include "psl_common.h"
#include "psl_device.h"

namespace psl {

typedef unsigned long long u64;

// ----------------------------------------------------------------- point growth
__global__ __launch_bounds__(256) void k_append_raw(const float* __restrict__ src, int n, float4* pos, int base) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pos[base + i] = make_float4(src[i * 3 + 0], src[i * 3 + 1], src[i * 3 + 2], 0.f);
}

__global__ __launch_bounds__(256) void k_download(const float4* __restrict__ pos, int n, float* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { float4 p = pos[i]; out[i * 3 + 0] = p.x; out[i * 3 + 1] = p.y; out[i * 3 + 2] = p.z; }
}

// surface points o + d*depth for rays with depth > 0 (neural_point.py:108-113); others get a far-away sentinel
__global__ __launch_bounds__(256) void k_surface_pts(const float* __restrict__ ro, const float* __restrict__ rd,
                                                     const float* __restrict__ dep, int n, float* q) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float d = dep[i];
  float x, y, z;
  sample_point(ro[i * 3], ro[i * 3 + 1], ro[i * 3 + 2], rd[i * 3], rd[i * 3 + 1], rd[i * 3 + 2], d, x, y, z);
  q[i * 3] = x; q[i * 3 + 1] = y; q[i * 3 + 2] = z;
}

// single-block ordered compaction: keep[i] = depth>0 && cnt==0 ; appends 3 points per kept location in
// ray order (neural_point.py:141-147: pts[mask].reshape(-1,3)).
__global__ __launch_bounds__(1024) void k_append_kept(const float* __restrict__ ro, const float* __restrict__ rd,
                                                      const float* __restrict__ dep, const int* __restrict__ cnt,
                                                      int has_index, int n, float near_e, float far_e, float4* pos,
                                                      int base, int capacity, unsigned char* keep_out,
                                                      int* n_kept_out) {
  __shared__ int wsum[16];
  __shared__ int running;
  if (threadIdx.x == 0) running = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i0 = 0; i0 < n; i0 += 1024) {
    int i = i0 + threadIdx.x;
    bool keep = false;
    float d = 0.f;
    if (i < n) { d = dep[i]; keep = d > 0.f && (!has_index || cnt[i] == 0); }
    u64 bal = __ballot(keep);
    int pre = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[w] = __popcll(bal);
    __syncthreads();
    int off = running;
    for (int j = 0; j < w; ++j) off += wsum[j];
    int rank = off + pre;
    if (i < n) keep_out[i] = keep ? 1 : 0;
    if (keep && base + 3 * rank + 2 < capacity) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        // z = near*d*(1-t) + far*d*t, t = linspace(0,1,3)  (neural_point.py:126-139)
        float t = 0.5f * (float)a;
        float z = __fadd_rn(__fmul_rn(__fmul_rn(near_e, d), 1.0f - t), __fmul_rn(__fmul_rn(far_e, d), t));
        float x, y, zz;
        sample_point(ro[i * 3], ro[i * 3 + 1], ro[i * 3 + 2], rd[i * 3], rd[i * 3 + 1], rd[i * 3 + 2], z, x, y, zz);
        pos[base + 3 * rank + a] = make_float4(x, y, zz, 0.f);
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int j = 0; j < 16; ++j) t += wsum[j]; running += t; }
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_kept_out = running;
}

}  // namespace psl

using namespace psl;

extern "C" int psl_points_reset(psl_ctx* ctx) {
  if (!ctx) return PSL_ERR_ARG;
  ctx->n_points = 0; ctx->index_points = -1;
  return PSL_OK;
}

extern "C" int psl_points_append(psl_ctx* ctx, const float* pos, int n, void* stream) {
  if (!ctx || n < 0) { set_error("psl_points_append: bad argument"); return PSL_ERR_ARG; }
  if (n == 0) return PSL_OK;
  if (ctx->n_points + n > ctx->cfg.max_points) {
    set_error("psl_points_append: capacity %d exceeded (%d + %d)", ctx->cfg.max_points, ctx->n_points, n);
    return PSL_ERR_CAPACITY;
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_append_raw, dim3((n + 255) / 256), dim3(256), 0, s, pos, n, ctx->pos, ctx->n_points);
  PSL_LAUNCH_CHECK();
  ctx->n_points += n;
  ctx->index_points = -1;
  return PSL_OK;
}

extern "C" int psl_points_truncate(psl_ctx* ctx, int n) {
  if (!ctx || n < 0 || n > ctx->n_points) { set_error("psl_points_truncate: bad count"); return PSL_ERR_ARG; }
  if (n != ctx->n_points) { ctx->n_points = n; ctx->index_points = -1; }
  return PSL_OK;
}

extern "C" int psl_points_count(psl_ctx* ctx) { return ctx ? ctx->n_points : PSL_ERR_ARG; }

extern "C" int psl_points_download(psl_ctx* ctx, float* pos_out, int capacity_points, void* stream) {
  if (!ctx || !pos_out) return PSL_ERR_ARG;
  int n = min(ctx->n_points, capacity_points);
  if (n > 0) hipLaunchKernelGGL(k_download, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, ctx->pos, n, pos_out);
  PSL_LAUNCH_CHECK();
  return n;
}

extern "C" int psl_points_download_range(psl_ctx* ctx, int first, int count, float* pos_out, void* stream) {
  if (!ctx || first < 0 || count < 0 || first + count > ctx->n_points || (count > 0 && !pos_out)) {
    set_error("psl_points_download_range: bad range [%d, %d) of %d", first, first + count, ctx ? ctx->n_points : -1);
    return PSL_ERR_ARG;
  }
  if (count > 0)
    hipLaunchKernelGGL(k_download, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, ctx->pos + first, count, pos_out);
  PSL_LAUNCH_CHECK();
  return count;
}




// psl_dedupe_blocks: the cross-rank half of the merge's admission test.  Locations arrive in rank blocks; block b's
// locations are tested against the points (three per location) of the locations of the blocks before it that are still
// kept.  One launch per block, in block order, so the keep flags a block reads are final; one wavefront per location,
// lanes stride over the earlier locations.  (dx*dx + dy*dy) + dz*dz unfused, as the grid search evaluates it.
__global__ __launch_bounds__(256) void k_dedupe_block(const float* __restrict__ pts, int pts_stride, const float* __restrict__ rad,
                                                      int rad_stride, int first, int last, unsigned char* __restrict__ keep) {
  const int l = first + __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  if (l >= last || !keep[l]) return;
  const int lane = threadIdx.x & 63;
  const float* qp = pts + (size_t)(3 * l + 1) * pts_stride;       // the surface point is the middle one of the triplet
  const float qx = qp[0], qy = qp[1], qz = qp[2];
  const float r = rad[(size_t)(3 * l + 1) * rad_stride], r2 = __fmul_rn(r, r);
  bool hit = false;
  for (int j0 = 0; j0 < first && !hit; j0 += 64) {
    const int j = j0 + lane;
    bool h = false;
    if (j < first && keep[j]) {
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const float* p = pts + (size_t)(3 * j + t) * pts_stride;
        const float dx = __fsub_rn(qx, p[0]), dy = __fsub_rn(qy, p[1]), dz = __fsub_rn(qz, p[2]);
        h |= __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)) < r2;
      }
    }
    hit = __any(h);
  }
  if (hit && lane == 0) keep[l] = 0;
}

extern "C" int psl_dedupe_blocks(psl_ctx* ctx, const float* pts, int pts_stride, const float* radius, int radius_stride,
                                 const int32_t* block_first, int n_blocks, uint8_t* keep, void* stream) {
  if (!ctx || !pts || !radius || !block_first || !keep || n_blocks < 0 || pts_stride < 3 || radius_stride < 1) {
    set_error("psl_dedupe_blocks: bad argument"); return PSL_ERR_ARG;
  }
  for (int b = 0; b < n_blocks; ++b)
    if (block_first[b] < 0 || block_first[b + 1] < block_first[b]) { set_error("psl_dedupe_blocks: block offsets must ascend"); return PSL_ERR_ARG; }
  for (int b = 1; b < n_blocks; ++b) {
    const int first = block_first[b], last = block_first[b + 1];
    if (last == first || first == 0) continue;
    hipLaunchKernelGGL(k_dedupe_block, dim3((last - first + 3) / 4), dim3(256), 0, (hipStream_t)stream, pts, pts_stride, radius,
                       radius_stride, first, last, keep);
  }
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}


extern "C" int psl_add_points_sync(psl_ctx* ctx, const float* rays_o, const float* rays_d, const float* depth,
                                   const float* radius_per_ray, float r_scalar, int n, float near_end, float far_end,
                                   uint8_t* keep_out, int* n_kept_host, void* stream) {
  if (!ctx || n < 0 || !keep_out || !n_kept_host) { set_error("psl_add_points_sync: bad argument"); return PSL_ERR_ARG; }
  *n_kept_host = 0;
  if (n == 0) return PSL_OK;
  hipStream_t s = (hipStream_t)stream;
  int has_index = ctx->n_points > 0;
  if (has_index && ctx->index_points != ctx->n_points) {
    set_error("psl_add_points_sync: index is stale, call psl_index_build"); return PSL_ERR_STATE;
  }
  if (ctx->scan_flags_cap < 4 * n) {
    if (ctx->scan_flags) (void)hipFree(ctx->scan_flags);
    PSL_HIP(hipMalloc(&ctx->scan_flags, sizeof(int) * 4 * (size_t)n)); psl::poison(ctx->scan_flags, sizeof(int) * 4 * (size_t)n);
    ctx->scan_flags_cap = 4 * n;
  }
  float* qsurf = (float*)ctx->scan_flags;            // [n][3]
  int* cnt = ctx->scan_flags + 3 * n;                // [n]
  if (has_index) {
    hipLaunchKernelGGL(k_surface_pts, dim3((n + 255) / 256), dim3(256), 0, s, rays_o, rays_d, depth, n, qsurf);
    int rc = knn_queries(ctx, qsurf, radius_per_ray, r_scalar, n, nullptr, nullptr, cnt, s);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_append_kept, dim3(1), dim3(1024), 0, s, rays_o, rays_d, depth, cnt, has_index, n, near_end,
                     far_end, ctx->pos, ctx->n_points, ctx->cfg.max_points, keep_out, ctx->d_counter);
  PSL_LAUNCH_CHECK();
  int kept = 0;
  PSL_HIP(hipMemcpyAsync(&kept, ctx->d_counter, sizeof(int), hipMemcpyDeviceToHost, s));
  PSL_HIP(hipStreamSynchronize(s));
  if (ctx->n_points + 3 * kept > ctx->cfg.max_points) {
    set_error("psl_add_points_sync: capacity %d exceeded", ctx->cfg.max_points);
    return PSL_ERR_CAPACITY;
  }
  ctx->n_points += 3 * kept;
  if (kept > 0) ctx->index_points = -1;
  *n_kept_host = kept;
  return PSL_OK;
}
