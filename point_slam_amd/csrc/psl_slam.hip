// Fused optimisation loops: the bodies of Tracker.optimize_cam_in_batch (src/Tracker.py:89-186, loop :332-371)
// and of the Mapper.optimize_map inner loop (src/Mapper.py:408-568) as back-to-back kernel launches with NO
// host synchronisation: pixel gather + ray set-up, depth-outlier mask (lower median), render forward, loss +
// cotangents, render backward, pose / feature / decoder Adam.  The reference pays ~2000 ATen launches, three
// .item() syncs and an RPC-ed FAISS search per iteration for the same work.
#include <cstring>
#include <algorithm>
#include <vector>
#include "psl_decode.h"
#include "psl_pose.h"

namespace psl {

int render_fwd_impl(psl_ctx* ctx, const psl_render_args* a, hipStream_t s, bool repack);
int render_bwd_impl(psl_ctx* ctx, const psl_render_args* a, const psl_render_grads* g, hipStream_t s);
int geo_iter_impl(psl_ctx* ctx, const psl_render_args* a, const psl_render_grads* g, const int* active, double* loss_acc,
                  const AdamWorklist* wl, hipStream_t s, bool repack);

struct FrameDev {            // one RGB-D frame resident in HBM
  const float* depth;        // [H][W]
  const float* color;        // [H][W][3]
  const float* r_query;      // [H][W] or null
  float c2w[12];             // row-major 3x4
};

struct RayBufs {             // per-iteration ray batch (n slots, inactive slots masked)
  float *rays_o, *rays_d, *dirs, *gd, *gc, *rq;
  int* active;
  float *depth, *var, *rgb;  // render outputs
  unsigned char* valid;
  float *g_depth, *g_rgb, *g_o, *g_d;
};

// get_samples + get_rays_from_uv (src/common.py:40-89,162-183) for pre-drawn flat pixel indices.
// cam_tensor != null: pose from (quat, T) (tracker); else from frames[f].c2w (mapper).
__device__ __forceinline__ void ray_setup_one(int i, const psl_cam_intr& cam, int H0, int H1, int W0, int W1,
                                              const FrameDev* __restrict__ frames, int n_frames, int pix_per_frame,
                                              const int* __restrict__ pix_idx, const float* __restrict__ cam_tensor,
                                              const RayBufs& b) {
  int n = n_frames * pix_per_frame;
  int f = (i % n) / pix_per_frame;
  const FrameDev& fr = frames[f];
  float R[3][3], T[3];
  if (cam_tensor) {
    quat_to_rot(cam_tensor, R);
    T[0] = cam_tensor[4]; T[1] = cam_tensor[5]; T[2] = cam_tensor[6];
  } else {
#pragma unroll
    for (int a = 0; a < 3; ++a) { R[a][0] = fr.c2w[a * 4]; R[a][1] = fr.c2w[a * 4 + 1]; R[a][2] = fr.c2w[a * 4 + 2]; T[a] = fr.c2w[a * 4 + 3]; }
  }
  int idx = pix_idx[i];
  int w = W1 - W0;
  int u = W0 + idx % w, v = H0 + idx / w;
  float d0 = ((float)u - cam.cx) / cam.fx, d1 = -((float)v - cam.cy) / cam.fy, d2 = -1.0f;
  if (b.rays_d) {      // (null: camera-frame part only -- k_track_rays_all; the tracker's k-NN launch turns the directions)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      b.rays_d[i * 3 + a] = __fadd_rn(__fadd_rn(__fmul_rn(d0, R[a][0]), __fmul_rn(d1, R[a][1])), __fmul_rn(d2, R[a][2]));
      b.rays_o[i * 3 + a] = T[a];
    }
  }
  size_t px = (size_t)v * cam.W + u;
  float dep = fr.depth[px];
  if (b.dirs) {
    b.dirs[i * 3] = d0; b.dirs[i * 3 + 1] = d1; b.dirs[i * 3 + 2] = d2;
    b.gc[i * 3] = fr.color[px * 3]; b.gc[i * 3 + 1] = fr.color[px * 3 + 1]; b.gc[i * 3 + 2] = fr.color[px * 3 + 2];
  }
  if (b.rq) b.rq[i] = fr.r_query ? fr.r_query[px] : 0.f;
  bool act = dep > 0.f;                      // depth_filter (common.py:173-179)
  if (b.active) b.active[i] = act ? 1 : 0;
  b.gd[i] = act ? dep : 1.0f;                // inactive slots get a harmless finite depth
}

__global__ __launch_bounds__(256) void k_ray_setup(psl_cam_intr cam, int H0, int H1, int W0, int W1,
                                                   const FrameDev* __restrict__ frames, int n_frames, int pix_per_frame,
                                                   const int* __restrict__ pix_idx, const float* __restrict__ cam_tensor,
                                                   RayBufs b, int n_batches = 1) {
  // n_batches > 1: the rays of several mapper iterations at once (block prefetch)
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_frames * pix_per_frame * n_batches) return;
  ray_setup_one(i, cam, H0, H1, W0, W1, frames, n_frames, pix_per_frame, pix_idx, cam_tensor, b);
}

__global__ void k_track_init(FrameDev* fdev, FrameDev fh, float* best) {
  if (threadIdx.x == 0) *fdev = fh;
  if (threadIdx.x < 8) best[threadIdx.x] = (threadIdx.x == 7) ? 1e20f : 0.f;
}

// inside_mask = d <= min(10*median(d), 1.2*max(d)) over the ACTIVE rays; torch.median = lower median
// (Tracker.py:142-144, Mapper.py:507-509).  One workgroup.  n <= 4096: every thread ranks its own element
// against all others held in LDS (n^2/1024 compares per thread, no sort, two barriers); larger batches use a
// 4-pass byte-wise radix select over the float bit patterns (positive floats order like their bits).
// Lower median (torch.median) of the non-negative floats v[i] over the elements with active[i] != 0, by one workgroup; also
// their count, minimum and maximum.  Every thread returns the same values.  (positive floats order like their bit patterns)
struct BlockStats { unsigned cnt; float mn, mx, med; };
__device__ __forceinline__ BlockStats block_lower_median(const float* __restrict__ v, const int* __restrict__ active, int n,
                                                         bool skip_median_if_inlier_rule_is_void) {
  __shared__ unsigned keys[4096];
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_rank, s_cnt, s_max, s_min, s_med;
  __syncthreads();           // a previous call's readers are done with the shared state
  if (threadIdx.x == 0) { s_cnt = 0; s_max = 0; s_min = 0xFFFFFFFFu; s_prefix = 0; s_med = 0; }
  __syncthreads();
  unsigned lc = 0, lm = 0, ln = 0xFFFFFFFFu;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    bool a = active[i] != 0;
    unsigned b = a ? __float_as_uint(v[i]) : 0xFFFFFFFFu;
    if (n <= 4096) keys[i] = b;
    if (a) { lc++; lm = max(lm, b); ln = min(ln, b); }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lc += __shfl_xor(lc, o);
    lm = max(lm, (unsigned)__shfl_xor((int)lm, o));
    ln = min(ln, (unsigned)__shfl_xor((int)ln, o));
  }
  if ((threadIdx.x & 63) == 0) { atomicAdd(&s_cnt, lc); atomicMax(&s_max, lm); atomicMin(&s_min, ln); }
  __syncthreads();
  BlockStats st;
  st.cnt = s_cnt; st.mn = __uint_as_float(s_min); st.mx = __uint_as_float(s_max); st.med = 0.f;
  if (st.cnt == 0) return st;
  // depth-outlier rule only: median >= min, so 10*min >= 1.2*max implies thr = 1.2*max >= every depth: nothing to mask and
  // no median needed (the usual indoor case); the result is identical to evaluating the reference expression
  if (skip_median_if_inlier_rule_is_void && 10.0f * st.mn >= 1.2f * st.mx) { st.med = st.mn; return st; }
  const unsigned want = (st.cnt - 1) >> 1;
  if (n <= 4096) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const unsigned mine = keys[i];
      if (mine == 0xFFFFFFFFu) continue;
      unsigned rank = 0;
      for (int j = 0; j < n; ++j) {
        unsigned o = keys[j];
        rank += (o < mine || (o == mine && j < i)) ? 1u : 0u;
      }
      if (rank == want) s_med = mine;
    }
    __syncthreads();
  } else {
    if (threadIdx.x == 0) s_rank = want;
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      if (threadIdx.x < 256) hist[threadIdx.x] = 0;
      __syncthreads();
      const unsigned prefix = s_prefix;
      const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        if (active[i]) {
          unsigned b = __float_as_uint(v[i]);
          if ((b & himask) == prefix) atomicAdd(&hist[(b >> shift) & 255u], 1u);
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        unsigned r = s_rank, acc = 0; int bkt = 0;
        for (; bkt < 256; ++bkt) { if (acc + hist[bkt] > r) break; acc += hist[bkt]; }
        s_rank = r - acc;
        s_prefix = prefix | ((unsigned)bkt << shift);
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) s_med = s_prefix;
    __syncthreads();
  }
  st.med = __uint_as_float(s_med);
  return st;
}

__device__ __forceinline__ void depth_inlier_block(const float* __restrict__ gd, int* active, int n) {
  const BlockStats st = block_lower_median(gd, active, n, true);
  if (st.cnt == 0 || 10.0f * st.mn >= 1.2f * st.mx) return;
  const float thr = fminf(10.0f * st.med, 1.2f * st.mx);
  for (int i = threadIdx.x; i < n; i += blockDim.x)
    if (active[i] && !(gd[i] <= thr)) active[i] = 0;
}

__global__ __launch_bounds__(1024) void k_depth_inlier(const float* __restrict__ gd, int* active, int n) {
  // one workgroup per ray batch
  depth_inlier_block(gd + (size_t)blockIdx.x * n, active + (size_t)blockIdx.x * n, n);
}

__device__ __forceinline__ float signf0(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

// block-wide sum of doubles (1024 threads)
__device__ __forceinline__ double block_sum_d(double v, double* lds) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) lds[w] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += lds[i];
  return t;
}

// Tracker loss + cotangents (Tracker.py:159-180): uncertainty-weighted L1 depth + w*L1 colour,
// mask = (tmp < 10*mean(tmp)) & (gt>0) & ~nan.  Keeps the best (lowest-loss) pose (Tracker.py:347-350).
__global__ __launch_bounds__(1024) void k_tracker_loss(RayBufs b, int n, float w_color, int handle_dynamic, int use_color,
                                                       const float* __restrict__ cam_tensor, float* best /*[8]*/,
                                                       float* loss_out /*[4]*/) {
  __shared__ double lds[16];
  __shared__ float s_thr;
  double s = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    if (b.active[i]) {
      float e = fabsf(b.gd[i] - b.depth[i]);
      if (handle_dynamic) e = e / sqrtf(b.var[i] + 1e-10f);
      s += (double)e; c += 1.0;
    }
  }
  double tot = block_sum_d(s, lds);
  double cnt = block_sum_d(c, lds);
  if (threadIdx.x == 0) s_thr = (cnt > 0.0) ? 10.0f * (float)(tot / cnt) : 0.f;
  __syncthreads();
  float thr = s_thr;
  if (!handle_dynamic) {
    // tracking.handle_dynamic = False (Tracker.py:166-168; no shipped config): tmp = |gt - d|, mask = tmp < 10 * tmp.median()
    // (torch.median: the lower median) over the rays of the batch.  tmp goes through g_depth, which the loop below overwrites.
    // A NaN among the batch's tmp makes torch's median NaN and the mask empty (tmp < NaN is False everywhere); the bit-pattern
    // selection below would rank it as the largest key instead, so it is looked for first (advisor, round 5).
    int any_nan = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const float tmp = b.active[i] ? fabsf(b.gd[i] - b.depth[i]) : 0.f;
      any_nan |= (tmp != tmp) ? 1 : 0;
      b.g_depth[i] = tmp;
    }
    any_nan = __syncthreads_or(any_nan);
    const BlockStats st = block_lower_median(b.g_depth, b.active, n, false);
    thr = (st.cnt && !any_nan) ? 10.0f * st.med : 0.f;
    __syncthreads();
  }
  double lg = 0.0, lc = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float gdp = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (b.active[i]) {
      float d = b.depth[i], u = b.var[i], gt = b.gd[i];
      float inv = 1.0f / sqrtf(u + 1e-10f);
      float diff = fabsf(gt - d);
      float tmp = handle_dynamic ? diff / sqrtf(u + 1e-10f) : diff;
      bool m = (tmp < thr) && (gt > 0.f) && (d == d) && (u == u);
      if (m) {
        float e = diff / sqrtf(u + 1e-10f);
        float ec = fminf(fmaxf(e, 0.f), 1e3f);
        lg += (double)ec;
        if (e <= 1e3f) gdp = signf0(d - gt) * inv;
        float r0 = b.rgb[i * 3], r1 = b.rgb[i * 3 + 1], r2 = b.rgb[i * 3 + 2];
        float c0 = b.gc[i * 3], c1 = b.gc[i * 3 + 1], c2 = b.gc[i * 3 + 2];
        lc += (double)fabsf(c0 - r0) + (double)fabsf(c1 - r1) + (double)fabsf(c2 - r2);
        if (use_color) { g0 = w_color * signf0(r0 - c0); g1 = w_color * signf0(r1 - c1); g2 = w_color * signf0(r2 - c2); }
      }
    }
    b.g_depth[i] = gdp;
    b.g_rgb[i * 3] = g0; b.g_rgb[i * 3 + 1] = g1; b.g_rgb[i * 3 + 2] = g2;
  }
  double Lg = block_sum_d(lg, lds);
  double Lc = block_sum_d(lc, lds);
  if (threadIdx.x == 0) {
    double L = use_color ? Lg + (double)w_color * Lc : Lg;
    loss_out[0] = (float)L; loss_out[1] = (float)Lg; loss_out[2] = (float)Lc; loss_out[3] = (float)cnt;
    if ((float)L < best[7]) {
      best[7] = (float)L;
#pragma unroll
      for (int j = 0; j < 7; ++j) best[j] = cam_tensor[j];
    }
  }
}

// per-iteration loss record of psl_map_iters from the double accumulators [iteration][kLossSlots][4] of the ray stage: (L, L_geo, L_col, #rays)
__global__ void k_map_loss_finalize(const double* __restrict__ acc, int n_iters, int n_geo_iters, float w_color,
                                    float* __restrict__ loss_out) {
  int it = blockIdx.x * blockDim.x + threadIdx.x;
  if (it >= n_iters) return;
  double Lg = 0.0, Lc = 0.0, cnt = 0.0;
  for (int k = 0; k < kLossSlots; ++k) {      // slot order: the same sum whichever workgroups contributed
    const double* q = acc + 4 * ((size_t)it * kLossSlots + k);
    Lg += q[0]; Lc += q[1]; cnt += q[2];
  }
  const double L = (it > n_geo_iters) ? Lg + (double)w_color * Lc : Lg;
  loss_out[4 * it] = (float)L; loss_out[4 * it + 1] = (float)Lg; loss_out[4 * it + 2] = (float)Lc;
  loss_out[4 * it + 3] = (float)cnt;
}

// d(loss)/d(pose) from the per-ray gradients, analytic quaternion chain, Adam on (T: lr, quat: 0.2 lr)
// (Tracker.py:305-311,323,183; get_camera_from_tensor common.py:251-267).
__global__ __launch_bounds__(256) void k_pose_step(RayBufs b, int n, float* cam_tensor, float* adam_mv /*[14]*/, int step,
                                                   float lr_T, float lr_q) {
  __shared__ float lds[4][12];
  float acc[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) acc[j] = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float d0 = b.dirs[i * 3], d1 = b.dirs[i * 3 + 1], d2 = b.dirs[i * 3 + 2];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float g = b.g_d[i * 3 + a];
      acc[a * 3 + 0] += d0 * g; acc[a * 3 + 1] += d1 * g; acc[a * 3 + 2] += d2 * g;   // dL/dR[a][k]
      acc[9 + a] += b.g_o[i * 3 + a];                                                  // dL/dT[a]
    }
  }
#pragma unroll
  for (int j = 0; j < 12; ++j) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc[j] += __shfl_xor(acc[j], o);
  }
  int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 12; ++j) lds[w][j] = acc[j];
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  float G[3][3], gT[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int k = 0; k < 3; ++k) G[a][k] = lds[0][a * 3 + k] + lds[1][a * 3 + k] + lds[2][a * 3 + k] + lds[3][a * 3 + k];
    gT[a] = lds[0][9 + a] + lds[1][9 + a] + lds[2][9 + a] + lds[3][9 + a];
  }
  pose_adam(G, gT, cam_tensor, adam_mv, step, lr_T, lr_q);
}

// ------------------------------------------------------------------ tracker, batches of <= 1024 rays: two fused launches
// A tracker iteration on a Replica-sized batch (200 rays) spent 7 of its 10 launches in single-workgroup kernels of
// ~5 us each (ray set-up, depth-outlier mask, compositing, loss, compositing backward, ray gradient, pose step).  With
// one ray per thread of ONE 1024-thread workgroup they collapse into two kernels around the decode:
//   k_track_pre : [ray gradient + pose step of the previous iteration]  ->  ray set-up + depth-outlier mask of this one
//   k_track_mid : compositing -> tracker loss (mask, best pose) -> compositing backward
// so an iteration is k_track_pre, k-NN, decode forward, k_track_mid, decode backward.  Same arithmetic, same order per
// ray as the separate kernels; the block reductions see one element per thread.
__global__ __launch_bounds__(1024) void k_track_pre(int do_step, int do_setup, const float4* __restrict__ dp,
                                                    const float4* __restrict__ dp2, float near_s, float far_s, RayBufs b,
                                                    int n, float* cam_tensor, float* adam_mv, int step, float lr_T,
                                                    float lr_q, psl_cam_intr cam, int H0, int H1, int W0, int W1,
                                                    const FrameDev* __restrict__ fdev, const int* __restrict__ pix_idx,
                                                    AdamBias bias, const float* __restrict__ pose_src = nullptr,
                                                    const float* __restrict__ adam_src = nullptr) {
  __shared__ float red[16][12];
  // one ray per thread up to 1 024 rays (the base mix: 200); written as strided loops (any n), launched for n <= 1 024 only:
  // measured in round 5, ONE workgroup striding over 1 500 / 5 000 rays loses to the parallel ray kernels (cfg 1 -4 %, TUM -9 %)
  if (do_step) {
    float acc[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) acc[j] = 0.f;
    for (int r = threadIdx.x; r < n; r += blockDim.x) {
      // g_rays_o = sum_s dp_s ; g_rays_d = sum_s z_s dp_s  (k_ray_grad)
      float go[3] = {0.f, 0.f, 0.f}, gd3[3] = {0.f, 0.f, 0.f};
      const float gt = b.gd[r];
#pragma unroll
      for (int s = 0; s < S; ++s) {
        float4 g = dp[r * S + s];
        if (dp2) { const float4 g2 = dp2[r * S + s]; g.x += g2.x; g.y += g2.y; g.z += g2.z; }
        const float z = sample_z(gt, s, near_s, far_s);
        go[0] += g.x; go[1] += g.y; go[2] += g.z;
        gd3[0] += z * g.x; gd3[1] += z * g.y; gd3[2] += z * g.z;
      }
      const float d0 = b.dirs[r * 3], d1 = b.dirs[r * 3 + 1], d2 = b.dirs[r * 3 + 2];
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
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < 12; ++j) red[w][j] = acc[j];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float G[3][3], gT[3];
      const int nw = min((n + 63) >> 6, 16);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { float t = 0.f; for (int q = 0; q < nw; ++q) t += red[q][a * 3 + k]; G[a][k] = t; }
        float t = 0.f; for (int q = 0; q < nw; ++q) t += red[q][9 + a]; gT[a] = t;
      }
      if (pose_src) {      // the last pose of a call whose iterations stepped it inside their k-NN launches sits in the second buffer
        for (int j = 0; j < 7; ++j) cam_tensor[j] = pose_src[j];
        for (int j = 0; j < 14; ++j) adam_mv[j] = adam_src[j];
      }
      pose_adam(G, gT, cam_tensor, adam_mv, step, lr_T, lr_q, &bias);
      __threadfence_block();
    }
    __syncthreads();          // the new pose is visible to every thread of the workgroup
  }
  if (!do_setup) return;
  for (int r = threadIdx.x; r < n; r += blockDim.x) ray_setup_one(r, cam, H0, H1, W0, W1, fdev, 1, n, pix_idx, cam_tensor, b);
  __syncthreads();
  depth_inlier_block(b.gd, b.active, n);
}

// Everything of the tracker's ray set-up that does not depend on the pose, for ALL iterations of a psl_track_iters call in one launch
// (workgroup = iteration): camera-frame directions, sensor depth / colour / query radius of the pre-drawn pixels, depth-outlier mask.
// b points at iteration 0's slices; iteration `it` lives `stride` floats further in each.
__global__ __launch_bounds__(1024) void k_track_rays_all(RayBufs b, int stride, int n, psl_cam_intr cam, int H0, int H1, int W0, int W1,
                                                         const FrameDev* __restrict__ fdev, const int* __restrict__ pix_idx) {
  const int it = blockIdx.x;
  b.dirs += (size_t)it * stride; b.gd += (size_t)it * stride; b.gc += (size_t)it * stride; b.active += (size_t)it * stride;
  if (b.rq) b.rq += (size_t)it * stride;
  for (int r = threadIdx.x; r < n; r += blockDim.x) ray_setup_one(r, cam, H0, H1, W0, W1, fdev, 1, n, pix_idx + (size_t)it * n, nullptr, b);
  __syncthreads();
  depth_inlier_block(b.gd, b.active, n);
}

__global__ __launch_bounds__(1024) void k_track_mid(const float4* __restrict__ raw, const int* __restrict__ cnt, RayBufs b, int n,
                                                    float near_s, float far_s, int min_nn, float coef, float w_color,
                                                    int handle_dynamic, int use_color, const float* __restrict__ cam_tensor,
                                                    float* best /*[8]*/, float* loss_out /*[4]*/, float4* __restrict__ d_raw,
                                                    float* __restrict__ zero64) {
  __shared__ double lds[16];
  __shared__ float s_thr;
  if (zero64 && threadIdx.x < 64) zero64[threadIdx.x] = 0.f;   // accumulators of the decode backward that follows
  const int r = threadIdx.x;
  const bool in = r < n;
  float w[S], z[S], al[S], Tt[S], c0[S], c1[S], c2[S];
  float W = 1.f, d = 0.f, v = 0.f, m0 = 0.f, m1 = 0.f, m2 = 0.f, gt = 1.f;
  bool act = false;
  if (in) {                                       // k_composite_fwd
    gt = b.gd[r];
    act = b.active[r] != 0;
    float T = 1.0f, wsum = 0.f;
    int nhas = 0;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const float4 q = raw[r * S + s];
      z[s] = sample_z(gt, s, near_s, far_s);
      al[s] = sigmoidf(coef * q.w);
      Tt[s] = T;
      w[s] = al[s] * T;
      T = T * (1.0f - al[s] + 1e-10f);
      wsum += w[s];
      c0[s] = q.x; c1[s] = q.y; c2[s] = q.z;
      nhas += (cnt[r * S + s] >= min_nn) ? 1 : 0;
    }
    W = wsum + 1e-10f;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, ad = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) { a0 += w[s] * c0[s]; a1 += w[s] * c1[s]; a2 += w[s] * c2[s]; ad += w[s] * z[s]; }
    d = ad / W;
#pragma unroll
    for (int s = 0; s < S; ++s) { const float tmp = z[s] - d; v += w[s] * tmp * tmp; }
    m0 = a0 / W; m1 = a1 / W; m2 = a2 / W;
    b.depth[r] = d; b.var[r] = v;
    b.rgb[r * 3] = m0; b.rgb[r * 3 + 1] = m1; b.rgb[r * 3 + 2] = m2;
    b.valid[r] = nhas >= (S / 2 + 1) ? 1 : 0;
  }
  // k_tracker_loss (Tracker.py:159-180)
  double se = 0.0, sc = 0.0;
  if (in && act) {
    float e = fabsf(gt - d);
    if (handle_dynamic) e = e / sqrtf(v + 1e-10f);
    se = (double)e; sc = 1.0;
  }
  const double tot = block_sum_d(se, lds);
  const double nact = block_sum_d(sc, lds);
  if (threadIdx.x == 0) s_thr = (nact > 0.0) ? 10.0f * (float)(tot / nact) : 0.f;
  __syncthreads();
  const float thr = s_thr;
  double lg = 0.0, lc = 0.0;
  float gdp = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f;
  if (in && act) {
    const float inv = 1.0f / sqrtf(v + 1e-10f);
    const float diff = fabsf(gt - d);
    const float tmp = handle_dynamic ? diff / sqrtf(v + 1e-10f) : diff;
    const bool m = (tmp < thr) && (gt > 0.f) && (d == d) && (v == v);
    if (m) {
      const float e = diff / sqrtf(v + 1e-10f);
      const float ec = fminf(fmaxf(e, 0.f), 1e3f);
      lg = (double)ec;
      if (e <= 1e3f) gdp = signf0(d - gt) * inv;
      const float q0 = b.gc[r * 3], q1 = b.gc[r * 3 + 1], q2 = b.gc[r * 3 + 2];
      lc = (double)fabsf(q0 - m0) + (double)fabsf(q1 - m1) + (double)fabsf(q2 - m2);
      if (use_color) { g0 = w_color * signf0(m0 - q0); g1 = w_color * signf0(m1 - q1); g2 = w_color * signf0(m2 - q2); }
    }
  }
  const double Lg = block_sum_d(lg, lds);
  const double Lc = block_sum_d(lc, lds);
  if (threadIdx.x == 0) {
    const double L = use_color ? Lg + (double)w_color * Lc : Lg;
    loss_out[0] = (float)L; loss_out[1] = (float)Lg; loss_out[2] = (float)Lc; loss_out[3] = (float)nact;
    if ((float)L < best[7]) {
      best[7] = (float)L;
#pragma unroll
      for (int j = 0; j < 7; ++j) best[j] = cam_tensor[j];
    }
  }
  if (!in) return;
  // k_composite_bwd with g_var = 0
  const float gv = 0.f;
  float dvd = 0.f;
#pragma unroll
  for (int s = 0; s < S; ++s) dvd += w[s] * (z[s] - d);
  const float gdt = gdp + gv * (-2.0f * dvd);
  float gw[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const float dz = z[s] - d;
    gw[s] = gv * dz * dz + (gdt * dz + g0 * (c0[s] - m0) + g1 * (c1[s] - m1) + g2 * (c2[s] - m2)) / W;
  }
  float suffix = 0.f;  // sum_{t>s} gw_t w_t
#pragma unroll
  for (int s = S - 1; s >= 0; --s) {
    const float ga = gw[s] * Tt[s] - suffix / (1.0f - al[s] + 1e-10f);
    const float gocc = ga * coef * al[s] * (1.0f - al[s]);
    const float ws = w[s] / W;
    d_raw[r * S + s] = make_float4(g0 * ws, g1 * ws, g2 * ws, gocc);
    suffix += gw[s] * w[s];
  }
}

// ------------------------------------------------------------------ per-frame exposure compensation
// MLP_exposure (decoder.py:243-258): aff = linear2(softplus100(linear1(e))), e [8] -> 128 -> 12.  The affine of every
// window frame is evaluated by one small workgroup each; its backward, the sum over frames and Adam on the MLP and on
// the trainable latent run in ONE single-workgroup kernel which also re-evaluates the affines for the next iteration.
constexpr int EXD = PSL_EXPOSURE_DIM, EXH = 128, EXO = 12;
constexpr int EX_W1 = 0, EX_B1 = EXH * EXD, EX_W2 = EX_B1 + EXH, EX_B2 = EX_W2 + EXO * EXH, EX_N = EX_B2 + EXO;
static_assert(EX_N == PSL_EXPOSURE_MLP_FLOATS, "exposure MLP layout");

// one thread per hidden unit; aff [12] and act [128] of frame f
__device__ __forceinline__ void exposure_eval(const float* __restrict__ mlp, const float* __restrict__ e, float* aff,
                                              float* act, float* lds_a /*[128]*/) {
  const int j = threadIdx.x;
  if (j < EXH) {
    float h = mlp[EX_B1 + j];
#pragma unroll
    for (int k = 0; k < EXD; ++k) h = fmaf(mlp[EX_W1 + j * EXD + k], e[k], h);
    const float a = softplus100(h);
    lds_a[j] = a;
    act[j] = a;
  }
  __syncthreads();
  if (j < EXO) {
    float o = mlp[EX_B2 + j];
    for (int k = 0; k < EXH; ++k) o = fmaf(mlp[EX_W2 + j * EXH + k], lds_a[k], o);
    aff[j] = o;
  }
  __syncthreads();
}

__global__ __launch_bounds__(128) void k_exposure_fwd(const float* __restrict__ mlp, const float* __restrict__ feats, float* aff,
                                                      float* act) {
  __shared__ float la[EXH];
  const int f = blockIdx.x;
  exposure_eval(mlp, feats + f * EXD, aff + f * EXO, act + f * EXH, la);
}

// g_aff [F][12] (consumed and cleared), act [F][128]; trainable latent = row F-1.  128 threads.
__global__ __launch_bounds__(128) void k_exposure_step(float* mlp, float* feats, int F, float* g_aff, float* aff, float* act,
                                                       float* adam_m, float* adam_v, int step, float lr_mlp, float lr_feat) {
  __shared__ float la[EXH];
  __shared__ float sg[64 * EXO];
  __shared__ float sde[EXH][EXD + 1];
  const int j = threadIdx.x;
  for (int e = j; e < F * EXO; e += EXH) sg[e] = g_aff[e];
  __syncthreads();
  float dW1[EXD], dW2[EXO], db1 = 0.f;
#pragma unroll
  for (int k = 0; k < EXD; ++k) dW1[k] = 0.f;
#pragma unroll
  for (int o = 0; o < EXO; ++o) dW2[o] = 0.f;
  float dh_last = 0.f;
  for (int f = 0; f < F; ++f) {
    const float a = act[f * EXH + j];
    float da = 0.f;
#pragma unroll
    for (int o = 0; o < EXO; ++o) {
      const float g = sg[f * EXO + o];
      da = fmaf(mlp[EX_W2 + o * EXH + j], g, da);
      dW2[o] = fmaf(g, a, dW2[o]);
    }
    const float dh = da * softplus100_grad_from_out(a);
    db1 += dh;
#pragma unroll
    for (int k = 0; k < EXD; ++k) dW1[k] = fmaf(dh, feats[f * EXD + k], dW1[k]);
    if (f == F - 1) dh_last = dh;
  }
  // d/d(latent of the current frame) = W1^T dh: per-thread products, reduced over the hidden units below
#pragma unroll
  for (int k = 0; k < EXD; ++k) sde[j][k] = mlp[EX_W1 + j * EXD + k] * dh_last;
  __syncthreads();
  // Adam (torch single-tensor op order, adam1) -- every thread owns row j of linear1 and column j of linear2
#pragma unroll
  for (int k = 0; k < EXD; ++k) { const int i = EX_W1 + j * EXD + k; adam1(mlp[i], dW1[k], adam_m[i], adam_v[i], lr_mlp, step); }
  { const int i = EX_B1 + j; adam1(mlp[i], db1, adam_m[i], adam_v[i], lr_mlp, step); }
#pragma unroll
  for (int o = 0; o < EXO; ++o) { const int i = EX_W2 + o * EXH + j; adam1(mlp[i], dW2[o], adam_m[i], adam_v[i], lr_mlp, step); }
  if (j < EXO) {
    float db2 = 0.f;
    for (int f = 0; f < F; ++f) db2 += sg[f * EXO + j];
    const int i = EX_B2 + j;
    adam1(mlp[i], db2, adam_m[i], adam_v[i], lr_mlp, step);
  }
  if (j < EXD) {
    float de = 0.f;
    for (int h = 0; h < EXH; ++h) de += sde[h][j];
    adam1(feats[(F - 1) * EXD + j], de, adam_m[EX_N + j], adam_v[EX_N + j], lr_feat, step);
  }
  for (int e = j; e < F * EXO; e += EXH) g_aff[e] = 0.f;
  __threadfence_block();
  __syncthreads();
  for (int f = 0; f < F; ++f) exposure_eval(mlp, feats + f * EXD, aff + f * EXO, act + f * EXH, la);
}

// ------------------------------------------------------------------ frustum feature selection
// Mapper.get_mask_from_c2w (src/Mapper.py:120-168): project every neural point with the frame pose, bilinear
// sensor-depth lookup (cv2.remap INTER_LINEAR, constant-0 border), keep points inside the (edge-enlarged) image
// whose camera depth lies in [0, depth+0.5].  Zero lookups are replaced by the maximum over the per-point lookups
// (:161-162).
// projection + bilinear sensor-depth lookup of one point (cv2.remap INTER_LINEAR, constant-0 border)
// cv2.remap(depth, u, v, INTER_LINEAR) with the default constant-0 border (src/Mapper.py:149-155), restated from OpenCV's
// imgproc/src/imgwarp.cpp (remap() with CV_32FC1 maps + remapBilinear<float>): the coordinates are converted to fixed point
// with INTER_BITS = 5 -- sx = cvRound(u * 32) (round half to even), integer part sx >> 5 saturated to int16, fraction
// (sx & 31) / 32 --, the four weights come from a float table w[ky][kx] = vy[ky] * vx[kx], v = {1 - f, f}, and the taps are
// summed left to right.  So the lookup position is quantised to 1/32 pixel; exact bilinear interpolation (the `exact`
// variant below, psl_debug_option("remap_cv2", 0)) differs by up to |grad depth| / 64 per pixel.
__device__ __forceinline__ float remap_linear_cv2(const float* __restrict__ img, int W, int H, float u, float v) {
  if (!(fabsf(u) < 1.0e6f) || !(fabsf(v) < 1.0e6f)) return 0.f;      // far outside (or NaN): every tap is border (cv2 saturates to int16)
  const int sx = (int)rintf(u * 32.0f), sy = (int)rintf(v * 32.0f);
  const int ix = max(-32768, min(32767, sx >> 5)), iy = max(-32768, min(32767, sy >> 5));
  const float fx = (float)(sx & 31) * (1.0f / 32.0f), fy = (float)(sy & 31) * (1.0f / 32.0f);
  const float vx[2] = {1.0f - fx, fx}, vy[2] = {1.0f - fy, fy};
  float d = 0.f;
#pragma unroll
  for (int ky = 0; ky < 2; ++ky)
#pragma unroll
    for (int kx = 0; kx < 2; ++kx) {
      const int x = ix + kx, y = iy + ky;
      const float val = (x >= 0 && x < W && y >= 0 && y < H) ? img[(size_t)y * W + x] : 0.f;
      d = __fadd_rn(d, __fmul_rn(val, __fmul_rn(vy[ky], vx[kx])));
    }
  return d;
}

struct FrustumPt { float u, v, d, mz; };
template <bool CV2>
__device__ __forceinline__ FrustumPt frustum_point(const float4 p, const float* __restrict__ w2c, const psl_cam_intr& cam,
                                                  const float* __restrict__ depth) {
  double x = (double)w2c[0] * p.x + (double)w2c[1] * p.y + (double)w2c[2] * p.z + (double)w2c[3];
  double y = (double)w2c[4] * p.x + (double)w2c[5] * p.y + (double)w2c[6] * p.z + (double)w2c[7];
  double zc = (double)w2c[8] * p.x + (double)w2c[9] * p.y + (double)w2c[10] * p.z + (double)w2c[11];
  x = -x;                                  // cam_cord[:, 0] *= -1
  double z = zc + 1e-5;
  FrustumPt o;
  o.u = (float)(((double)cam.fx * x + (double)cam.cx * zc) / z);
  o.v = (float)(((double)cam.fy * y + (double)cam.cy * zc) / z);
  o.mz = (float)(-z);
  if (CV2) { o.d = remap_linear_cv2(depth, cam.W, cam.H, o.u, o.v); return o; }
  float u0 = floorf(o.u), v0 = floorf(o.v);
  float fu = o.u - u0, fv = o.v - v0;
  float d = 0.f;
#pragma unroll
  for (int dv = 0; dv < 2; ++dv)
#pragma unroll
    for (int du = 0; du < 2; ++du) {
      // The bounds test is made on the FLOATS: a point in the camera plane (z ~ 0) projects to +-1e30 or NaN, for which
      // (int)u0 is undefined behaviour -- the compiler folded the saturated INT_MAX + 1 into a 64-bit index and the
      // "out of image" lookup became a wild load (one run in ~20 of bench.py, whenever a map point met z ~ 0).
      const float uf = u0 + (float)du, vf = v0 + (float)dv;
      const bool ok = uf >= 0.f && uf < (float)cam.W && vf >= 0.f && vf < (float)cam.H;   // false for NaN
      float val = 0.f;
      if (ok) val = depth[(size_t)(int)vf * cam.W + (int)uf];
      d += val * (du ? fu : 1.f - fu) * (dv ? fv : 1.f - fv);
    }
  o.d = d;
  return o;
}

// np.max(depths) of Mapper.py:161-162: the maximum over the PER-POINT lookups (not over the image); non-negative floats
// order like their bit patterns
template <bool CV2>
__global__ __launch_bounds__(256) void k_frustum_dmax(const float4* __restrict__ pos, int n, const float* __restrict__ w2c,
                                                      psl_cam_intr cam, const float* __restrict__ depth, unsigned* dmax_bits) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  float d = 0.f;
  if (i < n) { d = frustum_point<CV2>(pos[i], w2c, cam, depth).d; if (!(d >= 0.f)) d = 0.f; }
  unsigned b = __float_as_uint(d);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) b = max(b, (unsigned)__shfl_xor((int)b, o));
  // one address for the whole launch: a wavefront whose maximum cannot raise the value it reads there leaves without the atomic
  // (16 k same-address atomics were 120 us of a kernel that reads 16 MB)
  if ((threadIdx.x & 63) == 0 && b > __atomic_load_n(dmax_bits, __ATOMIC_RELAXED)) atomicMax(dmax_bits, b);
}

template <bool CV2>
__global__ __launch_bounds__(256) void k_frustum_flags(const float4* __restrict__ pos, int n, const float* __restrict__ w2c,
                                                       psl_cam_intr cam, const float* __restrict__ depth, float depth_max,
                                                       const unsigned* __restrict__ dmax_bits, float edge,
                                                       int* __restrict__ flags) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const FrustumPt f = frustum_point<CV2>(pos[i], w2c, cam, depth);
  float d = f.d;
  if (d == 0.f) d = dmax_bits ? __uint_as_float(*dmax_bits) : depth_max;
  bool inb = (f.u < (float)cam.W - edge) && (f.u > edge) && (f.v < (float)cam.H - edge) && (f.v > edge);
  flags[i] = (inb && f.mz >= 0.f && f.mz <= d + 0.5f) ? 1 : 0;
}

// ordered compaction of flags -> sel[n_sel] (ascending point index) and row_map[n] (-1 = not selected)
__global__ __launch_bounds__(256) void k_flag_block_sums(const int* __restrict__ flags, int n, int* block_sums) {
  int base = blockIdx.x * 1024;
  int s = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) { int i = base + threadIdx.x * 4 + j; if (i < n) s += flags[i]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  __shared__ int l[4];
  if ((threadIdx.x & 63) == 0) l[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) block_sums[blockIdx.x] = l[0] + l[1] + l[2] + l[3];
}

__global__ __launch_bounds__(1024) void k_flag_scan_top(int* block_sums, int nblocks, int* total_out) {
  // sequential-in-chunks exclusive scan by one workgroup (nblocks <= 16384)
  __shared__ int carry;
  __shared__ int wsum[16];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 1024) {
    int i = base + threadIdx.x;
    int v = (i < nblocks) ? block_sums[i] : 0;
    int x = v;
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    int off = carry;
    for (int j = 0; j < w; ++j) off += wsum[j];
    if (i < nblocks) block_sums[i] = off + x - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = off + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(256) void k_flag_compact(const int* __restrict__ flags, int n, const int* __restrict__ block_off,
                                                      int* __restrict__ sel, int* __restrict__ row_map) {
  __shared__ int l[4];
  int base = blockIdx.x * 1024;
  int f[4]; int s = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) { int i = base + threadIdx.x * 4 + j; f[j] = (i < n) ? flags[i] : 0; s += f[j]; }
  int x = s;
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { int y = __shfl_up(x, o); if (lane >= o) x += y; }
  if (lane == 63) l[w] = x;
  __syncthreads();
  int off = block_off[blockIdx.x];
  for (int j = 0; j < w; ++j) off += l[j];
  off += x - s;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int i = base + threadIdx.x * 4 + j;
    if (i < n) {
      if (f[j]) { sel[off] = i; row_map[i] = off; ++off; }
      else row_map[i] = -1;
    }
  }
}

}  // namespace psl

using namespace psl;

// A/B switches (psl_debug_option "lazy_adam" / "track_fused", or the environment at load time)
namespace psl {
static int env_flag(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
int g_color_split = env_flag("PSL_COLOR_SPLIT", 1);   // colour-stage launch structure: 0 fused tiles, 2 split kernels, 1 by launch size (psl_decode.h)
int g_wave_trunk_tiles = env_flag("PSL_WAVE_TRUNK", 1024);   // split structure: trunk as one wavefront per tile from this many tiles on
int g_lazy_adam = env_flag("PSL_LAZY_ADAM", 1);
int g_track_fused = env_flag("PSL_TRACK_FUSED", 3);   // 3: + the pose step inside the k-NN launch (TrackPose); 2: k_track_mid inside the decode backward (TrackFuse); 1: its own launch; 0: the ten-launch iteration
int g_dw_fused = env_flag("PSL_DW_FUSED", 1);
int g_knn_overlap = env_flag("PSL_KNN_OVERLAP", 1);
int g_knn_side_blocks = env_flag("PSL_KNN_SIDE_BLOCKS", 512);
// geometry-stage mapper iterations as ONE launch (psl_decode_geo.hip) instead of decode fwd / ray kernel / decode bwd
int g_geo_fused = env_flag("PSL_GEO_FUSED", 1);
// colour stage (no per-frame exposure): compositing + loss + compositing backward inside the decode backward, no ray kernel
int g_ray_in_bwd = env_flag("PSL_RAY_IN_BWD", 1);
// the frustum selection's depth lookup: 1 = cv2.remap's INTER_LINEAR as the reference calls it (coordinates quantised to 1/32
// pixel, the default: it is what src/Mapper.py:149-155 computes), 0 = exact bilinear interpolation (rounds 1-3)
int g_remap_cv2 = env_flag("PSL_REMAP_CV2", 1);
constexpr int kGeoIterMaxSamples = 10000;
}  // namespace psl

// ---------------------------------------------------------------------------------------------- C ABI
static RayBufs carve_rays(float*& p, int n) {
  RayBufs b;
  auto take = [&](size_t k) { float* r = p; p += (k + 3) / 4 * 4; return r; };
  b.rays_o = take(3 * (size_t)n); b.rays_d = take(3 * (size_t)n); b.dirs = take(3 * (size_t)n);
  b.gd = take(n); b.gc = take(3 * (size_t)n); b.rq = take(n); b.active = (int*)take(n);
  b.depth = take(n); b.var = take(n); b.rgb = take(3 * (size_t)n); b.valid = (unsigned char*)take(n);
  b.g_depth = take(n); b.g_rgb = take(3 * (size_t)n); b.g_o = take(3 * (size_t)n); b.g_d = take(3 * (size_t)n);
  return b;
}
static int64_t rays_floats(int n) {   // size of carve_rays' layout: carved from an aligned dummy base that is never dereferenced
  float* const base = reinterpret_cast<float*>((uintptr_t)1 << 20);
  float* p = base; (void)carve_rays(p, n); return (int64_t)(p - base);
}

// Initial pose of the next frame (src/Tracker.py:283-290, const_speed_assumption): delta = pre_c2w @ inv(c2w[idx-2]),
// estimate = delta @ pre_c2w, handed back as a camera tensor (get_tensor_from_camera, src/common.py:270-295) -- on the device,
// from the tracker's own two previous camera tensors, so that a closed loop never copies a pose to the host.  The poses are
// rigid: inv([R t]) = [R^T, -R^T t] (the reference inverts the 4x4 numerically; the two agree to fp32 rounding).  Quaternion
// by Shepperd's method (largest of trace / diagonal terms), w >= 0 (q and -q are the same rotation).
__global__ void k_pose_const_speed(const float* __restrict__ prev, const float* __restrict__ prev2, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float R1[3][3], t1[3] = {prev[4], prev[5], prev[6]};
  quat_to_rot(prev, R1);
  float Re[3][3], te[3];
  if (prev2) {
    float R0[3][3], t0[3] = {prev2[4], prev2[5], prev2[6]}, Rd[3][3], td[3];
    quat_to_rot(prev2, R0);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Rd[i][j] = (R1[i][0] * R0[j][0] + R1[i][1] * R0[j][1]) + R1[i][2] * R0[j][2];      // R1 R0^T
    for (int i = 0; i < 3; ++i) td[i] = t1[i] - ((Rd[i][0] * t0[0] + Rd[i][1] * t0[1]) + Rd[i][2] * t0[2]);
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) Re[i][j] = (Rd[i][0] * R1[0][j] + Rd[i][1] * R1[1][j]) + Rd[i][2] * R1[2][j];
      te[i] = ((Rd[i][0] * t1[0] + Rd[i][1] * t1[1]) + Rd[i][2] * t1[2]) + td[i];
    }
  } else {
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) Re[i][j] = R1[i][j]; te[i] = t1[i]; }
  }
  const float m00 = Re[0][0], m11 = Re[1][1], m22 = Re[2][2], tr = (m00 + m11) + m22;
  float q[4];
  if (tr >= m00 && tr >= m11 && tr >= m22) {
    q[0] = 1.f + tr; q[1] = Re[2][1] - Re[1][2]; q[2] = Re[0][2] - Re[2][0]; q[3] = Re[1][0] - Re[0][1];
  } else if (m00 >= m11 && m00 >= m22) {
    q[0] = Re[2][1] - Re[1][2]; q[1] = 1.f + m00 - m11 - m22; q[2] = Re[0][1] + Re[1][0]; q[3] = Re[0][2] + Re[2][0];
  } else if (m11 >= m22) {
    q[0] = Re[0][2] - Re[2][0]; q[1] = Re[0][1] + Re[1][0]; q[2] = 1.f - m00 + m11 - m22; q[3] = Re[1][2] + Re[2][1];
  } else {
    q[0] = Re[1][0] - Re[0][1]; q[1] = Re[0][2] + Re[2][0]; q[2] = Re[1][2] + Re[2][1]; q[3] = 1.f - m00 - m11 + m22;
  }
  const float nrm = sqrtf(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
  const float sgn = (q[0] < 0.f) ? -1.f : 1.f;
  for (int i = 0; i < 4; ++i) out[i] = sgn * q[i] / nrm;
  out[4] = te[0]; out[5] = te[1]; out[6] = te[2];
}

extern "C" int psl_pose_const_speed(const float* cam_prev, const float* cam_prev2, float* cam_out, void* stream) {
  if (!cam_prev || !cam_out) { set_error("psl_pose_const_speed: missing argument"); return PSL_ERR_ARG; }
  hipLaunchKernelGGL(k_pose_const_speed, dim3(1), dim3(64), 0, (hipStream_t)stream, cam_prev, cam_prev2, cam_out);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}

extern "C" int64_t psl_track_ws_floats(int n_pix) {
  if (n_pix < 0) return PSL_ERR_ARG;
  return rays_floats(n_pix) + psl_render_ws_floats(n_pix, PSL_STAGE_COLOR | PSL_PTS_GRAD) + 64 +
         (int64_t)((sizeof(FrameDev) + 3) / 4) + 16;
}

extern "C" int psl_track_iters(psl_ctx* ctx, const psl_track_args* t, void* stream) {
  if (!ctx || !t || !t->pix_idx || !t->cam_tensor || !t->adam_state || !t->ws || !t->frame.depth || !t->frame.color ||
      !t->fallback || !t->best_out) { set_error("psl_track_iters: missing argument"); return PSL_ERR_ARG; }
  if (t->n_pix <= 0 || t->n_pix > 65536) { set_error("psl_track_iters: n_pix must be in [1,65536]"); return PSL_ERR_ARG; }
  const psl_exposure_args* ex = t->exposure;
  if (ex && (!ex->mlp || !ex->feats || !ex->adam)) { set_error("psl_track_iters: incomplete exposure block"); return PSL_ERR_ARG; }
  if (ctx->index_points != ctx->n_points) { set_error("psl_track_iters: index is stale"); return PSL_ERR_STATE; }
  hipStream_t s = (hipStream_t)stream;
  const int n = t->n_pix;
  float* p = t->ws;
  RayBufs b = carve_rays(p, n);
  if (!t->frame.r_query) b.rq = nullptr;
  float* loss_scratch = p; p += 64;
  FrameDev* fdev = (FrameDev*)p; p += (sizeof(FrameDev) + 3) / 4 + 4;
  float* rws = (float*)(((uintptr_t)p + 15) & ~(uintptr_t)15);
  // the carve above must stay inside what psl_track_ws_floats() promises the caller
  if ((rws - t->ws) + psl_render_ws_floats(n, PSL_STAGE_COLOR | PSL_PTS_GRAD) > psl_track_ws_floats(n)) {
    set_error("psl_track_iters: internal workspace layout exceeds psl_track_ws_floats"); return PSL_ERR_STATE;
  }
  dbg_range("trk.ws", t->ws, sizeof(float) * (size_t)psl_track_ws_floats(n));
  dbg_range("trk.pix_idx", t->pix_idx, sizeof(int) * (size_t)t->n_iters * n);
  dbg_range("trk.fallback", t->fallback, sizeof(float) * (size_t)t->n_iters * 64);
  dbg_range("trk.cam", t->cam_tensor, 28); dbg_range("trk.adam", t->adam_state, 56); dbg_range("trk.best", t->best_out, 32);
  dbg_range("trk.depth", t->frame.depth, sizeof(float) * (size_t)t->cam.H * t->cam.W);
  dbg_range("trk.color", t->frame.color, sizeof(float) * 3 * (size_t)t->cam.H * t->cam.W);
  if (t->frame.r_query) dbg_range("trk.r_query", t->frame.r_query, sizeof(float) * (size_t)t->cam.H * t->cam.W);
  if (t->loss_out) dbg_range("trk.loss_out", t->loss_out, sizeof(float) * 4 * (size_t)t->n_iters);
  FrameDev fh;
  memset(&fh, 0, sizeof(fh));
  fh.depth = t->frame.depth; fh.color = t->frame.color; fh.r_query = t->frame.r_query;
  // The frame descriptor and best[0..6] = pose, best[7] = loss = +big go to the device as KERNEL ARGUMENTS (captured
  // at launch).  An asynchronous copy from these stack variables is read by the runtime whenever the stream gets
  // to it -- after this function has returned when the host runs ahead of the GPU (no sync between tracked frames).
  hipLaunchKernelGGL(k_track_init, dim3(1), dim3(64), 0, s, fdev, fh, t->best_out);
  PSL_LAUNCH_CHECK();
  psl_render_args ra{};
  memset(&ra, 0, sizeof(ra));
  ra.n_rays = n; ra.flags = PSL_STAGE_COLOR | PSL_PTS_GRAD; ra.sigmoid_coef = t->sigmoid_coef;
  ra.rays_o = b.rays_o; ra.rays_d = b.rays_d; ra.gt_depth = b.gd; ra.r_query = b.rq;
  ra.geo_feats = t->geo_feats; ra.col_feats = t->col_feats; ra.params = t->params; ra.col_embed_B = t->col_embed_B;
  ra.ws = rws; ra.depth = b.depth; ra.var = b.var; ra.rgb = b.rgb; ra.valid_ray = b.valid;
  psl_render_grads rg;
  memset(&rg, 0, sizeof(rg));
  rg.g_depth = b.g_depth; rg.g_rgb = b.g_rgb; rg.g_rays_o = b.g_o; rg.g_rays_d = b.g_d;
  // sample_with_color_grad (Tracker.py:115-128): indices address the whole image, no crop
  const int eh = t->pix_full_image ? 0 : t->edge_h, ew = t->pix_full_image ? 0 : t->edge_w;
  float *ex_aff = nullptr, *ex_act = nullptr, *ex_g = nullptr;
  if (ex) {   // per-frame exposure (Tracker.py:305-311): affine of this frame's latent, refreshed by k_exposure_step
    ex_aff = ctx->d_expo; ex_act = ex_aff + 64 * EXO; ex_g = ex_act + 64 * EXH;
    hipLaunchKernelGGL(k_exposure_fwd, dim3(1), dim3(128), 0, s, ex->mlp, ex->feats, ex_aff, ex_act);
    PSL_LAUNCH_CHECK();
    ra.flags |= PSL_HAS_AFFINE; ra.exposure_affine = ex_aff; rg.g_exposure_affine = ex_g;
  }
  // batches of <= 1024 rays: the seven single-workgroup kernels of an iteration collapse into k_track_pre / k_track_mid
  // ... and since round 5 k_track_mid's work runs inside the decode backward (TrackFuse, psl_decode.h): an iteration is
  // k_track_pre, k-NN, decode forward, decode backward -- four launches.  Larger batches (Replica 1 500 px, TUM / ScanNet
  // 5 000 px) keep the ten launches of rounds 1-4, measured in round 5 (profiles/r05_track_fuse_ab.txt): one workgroup
  // striding over thousands of rays loses to the parallel ray kernels (cfg 1 -4 %, TUM -9 %, ScanNet -7 %), and folding only
  // the compositing backward into the decode backward costs that kernel ~10 us at 5 000 rays for the 5-us launch it saves.
  // PSL_TRACK_FUSED=1 / psl_debug_option("track_fused", 1): rounds 3-4 (k_track_mid as a launch of its own); 0: ten launches.
  // tracking.handle_dynamic = False (the median mask, Tracker.py:166-168; no shipped config) takes the ten-launch path: the
  // median lives in k_tracker_loss only
  const bool fused = n <= 1024 && g_track_fused != 0 && t->handle_dynamic != 0;
  const bool mid_in_bwd = fused && g_track_fused >= 2;
  struct FusedGuard { psl_ctx* c; ~FusedGuard() { c->fused_ray = false; c->track_fuse = nullptr; c->fwd_zero64 = nullptr; } } fused_guard{ctx};
  const RenderWs rw = carve_ws(ra.ws, n, ra.flags | (ctx->cfg.encode_rel_pos ? 0x10000 : 0));
  if (fused) { ctx->fused_ray = true; rg.g_rays_o = nullptr; rg.g_rays_d = nullptr; }
  if (mid_in_bwd) ctx->fwd_zero64 = ctx->d_small;     // the forward clears the backward's accumulators (k_track_mid did)
  // ... and since round 6 k_track_pre's work is gone from the iteration too (three launches: k-NN, forward, backward): what does not
  // depend on the pose (pixels, camera-frame directions, depth mask) is prepared for all iterations by ONE launch per call, and the pose
  // step runs in the prologue of every workgroup of the k-NN launch (TrackPose, psl_pose.h), between two pose / Adam buffers.
  const bool pose_in_knn = mid_in_bwd && g_track_fused >= 3 && t->n_iters > 0;
  struct PoseGuard { psl_ctx* c; ~PoseGuard() { c->track_pose = nullptr; } } pose_guard{ctx};
  const int pstride = (9 * n + 3) & ~3;          // per iteration: dirs 3n | gd n | gc 3n | rq n | active n
  float* pose_buf[2] = {t->cam_tensor, nullptr};
  float* adam_buf[2] = {t->adam_state, nullptr};
  auto pref = [&](int it) {
    RayBufs q = b;
    float* base = ctx->trk_pref + (size_t)it * pstride;
    q.dirs = base; q.gd = base + 3 * n; q.gc = base + 4 * n; q.rq = b.rq ? base + 7 * n : nullptr; q.active = (int*)(base + 8 * n);
    return q;
  };
  // batches above 1 024 rays (and handle_dynamic = False) keep their ray stage and pose step as launches, but the per-call
  // preparation serves them too: ray set-up and depth mask leave the iteration, the k-NN launch turns the directions (POSE = 2)
  const bool prefetch = g_track_fused >= 3 && t->n_iters > 0 && (pose_in_knn || !fused);
  if (prefetch) {
    const size_t need = (size_t)t->n_iters * pstride + 32;
    if (ctx->trk_pref_cap < need) {
      if (ctx->trk_pref) { PSL_HIP(hipStreamSynchronize(s)); (void)hipFree(ctx->trk_pref); ctx->trk_pref = nullptr; ctx->trk_pref_cap = 0; }
      PSL_HIP(hipMalloc(&ctx->trk_pref, sizeof(float) * need)); psl::poison(ctx->trk_pref, sizeof(float) * need);
      ctx->trk_pref_cap = need;
      dbg_range("trk_pref", ctx->trk_pref, sizeof(float) * need);
    }
    pose_buf[1] = ctx->trk_pref + (size_t)t->n_iters * pstride; adam_buf[1] = pose_buf[1] + 8;
    ProfScope ps(ctx, PROF_MISC, s);
    RayBufs q = pref(0);
    q.rays_o = nullptr; q.rays_d = nullptr;
    hipLaunchKernelGGL(k_track_rays_all, dim3(t->n_iters), dim3(1024), 0, s, q, pstride, n, t->cam, eh, t->cam.H - eh, ew, t->cam.W - ew,
                       fdev, t->pix_idx);
    PSL_LAUNCH_CHECK();
  }
  auto adam_bias_host = [&](int it) {
    AdamBias bias;
    const int step = std::max(t->step0 + it, 1);
    bias.bc1 = 1.0 - pow((double)0.9f, (double)step);
    bias.sqrt_bc2 = (float)sqrt(1.0 - pow((double)0.999f, (double)step));
    return bias;
  };
  auto track_pre = [&](int it, int do_step, int do_setup) {
    ProfScope ps(ctx, PROF_MISC, s);
    // bias corrections of Adam step `step0 + it` with the formulas of adam_bias(), evaluated here instead of by one thread
    AdamBias bias;
    const int step = std::max(t->step0 + it, 1);
    bias.bc1 = 1.0 - pow((double)0.9f, (double)step);
    bias.sqrt_bc2 = (float)sqrt(1.0 - pow((double)0.999f, (double)step));
    hipLaunchKernelGGL(k_track_pre, dim3(1), dim3(1024), 0, s, do_step, do_setup, (const float4*)rw.dp, (const float4*)rw.dp2,
                       ctx->cfg.near_end_surface, ctx->cfg.far_end_surface, b, n, t->cam_tensor, t->adam_state, t->step0 + it,
                       t->lr_T, t->lr_quat, t->cam, eh, t->cam.H - eh, ew, t->cam.W - ew, fdev,
                       t->pix_idx + (size_t)std::min(it, t->n_iters - 1) * n, bias);
  };
  for (int it = 0; it < t->n_iters; ++it) {
    RayBufs bi = b;                            // this iteration's rays (pose_in_knn: its slices of the per-call buffers)
    TrackPose tp{};
    if (pose_in_knn) {
      bi = pref(it);
      const RayBufs bp = pref(std::max(it - 1, 0));
      ra.gt_depth = bi.gd; ra.r_query = bi.rq;
      tp.do_step = it > 0 ? 1 : 0;
      tp.dp = (const float4*)rw.dp; tp.dp2 = (const float4*)rw.dp2;
      tp.dirs_prev = bp.dirs; tp.gd_prev = bp.gd;
      tp.pose_in = pose_buf[(it > 0 ? it - 1 : 0) & 1]; tp.adam_in = adam_buf[(it > 0 ? it - 1 : 0) & 1];
      tp.pose_out = pose_buf[it & 1]; tp.adam_out = adam_buf[it & 1];
      tp.step = t->step0 + it; tp.lr_T = t->lr_T; tp.lr_q = t->lr_quat; tp.bias = adam_bias_host(it);
      tp.dirs = bi.dirs; tp.rays_o = b.rays_o; tp.rays_d = b.rays_d; tp.n = n;
      tp.near_s = ctx->cfg.near_end_surface; tp.far_s = ctx->cfg.far_end_surface;
      ctx->track_pose = &tp;
    } else if (fused) {
      track_pre(it, it > 0 ? 1 : 0, 1);       // pose step of iteration it-1 (Adam step number step0 + it), rays of iteration it
      PSL_LAUNCH_CHECK();
    } else if (prefetch) {
      bi = pref(it);
      ra.gt_depth = bi.gd; ra.r_query = bi.rq;
      tp.rotate_only = 1; tp.pose_in = t->cam_tensor; tp.dirs = bi.dirs; tp.rays_o = b.rays_o; tp.rays_d = b.rays_d; tp.n = n;
      ctx->track_pose = &tp;
    } else {
      ProfScope ps(ctx, PROF_MISC, s);
      hipLaunchKernelGGL(k_ray_setup, dim3((n + 255) / 256), dim3(256), 0, s, t->cam, eh, t->cam.H - eh,
                         ew, t->cam.W - ew, fdev, 1, n, t->pix_idx + (size_t)it * n, t->cam_tensor, b);
      hipLaunchKernelGGL(k_depth_inlier, dim3(1), dim3(1024), 0, s, b.gd, b.active, n);
      PSL_LAUNCH_CHECK();
    }
    ra.fallback_geo = t->fallback + (size_t)it * 64;
    ra.fallback_col = t->fallback + (size_t)it * 64 + 32;
    int rc = render_fwd_impl(ctx, &ra, s, it == 0);
    ctx->track_pose = nullptr;
    if (rc) return rc;
    float* lo = t->loss_out ? t->loss_out + 4 * (size_t)it : loss_scratch;
    TrackFuse tf{};
    if (mid_in_bwd) {
      tf = TrackFuse{bi.active, bi.gc, t->sigmoid_coef, t->w_color, t->handle_dynamic, t->use_color, n, b.depth, b.var, b.rgb, b.valid,
                     pose_in_knn ? pose_buf[it & 1] : t->cam_tensor, t->best_out, lo, 1};
      ctx->track_fuse = &tf;
    } else if (fused) {
      ProfScope ps(ctx, PROF_COMPOSITE, s, 324.0 * n, true);
      PSL_KLAUNCH(k_track_mid, dim3(1), dim3(1024), 0, s, (const float4*)rw.raw, rw.cnt, bi, n, ctx->cfg.near_end_surface,
                         ctx->cfg.far_end_surface, ctx->cfg.min_nn_num, t->sigmoid_coef, t->w_color, t->handle_dynamic,
                         t->use_color, t->cam_tensor, t->best_out, lo, (float4*)rw.d_raw, ctx->d_small);
    } else {
      hipLaunchKernelGGL(k_tracker_loss, dim3(1), dim3(1024), 0, s, bi, n, t->w_color, t->handle_dynamic, t->use_color,
                         t->cam_tensor, t->best_out, lo);
    }
    PSL_LAUNCH_CHECK();
    rc = render_bwd_impl(ctx, &ra, &rg, s);
    ctx->track_fuse = nullptr;
    if (rc) return rc;
    if (!fused)
      hipLaunchKernelGGL(k_pose_step, dim3(1), dim3(256), 0, s, bi, n, t->cam_tensor, t->adam_state, t->step0 + it + 1,
                         t->lr_T, t->lr_quat);
    if (ex)
      hipLaunchKernelGGL(k_exposure_step, dim3(1), dim3(128), 0, s, ex->mlp, ex->feats, 1, ex_g, ex_aff, ex_act, ex->adam,
                         ex->adam + (EX_N + EXD), ex->step0 + it + 1, ex->lr_mlp, ex->lr_feat);
    PSL_LAUNCH_CHECK();
  }
  if (pose_in_knn) {      // the last pose step, into the caller's buffers (the pose of the last iteration may sit in the second pair)
    ProfScope ps(ctx, PROF_MISC, s);
    const int last = (t->n_iters - 1) & 1;
    hipLaunchKernelGGL(k_track_pre, dim3(1), dim3(1024), 0, s, 1, 0, (const float4*)rw.dp, (const float4*)rw.dp2,
                       ctx->cfg.near_end_surface, ctx->cfg.far_end_surface, pref(t->n_iters - 1), n, t->cam_tensor, t->adam_state,
                       t->step0 + t->n_iters, t->lr_T, t->lr_quat, t->cam, eh, t->cam.H - eh, ew, t->cam.W - ew, fdev, t->pix_idx,
                       adam_bias_host(t->n_iters), last ? (const float*)pose_buf[1] : (const float*)nullptr,
                       last ? (const float*)adam_buf[1] : (const float*)nullptr);
    PSL_LAUNCH_CHECK();
  } else if (fused && t->n_iters > 0) { track_pre(t->n_iters, 1, 0); PSL_LAUNCH_CHECK(); }   // the last pose step
  return PSL_OK;
}

// kNN prefetch: the mapper's rays do not depend on what it optimises (fixed poses, pre-drawn pixels), so the 8-NN
// lookups of a block of iterations are answered by ONE launch (~10^5 queries) instead of one small launch each.
static int map_knn_block(int n_rays) { return std::max(1, std::min(64, 262144 / std::max(n_rays, 1))); }
static int64_t map_prefetch_floats(int n_rays) {
  int64_t rays = (int64_t)map_knn_block(n_rays) * n_rays;
  return rays * S * (K + 1) + 64;                  // I [.][5][8], cnt [.][5]
}

extern "C" int64_t psl_map_ws_floats(int n_rays, int n_frames) {
  if (n_rays < 0 || n_frames < 0) return PSL_ERR_ARG;
  // two prefetch sets (rays + neighbour lists): block b+1 is looked up on a second stream while block b iterates
  return 2 * (map_prefetch_floats(n_rays) + rays_floats(map_knn_block(n_rays) * n_rays) + 16) + psl_render_ws_floats(n_rays, PSL_STAGE_COLOR | PSL_FEAT_GRAD | PSL_PARAM_GRAD) + 64 +
         (int64_t)((sizeof(FrameDev) * (size_t)std::max(n_frames, 1) + 3) / 4) + 32 + psl_param_master_floats();
}

extern "C" int psl_map_iters(psl_ctx* ctx, const psl_map_args* m, void* stream) {
  if (!ctx || !m || !m->frames || !m->pix_idx || !m->ws || !m->fallback || !m->geo_feats || !m->col_feats || !m->params ||
      !m->sel_rows || !m->row_map || !m->adam_geo || !m->adam_col || !m->adam_params || !m->g_geo || !m->g_col) {
    set_error("psl_map_iters: missing argument"); return PSL_ERR_ARG;
  }
  const int n = m->n_frames * m->pix_per_frame;
  if (n <= 0 || n > 65536) { set_error("psl_map_iters: n_frames*pix_per_frame must be in [1,65536]"); return PSL_ERR_ARG; }
  if (ctx->index_points != ctx->n_points) { set_error("psl_map_iters: index is stale"); return PSL_ERR_STATE; }
  const psl_exposure_args* ex = m->exposure;
  if (ex && (!ex->mlp || !ex->feats || !ex->adam)) { set_error("psl_map_iters: incomplete exposure block"); return PSL_ERR_ARG; }
  if (ex && m->n_frames > 64) { set_error("psl_map_iters: exposure supports windows of <= 64 frames"); return PSL_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  float* p = m->ws;
  // Block prefetch: the mapper's rays depend on nothing it optimises (fixed poses, pre-drawn pixels), so ray set-up,
  // the depth-outlier mask and the 8-NN lookups of `kblock` iterations are done by three launches per block
  // instead of three per iteration; iteration `it` then works on slice it % kblock of these buffers.
  const int kblock = map_knn_block(n);
  RayBufs pbs[2];
  pbs[0] = carve_rays(p, kblock * n);
  pbs[1] = carve_rays(p, kblock * n);
  bool any_rq = false;
  for (int f = 0; f < m->n_frames; ++f) any_rq |= m->frames[f].r_query != nullptr;
  if (!any_rq) { pbs[0].rq = nullptr; pbs[1].rq = nullptr; }
  auto slice = [&](int set, int j) {
    RayBufs b = pbs[set];
    const size_t o = (size_t)j * n;
    b.rays_o += 3 * o; b.rays_d += 3 * o; b.dirs += 3 * o; b.gd += o; b.gc += 3 * o; if (b.rq) b.rq += o;
    b.active += o; b.depth += o; b.var += o; b.rgb += 3 * o; b.valid += o; b.g_depth += o; b.g_rgb += 3 * o;
    b.g_o += 3 * o; b.g_d += 3 * o;
    return b;
  };
  p += 64;
  FrameDev* fdev = (FrameDev*)p; p += (sizeof(FrameDev) * m->n_frames + 3) / 4 + 4;
  float* g_params = p; p += (psl_param_master_floats() + 3) / 4 * 4;
  int *pre_Is[2], *pre_cnts[2];
  for (int q = 0; q < 2; ++q) {
    p = (float*)(((uintptr_t)p + 15) & ~(uintptr_t)15);   // neighbour lists are read as 16-byte vectors
    pre_Is[q] = (int*)p; p += (size_t)kblock * n * S * K;
    pre_cnts[q] = (int*)p; p += ((size_t)kblock * n * S + 3) / 4 * 4;
  }
  float* rws = (float*)(((uintptr_t)p + 15) & ~(uintptr_t)15);
  // the carve above must stay inside what psl_map_ws_floats() promises the caller
  if ((rws - m->ws) + psl_render_ws_floats(n, PSL_STAGE_COLOR | PSL_FEAT_GRAD | PSL_PARAM_GRAD) >
      psl_map_ws_floats(n, m->n_frames)) {
    set_error("psl_map_iters: internal workspace layout exceeds psl_map_ws_floats"); return PSL_ERR_STATE;
  }
  struct PreGuard { psl_ctx* c; bool ok = false;
                    ~PreGuard() { if (!ok && c->stream2) (void)hipStreamSynchronize(c->stream2);   // error path: no prefetch left behind
                                  c->pre_I = nullptr; c->pre_cnt = nullptr; c->fused_ray = false;
                                               c->touched_geo = c->touched_col = nullptr;
                                               c->adam_upto = c->adam_need = c->adam_list = c->adam_count = nullptr;
                                               c->dw_defer_reduce = false; } } pre_guard{ctx};
  ctx->fused_ray = true;
  ctx->dw_defer_reduce = g_dw_fused != 0;   // the chunk partials of the dW kernel are summed inside the Adam launch
  // host staging of this call's tables: a pinned slot whose previous upload (four calls ago) has certainly been consumed
  const bool staged = ctx->h_stage && (size_t)m->n_iters <= kStageTabIters && sizeof(FrameDev) * (size_t)m->n_frames <= kStageFrames;
  char* slot = nullptr;
  if (staged) {
    const int si = ctx->stage_next; ctx->stage_next = (si + 1) & 3;
    PSL_HIP(hipEventSynchronize(ctx->ev_stage[si]));     // returns at once unless the host is > 3 mapped frames ahead
    slot = ctx->h_stage + (size_t)si * kStageSlot;
  }
  std::vector<float4> tab_vec;
  if (m->n_sel > 0) {
    // lazy Adam of the feature rows (k_map_adam_lazy): upto_geo | upto_col | stamp | count[n_iters] | list | touched_geo |
    // touched_col
    const size_t ns = (size_t)m->n_sel, lcap = std::min<size_t>(2 * (size_t)n * S * K, ns);
    const size_t n_int = 3 * ns + (size_t)m->n_iters + lcap, need = n_int * sizeof(int) + 2 * ns;
    if (ctx->touched_cap < need) {
      if (ctx->touched) (void)hipFree(ctx->touched);
      ctx->touched = nullptr; ctx->touched_cap = 0;
      PSL_HIP(hipMalloc(&ctx->touched, need + need / 4)); ctx->touched_cap = need + need / 4;
      dbg_range("touched", ctx->touched, ctx->touched_cap);
    }
    PSL_HIP(hipMemsetAsync(ctx->touched, 0xFF, 2 * ns * sizeof(int), s));
    PSL_HIP(hipMemsetAsync(ctx->touched + 2 * ns * sizeof(int), 0, (ns + (size_t)m->n_iters) * sizeof(int), s));
    ctx->adam_upto = (int*)ctx->touched; ctx->adam_need = ctx->adam_upto + 2 * ns;
    ctx->adam_count = ctx->adam_need + ns; ctx->adam_list = ctx->adam_count + m->n_iters; ctx->adam_list_cap = (long long)lcap;
    ctx->touched_geo = ctx->touched + n_int * sizeof(int); ctx->touched_col = ctx->touched_geo + ns;
    PSL_HIP(hipMemsetAsync(ctx->touched_geo, 0, 2 * ns, s));
    if (ctx->adam_tab_cap < (size_t)m->n_iters) {
      if (ctx->adam_tab) (void)hipFree(ctx->adam_tab);
      ctx->adam_tab = nullptr; ctx->adam_tab_cap = 0;
      PSL_HIP(hipMalloc(&ctx->adam_tab, sizeof(float4) * ((size_t)m->n_iters + 64))); ctx->adam_tab_cap = (size_t)m->n_iters + 64;
      dbg_range("adam_tab", ctx->adam_tab, sizeof(float4) * ctx->adam_tab_cap);
    }
    float4* tab_host = staged ? (float4*)(slot + kStageFrames) : (tab_vec.resize(m->n_iters), tab_vec.data());
    for (int it = 0; it < m->n_iters; ++it) {   // the constants launch_map_adam would compute for iteration `it`
      const bool cs = it > m->n_geo_iters;
      float4 t = make_float4(0.f, 1.f, 0.f, 1.f);
      adam_consts(m->step0_geo + it + 1, cs ? m->lr_geo_color_stage : m->lr_geo_geo_stage, 0.9f, 0.999f, t.x, t.y);
      if (cs) adam_consts(m->step0_col + (it - m->n_geo_iters), m->lr_col, 0.9f, 0.999f, t.z, t.w);
      tab_host[it] = t;
    }
    PSL_HIP(hipMemcpyAsync(ctx->adam_tab, tab_host, sizeof(float4) * m->n_iters, hipMemcpyHostToDevice, s));
  }
  if (ctx->loss_acc_cap < m->n_iters) {
    if (ctx->loss_acc) (void)hipFree(ctx->loss_acc);
    PSL_HIP(hipMalloc(&ctx->loss_acc, sizeof(double) * 4 * kLossSlots * (size_t)m->n_iters)); psl::poison(ctx->loss_acc, sizeof(double) * 4 * kLossSlots * (size_t)m->n_iters);
    ctx->loss_acc_cap = m->n_iters;
    dbg_range("loss_acc", ctx->loss_acc, sizeof(double) * 4 * kLossSlots * (size_t)m->n_iters);
  }
  PSL_HIP(hipMemsetAsync(ctx->loss_acc, 0, sizeof(double) * 4 * kLossSlots * (size_t)m->n_iters, s));
  dbg_range("map.ws", m->ws, sizeof(float) * (size_t)psl_map_ws_floats(n, m->n_frames));
  dbg_range("map.pix_idx", m->pix_idx, sizeof(int) * (size_t)m->n_iters * n);
  dbg_range("map.fallback", m->fallback, sizeof(float) * (size_t)m->n_iters * 64);
  dbg_range("map.sel_rows", m->sel_rows, sizeof(int) * (size_t)m->n_sel);
  dbg_range("map.row_map", m->row_map, sizeof(int) * (size_t)ctx->n_points);
  dbg_range("map.g_geo", m->g_geo, sizeof(float) * (size_t)m->n_sel * C);
  dbg_range("map.g_col", m->g_col, sizeof(float) * (size_t)m->n_sel * C);
  dbg_range("map.adam_geo", m->adam_geo, sizeof(float) * 2 * (size_t)m->n_sel * C);
  dbg_range("map.adam_col", m->adam_col, sizeof(float) * 2 * (size_t)m->n_sel * C);
  dbg_range("map.adam_par", m->adam_params, sizeof(float) * 2 * (size_t)kColorFloats);
  dbg_range("map.geo_feats", m->geo_feats, sizeof(float) * (size_t)ctx->n_points * C);
  dbg_range("map.col_feats", m->col_feats, sizeof(float) * (size_t)ctx->n_points * C);
  dbg_range("map.params", m->params, sizeof(float) * (size_t)kMasterFloats);
  if (m->loss_out) dbg_range("map.loss_out", m->loss_out, sizeof(float) * 4 * (size_t)m->n_iters);
  for (int f = 0; f < m->n_frames; ++f) {
    dbg_range("map.depth", m->frames[f].depth, sizeof(float) * (size_t)m->cam.H * m->cam.W);
    dbg_range("map.color", m->frames[f].color, sizeof(float) * 3 * (size_t)m->cam.H * m->cam.W);
    if (m->frames[f].r_query) dbg_range("map.r_query", m->frames[f].r_query, sizeof(float) * (size_t)m->cam.H * m->cam.W);
  }
  {
    std::vector<FrameDev> fvec;
    FrameDev* fh = staged ? (FrameDev*)slot : (fvec.resize(m->n_frames), fvec.data());
    for (int f = 0; f < m->n_frames; ++f) {
      fh[f].depth = m->frames[f].depth; fh[f].color = m->frames[f].color; fh[f].r_query = m->frames[f].r_query;
      memcpy(fh[f].c2w, m->frames[f].c2w, sizeof(float) * 12);
    }
    PSL_HIP(hipMemcpyAsync(fdev, fh, sizeof(FrameDev) * m->n_frames, hipMemcpyHostToDevice, s));
    if (staged) PSL_HIP(hipEventRecord(ctx->ev_stage[(ctx->stage_next + 3) & 3], s));   // both uploads of this slot are enqueued
    else PSL_HIP(hipStreamSynchronize(s));   // oversized call: the tables live on this stack frame
  }
  psl_render_args ra{};
  memset(&ra, 0, sizeof(ra));
  ra.n_rays = n; ra.sigmoid_coef = m->sigmoid_coef;
  ra.geo_feats = m->geo_feats; ra.col_feats = m->col_feats; ra.params = m->params; ra.col_embed_B = m->col_embed_B;
  ra.ws = rws;
  psl_render_grads rg;
  memset(&rg, 0, sizeof(rg));
  rg.g_geo_feats = m->g_geo; rg.g_col_feats = m->g_col;
  rg.feat_row_map = m->row_map; rg.g_params = g_params;
  const int ncol = psl::kColorFloats;
  float *ex_aff = nullptr, *ex_act = nullptr, *ex_g = nullptr;
  if (ex) {   // per-frame affines of the window (Mapper.py:530-548); refreshed by k_exposure_step after every update
    ex_aff = ctx->d_expo; ex_act = ex_aff + 64 * EXO; ex_g = ex_act + 64 * EXH;
    PSL_HIP(hipMemsetAsync(ex_g, 0, sizeof(float) * 64 * EXO, s));
    hipLaunchKernelGGL(k_exposure_fwd, dim3(m->n_frames), dim3(128), 0, s, ex->mlp, ex->feats, ex_aff, ex_act);
    PSL_LAUNCH_CHECK();
  }
  // k-NN prefetch of block b+1 on a second, low-priority stream while block b iterates (the lookups depend on nothing
  // the iterations change): ~0.9 ms per 64 iterations leave the critical path.  PSL_KNN_OVERLAP=0 switches it off.
  const bool overlap = g_knn_overlap != 0 && m->n_iters > kblock;
  if (overlap && !ctx->stream2) {
    int least = 0, greatest = 0;
    PSL_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    PSL_HIP(hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, least));
    PSL_HIP(hipEventCreateWithFlags(&ctx->ev_knn_ready[0], hipEventDisableTiming));
    PSL_HIP(hipEventCreateWithFlags(&ctx->ev_knn_ready[1], hipEventDisableTiming));
    PSL_HIP(hipEventCreateWithFlags(&ctx->ev_knn_free, hipEventDisableTiming));
  }
  auto prefetch = [&](int it0, int set, hipStream_t st) -> int {
    const int nb = std::min(kblock, m->n_iters - it0);
    { ProfScope ps(ctx, PROF_MISC, st);
      hipLaunchKernelGGL(k_ray_setup, dim3((nb * n + 255) / 256), dim3(256), 0, st, m->cam, 0, m->cam.H, 0, m->cam.W,
                         fdev, m->n_frames, m->pix_per_frame, m->pix_idx + (size_t)it0 * n, (const float*)nullptr, pbs[set],
                         nb);
      hipLaunchKernelGGL(k_depth_inlier, dim3(nb), dim3(1024), 0, st, pbs[set].gd, pbs[set].active, n);
      PSL_LAUNCH_CHECK(); }
    ProfScope ps(ctx, st == s ? PROF_KNN_PREFETCH : PROF_KNN_SIDE, st, 108.0 * nb * n * S, true);
    // on the side stream the lookup is THROTTLED (g_knn_side_blocks workgroups, ~2 per CU): left alone its 10^4
    // workgroups fill every CU and the decode kernels of the main stream wait for slots (one of them measured at
    // 845 us instead of 52); it has a whole block of iterations (~8 ms) to finish
    return knn_rays(ctx, pbs[set].rays_o, pbs[set].rays_d, pbs[set].gd, nullptr, pbs[set].rq, nb * n, pre_Is[set],
                    pre_cnts[set], st, st == s ? 0 : g_knn_side_blocks);
  };
  for (int it = 0; it < m->n_iters; ++it) {
    // stage switch (Mapper.py:420-423): joint_iter <= n_geo_iters -> geometry
    const bool color_stage = it > m->n_geo_iters;
    if (it % kblock == 0) {
      const int set = (it / kblock) & 1;
      if (it == 0 || !overlap) { int rc = prefetch(it, set, s); if (rc) return rc; }
      else PSL_HIP(hipStreamWaitEvent(s, ctx->ev_knn_ready[set], 0));       // looked up while the previous block iterated
      // the block after this one: on the second stream, into the other set, once the iterations that read it are over
      if (overlap && it + kblock < m->n_iters) {
        PSL_HIP(hipEventRecord(ctx->ev_knn_free, s));                       // everything enqueued so far used set^1 last
        PSL_HIP(hipStreamWaitEvent(ctx->stream2, ctx->ev_knn_free, 0));
        int rc = prefetch(it + kblock, set ^ 1, ctx->stream2);
        if (rc) return rc;
        PSL_HIP(hipEventRecord(ctx->ev_knn_ready[set ^ 1], ctx->stream2));
      }
    }
    const int cur = (it / kblock) & 1;
    ctx->pre_I = pre_Is[cur] + (size_t)(it % kblock) * n * S * K;
    ctx->pre_cnt = pre_cnts[cur] + (size_t)(it % kblock) * n * S;
    const RayBufs b = slice(cur, it % kblock);
    ra.rays_o = b.rays_o; ra.rays_d = b.rays_d; ra.gt_depth = b.gd; ra.r_query = b.rq;
    ra.depth = b.depth; ra.var = b.var; ra.rgb = b.rgb; ra.valid_ray = b.valid;
    ra.flags = PSL_FEAT_GRAD | (color_stage ? (PSL_STAGE_COLOR | (m->train_decoder ? PSL_PARAM_GRAD : 0)) : 0);
    if (ex && color_stage) ra.flags |= PSL_NO_SIGMOID;   // raw logits; affine + sigmoid per frame slice in the ray kernel
    ra.fallback_geo = m->fallback + (size_t)it * 64;
    ra.fallback_col = m->fallback + (size_t)it * 64 + 32;
    // the forward weights only change after the decoder was stepped (colour stage with train_decoder)
    const bool repack = it == 0;   // afterwards the fused Adam kernel keeps the forward-layout copy in step
    // lazy Adam: rows the next iteration reads must be up to date; its lists exist unless a prefetch block ends here
    const bool lazy = ctx->adam_upto != nullptr && g_lazy_adam != 0;
    const bool dense = !lazy || it + 1 == m->n_iters || (it + 1) % kblock == 0;
    AdamWorklist wl{};
    if (lazy && !dense)   // distinct rows of this iteration's and the next iteration's neighbour lists
      wl = AdamWorklist{ctx->pre_I, ctx->pre_I + (size_t)n * S * K, n * S * K / 4, m->row_map, ctx->adam_need, it + 1,
                        ctx->adam_list, ctx->adam_count + it};
    int rc;
    // one launch per geometry-stage iteration in the latency regime only: at 25 000 samples (Replica yaml) the one-wave
    // tiles of k_geo_iter take 81 us against 35 + 8 + 21 us for the three throughput-shaped launches (PSL_GEO_FUSED=2 forces it)
    const bool geo_one_launch = !color_stage && (g_geo_fused == 2 || (g_geo_fused == 1 && n * S <= kGeoIterMaxSamples));
    if (geo_one_launch) {
      // stage 'geometry': decode forward, compositing, loss, compositing backward and decode backward of whole ray triples
      // per wavefront, ONE launch (+ the work list of this iteration's Adam as extra workgroups)
      rc = geo_iter_impl(ctx, &ra, &rg, b.active, ctx->loss_acc + 4 * kLossSlots * (size_t)it, &wl, s, repack);
      if (rc) return rc;
    } else {
    // colour stage without per-frame exposure: the ray stage (compositing, loss, compositing backward) and the work list run
    // inside the decode backward; the forward clears the backward's accumulators, which the ray kernel used to do
    const bool ray_in_bwd = color_stage && !ex && g_ray_in_bwd != 0;
    ctx->fwd_zero64 = ray_in_bwd ? ctx->d_small : nullptr;
    rc = render_fwd_impl(ctx, &ra, s, repack);
    ctx->fwd_zero64 = nullptr;
    if (rc) return rc;
    RayFuse rf{};
    if (ray_in_bwd) {
      rf = RayFuse{b.active, b.gc, m->sigmoid_coef, m->w_color, n, b.depth, b.var, b.rgb, b.valid,
                   ctx->loss_acc + 4 * kLossSlots * (size_t)it, 1};
      ctx->ray_fuse = &rf; ctx->ray_wl = &wl;
    } else { // compositing + mapper loss + compositing backward in one launch (+ the work list of this iteration's Adam)
      ProfScope ps(ctx, PROF_COMPOSITE, s, 324.0 * n, true);
      const RenderWs rw = carve_ws(ra.ws, n, ra.flags | (ctx->cfg.encode_rel_pos ? 0x10000 : 0));
      rc = launch_map_ray_fused((const float4*)rw.raw, ctx->pre_cnt, b.gd, b.gc, b.active, ctx->cfg.near_end_surface,
                                ctx->cfg.far_end_surface, ctx->cfg.min_nn_num, n, m->sigmoid_coef, m->w_color,
                                color_stage ? 1 : 0, b.depth, b.var, b.rgb, b.valid, (float4*)rw.d_raw,
                                ctx->loss_acc + 4 * kLossSlots * (size_t)it, ctx->d_small, (ex && color_stage) ? ex_aff : nullptr,
                                m->pix_per_frame, ex_g, s, &wl);
      if (rc) return rc;
    }
    rc = render_bwd_impl(ctx, &ra, &rg, s);
    ctx->ray_fuse = nullptr; ctx->ray_wl = nullptr;
    if (rc) return rc;
    }
    // Adam (Mapper.py:394-402,425-439,556): geometry features every iteration; colour features and the colour
    // decoder only once they have received a gradient (colour stage) -- torch skips params whose .grad is None.
    const float lr_geo = color_stage ? m->lr_geo_color_stage : m->lr_geo_geo_stage;
    // dense Adam: 5 streams (p r/w, g r/w(zero), m r/w, v r/w ~ 7 accesses of 4 B; counted as 5 x 4 B per element as in
    // SURVEY.md §8d) over every selected row
    // (lazy path: the rows actually stepped are counted on the device and added by psl_profile_read)
    ProfScope psa(ctx, dense ? PROF_ADAM_DENSE : PROF_ADAM, s, 20.0 * ((lazy ? 0.0 : (double)m->n_sel * C * (color_stage ? 2 : 1)) +
                                             ((color_stage && m->train_decoder) ? (double)ncol : 0.0)), true);
    {
      AdamRowsSeg sg{}, sc{};
      AdamParSeg sp{};
      sg.feats = (float*)m->geo_feats; sg.rows = m->sel_rows; sg.g = (float4*)m->g_geo; sg.m = (float4*)m->adam_geo;
      sg.v = (float4*)(m->adam_geo + (size_t)m->n_sel * C); sg.n_rows = m->n_sel; sg.touched = ctx->touched_geo;
      sg.upto = ctx->adam_upto;
      int st = 1;
      if (color_stage) {
        st = m->step0_col + (it - m->n_geo_iters);
        sc.feats = (float*)m->col_feats; sc.rows = m->sel_rows; sc.g = (float4*)m->g_col; sc.m = (float4*)m->adam_col;
        sc.v = (float4*)(m->adam_col + (size_t)m->n_sel * C); sc.n_rows = m->n_sel; sc.touched = ctx->touched_col;
        sc.upto = ctx->adam_upto ? ctx->adam_upto + m->n_sel : nullptr;
        if (m->train_decoder) {
          sp.p = (float*)m->params; sp.g = g_params; sp.m = m->adam_params; sp.v = m->adam_params + ncol; sp.n = ncol;
          sp.wf_index = ctx->wf_index; sp.wf = ctx->wf; sp.wb_index = ctx->wb_index; sp.wb = ctx->wb;
          if (ctx->dw_defer_reduce) { sp.slabs = ctx->dw_slabs; sp.g_brel = ctx->d_small; sp.ra = ctx->dw_ra; }
        }
      }
      AdamLazy lz{lazy ? ctx->adam_tab : nullptr, (lazy && !dense) ? ctx->adam_list : nullptr,
                  lazy ? ctx->adam_count + it : nullptr, ctx->adam_list_cap, it, ctx->adam_rows + (dense ? kAdamRowSlots : 0), it - it % kblock};
      rc = launch_map_adam(sg, m->step0_geo + it + 1, lr_geo, sc, st, m->lr_col, sp, m->lr_decoder, s,
                           st + m->step0_params, lz);
      if (rc) return rc;
      if (ex && color_stage) {   // mlp_exposure is part of color_decoder.parameters() (decoders_lr); latent lr 0.001
        hipLaunchKernelGGL(k_exposure_step, dim3(1), dim3(128), 0, s, ex->mlp, ex->feats, m->n_frames, ex_g, ex_aff, ex_act,
                           ex->adam, ex->adam + (EX_N + EXD), ex->step0 + (it - m->n_geo_iters),
                           m->train_decoder ? ex->lr_mlp : 0.f, ex->lr_feat);
        PSL_LAUNCH_CHECK();
      }
    }
  }
  if (m->loss_out) {
    hipLaunchKernelGGL(k_map_loss_finalize, dim3((m->n_iters + 255) / 256), dim3(256), 0, s, ctx->loss_acc, m->n_iters,
                       m->n_geo_iters, m->w_color, m->loss_out);
    PSL_LAUNCH_CHECK();
  }
  pre_guard.ok = true;
  return PSL_OK;
}

extern "C" int psl_frustum_select_sync(psl_ctx* ctx, const float* c2w_host /*[16] row-major 4x4*/, psl_cam_intr cam,
                                       const float* depth, float depth_max, float edge, int32_t* sel_out,
                                       int32_t* row_map_out, int* n_sel_host, void* stream) {
  if (!ctx || !c2w_host || !depth || !sel_out || !row_map_out || !n_sel_host) { set_error("psl_frustum_select_sync: bad argument"); return PSL_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  const int n = ctx->n_points;
  *n_sel_host = 0;
  if (n == 0) return PSL_OK;
  // w2c = inverse of the rigid c2w in double
  double R[3][3], T[3];
  for (int a = 0; a < 3; ++a) { for (int k = 0; k < 3; ++k) R[a][k] = c2w_host[a * 4 + k]; T[a] = c2w_host[a * 4 + 3]; }
  // general 3x3 inverse (the pose is rigid up to rounding; np.linalg.inv in the reference)
  double det = R[0][0] * (R[1][1] * R[2][2] - R[1][2] * R[2][1]) - R[0][1] * (R[1][0] * R[2][2] - R[1][2] * R[2][0]) +
               R[0][2] * (R[1][0] * R[2][1] - R[1][1] * R[2][0]);
  double inv[3][3];
  inv[0][0] = (R[1][1] * R[2][2] - R[1][2] * R[2][1]) / det; inv[0][1] = (R[0][2] * R[2][1] - R[0][1] * R[2][2]) / det;
  inv[0][2] = (R[0][1] * R[1][2] - R[0][2] * R[1][1]) / det; inv[1][0] = (R[1][2] * R[2][0] - R[1][0] * R[2][2]) / det;
  inv[1][1] = (R[0][0] * R[2][2] - R[0][2] * R[2][0]) / det; inv[1][2] = (R[0][2] * R[1][0] - R[0][0] * R[1][2]) / det;
  inv[2][0] = (R[1][0] * R[2][1] - R[1][1] * R[2][0]) / det; inv[2][1] = (R[0][1] * R[2][0] - R[0][0] * R[2][1]) / det;
  inv[2][2] = (R[0][0] * R[1][1] - R[0][1] * R[1][0]) / det;
  float w2c[12];
  for (int a = 0; a < 3; ++a) {
    for (int k = 0; k < 3; ++k) w2c[a * 4 + k] = (float)inv[a][k];
    w2c[a * 4 + 3] = (float)(-(inv[a][0] * T[0] + inv[a][1] * T[1] + inv[a][2] * T[2]));
  }
  // scratch: reuse cell_of (int[max_points]) for flags, scan_tmp-like block sums in scan_flags
  const int nblk = (n + 1023) / 1024;
  if (ctx->scan_flags_cap < nblk + 64) {
    if (ctx->scan_flags) (void)hipFree(ctx->scan_flags);
    PSL_HIP(hipMalloc(&ctx->scan_flags, sizeof(int) * (size_t)(nblk + 64) * 4)); psl::poison(ctx->scan_flags, sizeof(int) * (size_t)(nblk + 64) * 4);
    ctx->scan_flags_cap = (nblk + 64) * 4;
  }
  float* w2c_dev = ctx->d_small;   // 12 floats
  PSL_HIP(hipMemcpyAsync(w2c_dev, w2c, sizeof(w2c), hipMemcpyHostToDevice, s));
  int* flags = ctx->cell_of;   // free between index builds
  unsigned* dmax_bits = nullptr;
  if (depth_max < 0.f) {       // the reference's rule: maximum over the per-point lookups, taken on the device
    dmax_bits = (unsigned*)(ctx->d_counter + 1);
    PSL_HIP(hipMemsetAsync(dmax_bits, 0, sizeof(unsigned), s));
    if (g_remap_cv2) hipLaunchKernelGGL(k_frustum_dmax<true>, dim3((n + 255) / 256), dim3(256), 0, s, ctx->pos, n, w2c_dev, cam, depth, dmax_bits);
    else hipLaunchKernelGGL(k_frustum_dmax<false>, dim3((n + 255) / 256), dim3(256), 0, s, ctx->pos, n, w2c_dev, cam, depth, dmax_bits);
  }
  if (g_remap_cv2) hipLaunchKernelGGL(k_frustum_flags<true>, dim3((n + 255) / 256), dim3(256), 0, s, ctx->pos, n, w2c_dev, cam, depth,
                                      depth_max, dmax_bits, edge, flags);
  else hipLaunchKernelGGL(k_frustum_flags<false>, dim3((n + 255) / 256), dim3(256), 0, s, ctx->pos, n, w2c_dev, cam, depth,
                          depth_max, dmax_bits, edge, flags);
  hipLaunchKernelGGL(k_flag_block_sums, dim3(nblk), dim3(256), 0, s, flags, n, ctx->scan_flags);
  hipLaunchKernelGGL(k_flag_scan_top, dim3(1), dim3(1024), 0, s, ctx->scan_flags, nblk, ctx->d_counter);
  hipLaunchKernelGGL(k_flag_compact, dim3(nblk), dim3(256), 0, s, flags, n, ctx->scan_flags, sel_out, row_map_out);
  PSL_LAUNCH_CHECK();
  int tot = 0;
  PSL_HIP(hipMemcpyAsync(&tot, ctx->d_counter, sizeof(int), hipMemcpyDeviceToHost, s));
  PSL_HIP(hipStreamSynchronize(s));
  *n_sel_host = tot;
  return PSL_OK;
}
