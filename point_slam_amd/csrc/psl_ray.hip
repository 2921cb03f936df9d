// Per-ray kernels: occupancy alpha-compositing (forward + backward), ray-gradient assembly, Adam.
//
// raw2outputs_nerf_color (src/common.py:298-336): alpha_s = sigmoid(coef*occ_s),
// T_s = prod_{j<s}(1 - alpha_j + 1e-10), w = alpha*T, W = sum(w)+1e-10, rgb = sum(w c)/W,
// depth = sum(w z)/W, var = sum(w (z-depth)^2).  S = 5 samples fit in registers: one lane per ray.
#include "psl_common.h"
#include "psl_device.h"
#include "psl_adam.h"

namespace psl {

__global__ __launch_bounds__(256) void k_composite_fwd(const float4* __restrict__ raw, const float* __restrict__ z_in,
                                                       const float* __restrict__ gt_depth, float near_s, float far_s,
                                                       const int* __restrict__ cnt, int min_nn, int n_rays, float coef,
                                                       float* __restrict__ depth, float* __restrict__ var,
                                                       float* __restrict__ rgb, unsigned char* __restrict__ valid,
                                                       float* __restrict__ cw, float* __restrict__ ray_aux) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  float w[S], z[S], c0[S], c1[S], c2[S];
  float T = 1.0f, wsum = 0.f;
  int nhas = 0;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    float4 q = raw[r * S + s];
    z[s] = z_in ? z_in[r * S + s] : sample_z(gt_depth[r], s, near_s, far_s);
    float alpha = sigmoidf(coef * q.w);
    w[s] = alpha * T;
    T = T * (1.0f - alpha + 1e-10f);
    wsum += w[s];
    c0[s] = q.x; c1[s] = q.y; c2[s] = q.z;
    if (cnt) nhas += (cnt[r * S + s] >= min_nn) ? 1 : 0;
  }
  float W = wsum + 1e-10f;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, ad = 0.f;
#pragma unroll
  for (int s = 0; s < S; ++s) { a0 += w[s] * c0[s]; a1 += w[s] * c1[s]; a2 += w[s] * c2[s]; ad += w[s] * z[s]; }
  float d = ad / W;
  float v = 0.f;
#pragma unroll
  for (int s = 0; s < S; ++s) { float tmp = z[s] - d; v += w[s] * tmp * tmp; }
  depth[r] = d; var[r] = v;
  rgb[r * 3 + 0] = a0 / W; rgb[r * 3 + 1] = a1 / W; rgb[r * 3 + 2] = a2 / W;
  // valid ray: at least int(S/2+1) = 3 samples with >= min_nn neighbours (decoder.py:200-201)
  if (valid) valid[r] = nhas >= (S / 2 + 1) ? 1 : 0;
  if (cw) {
#pragma unroll
    for (int s = 0; s < S; ++s) cw[r * S + s] = w[s];
  }
  if (ray_aux) { ray_aux[r * 4 + 0] = d; ray_aux[r * 4 + 1] = W; ray_aux[r * 4 + 2] = v; ray_aux[r * 4 + 3] = 0.f; }
}

// d(raw) from d(depth), d(var), d(rgb).  The -100 written into masked samples (Renderer.py:189-190) replaces the
// VALUE only; autograd still routes d/d(occ) to the decoder output (in-place write under no_grad), so d_raw.w is
// produced for masked samples as well.
__global__ __launch_bounds__(256) void k_composite_bwd(const float4* __restrict__ raw, const float* __restrict__ z_in,
                                                       const float* __restrict__ gt_depth, float near_s, float far_s, int n_rays, float coef,
                                                       const float* __restrict__ g_depth, const float* __restrict__ g_var,
                                                       const float* __restrict__ g_rgb, float4* __restrict__ d_raw,
                                                       float* __restrict__ zero64) {
  // the decode backward that follows accumulates dB_rel / d(affine) into 64 floats: cleared here, not by a memset
  if (zero64 && blockIdx.x == 0 && threadIdx.x < 64) zero64[threadIdx.x] = 0.f;
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  float w[S], z[S], al[S], Tt[S], c0[S], c1[S], c2[S];
  float T = 1.0f, wsum = 0.f;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    float4 q = raw[r * S + s];
    z[s] = z_in ? z_in[r * S + s] : sample_z(gt_depth[r], s, near_s, far_s);
    al[s] = sigmoidf(coef * q.w);
    Tt[s] = T;
    w[s] = al[s] * T;
    T = T * (1.0f - al[s] + 1e-10f);
    wsum += w[s];
    c0[s] = q.x; c1[s] = q.y; c2[s] = q.z;
  }
  float W = wsum + 1e-10f;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, ad = 0.f;
#pragma unroll
  for (int s = 0; s < S; ++s) { a0 += w[s] * c0[s]; a1 += w[s] * c1[s]; a2 += w[s] * c2[s]; ad += w[s] * z[s]; }
  float d = ad / W, m0 = a0 / W, m1 = a1 / W, m2 = a2 / W;
  float gd = g_depth ? g_depth[r] : 0.f, gv = g_var ? g_var[r] : 0.f;
  float gr0 = g_rgb ? g_rgb[r * 3] : 0.f, gr1 = g_rgb ? g_rgb[r * 3 + 1] : 0.f, gr2 = g_rgb ? g_rgb[r * 3 + 2] : 0.f;
  // var = sum w (z-d)^2 : d var / d depth = -2 sum w (z - d)
  float dvd = 0.f;
#pragma unroll
  for (int s = 0; s < S; ++s) dvd += w[s] * (z[s] - d);
  float gdt = gd + gv * (-2.0f * dvd);
  float gw[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    float dz = z[s] - d;
    gw[s] = gv * dz * dz + (gdt * dz + gr0 * (c0[s] - m0) + gr1 * (c1[s] - m1) + gr2 * (c2[s] - m2)) / W;
  }
  // w_s = alpha_s T_s, T_s = prod_{j<s} (1 - alpha_j + 1e-10)
  float suffix = 0.f;  // sum_{t>s} gw_t w_t
#pragma unroll
  for (int s = S - 1; s >= 0; --s) {
    float ga = gw[s] * Tt[s] - suffix / (1.0f - al[s] + 1e-10f);
    float gocc = ga * coef * al[s] * (1.0f - al[s]);
    float ws = w[s] / W;
    d_raw[r * S + s] = make_float4(gr0 * ws, gr1 * ws, gr2 * ws, gocc);
    suffix += gw[s] * w[s];
  }
}

// g_rays_o = sum_s dp_s ; g_rays_d = sum_s z_s dp_s  (pts = o + d*z, Renderer.py:172-173)
__global__ __launch_bounds__(256) void k_ray_grad(const float4* __restrict__ dp, const float4* __restrict__ dp2, const float* __restrict__ z_in,
                                                  const float* __restrict__ gt_depth, float near_s, float far_s, int n_rays, float* __restrict__ g_o,
                                                  float* __restrict__ g_d) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  float o0 = 0.f, o1 = 0.f, o2 = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    float4 g = dp[r * S + s];
    if (dp2) { const float4 g2 = dp2[r * S + s]; g.x += g2.x; g.y += g2.y; g.z += g2.z; }   // colour + geometry branch
    float z = z_in ? z_in[r * S + s] : sample_z(gt_depth[r], s, near_s, far_s);
    o0 += g.x; o1 += g.y; o2 += g.z;
    d0 += z * g.x; d1 += z * g.y; d2 += z * g.z;
  }
  if (g_o) { g_o[r * 3] = o0; g_o[r * 3 + 1] = o1; g_o[r * 3 + 2] = o2; }
  if (g_d) { g_d[r * 3] = d0; g_d[r * 3 + 1] = d1; g_d[r * 3 + 2] = d2; }
}

// Work list of a lazy Adam step (k_map_adam_lazy): the distinct selected rows named by the neighbour lists of this
// iteration (I_a) and of the next one (I_b, may be null).  `stamp_arr[row] == stamp` marks rows already listed (stamps
// grow with the iteration, the array is cleared once per call); appends are aggregated per wavefront.  Runs as extra
// workgroups of k_map_ray_fused: no launch of its own, and the four entries of a thread are in flight together.
__device__ __forceinline__ void adam_worklist_role(const AdamWorklist& wl, int i) {
  const int lane = threadIdx.x & 63;
  int4 v = make_int4(-1, -1, -1, -1);
  if (i < wl.n4) v = reinterpret_cast<const int4*>(wl.I_a)[i];
  else if (wl.I_b && i < 2 * wl.n4) v = reinterpret_cast<const int4*>(wl.I_b)[i - wl.n4];
  const int ent[4] = {v.x, v.y, v.z, v.w};
  int r[4];
  bool fresh[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) r[c] = (ent[c] >= 0) ? wl.row_map[ent[c]] : -1;
#pragma unroll
  for (int c = 0; c < 4; ++c) fresh[c] = (r[c] >= 0) ? (atomicExch(&wl.stamp_arr[r[c]], wl.stamp) != wl.stamp) : false;
  unsigned long long mask[4];
  int tot = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) { mask[c] = __ballot(fresh[c]); tot += __popcll(mask[c]); }
  if (tot == 0) return;
  int base = 0;
  if (lane == 0) base = atomicAdd(wl.count, tot);
  base = __shfl(base, 0);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (fresh[c]) wl.list[base + __popcll(mask[c] & ((1ull << lane) - 1ull))] = r[c];
    base += __popcll(mask[c]);
  }
}

// Mapper iteration, everything between the two decode kernels in ONE launch: compositing (common.py:298-336), the
// mapper loss with its mask (Mapper.py:524-553: L1 sums, so every ray's cotangent is local) and the compositing
// backward.  Replaces k_composite_fwd + k_mapper_loss + k_composite_bwd (three ~6 us launches per iteration).
// loss_acc[slot][0..2] += (sum |d_gt - d|, sum |c_gt - c|, #rays in the mask) in double, slot = workgroup & (kLossSlots - 1).
__global__ __launch_bounds__(256) void k_map_ray_fused(const float4* __restrict__ raw, const int* __restrict__ cnt,
                                                       const float* __restrict__ gt_depth, const float* __restrict__ gt_color,
                                                       const int* __restrict__ active, float near_s, float far_s,
                                                       int min_nn, int n_rays, float coef, float w_color, int color_stage,
                                                       float* __restrict__ depth, float* __restrict__ var,
                                                       float* __restrict__ rgb, unsigned char* __restrict__ valid,
                                                       float4* __restrict__ d_raw, double* __restrict__ loss_acc,
                                                       float* __restrict__ zero64, const float* __restrict__ frame_affine,
                                                       int pix_per_frame, float* __restrict__ g_frame_affine, AdamWorklist wl,
                                                       int nb_ray) {
  __builtin_amdgcn_s_setprio(1);      // above the side-stream k-NN prefetch (see k_decode_fwd2)
  if ((int)blockIdx.x >= nb_ray) { adam_worklist_role(wl, ((int)blockIdx.x - nb_ray) * (int)blockDim.x + (int)threadIdx.x); return; }
  // frame_affine != null (ScanNet, colour stage): the decoder returned raw colour logits; the affine of the ray's window
  // frame (slot r / pix_per_frame) and the sigmoid are applied to the COMPOSITED logits here (Mapper.py:530-548), and
  // d(loss)/d(affine) is accumulated per frame into g_frame_affine [F][12].
  __shared__ double red[3][4];
  if (zero64 && blockIdx.x == 0 && threadIdx.x < 64) zero64[threadIdx.x] = 0.f;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  double lg = 0.0, lc = 0.0, lcnt = 0.0;
  float ga[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) ga[j] = 0.f;
  if (r < n_rays) {
    float w[S], z[S], al[S], Tt[S], c0[S], c1[S], c2[S];
    float T = 1.0f, wsum = 0.f;
    int nhas = 0;
    const float gt = gt_depth[r];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      float4 q = raw[r * S + s];
      z[s] = sample_z(gt, s, near_s, far_s);
      al[s] = sigmoidf(coef * q.w);
      Tt[s] = T;
      w[s] = al[s] * T;
      T = T * (1.0f - al[s] + 1e-10f);
      wsum += w[s];
      c0[s] = q.x; c1[s] = q.y; c2[s] = q.z;
      nhas += (cnt[r * S + s] >= min_nn) ? 1 : 0;
    }
    const float W = wsum + 1e-10f;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, ad = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) { a0 += w[s] * c0[s]; a1 += w[s] * c1[s]; a2 += w[s] * c2[s]; ad += w[s] * z[s]; }
    const float d = ad / W, m0 = a0 / W, m1 = a1 / W, m2 = a2 / W;
    float v = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) { float tmp = z[s] - d; v += w[s] * tmp * tmp; }
    const bool vr = nhas >= (S / 2 + 1);
    // colour as the loss sees it: composited value, or sigmoid(composited logits @ A_f + t_f) with per-frame exposure
    float e0 = m0, e1 = m1, e2 = m2;
    const float* A = nullptr;
    if (frame_affine && color_stage) {
      A = frame_affine + 12 * (r / pix_per_frame);
      e0 = sigmoidf(m0 * A[0] + m1 * A[3] + m2 * A[6] + A[9]);
      e1 = sigmoidf(m0 * A[1] + m1 * A[4] + m2 * A[7] + A[10]);
      e2 = sigmoidf(m0 * A[2] + m1 * A[5] + m2 * A[8] + A[11]);
    }
    depth[r] = d; var[r] = v; rgb[r * 3] = e0; rgb[r * 3 + 1] = e1; rgb[r * 3 + 2] = e2; valid[r] = vr ? 1 : 0;
    // loss + cotangents
    float gd = 0.f, gr0 = 0.f, gr1 = 0.f, gr2 = 0.f;
    if (active[r] && gt > 0.f && vr && d == d) {
      lg = (double)fabsf(gt - d);
      gd = (d > gt) ? 1.f : ((d < gt) ? -1.f : 0.f);
      lcnt = 1.0;
      if (color_stage) {
        const float g0 = gt_color[r * 3], g1 = gt_color[r * 3 + 1], g2 = gt_color[r * 3 + 2];
        lc = (double)fabsf(g0 - e0) + (double)fabsf(g1 - e1) + (double)fabsf(g2 - e2);
        gr0 = w_color * ((e0 > g0) ? 1.f : ((e0 < g0) ? -1.f : 0.f));
        gr1 = w_color * ((e1 > g1) ? 1.f : ((e1 < g1) ? -1.f : 0.f));
        gr2 = w_color * ((e2 > g2) ? 1.f : ((e2 < g2) ? -1.f : 0.f));
      }
    }
    if (A) {
      // through the sigmoid, then out' = out @ A + t: dA[i][j] = out_i d_j, dt_j = d_j, d out_i = sum_j A[i][j] d_j
      const float q0 = gr0 * e0 * (1.f - e0), q1 = gr1 * e1 * (1.f - e1), q2 = gr2 * e2 * (1.f - e2);
      ga[0] = m0 * q0; ga[1] = m0 * q1; ga[2] = m0 * q2; ga[3] = m1 * q0; ga[4] = m1 * q1; ga[5] = m1 * q2;
      ga[6] = m2 * q0; ga[7] = m2 * q1; ga[8] = m2 * q2; ga[9] = q0; ga[10] = q1; ga[11] = q2;
      gr0 = A[0] * q0 + A[1] * q1 + A[2] * q2;
      gr1 = A[3] * q0 + A[4] * q1 + A[5] * q2;
      gr2 = A[6] * q0 + A[7] * q1 + A[8] * q2;
    }
    // compositing backward (no variance cotangent in the mapper loss)
    float gw[S];
#pragma unroll
    for (int s = 0; s < S; ++s)
      gw[s] = (gd * (z[s] - d) + gr0 * (c0[s] - m0) + gr1 * (c1[s] - m1) + gr2 * (c2[s] - m2)) / W;
    float suffix = 0.f;
#pragma unroll
    for (int s = S - 1; s >= 0; --s) {
      float ga = gw[s] * Tt[s] - suffix / (1.0f - al[s] + 1e-10f);
      float gocc = ga * coef * al[s] * (1.0f - al[s]);
      float ws = w[s] / W;
      d_raw[r * S + s] = make_float4(gr0 * ws, gr1 * ws, gr2 * ws, gocc);
      suffix += gw[s] * w[s];
    }
  }
  if (frame_affine && color_stage) {
    // d(loss)/d(affine of the ray's frame): a wavefront's 64 rays almost always belong to ONE window frame, so the twelve
    // sums are reduced across the wave and leave as 12 atomics; a wave that straddles two frames falls back to per-lane
    // atomics (one atomic per ray and entry would serialise thousands of updates on F x 12 addresses)
    const int fr = min(r, n_rays - 1) / pix_per_frame;
    const int fr0 = __builtin_amdgcn_readfirstlane(fr);
    if (__ballot(fr != fr0) == 0ull) {
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const float v = wave_sum(ga[j]);
        if ((threadIdx.x & 63) == 0 && v != 0.f) atomic_add_f32(g_frame_affine + 12 * fr0 + j, v);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 12; ++j) if (ga[j] != 0.f) atomic_add_f32(g_frame_affine + 12 * fr + j, ga[j]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lg += __shfl_xor(lg, o); lc += __shfl_xor(lc, o); lcnt += __shfl_xor(lcnt, o); }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wv] = lg; red[1][wv] = lc; red[2][wv] = lcnt; }
  __syncthreads();
  if (threadIdx.x < 3) {
    double t = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    if (t != 0.0) atomicAdd(&loss_acc[4 * (blockIdx.x & (kLossSlots - 1)) + threadIdx.x], t);
  }
}

int launch_map_ray_fused(const float4* raw, const int* cnt, const float* gt_depth, const float* gt_color, const int* active,
                         float near_s, float far_s, int min_nn, int n_rays, float coef, float w_color, int color_stage,
                         float* depth, float* var, float* rgb, unsigned char* valid, float4* d_raw, double* loss_acc,
                         float* zero64, const float* frame_affine, int pix_per_frame, float* g_frame_affine, hipStream_t s,
                         const AdamWorklist* wl) {
  if (n_rays <= 0) return PSL_OK;
  AdamWorklist w{};
  if (wl) w = *wl;
  const int nb_ray = (n_rays + 255) / 256;
  const int nb_wl = (w.I_a && w.n4 > 0) ? ((w.I_b ? 2 : 1) * w.n4 + 255) / 256 : 0;
  PSL_KLAUNCH(k_map_ray_fused, dim3(nb_ray + nb_wl), dim3(256), 0, s, raw, cnt, gt_depth, gt_color, active,
                     near_s, far_s, min_nn, n_rays, coef, w_color, color_stage, depth, var, rgb, valid, d_raw, loss_acc,
                     zero64, frame_affine, pix_per_frame, g_frame_affine, w, nb_ray);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}

int launch_composite_fwd(const float4* raw, const float* z, const float* gt_depth, float near_s, float far_s,
                         const int* cnt, int min_nn, int n_rays, float coef, float* depth, float* var, float* rgb,
                         unsigned char* valid, float* cw, float* ray_aux, hipStream_t s) {
  if (n_rays <= 0) return PSL_OK;
  PSL_KLAUNCH(k_composite_fwd, dim3((n_rays + 255) / 256), dim3(256), 0, s, raw, z, gt_depth, near_s, far_s,
                     cnt, min_nn, n_rays, coef, depth, var, rgb, valid, cw, ray_aux);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}

int launch_composite_bwd(const float4* raw, const float* z, const float* gt_depth, float near_s, float far_s, int n_rays, float coef,
                         const float* g_depth, const float* g_var, const float* g_rgb, float4* d_raw, float* zero64,
                         hipStream_t s) {
  if (n_rays <= 0) return PSL_OK;
  PSL_KLAUNCH(k_composite_bwd, dim3((n_rays + 255) / 256), dim3(256), 0, s, raw, z, gt_depth, near_s, far_s,
                     n_rays, coef, g_depth, g_var, g_rgb, d_raw, zero64);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}

int launch_ray_grad(const float4* dp, const float4* dp2, const float* z, const float* gt_depth, float near_s, float far_s, int n_rays,
                    float* g_o, float* g_d, hipStream_t s) {
  if (n_rays <= 0) return PSL_OK;
  hipLaunchKernelGGL(k_ray_grad, dim3((n_rays + 255) / 256), dim3(256), 0, s, dp, dp2, z, gt_depth, near_s, far_s, n_rays,
                     g_o, g_d);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}

// ------------------------------------------------------------------------ Adam
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, long long n, float lr_bc1, float sqrt_bc2,
                                              float b1, float b2, float eps, int zero_grad) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float pp = p[i], gg = g[i], mm = m[i], vv = v[i];
    adam_update(pp, gg, mm, vv, lr_bc1, sqrt_bc2, b1, b2, eps);
    p[i] = pp; m[i] = mm; v[i] = vv;
    if (zero_grad) g[i] = 0.f;
  }
}

// rows of a [N][32] feature matrix through an index list; g/m/v compact [n_rows][32]; float4 per lane
__global__ __launch_bounds__(256) void k_adam_rows(float* __restrict__ feats, const int* __restrict__ rows,
                                                   float4* __restrict__ g, float4* __restrict__ m,
                                                   float4* __restrict__ v, int n_rows, float lr_bc1,
                                                   float sqrt_bc2, float b1, float b2, float eps, int zero_grad) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n4 = (long long)n_rows * (C / 4);
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    int row = (int)(i >> 3), q = (int)(i & 7);
    int dst = rows ? rows[row] : row;
    float4* pp4 = reinterpret_cast<float4*>(feats + (size_t)dst * C) + q;
    float4 pp = *pp4, gg = g[i], mm = m[i], vv = v[i];
    adam_update(pp.x, gg.x, mm.x, vv.x, lr_bc1, sqrt_bc2, b1, b2, eps);
    adam_update(pp.y, gg.y, mm.y, vv.y, lr_bc1, sqrt_bc2, b1, b2, eps);
    adam_update(pp.z, gg.z, mm.z, vv.z, lr_bc1, sqrt_bc2, b1, b2, eps);
    adam_update(pp.w, gg.w, mm.w, vv.w, lr_bc1, sqrt_bc2, b1, b2, eps);
    *pp4 = pp; m[i] = mm; v[i] = vv;
    if (zero_grad) g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// One launch for all Adam groups of a mapper iteration (Mapper.py:394-402,556): geometry feature rows, colour
// feature rows (colour stage) and the colour-decoder parameters.  The parameter segment also refreshes the two
// fragment-major copies of each weight it steps (wf_index / wb_index: master element -> element of the forward / backward
// fragment buffer, -1 for tensors without a copy), which replaces a re-pack launch before the next forward.
__device__ __forceinline__ void adam_par_apply(const AdamParSeg& par, int i, float g, float b1, float b2, float eps) {
  float pp = par.p[i], mm = par.m[i], vv = par.v[i];
  adam_update(pp, g, mm, vv, par.lr_bc1, par.sqrt_bc2, b1, b2, eps);
  par.p[i] = pp; par.m[i] = mm; par.v[i] = vv;
  const int wf = par.wf_index[i], wb = par.wb_index[i];
  if (wf >= 0) par.wf[wf] = pp;
  if (wb >= 0) par.wb[wb] = pp;
}
// workgroup `blk` of the parameter segment.  par.slabs == null: 256 parameters, gradient read from par.g.  Otherwise
// 32 parameters x 8 chunk lanes: the chunk partials of the dW kernel are summed here in k_dw_reduce's fixed order
// (lane c takes chunks c, c+8, ...; the 8 sums are added in order) -- one launch and one pass over g_params less.
__device__ __forceinline__ void adam_par_segment(const AdamParSeg& par, int blk, float b1, float b2, float eps) {
  if (!par.slabs) {
    const int i = blk * 256 + (int)threadIdx.x;
    if (i < par.n) adam_par_apply(par, i, par.g[i], b1, b2, eps);
    return;
  }
  __shared__ float part[8][32];
  const int el = threadIdx.x & 31, cl = threadIdx.x >> 5;
  const int e = blk * 32 + el;
  float v = 0.f;
  if (e < par.n) {
    constexpr int b0 = poff(PI_C_BREL);
    if (e >= b0 && e < b0 + 3 * ERF) { if (cl == 0) v = par.g_brel[e - b0]; }
    else {
      int ent = 0;
#pragma unroll
      for (int j = 1; j < kNumColorParams; ++j) if (e >= poff(j)) ent = j;
      const int n_chunks = par.ra.chunks_of_entry[ent];
      const int se = par.ra.slab_off[ent] + (e - poff(ent));
      for (int c = cl; c < n_chunks; c += 8) v += par.slabs[(size_t)c * kDwSlabStride + se];
    }
  }
  part[cl][el] = v;
  __syncthreads();
  if (cl == 0 && e < par.n) {
    float t = part[0][el];
#pragma unroll
    for (int c = 1; c < 8; ++c) t += part[c][el];
    adam_par_apply(par, e, t, b1, b2, eps);
  }
}

// dense sweep: every selected row that ever had a gradient, one step (A/B baseline of the lazy kernel below)
__global__ __launch_bounds__(256) void k_map_adam(AdamRowsSeg geo, AdamRowsSeg col, AdamParSeg par, int nb_geo, int nb_col,
                                                  float b1, float b2, float eps) {
  int blk = blockIdx.x;
  if (blk < nb_geo + nb_col) {
    const AdamRowsSeg& sg = (blk < nb_geo) ? geo : col;
    if (blk >= nb_geo) blk -= nb_geo;
    const long long i = (long long)blk * blockDim.x + threadIdx.x;
    if (i >= (long long)sg.n_rows * (C / 4)) return;
    const int row = (int)(i >> 3), q = (int)(i & 7);
    if (sg.touched && !sg.touched[row]) return;         // never had a gradient: g = m = v = 0, the update is exactly zero
    const int dst = sg.rows ? sg.rows[row] : row;
    float4* pp4 = reinterpret_cast<float4*>(sg.feats + (size_t)dst * C) + q;
    float4 pp = *pp4, gg = sg.g[i], mm = sg.m[i], vv = sg.v[i];
    adam_update(pp.x, gg.x, mm.x, vv.x, sg.lr_bc1, sg.sqrt_bc2, b1, b2, eps);
    adam_update(pp.y, gg.y, mm.y, vv.y, sg.lr_bc1, sg.sqrt_bc2, b1, b2, eps);
    adam_update(pp.z, gg.z, mm.z, vv.z, sg.lr_bc1, sg.sqrt_bc2, b1, b2, eps);
    adam_update(pp.w, gg.w, mm.w, vv.w, sg.lr_bc1, sg.sqrt_bc2, b1, b2, eps);
    *pp4 = pp; sg.m[i] = mm; sg.v[i] = vv;
    sg.g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    adam_par_segment(par, blk - nb_geo - nb_col, b1, b2, eps);
  }
}

template <int LPR>
__global__ __launch_bounds__(256) void k_map_adam_lazy(AdamRowsSeg geo, AdamRowsSeg col, AdamParSeg par, int nb_rows, int n_groups,
                                                       float b1, float b2, float eps, AdamLazy lz) {
  __builtin_amdgcn_s_setprio(1);      // above the side-stream k-NN prefetch (see k_decode_fwd2)
  int blk0 = blockIdx.x;
  if (blk0 >= nb_rows * n_groups) { adam_par_segment(par, blk0 - nb_rows * n_groups, b1, b2, eps); return; }
  __shared__ float2 stab[kAdamTabLds];
  const bool is_col = blk0 >= nb_rows;
  if (is_col) blk0 -= nb_rows;
  adam_lazy_rows_block<LPR>(is_col ? col : geo, is_col, blk0, nb_rows, b1, b2, eps, lz, stab);
}

// Lanes per feature row of the lazy launch: 16 (two channels, one 8-byte access per stream and lane).  Round 4 measured 32 / 16 / 8
// lanes per row on one box (profiles/r04_adam_lanes_per_row.txt): 25.6 / 23.3 / 23.9 us per launch by the dispatch events, 95.1-95.2
// frames/s end to end for all three -- the launch is bound by its fixed costs (dispatch of the grid, the write-back at the kernel
// boundary), not by the access width; round 5 kept the one instantiation (advisor r4) and dropped the PSL_ADAM_LPR switch.
constexpr int kAdamLpr = 16;

}  // namespace psl

using namespace psl;

extern "C" int psl_composite_fwd(const float* raw, const float* z, int n_rays, float coef, float* depth, float* var,
                                 float* rgb, float* weights, void* stream) {
  if (!raw || !z || !depth || !var || !rgb || n_rays < 0) { set_error("psl_composite_fwd: bad argument"); return PSL_ERR_ARG; }
  return launch_composite_fwd((const float4*)raw, z, nullptr, 0.f, 0.f, nullptr, 0, n_rays, coef, depth, var, rgb,
                              nullptr, weights, nullptr, (hipStream_t)stream);
}

void psl::adam_consts(int step, float lr, float b1, float b2, float& lr_bc1, float& sqrt_bc2) {
  double bc1 = 1.0 - pow((double)b1, (double)step);
  double bc2 = 1.0 - pow((double)b2, (double)step);
  lr_bc1 = (float)((double)lr / bc1);
  sqrt_bc2 = (float)sqrt(bc2);
}

namespace psl {
int adam_lazy_row_blocks(const AdamLazy& lazy, int n_rows) {
  // a fixed grid of at most 2 048 workgroups per group (256 / lanes-per-row rows each per trip) walks the list with a stride
  const long long rows = lazy.list ? std::min<long long>(lazy.list_cap, n_rows) : n_rows;
  const int rpt = 256 / kAdamLpr;
  return (int)std::min<long long>((rows + rpt - 1) / rpt, 2048);
}

int launch_map_adam(AdamRowsSeg geo, int step_geo, float lr_geo, AdamRowsSeg col, int step_col, float lr_col, AdamParSeg par,
                    float lr_par, hipStream_t s, int step_par, AdamLazy lazy) {
  adam_consts(step_geo, lr_geo, 0.9f, 0.999f, geo.lr_bc1, geo.sqrt_bc2);
  if (col.n_rows > 0) adam_consts(step_col, lr_col, 0.9f, 0.999f, col.lr_bc1, col.sqrt_bc2);
  if (par.n > 0) adam_consts(step_par > 0 ? step_par : step_col, lr_par, 0.9f, 0.999f, par.lr_bc1, par.sqrt_bc2);
  const int nb_par = par.n <= 0 ? 0 : (par.slabs ? (par.n + 31) / 32 : (par.n + 255) / 256);
  if (lazy.tab) {
    // work-list mode: the grid covers the list's capacity, workgroups past its length leave after one load
    const int nb_rows = adam_lazy_row_blocks(lazy, geo.n_rows), n_groups = col.n_rows > 0 ? 2 : 1;
    if (nb_rows * n_groups + nb_par == 0) return PSL_OK;
    const dim3 grid(nb_rows * n_groups + nb_par);
    PSL_KLAUNCH(k_map_adam_lazy<kAdamLpr>, grid, dim3(256), 0, s, geo, col, par, nb_rows, n_groups, 0.9f, 0.999f, 1e-8f, lazy);
    PSL_LAUNCH_CHECK();
    return PSL_OK;
  }
  const int nb_geo = (int)(((long long)geo.n_rows * (C / 4) + 255) / 256);
  const int nb_col = (int)(((long long)col.n_rows * (C / 4) + 255) / 256);
  if (nb_geo + nb_col + nb_par == 0) return PSL_OK;
  PSL_KLAUNCH(k_map_adam, dim3(nb_geo + nb_col + nb_par), dim3(256), 0, s, geo, col, par, nb_geo, nb_col, 0.9f,
                     0.999f, 1e-8f);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}
}  // namespace psl

extern "C" int psl_adam_step(float* p, float* g, float* m, float* v, int64_t n, int step, float lr, float beta1,
                             float beta2, float eps, int zero_grad, void* stream) {
  if (!p || !g || !m || !v || n < 0 || step < 1) { set_error("psl_adam_step: bad argument"); return PSL_ERR_ARG; }
  if (n == 0) return PSL_OK;
  float a, b;
  adam_consts(step, lr, beta1, beta2, a, b);
  int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n, a, b, beta1,
                     beta2, eps, zero_grad);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}

extern "C" int psl_adam_step_rows(float* feats, const int32_t* rows, float* g, float* m, float* v, int n_rows,
                                  int step, float lr, float beta1, float beta2, float eps, int zero_grad, void* stream) {
  if (!feats || !g || !m || !v || n_rows < 0 || step < 1) { set_error("psl_adam_step_rows: bad argument"); return PSL_ERR_ARG; }
  if (n_rows == 0) return PSL_OK;
  float a, b;
  adam_consts(step, lr, beta1, beta2, a, b);
  long long n4 = (long long)n_rows * (C / 4);
  int blocks = (int)std::min<long long>((n4 + 255) / 256, 8192);
  hipLaunchKernelGGL(k_adam_rows, dim3(blocks), dim3(256), 0, (hipStream_t)stream, feats, rows, (float4*)g, (float4*)m,
                     (float4*)v, n_rows, a, b, beta1, beta2, eps, zero_grad);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}
