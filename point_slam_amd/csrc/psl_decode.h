// Shared between the decode forward / backward translation units.
#pragma once
#include "psl_common.h"
#include "psl_device.h"

namespace psl {

struct DecodeArgs {
  int P, n_rays, flags, min_nn;
  float near_s, far_s, r2_fixed;
  const float *rays_o, *rays_d, *depth, *r_query;
  const float* zv;       // explicit sample depths [P] (rays without sensor depth) or null
  const float4* pos;
  const float *geo_feats, *col_feats;
  const float* master;   // torch-layout parameter blob
  const float* Bcol;     // [3][20]
  const float *fb_geo, *fb_col, *affine;
  RenderWs ws;
  float* zero64;             // psl_map_iters, ray stage inside the backward: the forward clears the backward's 64 accumulators
  unsigned long long* dbg;   // optional phase timestamps (PSL_DEBUG_PHASES=1)
  unsigned long long* blk;   // optional per-workgroup trace [grid][4]: wall start, wall end, hw id, shader cycles (PSL_DEBUG_BLOCKS=<file>)
};
// per-workgroup trace: where the workgroups of a launch ran (XCC / SE / CU), when they started and how long they took
struct BlkTrace {     // the start stamps go straight to memory: nothing is carried in registers across the kernel
  __device__ __forceinline__ BlkTrace(const DecodeArgs& a) {
    if (a.blk != nullptr && threadIdx.x == 0) { unsigned long long* o = a.blk + 4 * (size_t)blockIdx.x; o[0] = wall_clock64(); o[3] = clock64(); }
  }
  __device__ __forceinline__ void done(const DecodeArgs& a) {
    if (a.blk == nullptr || threadIdx.x != 0) return;
    unsigned long long* o = a.blk + 4 * (size_t)blockIdx.x;
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    o[1] = wall_clock64(); o[2] = ((unsigned long long)xcc << 32) | hw; o[3] = clock64() - o[3];
  }
};
// ---- the mapper's ray stage inside the colour-stage decode backward (psl_map_iters without per-frame exposure) --------
// k_map_ray_fused (psl_ray.hip) is 8 us of launch latency between the two decode kernels, 240 times per mapped frame.  Its
// work per ray is ~150 instructions on 5 samples, and a ray's cotangents depend on nothing but that ray: the thread that
// owns a sample in the backward's set-up phase evaluates the sample's ray itself (compositing common.py:298-336, mapper loss
// Mapper.py:524-553, compositing backward) and keeps the cotangent of its own sample; the thread of a ray's FIRST sample
// also writes the ray's outputs and adds its loss terms.  Same expressions, same order as k_map_ray_fused.
struct RayFuse {
  const int* active; const float* gt_color; float coef, w_color; int n_rays;
  float *depth, *var, *rgb; unsigned char* valid;
  double* loss_acc;        // this iteration's [kLossSlots][4]
  int on;
};

// cotangent of raw[p] (rgb after sigmoid, occupancy logit); owner (first sample of its ray): outputs + loss terms
__device__ __forceinline__ float4 ray_cotangent(const DecodeArgs& a, const RayFuse& rf, int p, bool owner_writes, double& lg,
                                                double& lc, double& lcnt) {
  const int r = p / S, sj = p - r * S;
  float w[S], z[S], al[S], Tt[S], c0[S], c1[S], c2[S];
  float T = 1.0f, wsum = 0.f;
  int nhas = 0;
  const float gt = a.depth[r];
  const float4* raw = reinterpret_cast<const float4*>(a.ws.raw);
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const float4 q = raw[r * S + s];
    z[s] = sample_z(gt, s, a.near_s, a.far_s);
    al[s] = sigmoidf(rf.coef * q.w);
    Tt[s] = T;
    w[s] = al[s] * T;
    T = T * (1.0f - al[s] + 1e-10f);
    wsum += w[s];
    c0[s] = q.x; c1[s] = q.y; c2[s] = q.z;
    nhas += (a.ws.cnt[r * S + s] >= a.min_nn) ? 1 : 0;
  }
  const float W = wsum + 1e-10f;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, ad = 0.f;
#pragma unroll
  for (int s = 0; s < S; ++s) { a0 += w[s] * c0[s]; a1 += w[s] * c1[s]; a2 += w[s] * c2[s]; ad += w[s] * z[s]; }
  const float d = ad / W, m0 = a0 / W, m1 = a1 / W, m2 = a2 / W;
  const bool vr = nhas >= (S / 2 + 1);
  const bool owner = owner_writes && sj == 0;
  if (owner) {
    float v = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) { const float tmp = z[s] - d; v += w[s] * tmp * tmp; }
    rf.depth[r] = d; rf.var[r] = v; rf.rgb[r * 3] = m0; rf.rgb[r * 3 + 1] = m1; rf.rgb[r * 3 + 2] = m2; rf.valid[r] = vr ? 1 : 0;
  }
  float gd = 0.f, gr0 = 0.f, gr1 = 0.f, gr2 = 0.f;
  if (rf.active[r] && gt > 0.f && vr && d == d) {
    gd = (d > gt) ? 1.f : ((d < gt) ? -1.f : 0.f);
    const float g0 = rf.gt_color[r * 3], g1 = rf.gt_color[r * 3 + 1], g2 = rf.gt_color[r * 3 + 2];
    gr0 = rf.w_color * ((m0 > g0) ? 1.f : ((m0 < g0) ? -1.f : 0.f));
    gr1 = rf.w_color * ((m1 > g1) ? 1.f : ((m1 < g1) ? -1.f : 0.f));
    gr2 = rf.w_color * ((m2 > g2) ? 1.f : ((m2 < g2) ? -1.f : 0.f));
    if (owner) {
      lg = (double)fabsf(gt - d);
      lc = (double)fabsf(g0 - m0) + (double)fabsf(g1 - m1) + (double)fabsf(g2 - m2);
      lcnt = 1.0;
    }
  }
  float gw[S];
#pragma unroll
  for (int s = 0; s < S; ++s)
    gw[s] = (gd * (z[s] - d) + gr0 * (c0[s] - m0) + gr1 * (c1[s] - m1) + gr2 * (c2[s] - m2)) / W;
  float suffix = 0.f;
  float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int s = S - 1; s >= 0; --s) {
    const float ga = gw[s] * Tt[s] - suffix / (1.0f - al[s] + 1e-10f);
    const float gocc = ga * rf.coef * al[s] * (1.0f - al[s]);
    const float ws = w[s] / W;
    if (s == sj) out = make_float4(gr0 * ws, gr1 * ws, gr2 * ws, gocc);
    suffix += gw[s] * w[s];
  }
  return out;
}

// ---- the TRACKER's ray stage inside the pose-gradient decode backward (psl_track_iters, batches <= 1024 rays) -------------
// k_track_mid (psl_slam.hip) -- compositing, tracker loss with its 10 x mean mask (Tracker.py:159-180), lowest-loss pose,
// compositing backward -- was one more single-workgroup launch between the two decode kernels: 7.9 us + a launch gap, twenty
// times per frame, for ~200 instructions per ray.  As with the mapper's RayFuse the thread that owns a sample evaluates the
// sample's ray itself; what the tracker adds is the mask threshold 10 * mean(e) over ALL active rays: every wavefront that
// needs it composites all rays once (<= 16 per lane, the 16 KB of `raw` are L2-resident) and reduces with shuffles -- the
// same order in every wavefront of every workgroup, so that all tiles (and both roles) apply one and the same threshold.
// Same per-ray expressions and order as k_track_mid; the sums over rays (mean, loss) are taken in a different order (double).
struct TrackFuse {
  const int* active; const float* gt_color; float coef, w_color; int handle_dynamic, use_color, n_rays;
  float *depth, *var, *rgb; unsigned char* valid;
  const float* cam_tensor; float* best; float* loss_out;
  int on;
};
struct RayComp { float w[S], z[S], al[S], Tt[S], c0[S], c1[S], c2[S]; float W, d, v, m0, m1, m2, gt; int nhas; };

__device__ __forceinline__ float signf0_(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

// k_composite_fwd of ray r (common.py:298-336), as k_track_mid evaluates it
__device__ __forceinline__ void track_composite(const DecodeArgs& a, float coef, int r, RayComp& c) {
  const float4* raw = reinterpret_cast<const float4*>(a.ws.raw);
  c.gt = a.depth[r];
  float T = 1.0f, wsum = 0.f;
  c.nhas = 0;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const float4 q = raw[r * S + s];
    c.z[s] = sample_z(c.gt, s, a.near_s, a.far_s);
    c.al[s] = sigmoidf(coef * q.w);
    c.Tt[s] = T;
    c.w[s] = c.al[s] * T;
    T = T * (1.0f - c.al[s] + 1e-10f);
    wsum += c.w[s];
    c.c0[s] = q.x; c.c1[s] = q.y; c.c2[s] = q.z;
    c.nhas += (a.ws.cnt[r * S + s] >= a.min_nn) ? 1 : 0;
  }
  c.W = wsum + 1e-10f;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, ad = 0.f;
#pragma unroll
  for (int s = 0; s < S; ++s) { a0 += c.w[s] * c.c0[s]; a1 += c.w[s] * c.c1[s]; a2 += c.w[s] * c.c2[s]; ad += c.w[s] * c.z[s]; }
  c.d = ad / c.W;
  c.v = 0.f;
#pragma unroll
  for (int s = 0; s < S; ++s) { const float tmp = c.z[s] - c.d; c.v += c.w[s] * tmp * tmp; }
  c.m0 = a0 / c.W; c.m1 = a1 / c.W; c.m2 = a2 / c.W;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// 10 * mean(e) over the active rays (Tracker.py:165), by ALL 512 threads of a workgroup (both roles call it on entry: the
// seven wavefronts of a geometry-role workgroup that exit at once otherwise lend a hand first): thread t composites rays t and
// t + 512 -- every load of the pass in flight at once, one cache round trip instead of n / 64 serial ones in one wavefront
// (measured: with one wavefront per tile doing the pass the saved launch was paid back in full) --, 64-ray chunks are summed
// by wave shuffles, the chunk sums in chunk order by every thread: the same arithmetic in every workgroup of either role, so
// that all tiles apply ONE threshold.  LOSS (workgroup 0 of the colour role): the iteration's loss and the lowest-loss pose
// (Tracker.py:176-180,347-350) from the same composites.  n_rays <= 1024.  red: >= 128 floats of LDS nobody else touches yet.
__device__ __forceinline__ float track_threshold_block(const DecodeArgs& a, const TrackFuse& tf, float* red, bool loss) {
  double* part = reinterpret_cast<double*>(red);        // [16 chunks][2] (sum e, count), then [16][2] (loss geo, colour)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, n = tf.n_rays;
  float gt[2], d[2], v[2], m0[2], m1[2], m2[2];
  bool act[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = t + 512 * j;
    act[j] = r < n && tf.active[min(r, n - 1)] != 0;
    gt[j] = d[j] = v[j] = m0[j] = m1[j] = m2[j] = 0.f;
    if (r < n) {
      RayComp c;
      track_composite(a, tf.coef, r, c);
      gt[j] = c.gt; d[j] = c.d; v[j] = c.v; m0[j] = c.m0; m1[j] = c.m1; m2[j] = c.m2;
    }
    double se = 0.0, sc = 0.0;
    if (act[j]) {
      float e = fabsf(gt[j] - d[j]);
      if (tf.handle_dynamic) e = e / sqrtf(v[j] + 1e-10f);
      se = (double)e; sc = 1.0;
    }
    se = wave_sum_d(se); sc = wave_sum_d(sc);
    if (lane == 0) { part[2 * (wave + 8 * j)] = se; part[2 * (wave + 8 * j) + 1] = sc; }
  }
  __syncthreads();
  double tot = 0.0, cnt = 0.0;
#pragma unroll
  for (int c = 0; c < 16; ++c) { tot += part[2 * c]; cnt += part[2 * c + 1]; }
  const float thr = (cnt > 0.0) ? 10.0f * (float)(tot / cnt) : 0.f;
  if (loss) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = t + 512 * j;
      double lg = 0.0, lc = 0.0;
      if (act[j]) {
        const float diff = fabsf(gt[j] - d[j]);
        const float tmp = tf.handle_dynamic ? diff / sqrtf(v[j] + 1e-10f) : diff;
        if ((tmp < thr) && (gt[j] > 0.f) && (d[j] == d[j]) && (v[j] == v[j])) {
          const float e = diff / sqrtf(v[j] + 1e-10f);
          lg = (double)fminf(fmaxf(e, 0.f), 1e3f);
          const float q0 = tf.gt_color[r * 3], q1 = tf.gt_color[r * 3 + 1], q2 = tf.gt_color[r * 3 + 2];
          lc = (double)fabsf(q0 - m0[j]) + (double)fabsf(q1 - m1[j]) + (double)fabsf(q2 - m2[j]);
        }
      }
      lg = wave_sum_d(lg); lc = wave_sum_d(lc);
      if (lane == 0) { part[32 + 2 * (wave + 8 * j)] = lg; part[32 + 2 * (wave + 8 * j) + 1] = lc; }
    }
    __syncthreads();
    if (t == 0) {
      double Lg = 0.0, Lc = 0.0;
      for (int c = 0; c < 16; ++c) { Lg += part[32 + 2 * c]; Lc += part[32 + 2 * c + 1]; }
      const double L = tf.use_color ? Lg + (double)tf.w_color * Lc : Lg;
      tf.loss_out[0] = (float)L; tf.loss_out[1] = (float)Lg; tf.loss_out[2] = (float)Lc; tf.loss_out[3] = (float)cnt;
      if ((float)L < tf.best[7]) {
        tf.best[7] = (float)L;
#pragma unroll
        for (int j = 0; j < 7; ++j) tf.best[j] = tf.cam_tensor[j];
      }
    }
  }
  __syncthreads();          // the scratch may be reused from here on
  return thr;
}

// cotangent of raw[p]; owner (first sample of its ray, colour role only): the ray's render outputs
__device__ __forceinline__ float4 track_cotangent(const DecodeArgs& a, const TrackFuse& tf, int p, float thr, bool owner_writes) {
  const int r = p / S, sj = p - r * S;
  RayComp c;
  track_composite(a, tf.coef, r, c);
  if (owner_writes && sj == 0) {
    tf.depth[r] = c.d; tf.var[r] = c.v; tf.rgb[r * 3] = c.m0; tf.rgb[r * 3 + 1] = c.m1; tf.rgb[r * 3 + 2] = c.m2;
    tf.valid[r] = c.nhas >= (S / 2 + 1) ? 1 : 0;
  }
  float gdp = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f;
  if (tf.active[r] != 0) {
    const float inv = 1.0f / sqrtf(c.v + 1e-10f);
    const float diff = fabsf(c.gt - c.d);
    const float tmp = tf.handle_dynamic ? diff / sqrtf(c.v + 1e-10f) : diff;
    const bool m = (tmp < thr) && (c.gt > 0.f) && (c.d == c.d) && (c.v == c.v);
    if (m) {
      const float e = diff / sqrtf(c.v + 1e-10f);
      if (e <= 1e3f) gdp = signf0_(c.d - c.gt) * inv;
      if (tf.use_color) {
        const float q0 = tf.gt_color[r * 3], q1 = tf.gt_color[r * 3 + 1], q2 = tf.gt_color[r * 3 + 2];
        g0 = tf.w_color * signf0_(c.m0 - q0); g1 = tf.w_color * signf0_(c.m1 - q1); g2 = tf.w_color * signf0_(c.m2 - q2);
      }
    }
  }
  // k_composite_bwd with g_var = 0
  float gw[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const float dz = c.z[s] - c.d;
    gw[s] = (gdp * dz + g0 * (c.c0[s] - c.m0) + g1 * (c.c1[s] - c.m1) + g2 * (c.c2[s] - c.m2)) / c.W;
  }
  float suffix = 0.f;
  float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int s = S - 1; s >= 0; --s) {
    const float ga = gw[s] * c.Tt[s] - suffix / (1.0f - c.al[s] + 1e-10f);
    const float gocc = ga * tf.coef * c.al[s] * (1.0f - c.al[s]);
    const float ws = c.w[s] / c.W;
    if (s == sj) out = make_float4(g0 * ws, g1 * ws, g2 * ws, gocc);
    suffix += gw[s] * c.w[s];
  }
  return out;
}

// work list of the lazy Adam (as adam_worklist_role, psl_ray.hip), one int4 of neighbour indices per lane; wave-level appends
__device__ __forceinline__ void worklist_role_wave(const AdamWorklist& wl, int i) {
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


// Colour-stage launch structure (psl_debug_option("color_split", v) / PSL_COLOR_SPLIT): 0 = always the fused 16-sample tile
// kernels (k_decode_fwd2<true> / k_decode_bwd2<., true>), 2 = always the split kernels (k_nbr_* + k_trunk_*), 1 = by launch
// size: the split pays once a launch holds clearly more tiles than the chip has CUs -- its four kernels each carry ~5 us of
// dependent-load set-up and tail that the fused tile pays once (measured on one box, gpurun r06h / r06j: 5 000 samples fused
// 48.2 + 46.6 us against 49.1 + 51.8 split; 25 000: 185 + 203 against 180 + 190; 125 000: 872 + 978 against 763 + 937)
extern int g_color_split;
constexpr int kSplitMinTiles = 384;
// trunk of the split structure: one wavefront per tile (psl_trunk_wave.hip) from this many tiles on (PSL_WAVE_TRUNK / option
// "wave_trunk": tiles from which it is used; 0 = never)
extern int g_wave_trunk_tiles;
inline bool wave_trunk_on(int tiles) { return g_wave_trunk_tiles > 0 && tiles >= g_wave_trunk_tiles; }
inline bool color_split_on(int tiles) { return g_color_split == 2 || (g_color_split == 1 && tiles > kSplitMinTiles); }

// ray-level arguments of the one-launch geometry-stage iteration (psl_decode_geo.hip)
struct GeoIterRays {
  const int* active;            // [R] 1 = ray passed the depth filters (common.py:173-179, Mapper.py:507-514)
  float coef;                   // sigmoid coefficient of the mapper (Mapper.py:45)
  float *depth, *var, *rgb;     // [R], [R], [R][3] render outputs (rgb = 0 in this stage)
  unsigned char* valid;         // [R]
  double* loss_acc;             // [kLossSlots][4]: sum |d_gt - d|, (colour: unused), #rays in the mask; slot = workgroup & 31
  float* zero64;                // accumulators of a later colour-stage backward (cleared here as the ray kernel does)
  int n_rays;
};

int blk_trace_begin(DecodeArgs& a, int grid, hipStream_t s);                                  // psl_api.hip
int blk_trace_end(const DecodeArgs& a, const char* kernel, int grid, int color_tiles, int threads);
#define PSL_STAMP(i) do { if (a.dbg && blockIdx.x == 0 && threadIdx.x == 0) a.dbg[i] = clock64(); } while (0)
#define PSL_STAMPB(i, blk) do { if (a.dbg && (int)blockIdx.x == (blk) && threadIdx.x == 0) a.dbg[i] = clock64(); } while (0)
#ifdef PSL_FINE_STAMPS   // stamps inside the GEMM loops perturb scheduling: opt-in build (make EXTRA=-DPSL_FINE_STAMPS)
#define PSL_STAMPF(i) PSL_STAMP(i)
#else
#define PSL_STAMPF(i) do { } while (0)
#endif

// Algorithmic work per sample point (2 FLOP per MAC; unpadded layer sizes)
constexpr double MAC_GEO = 15479.0, MAC_COL = 96700.0, MAC_NBR = 86256.0, MAC_INTERP = 256.0;
inline double fwd_flops_per_sample(int flags) {
  double m = MAC_GEO + MAC_INTERP;
  if (flags & PSL_STAGE_COLOR) m += MAC_COL + MAC_INTERP + ((flags & 0x10000) ? MAC_NBR : 0.0);
  return 2.0 * m;
}
// dX chain: mirrors the forward for the tracker instantiation (pose gradient: every input of every layer); the mapper
// instantiation never differentiates w.r.t. the Fourier embeddings or the rel-pos rows unless parameter gradients need
// dB_rel, and never w.r.t. the first colour layer's input -- those products are not issued and not counted
inline double bwd_flops_per_sample(int flags) {
  if (flags & PSL_PTS_GRAD) return fwd_flops_per_sample(flags);
  const bool color = flags & PSL_STAGE_COLOR, relpos = flags & 0x10000, parg = flags & PSL_PARAM_GRAD;
  double m = (MAC_GEO - 93.0 * 32.0 * 2.0) + MAC_INTERP;            // geometry: no d/d(embedding) (layers 0 and 3)
  if (color) {
    m += (MAC_COL - 40.0 * 128.0 * 2.0) + MAC_INTERP;               // colour trunk without the two embedding products
    if (relpos) m += MAC_NBR - (parg ? 0.0 : 8.0 * 20.0 * 128.0);   // F_theta; rel-pos rows only for dB_rel
  }
  return 2.0 * m;
}
inline double dw_flops_per_sample(int flags) {
  return (flags & PSL_STAGE_COLOR) ? 2.0 * (MAC_COL + ((flags & 0x10000) ? MAC_NBR : 0.0)) : 0.0;
}
// algorithmic HBM bytes per sample: query 12 + neighbour positions 8*12 + 8 feature rows of 128 B per feature set
inline double gather_bytes_per_sample(int flags) { return 12.0 + 96.0 + ((flags & PSL_STAGE_COLOR) ? 2048.0 : 1024.0); }
inline double scatter_bytes_per_sample(int flags) { return (flags & PSL_STAGE_COLOR) ? 4096.0 : 2048.0; }

// master offsets (floats)
constexpr int MO(int pi) { return poff(pi); }

// per-sample geometry shared by fwd and bwd: sample position, squared radius
struct SampleGeom { float x, y, z, r2, zval; int ray; };

__device__ __forceinline__ SampleGeom sample_geom(const DecodeArgs& a, int p) {
  SampleGeom g;
  int ray = p / S, si = p - ray * S;
  g.ray = ray;
  // every load of the set-up is requested before the first use (round 4).  Written with `cond ? load : load` and `if (ptr)`,
  // hipcc put each load into its own basic block with a full wait behind it: depth -> wait -> rays -> radius -> wait, two
  // serial cache round trips in front of EVERY tile of every decode kernel before its neighbour lists were even requested.
  const float* dptr = a.zv ? a.zv + p : a.depth + ray;                 // one address, one load
  const float* rptr = a.r_query ? a.r_query + ray : dptr;              // (a valid dummy address when the radius is fixed)
  const float dval = *dptr, rq = *rptr;
  const float ox = a.rays_o[ray * 3], oy = a.rays_o[ray * 3 + 1], oz = a.rays_o[ray * 3 + 2];
  const float dx = a.rays_d[ray * 3], dy = a.rays_d[ray * 3 + 1], dz = a.rays_d[ray * 3 + 2];
  g.zval = a.zv ? dval : sample_z(dval, si, a.near_s, a.far_s);
  sample_point(ox, oy, oz, dx, dy, dz, g.zval, g.x, g.y, g.z);
  g.r2 = a.r_query ? __fmul_rn(rq, rq) : a.r2_fixed;
  return g;
}

// Interpolation weight of a neighbour at squared distance D before the L1 normalisation (decoder.py:152-157 geometry, :362-367 colour):
// pointcloud.nn_weighting 'distance' = 1 / (D + 1e-10), 'expo' = exp(-20 sqrt(D)); 0 beyond the query radius.  DecodeArgs.flags bit
// kFlagExpoW selects 'expo' (psl_config.nn_weighting; no shipped config uses it).
constexpr int kFlagExpoW = 0x20000;
__device__ __forceinline__ float nn_weight(float D, float r2, bool expo) {
  if (D > r2) return 0.f;
  return expo ? expf(__fmul_rn(-20.0f, sqrtf(D))) : 1.0f / (D + 1e-10f);
}
// (No derivative with respect to D for 'expo': the tracker's pose gradient through exp(-20 sqrt(D)) does not exist in the reference either --
// decoder.py:157 zeroes the exp's output in place and autograd raises in backward --, so psl_render_fwd refuses PSL_PTS_GRAD with it.)

// (2*pi*p) . B[:, f] -- the Fourier phase of decoder.py:33 (matmul of [.,3] by [3,F])
__device__ __forceinline__ float fourier_phase(float x, float y, float z, const float* __restrict__ B, int F, int f) {
  float x2 = __fmul_rn(TWO_PI, x), y2 = __fmul_rn(TWO_PI, y), z2 = __fmul_rn(TWO_PI, z);
  return fmaf(z2, B[2 * F + f], fmaf(y2, B[F + f], __fmul_rn(x2, B[f])));
}

}  // namespace psl
