// One geometry-stage mapper iteration in ONE launch: geometry decoder forward, compositing, mapper loss, compositing
// backward and geometry decoder backward of a whole ray triple per wavefront.
//
// Reference: Mapper.optimize_map, stage 'geometry' (src/Mapper.py:420-423,455-556): MLP_geometry.forward on the five samples
// of every ray (src/conv_onet/models/decoder.py:130-222), raw2outputs_nerf_color with rgb = 0 (src/common.py:298-336), the
// depth L1 term over the valid rays (src/Mapper.py:524-529), autograd back to the geometry feature rows.
//
// Why one kernel.  In stage 'geometry' (40 % of a mapped frame's iterations) nothing couples samples of different rays
// except the compositing of a ray's own five samples, and the geometry decoder of a 16-sample tile is wave-private
// (registers only, psl_decode_fwd2.hip).  Three launches per iteration -- decode forward (14-16 us), the ray kernel
// (8-11 us), decode backward (6-17 us) -- each a chain of dependent loads on ~330 single-wave workgroups, cost ~50 us with
// their launch gaps.  Here a wavefront owns 15 samples = THREE WHOLE RAYS (lane slot 15 idles): the occupancy logits
// never leave the register file, the five samples of a ray meet through wave shuffles, the ReLU masks the backward needs
// are 40 bits per lane, and neither g_y nor raw / d_raw are written or read back.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include "psl_decode.h"
#include "psl_frag.h"
#include "psl_decode2.h"

namespace psl {

__device__ __forceinline__ f32x4 ldfragb_g(const float* __restrict__ WB, int frag, int lane) {
  return *reinterpret_cast<const f32x4*>(WB + (size_t)frag * FRAG + lane * 4);
}

constexpr int GI_TILE = 15;     // samples per wavefront: three rays

__device__ __forceinline__ void geo_iter_tile(const DecodeArgs& a, const float* __restrict__ WF, const float* __restrict__ WB,
                                              const GeoIterRays& gr, float* g_geo, const int* __restrict__ row_map,
                                              unsigned char* t_geo, int tile, ScatterLds& sl) {
  constexpr int GEO_AHEAD = 6;
  const int lane = threadIdx.x & 63, rl = lane & 15, g = lane >> 4;
  const int p0 = tile * GI_TILE;
  const bool slot = rl < GI_TILE;
  const bool live = slot && p0 + rl < a.P;
  const int p = min(p0 + min(rl, GI_TILE - 1), a.P - 1);
  const float* __restrict__ M = a.master;
  f32x4 W0[kGeo.n], W1[kGeo.n];
  PSL_STAMP(0);
  // ---- neighbours, inverse-distance weights (decoder.py:152-160), interpolation (:162-171).  The lists (and the count) are
  // requested FIRST: they depend on nothing but the sample index, and the positions they name are the next dependent trip
  int nb[K];
  {
    const int4 i0 = *reinterpret_cast<const int4*>(a.ws.I + (size_t)p * K);
    const int4 i1 = *reinterpret_cast<const int4*>(a.ws.I + (size_t)p * K + 4);
    nb[0] = i0.x; nb[1] = i0.y; nb[2] = i0.z; nb[3] = i0.w; nb[4] = i1.x; nb[5] = i1.y; nb[6] = i1.z; nb[7] = i1.w;
  }
  const int cnt_p = a.ws.cnt[p];
  const SampleGeom sg = sample_geom(a, p);
  // The Fourier matrix columns of this lane and the first weight fragments depend on nothing: they are requested behind the
  // lists, so that the sine phase below runs while the positions and feature rows (the second dependent trip) are in flight
  const float* __restrict__ Bg = M + MO(PI_G_B);
  float Bv[6][4][3];
#pragma unroll
  for (int q = 0; q < 6; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = min(16 * q + 4 * g + r, EG - 1);
      Bv[q][r][0] = Bg[f]; Bv[q][r][1] = Bg[EG + f]; Bv[q][r][2] = Bg[2 * EG + f];
    }
  sched_fence();
  float4 qp[K];
  f32x4 f0[K], f1[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    qp[k] = a.pos[max(nb[k], 0)];
    const float* row = a.geo_feats + (size_t)max(nb[k], 0) * C + 4 * g;
    f0[k] = *reinterpret_cast<const f32x4*>(row); f1[k] = *reinterpret_cast<const f32x4*>(row + 16);
  }
  const f32x4 fb0 = *reinterpret_cast<const f32x4*>(a.fb_geo + 4 * g), fb1 = *reinterpret_cast<const f32x4*>(a.fb_geo + 16 + 4 * g);
  sched_fence();
  PSL_STAMP(1);
  // ---- Fourier features sin(2 pi p . B) (decoder.py:8-37), channel 16 q + 4 g + r
  f32x4 eg[6];
  {
    const float x2 = __fmul_rn(TWO_PI, sg.x), y2 = __fmul_rn(TWO_PI, sg.y), z2 = __fmul_rn(TWO_PI, sg.z);
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = fast_sinf(fmaf(z2, Bv[q][r][2], fmaf(y2, Bv[q][r][1], __fmul_rn(x2, Bv[q][r][0]))));   // = fourier_phase
        eg[q][r] = (16 * q + 4 * g + r < EG) ? v : 0.f;
      }
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) pin(eg[q]);
  sched_fence();
  PSL_STAMP(2);
#pragma unroll
  for (int st = 0; st < GEO_AHEAD; ++st) {
    W0[st] = ldfrag(WF, kGeo.s[st].f0, lane);
    if (kGeo.s[st].f1 >= 0) W1[st] = ldfrag(WF, kGeo.s[st].f1, lane);
  }
  sched_fence();
  // ---- inverse-distance weights (decoder.py:152-160), interpolation (:162-171)
  float w[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float D = (nb[k] >= 0) ? dist2(qp[k].x, qp[k].y, qp[k].z, sg.x, sg.y, sg.z) : __int_as_float(0x7F800000);
    w[k] = nn_weight(D, sg.r2, (a.flags & kFlagExpoW) != 0);
  }
  const float wsum = ((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]));
  const float inv = fmaxf(wsum, 1e-12f);
#pragma unroll
  for (int k = 0; k < K; ++k) w[k] = w[k] / inv;
  const bool has = cnt_p >= a.min_nn;     // has_neighbors (decoder.py:150)
  f32x4 cg[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      cg[0][r] = __fadd_rn(cg[0][r], __fmul_rn(w[k], f0[k][r]));
      cg[1][r] = __fadd_rn(cg[1][r], __fmul_rn(w[k], f1[k][r]));
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) { cg[0][r] = has ? cg[0][r] : fb0[r]; cg[1][r] = has ? cg[1][r] : fb1[r]; }
  pin(cg[0]); pin(cg[1]);
  sched_fence();
  PSL_STAMP(3);
  // ---- five blocks: h = relu(W_i h + b_i) + (Wc_i c + bc_i); the embedding is re-attached after block 2
  f32x4 h[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  f32x4 acc[2], u[2], oo[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  unsigned ym[5] = {0u, 0u, 0u, 0u, 0u};       // ReLU masks of the five layers: bit 4 nt + r = (pre-activation > 0)
  acc[0] = ldbias(WF, fbias(FL_G0), 0, g); acc[1] = ldbias(WF, fbias(FL_G0), 1, g);
  u[0] = ldbias(WF, fbias(FL_GF0), 0, g); u[1] = ldbias(WF, fbias(FL_GF0), 1, g);
#pragma unroll
  for (int st = 0; st < kGeo.n; ++st) {
    sched_fence();
    if (st + GEO_AHEAD < kGeo.n) {
      W0[st + GEO_AHEAD] = ldfrag(WF, kGeo.s[st + GEO_AHEAD].f0, lane);
      if (kGeo.s[st + GEO_AHEAD].f1 >= 0) W1[st + GEO_AHEAD] = ldfrag(WF, kGeo.s[st + GEO_AHEAD].f1, lane);
    }
    const int bs = kGeo.s[st].bsel;
    const f32x4 b = bs < 6 ? eg[bs < 6 ? bs : 0] : (bs < 8 ? h[bs < 8 ? (bs >= 6 ? bs - 6 : 0) : 0] : cg[bs >= 8 ? bs - 8 : 0]);
    if (kGeo.s[st].dst == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { acc[0] = mfma16(W0[st][r], b[r], acc[0]); acc[1] = mfma16(W1[st][r], b[r], acc[1]); }
    } else if (kGeo.s[st].dst == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { u[0] = mfma16(W0[st][r], b[r], u[0]); u[1] = mfma16(W1[st][r], b[r], u[1]); }
    } else {
      mma4(oo[st & 1], W0[st], b);
    }
    if (kGeo.s[st].layer_end) {
      const int i = kGeo.s[st].layer_end - 1;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float y = fmaxf(acc[nt][r], 0.f);
          ym[i] |= (y > 0.f ? 1u : 0u) << (4 * nt + r);        // what the backward's `y > 0` test reads
          h[nt][r] = y + u[nt][r];
        }
      if (i < 4) {
        constexpr int FLs[5] = {FL_G0, FL_G1, FL_G2, FL_G3, FL_G4};
        constexpr int FLf[5] = {FL_GF0, FL_GF1, FL_GF2, FL_GF3, FL_GF4};
        acc[0] = ldbias(WF, fbias(FLs[i + 1]), 0, g); acc[1] = ldbias(WF, fbias(FLs[i + 1]), 1, g);
        u[0] = ldbias(WF, fbias(FLf[i + 1]), 0, g); u[1] = ldbias(WF, fbias(FLf[i + 1]), 1, g);
      }
    }
  }
  // occupancy logit of sample rl (lanes g == 0 hold it); raw[~point_mask, -1] = -100 (Renderer.py:189-190)
  const float occ = has ? (oo[0][0] + oo[1][0]) + M[MO(PI_G_OUT + 1)] : -100.0f;
  PSL_STAMP(4);
  // the backward's first layer of fragments flies during the compositing
  f32x4 bwf[4], bwb[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { bwf[e] = ldfragb_g(WB, bfirst(BL_GF4) + e, lane); bwb[e] = ldfragb_g(WB, bfirst(BL_G4) + e, lane); }

  // ---------------------------------------------------------------- compositing + loss + compositing backward
  // Every lane evaluates its OWN ray (the five samples sit in lanes 5 j .. 5 j + 4 of lane group 0) and keeps the
  // occupancy cotangent of its own sample: raw2outputs_nerf_color (common.py:298-336) with rgb = 0.
  const int j = min(rl, GI_TILE - 1) / S, sj = min(rl, GI_TILE - 1) - j * S;
  const int ray = tile * 3 + j;
  const bool ray_ok = slot && ray < gr.n_rays;
  const int rayc = min(ray, gr.n_rays - 1);
  const float gt = a.depth[rayc];
  float al[S], Tt[S], wq[S], z[S];
  float T = 1.0f, wsumc = 0.f;
  int nhas = 0;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const float o_s = __shfl(occ, 5 * j + s);
    const int h_s = __shfl(has ? 1 : 0, 5 * j + s);
    z[s] = sample_z(gt, s, a.near_s, a.far_s);
    al[s] = sigmoidf(gr.coef * o_s);
    Tt[s] = T;
    wq[s] = al[s] * T;
    T = T * (1.0f - al[s] + 1e-10f);
    wsumc += wq[s];
    nhas += h_s;
  }
  const float Wn = wsumc + 1e-10f;
  float ad = 0.f;
#pragma unroll
  for (int s = 0; s < S; ++s) ad += wq[s] * z[s];
  const float d = ad / Wn;
  float v = 0.f;
#pragma unroll
  for (int s = 0; s < S; ++s) { const float tmp = z[s] - d; v += wq[s] * tmp * tmp; }
  const bool vr = nhas >= (S / 2 + 1);
  double lg = 0.0, lcnt = 0.0;
  float gd = 0.f;
  const bool head = ray_ok && g == 0 && sj == 0;     // one lane per ray writes the outputs and counts the loss
  if (ray_ok && gr.active[rayc] && gt > 0.f && vr && d == d) {
    gd = (d > gt) ? 1.f : ((d < gt) ? -1.f : 0.f);
    if (head) { lg = (double)fabsf(gt - d); lcnt = 1.0; }
  }
  if (head) {
    if (gr.depth) gr.depth[ray] = d;
    if (gr.var) gr.var[ray] = v;
    if (gr.rgb) { gr.rgb[ray * 3] = 0.f; gr.rgb[ray * 3 + 1] = 0.f; gr.rgb[ray * 3 + 2] = 0.f; }
    if (gr.valid) gr.valid[ray] = vr ? 1 : 0;
  }
  float gw[S];
#pragma unroll
  for (int s = 0; s < S; ++s) gw[s] = (gd * (z[s] - d)) / Wn;
  float suffix = 0.f, docc = 0.f;
#pragma unroll
  for (int s = S - 1; s >= 0; --s) {
    const float ga = gw[s] * Tt[s] - suffix / (1.0f - al[s] + 1e-10f);
    const float gocc = ga * gr.coef * al[s] * (1.0f - al[s]);
    if (s == sj) docc = gocc;
    suffix += gw[s] * wq[s];
  }
  docc = live ? docc : 0.f;        // d_occ flows for masked samples too (straight-through of the -100 write)
  // the three head lanes (0, 5, 10) hold the tile's terms: summed in the order the xor butterfly over the wavefront gave lane 0,
  // (ray 0 + ray 2) + ray 1, without its twelve dependent 64-bit shuffles
  lg = (readlane_d(lg, 0) + readlane_d(lg, 10)) + readlane_d(lg, 5);
  lcnt = (readlane_d(lcnt, 0) + readlane_d(lcnt, 10)) + readlane_d(lcnt, 5);
  if (lane == 0 && gr.loss_acc) {
    double* acc = gr.loss_acc + 4 * (blockIdx.x & (kLossSlots - 1));
    if (lg != 0.0) atomicAdd(&acc[0], lg);
    if (lcnt != 0.0) atomicAdd(&acc[2], lcnt);
  }

  PSL_STAMP(5);
  // ---------------------------------------------------------------- geometry decoder backward (psl_decode_bwd2.hip geo role)
  f32x4 G[2], dcg[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) G[nt][r] = docc * M[MO(PI_G_OUT) + nt * 16 + 4 * g + r];
  // (the eight fragments of layer i - 1 are requested before the MFMAs of layer i: a layer's weights used to be fetched
  //  where they were needed, ~3.5 k cycles of exposed L2 latency per layer on a lone wavefront)
#pragma unroll
  for (int i = 4; i >= 0; --i) {
    constexpr int BLs[5] = {BL_G0, BL_G1, BL_G2, BL_G3, BL_G4};
    constexpr int BLf[5] = {BL_GF0, BL_GF1, BL_GF2, BL_GF3, BL_GF4};
    sched_fence();
    f32x4 wfc[4], wbc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { wfc[e] = bwf[e]; wbc[e] = bwb[e]; }
    if (i > 0) {
      const int ffn = bfirst(BLf[i > 0 ? i - 1 : 0]), fbn = bfirst(BLs[i > 0 ? i - 1 : 0]);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bwf[e] = ldfragb_g(WB, ffn + e, lane); if (i > 1) bwb[e] = ldfragb_g(WB, fbn + e, lane); }
    }
    sched_fence();
    f32x4 dz[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) dz[nt][r] = ((ym[i] >> (4 * nt + r)) & 1u) ? G[nt][r] : 0.f;      // ReLU
    // dL/dc += Wc_i^T G   (fc_c.i.weight [32][32]); fragment (it, q) of a layer sits at first + 2 it + q
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int it = 0; it < 2; ++it) mma4(dcg[it], wfc[it * 2 + q], G[q]);
    // dL/d(input of layer i) = W_i^T dz
    if (i > 0) {
      f32x4 Gn[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int it = 0; it < 2; ++it) mma4(Gn[it], wbc[it * 2 + q], dz[q]);   // hidden tiles come first
      G[0] = Gn[0]; G[1] = Gn[1];
    }
  }
  sched_fence();
  PSL_STAMP(6);
  // ---- scatter w_k * dC into the geometry feature rows (coalesced: psl_decode2.h)
  const bool hasl = live && has;
  int dst[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int i = nb[k];
    dst[k] = (i >= 0 && hasl && w[k] != 0.f) ? (row_map ? row_map[i] : i) : -1;
  }
  scatter_interp_rows(sl, g_geo, t_geo, dcg, w, dst);
  PSL_STAMP(7);
}

// grid: [0, n_tiles) one wavefront per ray triple, then the work-list role (four list entries per thread, 8 waves' worth of
// entries per 64-thread workgroup: each walks its share with a stride)
__global__ __launch_bounds__(64, 2) void k_geo_iter(DecodeArgs a, const float* __restrict__ WF, const float* __restrict__ WB,
                                                    GeoIterRays gr, float* g_geo, const int* __restrict__ row_map,
                                                    unsigned char* t_geo, AdamWorklist wl, int n_tiles, int n_wl_blocks) {
  // above the mapper's side-stream k-NN prefetch (priority 0), whose waves share the SIMDs of this launch for 2 of every 7 ms of a mapped frame
  __builtin_amdgcn_s_setprio(2);
  __shared__ ScatterLds sl;
  BlkTrace bt(a);
  if ((int)blockIdx.x < n_tiles) {
    if (blockIdx.x == 0 && gr.zero64) gr.zero64[threadIdx.x] = 0.f;
    geo_iter_tile(a, WF, WB, gr, g_geo, row_map, t_geo, (int)blockIdx.x, sl);
  } else {
    const int total = (wl.I_b ? 2 : 1) * wl.n4;
    for (int i = ((int)blockIdx.x - n_tiles) * 64 + (int)threadIdx.x; i - (int)threadIdx.x < total; i += n_wl_blocks * 64)
      worklist_role_wave(wl, i);
  }
  bt.done(a);
}

int launch_geo_iter(psl_ctx* ctx, const DecodeArgs& a_in, const GeoIterRays& gr, float* g_geo, const int* row_map,
                    const AdamWorklist* wl, hipStream_t s) {
  if (a_in.P <= 0) return PSL_OK;
  DecodeArgs a = a_in;
  AdamWorklist w{};
  if (wl) w = *wl;
  const int n_tiles = (gr.n_rays + 2) / 3;
  const int n_wl = (w.I_a && w.n4 > 0) ? std::min(((w.I_b ? 2 : 1) * w.n4 + 63) / 64, 512) : 0;
  static unsigned long long* dbg = nullptr;
  static int dbg_on = -1;
  if (dbg_on < 0) { const char* e = getenv("PSL_DEBUG_PHASES"); dbg_on = (e && e[0] == '1') ? 1 : 0; }
  if (dbg_on) {
    if (!dbg) PSL_HIP(hipMalloc(&dbg, 64 * sizeof(unsigned long long)));
    PSL_HIP(hipMemsetAsync(dbg, 0, 64 * sizeof(unsigned long long), s));
    a.dbg = dbg;
  }
  { int rc = blk_trace_begin(a, n_tiles + n_wl, s); if (rc) return rc; }
  PSL_KLAUNCH(k_geo_iter, dim3(n_tiles + n_wl), dim3(64), 0, s, a, (const float*)ctx->wf, (const float*)ctx->wb, gr, g_geo,
                     row_map, ctx->touched_geo, w, n_tiles, n_wl);
  PSL_LAUNCH_CHECK();
  { int rc = blk_trace_end(a, "geo_iter", n_tiles + n_wl, n_tiles, 64); if (rc) return rc; }
  if (dbg_on) {
    unsigned long long h[8];
    PSL_HIP(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
    fprintf(stderr, "[psl geo_iter P=%d] nbr+weights %llu gather %llu frag-prefetch+sin %llu layers %llu | composite %llu | bwd layers %llu scatter %llu | total %llu\n",
            a.P, h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[7] - h[6], h[7] - h[0]);
  }
  return PSL_OK;
}

}  // namespace psl
