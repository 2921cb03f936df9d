// C ABI entry points: context lifecycle, parameter table, render forward/backward orchestration.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include "psl_decode.h"
#include "psl_adam.h"
#include "psl_frag.h"

namespace psl {
thread_local ProfArm g_prof_arm = {nullptr, nullptr, false};

static thread_local char g_err[512] = "";

bool debug_sync() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("PSL_DEBUG_SYNC"); v = (e && e[0] == '1') ? 1 : 0; }
  return v == 1;
}

void poison(void* p, size_t bytes) {
  static int v = -1;
  if (v < 0) { const char* e = getenv("PSL_POISON"); v = (e && e[0] == '1') ? 1 : 0; }
  if (v == 1 && p) { (void)hipMemset(p, 0x7F, bytes); (void)hipDeviceSynchronize(); }
}

void dbg_range(const char* name, const void* p, size_t bytes) {
  static int v = -1;
  if (v < 0) { const char* e = getenv("PSL_DEBUG_ADDRS"); v = (e && e[0] == '1') ? 1 : 0; }
  if (v == 1) fprintf(stderr, "[psl addr] %-14s %p .. %p (%zu B)\n", name, p, (const void*)((const char*)p + bytes), bytes);
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// PSL_DEBUG_BLOCKS=<file>: per-workgroup trace of the first launches of every decode kernel / launch size, appended to
// <file> as JSON lines {kernel, P, grid, color_tiles, threads, blocks: [[wall_start, wall_end, hw_id, cycles], ...]}
// (wall clock: 100 MHz constant counter; tools/block_trace.py turns them into per-CU timelines)
static const char* blk_trace_path() {
  static const char* p = nullptr; static int init = 0;
  if (!init) { p = getenv("PSL_DEBUG_BLOCKS"); if (p && !p[0]) p = nullptr; init = 1; }
  return p;
}
static unsigned long long* g_blk_buf = nullptr; static int g_blk_cap = 0;
int blk_trace_begin(DecodeArgs& a, int grid, hipStream_t s) {
  a.blk = nullptr;
  if (!blk_trace_path()) return PSL_OK;
  if (g_blk_cap < grid) {
    if (g_blk_buf) (void)hipFree(g_blk_buf);
    PSL_HIP(hipMalloc(&g_blk_buf, sizeof(unsigned long long) * 4 * (size_t)grid)); g_blk_cap = grid;
  }
  PSL_HIP(hipMemsetAsync(g_blk_buf, 0, sizeof(unsigned long long) * 4 * (size_t)grid, s));
  a.blk = g_blk_buf;
  return PSL_OK;
}
int blk_trace_end(const DecodeArgs& a, const char* kernel, int grid, int color_tiles, int threads) {
  if (!a.blk) return PSL_OK;
  static int seen[64][3]; static int n_seen = 0;      // (kernel hash, P) -> launches written
  int kh = 0; for (const char* c = kernel; *c; ++c) kh = kh * 31 + *c;
  int slot = -1;
  kh = kh * 31 + a.flags;                              // colour-stage and geometry-stage launches of one size are different kernels
  for (int i = 0; i < n_seen; ++i) if (seen[i][0] == kh && seen[i][1] == a.P) slot = i;
  if (slot < 0 && n_seen < 64) { slot = n_seen++; seen[slot][0] = kh; seen[slot][1] = a.P; seen[slot][2] = 0; }
  PSL_HIP(hipDeviceSynchronize());
  if (slot < 0 || seen[slot][2] >= 3) return PSL_OK;
  seen[slot][2]++;
  unsigned long long* h = (unsigned long long*)malloc(sizeof(unsigned long long) * 4 * (size_t)grid);
  PSL_HIP(hipMemcpy(h, a.blk, sizeof(unsigned long long) * 4 * (size_t)grid, hipMemcpyDeviceToHost));
  FILE* f = fopen(blk_trace_path(), "a");
  if (f) {
    fprintf(f, "{\"kernel\": \"%s\", \"P\": %d, \"flags\": %d, \"grid\": %d, \"color_tiles\": %d, \"threads\": %d, \"blocks\": [", kernel, a.P, a.flags,
            grid, color_tiles, threads);
    for (int b = 0; b < grid; ++b)
      fprintf(f, "%s[%llu,%llu,%llu,%llu]", b ? "," : "", h[4 * b], h[4 * b + 1], h[4 * b + 2], h[4 * b + 3]);
    fprintf(f, "]}\n");
    fclose(f);
  }
  free(h);
  return PSL_OK;
}

int launch_decode_fwd2(psl_ctx* ctx, const DecodeArgs& a, hipStream_t s);
int launch_decode_bwd(psl_ctx* ctx, const DecodeArgs& a, const psl_render_grads& g, hipStream_t s);
int launch_composite_fwd(const float4* raw, const float* z, const float* gt_depth, float near_s, float far_s,
                         const int* cnt, int min_nn, int n_rays, float coef, float* depth, float* var, float* rgb,
                         unsigned char* valid, float* cw, float* ray_aux, hipStream_t s);
int launch_composite_bwd(const float4* raw, const float* z, const float* gt_depth, float near_s, float far_s, int n_rays, float coef,
                         const float* g_depth, const float* g_var, const float* g_rgb, float4* d_raw, float* zero64,
                         hipStream_t s);
int launch_ray_grad(const float4* dp, const float4* dp2, const float* z, const float* gt_depth, float near_s, float far_s, int n_rays,
                    float* g_o, float* g_d, hipStream_t s);

static inline int64_t al4(int64_t x) { return (x + 3) & ~int64_t(3); }

RenderWs carve_ws(float* base, int n_rays, int flags) {
  RenderWs w;
  memset(&w, 0, sizeof(w));
  w.P = n_rays * S;
  w.Ppad = ((w.P + TILE - 1) / TILE) * TILE;
  const int64_t Pp = w.Ppad;
  const bool grad = flags & (PSL_PTS_GRAD | PSL_PARAM_GRAD | PSL_FEAT_GRAD);
  const bool color = flags & PSL_STAGE_COLOR;
  const bool relpos = flags & 0x10000;
  const bool pgrad = flags & PSL_PARAM_GRAD;
  int64_t off = 0;
  auto take = [&](int64_t n) -> float* { float* p = base ? base + off : nullptr; off += al4(n); return p; };
  w.I = (int*)take(Pp * K);
  w.cnt = (int*)take(Pp);
  w.raw = take(Pp * 4);
  w.w = take(Pp * K);
  w.dcc = take(Pp * C);
  w.cc = take(Pp * C);
  w.out3 = take(Pp * 4);
  if (color) w.c_emb2 = take(Pp * EC);
  w.cw = take((int64_t)n_rays * S);
  w.ray_aux = take((int64_t)n_rays * 4);
  if (grad) {
    w.g_y = take(Pp * 5 * HG);
    w.d_raw = take(Pp * 4);
    w.dp = take(Pp * 4);
    w.dp2 = take(Pp * 4);
    if (color) {
      w.c_y = take(Pp * 5 * HC);
      w.d_out3 = take(Pp * 4);
      if (relpos) {
        w.n_h1 = take(Pp * K * HC);
        if (flags & PSL_PTS_GRAD) w.n_out = take(Pp * K * C);     // only dL/dw -> dL/dp needs F_theta's outputs
      }
      if (pgrad) {      // operands of the parameter-gradient GEMM
        w.c_hin = take(Pp * 5 * HC);
        w.c_emb = take(Pp * EC);
        w.c_dz = take(Pp * 5 * HC);
        w.c_g = take(Pp * 5 * HC);
        if (relpos) { w.n_x = take(Pp * K * NX); w.n_dz1 = take(Pp * K * HC); w.n_dnf = take(Pp * K * C); }
      }
    }
  }
  w.total = off;
  return w;
}

static int fill_decode_args(psl_ctx* ctx, const psl_render_args* a, DecodeArgs& d) {
  memset(&d, 0, sizeof(d));
  int flags = a->flags;
  if (ctx->cfg.encode_rel_pos) flags |= 0x10000;
  if (ctx->cfg.nn_weighting == 1) flags |= kFlagExpoW;
  d.P = a->n_rays * S;
  d.n_rays = a->n_rays;
  d.flags = flags;
  d.min_nn = ctx->cfg.min_nn_num;
  d.near_s = ctx->cfg.near_end_surface;
  d.far_s = ctx->cfg.far_end_surface;
  d.r2_fixed = (float)((double)ctx->cfg.radius_query * (double)ctx->cfg.radius_query);
  d.rays_o = a->rays_o; d.rays_d = a->rays_d; d.depth = a->gt_depth; d.zv = a->z_vals; d.r_query = a->r_query;
  d.pos = ctx->pos;
  d.zero64 = ctx->fwd_zero64;
  d.geo_feats = a->geo_feats; d.col_feats = a->col_feats;
  d.master = a->params; d.Bcol = a->col_embed_B;
  d.fb_geo = a->fallback_geo; d.fb_col = a->fallback_col; d.affine = a->exposure_affine;
  d.ws = carve_ws(a->ws, a->n_rays, flags);
  if (ctx->pre_I) { d.ws.I = ctx->pre_I; d.ws.cnt = ctx->pre_cnt; }   // neighbours answered ahead (psl_map_iters)
  return PSL_OK;
}

static int check_render_args(psl_ctx* ctx, const psl_render_args* a, const char* who) {
  if (!ctx || !a) { set_error("%s: null argument", who); return PSL_ERR_ARG; }
  if (a->n_rays < 0) { set_error("%s: n_rays < 0", who); return PSL_ERR_ARG; }
  if (!a->rays_o || !a->rays_d || !a->gt_depth || !a->geo_feats || !a->params || !a->fallback_geo || !a->ws) {
    set_error("%s: missing required pointer", who); return PSL_ERR_ARG;
  }
  if ((a->flags & PSL_STAGE_COLOR) && (!a->col_feats || !a->col_embed_B || !a->fallback_col)) {
    set_error("%s: colour stage needs col_feats, col_embed_B, fallback_col", who); return PSL_ERR_ARG;
  }
  if ((a->flags & PSL_PTS_GRAD) && ctx->cfg.nn_weighting == 1) {
    // the reference cannot do it either: weights = exp(-20 sqrt(D)); weights[D > bound] = 0 modifies the exp's output in place and
    // autograd raises when the tracker back-propagates to the pose (decoder.py:154-157, 364-367)
    set_error("%s: nn_weighting 'expo' has no pose gradient (the reference raises in backward: in-place write on exp's output, decoder.py:157)", who);
    return PSL_ERR_UNSUPPORTED;
  }
  if ((a->flags & PSL_HAS_AFFINE) && !a->exposure_affine) { set_error("%s: PSL_HAS_AFFINE without affine", who); return PSL_ERR_ARG; }
  if (ctx->index_points != ctx->n_points) { set_error("%s: index is stale, call psl_index_build", who); return PSL_ERR_STATE; }
  return PSL_OK;
}

}  // namespace psl

using namespace psl;

extern "C" const char* psl_last_error(void) { return g_err; }
extern "C" int psl_abi_version(void) { return 7; }

extern "C" int psl_param_count(void) { return kNumParams; }
extern "C" int psl_param_color_count(void) { return kNumColorParams; }
extern "C" int psl_param_master_floats(void) { return kMasterFloats; }
extern "C" int psl_param_entry(int i, char* name_out, int name_cap, int* rows, int* cols, int* offset) {
  if (i < 0 || i >= kNumParams) { set_error("psl_param_entry: index %d out of range", i); return PSL_ERR_ARG; }
  if (name_out && name_cap > 0) { strncpy(name_out, kParams[i].name, name_cap - 1); name_out[name_cap - 1] = 0; }
  if (rows) *rows = kParams[i].rows;
  if (cols) *cols = kParams[i].cols;
  if (offset) *offset = poff(i);
  return PSL_OK;
}

extern "C" int psl_create(int device, const psl_config* cfg, psl_ctx** out) {
  if (!cfg || !out) { set_error("psl_create: null argument"); return PSL_ERR_ARG; }
  if (cfg->n_surface != S || cfg->nn_num != K || cfg->c_dim != C) {
    set_error("psl_create: this build is specialised for N_surface=5, nn_num=8, c_dim=32 (got %d,%d,%d)",
              cfg->n_surface, cfg->nn_num, cfg->c_dim);
    return PSL_ERR_UNSUPPORTED;
  }
  if (cfg->max_points <= 0 || cfg->max_query_radius <= 0.f) { set_error("psl_create: bad capacity/radius"); return PSL_ERR_ARG; }
  if (cfg->nn_weighting != 0 && cfg->nn_weighting != 1) { set_error("psl_create: nn_weighting must be 0 ('distance') or 1 ('expo')"); return PSL_ERR_ARG; }
  if (cfg->max_points > kMaxPointsScatter) {          // psl_decode2.h: gradient rows are addressed by 32-bit byte offsets
    set_error("psl_create: max_points %d exceeds %d", cfg->max_points, kMaxPointsScatter); return PSL_ERR_ARG;
  }
  PSL_HIP(hipSetDevice(device));
  psl_ctx* c = new psl_ctx();
  memset(c, 0, sizeof(*c));
  c->device = device;
  c->cfg = *cfg;
  c->index_points = -1;
  size_t np = (size_t)cfg->max_points;
  PSL_HIP(hipMalloc(&c->pos, sizeof(float4) * np)); psl::poison(c->pos, sizeof(float4) * np);
  PSL_HIP(hipMalloc(&c->spos, sizeof(float4) * np)); psl::poison(c->spos, sizeof(float4) * np);
  PSL_HIP(hipMalloc(&c->cell_of, sizeof(int) * np)); psl::poison(c->cell_of, sizeof(int) * np);
  PSL_HIP(hipMalloc(&c->cell_start, sizeof(int) * (kMaxCells + 1))); psl::poison(c->cell_start, sizeof(int) * (kMaxCells + 1));
  PSL_HIP(hipMalloc(&c->cell_fill, sizeof(int) * kMaxCells)); psl::poison(c->cell_fill, sizeof(int) * kMaxCells);
  PSL_HIP(hipMalloc(&c->coarse, sizeof(int) * kMaxCoarse)); psl::poison(c->coarse, sizeof(int) * kMaxCoarse);
  PSL_HIP(hipMalloc(&c->scan_tmp, sizeof(int) * 4096)); psl::poison(c->scan_tmp, sizeof(int) * 4096);
  PSL_HIP(hipMalloc(&c->bounds, sizeof(int) * 8)); psl::poison(c->bounds, sizeof(int) * 8);
  PSL_HIP(hipMalloc(&c->meta, sizeof(GridMeta))); psl::poison(c->meta, sizeof(GridMeta));
  PSL_HIP(hipMalloc(&c->wf, sizeof(float) * kFFloats)); psl::poison(c->wf, sizeof(float) * kFFloats);
  PSL_HIP(hipMalloc(&c->wb, sizeof(float) * kBFloats)); psl::poison(c->wb, sizeof(float) * kBFloats);
  PSL_HIP(hipMalloc(&c->wf_index, sizeof(int) * kColorFloats));
  PSL_HIP(hipMalloc(&c->wb_index, sizeof(int) * kColorFloats));
  { int rc = build_frag_index(c, nullptr); if (rc) return rc;
    PSL_HIP(hipStreamSynchronize(nullptr)); }
  PSL_HIP(hipMalloc(&c->knn_cand, sizeof(unsigned long long) * 8 * kKnnCandSlots)); PSL_HIP(hipMemset(c->knn_cand, 0, sizeof(unsigned long long) * 8 * kKnnCandSlots));
  PSL_HIP(hipMalloc(&c->adam_rows, sizeof(unsigned long long) * 2 * kAdamRowSlots)); PSL_HIP(hipMemset(c->adam_rows, 0, sizeof(unsigned long long) * 2 * kAdamRowSlots));
  PSL_HIP(hipMalloc(&c->d_counter, sizeof(int) * 4)); psl::poison(c->d_counter, sizeof(int) * 4);
  {
    // scratch of the fused loops, sized once for the capacity of this context (nothing is allocated inside psl_map_iters
    // unless a call exceeds these defaults: 8 192 iterations, 16 384 rays per iteration)
    const size_t ns = np, lcap = std::min<size_t>(2 * (size_t)16384 * S * K, ns);
    const size_t need = (3 * ns + kStageTabIters + lcap) * sizeof(int) + 2 * ns;
    PSL_HIP(hipMalloc(&c->touched, need)); c->touched_cap = need;
    PSL_HIP(hipMalloc(&c->adam_tab, sizeof(float4) * (kStageTabIters + 64))); c->adam_tab_cap = kStageTabIters + 64;
    PSL_HIP(hipMalloc(&c->loss_acc, sizeof(double) * 4 * kLossSlots * kStageTabIters)); c->loss_acc_cap = (int)kStageTabIters;
    PSL_HIP(hipHostMalloc((void**)&c->h_stage, 4 * kStageSlot, hipHostMallocDefault));
    for (int i = 0; i < 4; ++i) PSL_HIP(hipEventCreateWithFlags(&c->ev_stage[i], hipEventDisableTiming));
  }
  PSL_HIP(hipMalloc(&c->d_small, sizeof(float) * 64)); psl::poison(c->d_small, sizeof(float) * 64);
  PSL_HIP(hipMalloc(&c->d_expo, sizeof(float) * 64 * (12 + 128 + 12))); psl::poison(c->d_expo, sizeof(float) * 64 * (12 + 128 + 12));
  dbg_range("pos", c->pos, sizeof(float4) * np); dbg_range("spos", c->spos, sizeof(float4) * np);
  dbg_range("cell_of", c->cell_of, sizeof(int) * np); dbg_range("cell_start", c->cell_start, sizeof(int) * (kMaxCells + 1));
  dbg_range("cell_fill", c->cell_fill, sizeof(int) * kMaxCells); dbg_range("coarse", c->coarse, sizeof(int) * kMaxCoarse);
  dbg_range("scan_tmp", c->scan_tmp, sizeof(int) * 4096);
  dbg_range("wf", c->wf, sizeof(float) * kFFloats);
  dbg_range("wb", c->wb, sizeof(float) * kBFloats); dbg_range("wf_index", c->wf_index, sizeof(int) * kColorFloats);
  dbg_range("wb_index", c->wb_index, sizeof(int) * kColorFloats); dbg_range("d_small", c->d_small, 256);
  dbg_range("d_expo", c->d_expo, sizeof(float) * 64 * (12 + 128 + 12)); dbg_range("adam_rows", c->adam_rows, 16 * kAdamRowSlots);
  *out = c;
  return PSL_OK;
}

extern "C" int psl_comm_destroy(psl_ctx* ctx);
extern "C" void psl_destroy(psl_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  (void)psl_comm_destroy(c);
  (void)hipFree(c->pos); (void)hipFree(c->spos); (void)hipFree(c->cell_of); (void)hipFree(c->cell_start);
  (void)hipFree(c->cell_fill); (void)hipFree(c->coarse); (void)hipFree(c->scan_tmp); (void)hipFree(c->bounds); (void)hipFree(c->meta);
  (void)hipFree(c->wf); (void)hipFree(c->wb); (void)hipFree(c->wf_index);
  (void)hipFree(c->wb_index); (void)hipFree(c->d_counter); (void)hipFree(c->d_small); (void)hipFree(c->d_expo); if (c->trk_pref) (void)hipFree(c->trk_pref); (void)hipFree(c->knn_cand); (void)hipFree(c->adam_rows); if (c->adam_tab) (void)hipFree(c->adam_tab); if (c->touched) (void)hipFree(c->touched); if (c->loss_acc) (void)hipFree(c->loss_acc); if (c->img_hist) (void)hipFree(c->img_hist);
  if (c->dw_slabs) (void)hipFree(c->dw_slabs);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  for (int i = 0; i < 4; ++i) if (c->ev_stage[i]) (void)hipEventDestroy(c->ev_stage[i]);
  if (c->stream2) {
    (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2);
    (void)hipEventDestroy(c->ev_knn_ready[0]); (void)hipEventDestroy(c->ev_knn_ready[1]); (void)hipEventDestroy(c->ev_knn_free);
  }
  if (c->scan_flags) (void)hipFree(c->scan_flags);
  if (c->ev) { for (size_t i = 0; i < (size_t)PROF_N * PROF_RING * 2; ++i) (void)hipEventDestroy(c->ev[i]); delete[] c->ev; }
  delete c;
}

extern "C" int64_t psl_render_ws_floats(int n_rays, int flags) {
  if (n_rays < 0) return PSL_ERR_ARG;
  // size for the worst case of the rel-pos bit (the ctx decides it at run time)
  return carve_ws(nullptr, n_rays, flags | 0x10000).total;
}

namespace psl {
int render_fwd_impl(psl_ctx* ctx, const psl_render_args* a, hipStream_t s, bool repack) {
  int rc = check_render_args(ctx, a, "psl_render_fwd");
  if (rc) return rc;
  if (!a->depth || !a->var || !a->rgb) { set_error("psl_render_fwd: missing output pointer"); return PSL_ERR_ARG; }
  if (a->n_rays == 0) return PSL_OK;
  DecodeArgs d;
  fill_decode_args(ctx, a, d);
  if (repack) { ProfScope ps(ctx, PROF_MISC, s); rc = repack_frags(ctx, a->params, s); if (rc) return rc; }
  if (!ctx->pre_I)
  { ProfScope ps(ctx, PROF_KNN, s, 108.0 * d.P, true);   // lower bound: query + 8 neighbour positions
    rc = knn_rays(ctx, a->rays_o, a->rays_d, a->gt_depth, a->z_vals, a->r_query, a->n_rays, d.ws.I, d.ws.cnt, s);
    if (rc) return rc; }
  { ProfScope ps(ctx, prof_decode_slot(d.flags, false), s, fwd_flops_per_sample(d.flags) * d.P, true);
    rc = launch_decode_fwd2(ctx, d, s); if (rc) return rc; }
  if (!ctx->fused_ray)     // psl_map_iters composites, takes the loss and back-propagates it in one kernel of its own
  { ProfScope ps(ctx, PROF_COMPOSITE, s, 124.0 * a->n_rays, true);
    rc = launch_composite_fwd((const float4*)d.ws.raw, a->z_vals, a->gt_depth, d.near_s, d.far_s, d.ws.cnt, d.min_nn,
                              a->n_rays, a->sigmoid_coef, a->depth, a->var, a->rgb, a->valid_ray, d.ws.cw,
                              d.ws.ray_aux, s);
    if (rc) return rc; }
  return PSL_OK;
}

int render_bwd_impl(psl_ctx* ctx, const psl_render_args* a, const psl_render_grads* g, hipStream_t s) {
  int rc = check_render_args(ctx, a, "psl_render_bwd");
  if (rc) return rc;
  if (!g || (!g->g_depth && !ctx->fused_ray)) { set_error("psl_render_bwd: missing cotangents"); return PSL_ERR_ARG; }
  if (!(a->flags & (PSL_PTS_GRAD | PSL_PARAM_GRAD | PSL_FEAT_GRAD))) {
    set_error("psl_render_bwd: forward was run without any gradient flag"); return PSL_ERR_STATE;
  }
  if (a->n_rays == 0) return PSL_OK;
  DecodeArgs d;
  fill_decode_args(ctx, a, d);
  if (!ctx->fused_ray)
  { ProfScope ps(ctx, PROF_COMPOSITE_BWD, s, 200.0 * a->n_rays, true);
    rc = launch_composite_bwd((const float4*)d.ws.raw, a->z_vals, a->gt_depth, d.near_s, d.far_s, a->n_rays, a->sigmoid_coef,
                              g->g_depth, g->g_var, g->g_rgb, (float4*)d.ws.d_raw, ctx->d_small, s);
    if (rc) return rc; }
  rc = launch_decode_bwd(ctx, d, *g, s);
  if (rc) return rc;
  if ((a->flags & PSL_PTS_GRAD) && (g->g_rays_o || g->g_rays_d)) {
    ProfScope ps(ctx, PROF_MISC, s);
    rc = launch_ray_grad((const float4*)d.ws.dp, (const float4*)d.ws.dp2, a->z_vals,
                         a->gt_depth, d.near_s, d.far_s, a->n_rays, g->g_rays_o, g->g_rays_d, s);
    if (rc) return rc;
  }
  return PSL_OK;
}
}  // namespace psl

namespace psl {
int launch_geo_iter(psl_ctx* ctx, const DecodeArgs& a, const GeoIterRays& gr, float* g_geo, const int* row_map,
                    const AdamWorklist* wl, hipStream_t s);
// one geometry-stage mapper iteration (psl_decode_geo.hip): needs the neighbour lists answered ahead (psl_map_iters)
int geo_iter_impl(psl_ctx* ctx, const psl_render_args* a, const psl_render_grads* g, const int* active, double* loss_acc,
                  const AdamWorklist* wl, hipStream_t s, bool repack) {
  int rc = check_render_args(ctx, a, "psl_map_iters(geometry iteration)");
  if (rc) return rc;
  if (!ctx->pre_I || !g || !g->g_geo_feats) { set_error("geo_iter_impl: neighbour lists / gradient buffer missing"); return PSL_ERR_STATE; }
  DecodeArgs d;
  fill_decode_args(ctx, a, d);
  if (repack) { ProfScope ps(ctx, PROF_MISC, s); rc = repack_frags(ctx, a->params, s); if (rc) return rc; }
  GeoIterRays gr{active, a->sigmoid_coef, a->depth, a->var, a->rgb, a->valid_ray, loss_acc, ctx->d_small, a->n_rays};
  ProfScope ps(ctx, PROF_GEO_ITER, s, (fwd_flops_per_sample(d.flags) + bwd_flops_per_sample(d.flags)) * d.P, true);
  return launch_geo_iter(ctx, d, gr, g->g_geo_feats, g->feat_row_map, wl, s);
}
}  // namespace psl

extern "C" int psl_render_fwd(psl_ctx* ctx, const psl_render_args* a, void* stream) {
  return render_fwd_impl(ctx, a, (hipStream_t)stream, true);
}
extern "C" int psl_render_bwd(psl_ctx* ctx, const psl_render_args* a, const psl_render_grads* g, void* stream) {
  return render_bwd_impl(ctx, a, g, (hipStream_t)stream);
}

extern "C" int psl_sync(psl_ctx* ctx, void* stream) {
  if (!ctx) return PSL_ERR_ARG;
  PSL_HIP(hipStreamSynchronize((hipStream_t)stream));
  return PSL_OK;
}

static const char* kProfNames[PROF_N] = {"knn", "decode_fwd", "composite_fwd", "composite_bwd", "decode_bwd", "dw_gemm",
                                         "adam", "misc", "decode_fwd_geo", "decode_bwd_geo", "decode_fwd_track",
                                         "decode_bwd_track", "knn_side_stream", "knn_prefetch", "geo_iter", "adam_dense"};
extern "C" const char* psl_profile_name(int i) { return (i >= 0 && i < PROF_N) ? kProfNames[i] : ""; }
extern "C" int psl_profile_classes(void) { return PROF_N; }

namespace psl { extern int g_knn_start_hint; extern int g_knn_version, g_lazy_adam, g_track_fused, g_dw_fused, g_knn_overlap, g_geo_fused, g_ray_in_bwd, g_remap_cv2; int knn_trace_dump(); }
// debug / A-B switch settable at run time (tests compare kernel generations inside one process)
extern "C" int psl_debug_option(const char* name, int value) {
  if (!name) return PSL_ERR_ARG;
  if (!strcmp(name, "knn")) { psl::g_knn_version = value; return PSL_OK; }
  if (!strcmp(name, "color_split")) { psl::g_color_split = value; return PSL_OK; }
  if (!strcmp(name, "wave_trunk")) { psl::g_wave_trunk_tiles = value; return PSL_OK; }
  if (!strcmp(name, "lazy_adam")) { psl::g_lazy_adam = value; return PSL_OK; }
  if (!strcmp(name, "track_fused")) { psl::g_track_fused = value; return PSL_OK; }
  if (!strcmp(name, "dw_fused")) { psl::g_dw_fused = value; return PSL_OK; }
  if (!strcmp(name, "knn_overlap")) { psl::g_knn_overlap = value; return PSL_OK; }
  if (!strcmp(name, "geo_fused")) { psl::g_geo_fused = value; return PSL_OK; }
  if (!strcmp(name, "ray_in_bwd")) { psl::g_ray_in_bwd = value; return PSL_OK; }
  if (!strcmp(name, "remap_cv2")) { psl::g_remap_cv2 = value; return PSL_OK; }
  if (!strcmp(name, "knn_trace_dump")) return psl::knn_trace_dump();
  if (!strcmp(name, "knn_start_hint")) { psl::g_knn_start_hint = value; return PSL_OK; }
  set_error("psl_debug_option: unknown option %s", name);
  return PSL_ERR_ARG;
}

// candidates (16-byte position records) the ray k-NN has examined since the previous call; synchronises and resets
extern "C" int64_t psl_knn_candidates(psl_ctx* ctx) {
  if (!ctx) return PSL_ERR_ARG;
  unsigned long long v = 0, slots[8 * kKnnCandSlots];
  if (hipDeviceSynchronize() != hipSuccess) return PSL_ERR_HIP;
  if (hipMemcpy(slots, ctx->knn_cand, sizeof(slots), hipMemcpyDeviceToHost) != hipSuccess) return PSL_ERR_HIP;
  if (hipMemset(ctx->knn_cand, 0, sizeof(slots)) != hipSuccess) return PSL_ERR_HIP;
  for (int k = 0; k < 8 * kKnnCandSlots; k += 8) v += slots[k];
  return (int64_t)v;
}

extern "C" int psl_profile_enable(psl_ctx* ctx, int on) {
  if (!ctx) return PSL_ERR_ARG;
  if (on && !ctx->ev) {
    size_t n = (size_t)PROF_N * PROF_RING * 2;
    ctx->ev = new hipEvent_t[n];
    for (size_t i = 0; i < n; ++i) PSL_HIP(hipEventCreate(&ctx->ev[i]));
  }
  ctx->prof_on = on;
  if (on) {
    memset(ctx->prof_count, 0, sizeof(ctx->prof_count)); memset(ctx->prof_work, 0, sizeof(ctx->prof_work));
    memset(ctx->prof_seen, 0, sizeof(ctx->prof_seen));
    PSL_HIP(hipMemset(ctx->adam_rows, 0, sizeof(unsigned long long) * 2 * kAdamRowSlots));
  }
  return PSL_OK;
}

// Per kernel class since psl_profile_enable(n): total device time (ms), launch count and the algorithmic work (FLOP or
// bytes) of ALL launches of the class.  Synchronises the device.  The time is the mean of the bracketed launches (one in
// n; at most the last PROF_RING of them) times the launch count.
extern "C" int psl_profile_read(psl_ctx* ctx, double* ms_out, int* count_out, double* work_out, int cap) {
  if (!ctx || !ms_out || !count_out || !work_out) return PSL_ERR_ARG;
  PSL_HIP(hipDeviceSynchronize());
  int n = std::min(cap, (int)PROF_N);
  for (int i = 0; i < n; ++i) {
    int cnt = ctx->prof_count[i];
    int have = std::min(cnt, PROF_RING);
    double tot = 0.0;
    for (int k = 0; k < have; ++k) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ctx->ev[((size_t)i * PROF_RING + k) * 2], ctx->ev[((size_t)i * PROF_RING + k) * 2 + 1]) == hipSuccess)
        tot += ms;
    }
    ms_out[i] = have > 0 ? tot * ((double)ctx->prof_seen[i] / have) : 0.0;
    count_out[i] = ctx->prof_seen[i];
    work_out[i] = ctx->prof_work[i];
  }
  // feature rows the lazy Adam stepped: 5 streams x 4 B x 32 channels each; work-list launches and the
  // block-end catch-ups count into separate halves of adam_rows
  for (int half = 0; half < 2; ++half) {
    const int cls = half ? (int)PROF_ADAM_DENSE : (int)PROF_ADAM;
    if (n <= cls) continue;
    unsigned long long rows = 0, slots[kAdamRowSlots];
    PSL_HIP(hipMemcpy(slots, ctx->adam_rows + (size_t)half * kAdamRowSlots, sizeof(slots), hipMemcpyDeviceToHost));
    for (int k = 0; k < kAdamRowSlots; k += 8) rows += slots[k];
    work_out[cls] += 20.0 * C * (double)rows;
  }
  return n;
}

// ---------------------------------------------------------------------------------------------- fast-math unit harness
// The hot kernels replace libm / IEEE sequences by hardware transcendentals (psl_device.h, psl_adam.h).  They are bounded
// end to end by the parity tests; this entry point lets tests/test_hip_fastmath.py pin each helper BY ITSELF against its
// exact counterpart (torch in float64), so that the next person to touch one has a unit bound to keep.
namespace psl {
__global__ void k_selftest_math(int kind, const float* __restrict__ in, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  switch (kind) {
    case PSL_SELFTEST_SINCOS: { float s, c; fast_sincosf(in[i], s, c); out[2 * i] = s; out[2 * i + 1] = c; break; }
    case PSL_SELFTEST_SOFTPLUS: out[i] = softplus100(in[i]); break;
    case PSL_SELFTEST_SOFTPLUS_NB: out[i] = softplus100_nb(in[i]); break;
    case PSL_SELFTEST_SOFTPLUS_GRAD: out[i] = softplus100_grad_from_out(in[i]); break;
    case PSL_SELFTEST_ADAM_REPLAY: {
      // in[i] = (p, m, v, lr_bc1, sqrt_bc2, n_steps): n_steps gradient-free steps, once with the replay arithmetic of the
      // lazy Adam and once with the dense kernel's IEEE sequence (g = 0); out[i] = (p, m, v)_replay, (p, m, v)_ieee
      const float* a = in + 6 * (size_t)i;
      float p = a[0], m = a[1], v = a[2], p2 = a[0], m2 = a[1], v2 = a[2];
      const int steps = (int)a[5];
      for (int t = 0; t < steps; ++t) {
        adam_replay(p, m, v, a[3], __builtin_amdgcn_rcpf(a[4]), 0.9f, 0.999f, 1e-8f);
        adam_update(p2, 0.f, m2, v2, a[3], a[4], 0.9f, 0.999f, 1e-8f);
      }
      float* o = out + 6 * (size_t)i;
      o[0] = p; o[1] = m; o[2] = v; o[3] = p2; o[4] = m2; o[5] = v2;
      break;
    }
    default: break;
  }
}
}  // namespace psl

// ---- known-traffic kernels in the access patterns of the hot path: what FETCH_SIZE / WRITE_SIZE report is calibrated on THEM
// (the microarchitecture guide calibrates only the 16 B/lane streaming read and says "calibrate on a known byte count in your
// own access pattern before trusting an absolute").  Launch them under the same rocprofv3 --pmc passes as the probe.
namespace psl {
// kind 0: streaming read, 16 B per lane (weight fragments, saved activations): n float4 read, one float written per workgroup
__global__ __launch_bounds__(256) void k_traffic_stream_read(const float4* __restrict__ src, float* __restrict__ dst, long long n) {
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float4 v = src[i];
    acc += (v.x + v.y) + (v.z + v.w);
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(dst + (blockIdx.x & 1023), acc);
}
// kind 1: streaming write, 16 B per lane (the forward's saved rows)
__global__ __launch_bounds__(256) void k_traffic_stream_write(float4* __restrict__ dst, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dst[i] = make_float4((float)i, 1.f, 2.f, 3.f);
}
// kind 2: row gather as the decode tiles gather feature rows: lane (row slot = lane & 15, g = lane >> 4) reads 16 B at
// rows[.] * 128 B + 16 g and + 64 + 16 g: four lanes cover one 128-byte row with two loads each; n rows
__global__ __launch_bounds__(256) void k_traffic_row_gather(const float* __restrict__ table, const int* __restrict__ rows,
                                                            float* __restrict__ dst, long long n) {
  const int lane = threadIdx.x & 63, rl = lane & 15, g = lane >> 4;
  float acc = 0.f;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long b = wave * 16; b < n; b += nw * 16) {
    const long long i = b + rl;
    if (i < n) {
      const float* row = table + (size_t)rows[i] * 32 + 4 * g;
      const float4 v0 = *reinterpret_cast<const float4*>(row), v1 = *reinterpret_cast<const float4*>(row + 16);
      acc += (v0.x + v0.y) + (v0.z + v0.w) + (v1.x + v1.y) + (v1.z + v1.w);
    }
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) atomicAdd(dst + (blockIdx.x & 1023), acc);
}
// kind 3: gradient scatter as scatter_interp_rows issues it: 32 consecutive lanes add one float each to ONE 128-byte row
// (two rows per wave-instruction), float atomics; n rows
__global__ __launch_bounds__(256) void k_traffic_row_scatter(float* __restrict__ table, const int* __restrict__ rows, long long n) {
  const int lane = threadIdx.x & 63, half = lane >> 5, c = lane & 31;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long b = wave * 2; b < n; b += nw * 2) {
    const long long i = b + half;
    if (i < n) atomicAdd(table + (size_t)rows[i] * 32 + c, 1.0f);
  }
}
}  // namespace psl

extern "C" int psl_selftest_traffic(int kind, float* table, const int32_t* rows, float* scratch, long long n, void* stream) {
  // kind | 0x100: the same kernel on a slightly smaller grid, so that a profile keeps two uses of one kernel apart
  const int blocks = (kind & 0x100) ? 4064 : 4096;
  kind &= 0xff;
  if (kind < 0 || kind > 3 || !table || n <= 0 || ((kind == 2 || kind == 3) && !rows) || ((kind == 0 || kind == 2) && !scratch)) {
    set_error("psl_selftest_traffic: bad argument"); return PSL_ERR_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  if (kind == 0) hipLaunchKernelGGL(psl::k_traffic_stream_read, dim3(blocks), dim3(256), 0, s, (const float4*)table, scratch, n);
  else if (kind == 1) hipLaunchKernelGGL(psl::k_traffic_stream_write, dim3(blocks), dim3(256), 0, s, (float4*)table, n);
  else if (kind == 2) hipLaunchKernelGGL(psl::k_traffic_row_gather, dim3(blocks), dim3(256), 0, s, (const float*)table, rows, scratch, n);
  else hipLaunchKernelGGL(psl::k_traffic_row_scatter, dim3(blocks), dim3(256), 0, s, table, rows, n);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}

extern "C" int psl_selftest_math(int kind, const float* in, float* out, int n, void* stream) {
  if (kind < 0 || kind > PSL_SELFTEST_ADAM_REPLAY || !in || !out || n < 0) { set_error("psl_selftest_math: bad argument"); return PSL_ERR_ARG; }
  if (n == 0) return PSL_OK;
  hipLaunchKernelGGL(psl::k_selftest_math, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, kind, in, out, n);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}
