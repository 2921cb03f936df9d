// Internal definitions shared by the HIP translation units of libpointslam_hip.so.
// gfx950 only: 64-wide wavefronts, f32 MFMA 16x16x4, 160 KiB LDS per CU.
#pragma once
#include <cstdio>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include "../../include/pointslam_hip.h"

namespace psl {

// ---------------------------------------------------------------- constants
constexpr int S = 5;    // samples per ray        (configs/point_slam.yaml:95)
constexpr int K = 8;    // neighbours per sample   (configs/point_slam.yaml:107)
constexpr int C = 32;   // feature channels        (configs/point_slam.yaml:10)
constexpr int HG = 32;  // geometry decoder hidden (decoder.py:468-470)
constexpr int HC = 128; // colour decoder hidden   (decoder.py:471-474)
constexpr int EG = 93;  // geometry Fourier size (sin only)   (decoder.py:99-104)
constexpr int EGP = 96; // ... padded to a multiple of 4 for the MFMA k-step
constexpr int ECF = 20; // colour Fourier frequencies (sin+cos -> 40) (decoder.py:302-306)
constexpr int EC = 40;
constexpr int kMaxPointsScatter = 1 << 25;   // the gradient scatters address a row by a 32-bit byte offset (row * 128 B): psl_decode2.h
constexpr int ERF = 10; // rel-pos Fourier frequencies (sin+cos -> 20) (decoder.py:314-315)
constexpr int ER = 20;
constexpr int NX = ER + C;  // neighbour-MLP input width 52 (decoder.py:316-317)
constexpr int TILE = 16;    // samples per workgroup tile (= M of mfma_f32_16x16x4f32)
constexpr int WG = 512;     // threads per decode workgroup (8 waves)
constexpr float TWO_PI = 6.2831855f;  // float32(2*pi), decoder.py:33

// ------------------------------------------------ master parameter blob layout
// torch layouts ([out][in]) concatenated; colour group first (see include/pointslam_hip.h)
struct PEntry { const char* name; int rows, cols; };
constexpr PEntry kParams[] = {
    {"color_decoder.fc_c.0.weight", HC, C}, {"color_decoder.fc_c.0.bias", HC, 1},
    {"color_decoder.fc_c.1.weight", HC, C}, {"color_decoder.fc_c.1.bias", HC, 1},
    {"color_decoder.fc_c.2.weight", HC, C}, {"color_decoder.fc_c.2.bias", HC, 1},
    {"color_decoder.fc_c.3.weight", HC, C}, {"color_decoder.fc_c.3.bias", HC, 1},
    {"color_decoder.fc_c.4.weight", HC, C}, {"color_decoder.fc_c.4.bias", HC, 1},
    {"color_decoder.embedder_rel_pos._B", 3, ERF},
    {"color_decoder.mlp_col_neighbor.linear1.weight", HC, NX}, {"color_decoder.mlp_col_neighbor.linear1.bias", HC, 1},
    {"color_decoder.mlp_col_neighbor.linear2.weight", C, HC}, {"color_decoder.mlp_col_neighbor.linear2.bias", C, 1},
    {"color_decoder.pts_linears.0.weight", HC, EC}, {"color_decoder.pts_linears.0.bias", HC, 1},
    {"color_decoder.pts_linears.1.weight", HC, HC}, {"color_decoder.pts_linears.1.bias", HC, 1},
    {"color_decoder.pts_linears.2.weight", HC, HC}, {"color_decoder.pts_linears.2.bias", HC, 1},
    {"color_decoder.pts_linears.3.weight", HC, EC + HC}, {"color_decoder.pts_linears.3.bias", HC, 1},
    {"color_decoder.pts_linears.4.weight", HC, HC}, {"color_decoder.pts_linears.4.bias", HC, 1},
    {"color_decoder.output_linear.weight", 3, HC}, {"color_decoder.output_linear.bias", 3, 1},
    // ---- geometry group
    {"geo_decoder.fc_c.0.weight", HG, C}, {"geo_decoder.fc_c.0.bias", HG, 1},
    {"geo_decoder.fc_c.1.weight", HG, C}, {"geo_decoder.fc_c.1.bias", HG, 1},
    {"geo_decoder.fc_c.2.weight", HG, C}, {"geo_decoder.fc_c.2.bias", HG, 1},
    {"geo_decoder.fc_c.3.weight", HG, C}, {"geo_decoder.fc_c.3.bias", HG, 1},
    {"geo_decoder.fc_c.4.weight", HG, C}, {"geo_decoder.fc_c.4.bias", HG, 1},
    {"geo_decoder.embedder._B", 3, EG},
    {"geo_decoder.pts_linears.0.weight", HG, EG}, {"geo_decoder.pts_linears.0.bias", HG, 1},
    {"geo_decoder.pts_linears.1.weight", HG, HG}, {"geo_decoder.pts_linears.1.bias", HG, 1},
    {"geo_decoder.pts_linears.2.weight", HG, HG}, {"geo_decoder.pts_linears.2.bias", HG, 1},
    {"geo_decoder.pts_linears.3.weight", HG, EG + HG}, {"geo_decoder.pts_linears.3.bias", HG, 1},
    {"geo_decoder.pts_linears.4.weight", HG, HG}, {"geo_decoder.pts_linears.4.bias", HG, 1},
    {"geo_decoder.output_linear.weight", 1, HG}, {"geo_decoder.output_linear.bias", 1, 1},
};
constexpr int kNumParams = sizeof(kParams) / sizeof(kParams[0]);
constexpr int kNumColorParams = 27;

constexpr int poff(int i) {  // master offset (floats) of entry i
  int o = 0;
  for (int j = 0; j < i; ++j) o += kParams[j].rows * kParams[j].cols;
  return o;
}
constexpr int kMasterFloats = poff(kNumParams);
constexpr int kColorFloats = poff(kNumColorParams);
// per-chunk partial dW slabs of the colour decoder (psl_dw.hip): master layout with every tensor start rounded up to 4
struct DwReduceArgs { int chunks_of_entry[kNumColorParams]; int slab_off[kNumColorParams]; };
constexpr int kDwSlabStride = kColorFloats + 4 * kNumColorParams;

// indices into kParams
constexpr int PI_C_FCC = 0;    // + 2*i (weight), +2*i+1 (bias)
constexpr int PI_C_BREL = 10;
constexpr int PI_C_N1 = 11, PI_C_N2 = 13;
constexpr int PI_C_L = 15;     // + 2*i
constexpr int PI_C_OUT = 25;
constexpr int PI_G_FCC = 27;
constexpr int PI_G_B = 37;
constexpr int PI_G_L = 38;
constexpr int PI_G_OUT = 48;

// ---------------------------------------------------------- grid (spatial index)
constexpr int kMaxCells = 1 << 22;
struct GridMeta {      // lives in device memory; written by k_grid_meta
  float ox, oy, oz;    // origin (min corner)
  float inv_cell;      // 1 / cell size
  float cell;
  int nx, ny, nz;
  int ncells;
  int npts;
  int cnx, cny, cnz;   // coarse occupancy grid: blocks of 4 x 4 x 4 cells (block edge >= the largest query radius)
};
constexpr int kMaxCoarse = kMaxCells / 4 + 1024;   // worst case: a grid that is one cell thick in two dimensions

// --------------------------------------------------------------------- context
// decode classes are kept apart by launch type: colour-stage mapper batch, geometry-stage mapper batch (13x fewer
// FLOP per sample, another kernel), tracker batch (a fifth of the samples, pose gradients)
enum ProfSlot { PROF_KNN = 0, PROF_DECODE_FWD, PROF_COMPOSITE, PROF_COMPOSITE_BWD, PROF_DECODE_BWD, PROF_DW,
                PROF_ADAM, PROF_MISC, PROF_DECODE_FWD_GEO, PROF_DECODE_BWD_GEO, PROF_DECODE_FWD_TRK, PROF_DECODE_BWD_TRK,
                PROF_KNN_SIDE,   // the mapper's k-NN block prefetch while it runs on the side stream (off the critical path)
                PROF_KNN_PREFETCH,   // ... and on the main stream (first block of a call): the per-ray kernel, 10^4..10^5 rays per launch
                PROF_GEO_ITER,       // geometry-stage mapper iteration in one launch (decode fwd + compositing + loss + decode bwd)
                PROF_ADAM_DENSE,     // the lazy Adam's block-end catch-up over every selected row (and the dense Adam when lazy is off)
                PROF_N };
inline int prof_decode_slot(int flags, bool bwd) {
  if (!(flags & PSL_STAGE_COLOR)) return bwd ? PROF_DECODE_BWD_GEO : PROF_DECODE_FWD_GEO;
  if (flags & PSL_PTS_GRAD) return bwd ? PROF_DECODE_BWD_TRK : PROF_DECODE_FWD_TRK;
  return bwd ? PROF_DECODE_BWD : PROF_DECODE_FWD;
}
constexpr int PROF_RING = 4096;
constexpr size_t kStageFrames = 64 * 96, kStageTabIters = 8192, kStageSlot = kStageFrames + kStageTabIters * 16;   // bytes

}  // namespace psl

struct psl_ctx {
  int device;
  psl_config cfg;
  // positions, original (append) order, float4 {x,y,z,0}
  float4* pos;
  int n_points;          // host mirror
  int index_points;      // number of points covered by the current index (-1: none)
  // grid index
  psl::GridMeta* meta;   // device
  float4* spos;          // positions sorted by cell; .w = original index (int bits)
  int* cell_of;          // [max_points]
  int* cell_start;       // [kMaxCells + 1] exclusive prefix sums
  int* cell_fill;        // [kMaxCells]
  int* coarse = nullptr; // [kMaxCoarse] points per 4x4x4 block of cells: lets a query in empty space stop at once
  int* scan_tmp;         // block sums for the scan
  int* bounds;           // 6 ints: ordered-int min/max
  // fragment-major weight copies of the register-chained decode kernels (psl_frag.h) and their inverse maps
  float* wf = nullptr;       // forward fragments + aligned biases
  float* wb = nullptr;       // backward (transposed) fragments
  int* wf_index = nullptr;   // [kColorFloats] master element -> element of wf (or -1)
  int* wb_index = nullptr;   // [kColorFloats] master element -> element of wb (or -1)
  // dW partial slabs
  float* dw_slabs;
  int dw_slab_cap;       // number of slabs allocated
  // small device scratch
  // psl_map_iters: rows of the compact gradient / Adam state that have received a gradient since the frame started.
  // A row that never did has g = m = v = 0, so its Adam update is exactly zero and the row is skipped (the reference
  // steps every frustum-selected row densely, Mapper.py:394-402; the result is bit-identical)
  unsigned char* touched = nullptr; size_t touched_cap = 0;
  unsigned char *touched_geo = nullptr, *touched_col = nullptr;   // non-null only inside psl_map_iters
  int *adam_upto = nullptr, *adam_need = nullptr;                 // views into `touched` (lazy Adam bookkeeping)
  int *adam_list = nullptr, *adam_count = nullptr; long long adam_list_cap = 0;   // work list [cap] and its per-iteration lengths
  hipStream_t stream2 = nullptr;                 // low-priority stream of the mapper's k-NN block prefetch
  hipEvent_t ev_knn_ready[2] = {nullptr, nullptr}, ev_knn_free = nullptr;
  bool dw_defer_reduce = false;   // psl_map_iters: launch_dw leaves the chunk reduction to the Adam launch (dw_ra)
  psl::DwReduceArgs dw_ra{};
  // pinned host staging of psl_map_iters' per-call tables (frame descriptors, Adam constants): four slots, one event each,
  // so that the asynchronous uploads never read memory that has gone out of scope and the call needs no stream sync
  char* h_stage = nullptr; hipEvent_t ev_stage[4] = {nullptr, nullptr, nullptr, nullptr}; int stage_next = 0;
  float4* adam_tab = nullptr; size_t adam_tab_cap = 0;            // per-iteration (lr/bc1, sqrt(bc2)) of the two row groups
  unsigned long long* adam_rows = nullptr;                        // feature rows stepped by the lazy Adam since the last profile read
  unsigned long long* knn_cand = nullptr;   // [kKnnCandSlots * 8] candidates examined by the ray k-NN since the last psl_knn_candidates() read
  int* d_counter;
  // multi-GPU exchange (psl_comm.hip): RCCL communicator (ncclComm_t), device (rows, capacity) pairs [world + 1][2], staging buffer
  void* comm = nullptr; int comm_rank = 0, comm_world = 0; int* comm_counts = nullptr; int comm_counts_world = 0; float* comm_stage = nullptr; size_t comm_stage_cap = 0;
  int* pre_I = nullptr;      // neighbour lists answered ahead of the render call (psl_map_iters block prefetch)
  int* pre_cnt = nullptr;
  unsigned* img_hist = nullptr;   // 65536-bin histogram + select state of psl_topgrad_select_sync
  bool fused_ray = false;    // psl_map_iters: compositing fwd/bwd + loss run in its own fused kernel
  double* loss_acc = nullptr; int loss_acc_cap = 0;   // per-iteration loss sums of psl_map_iters: [iteration][kLossSlots][4]
  float* fwd_zero64 = nullptr;   // psl_map_iters, ray stage inside the backward: the colour-stage forward clears the backward's accumulators
  const void* track_fuse = nullptr;   // psl_track_iters (<= 1024 rays): TrackFuse* handed to launch_decode_bwd2 -- the tracker's ray stage inside the backward
  const void* track_pose = nullptr;   // psl_track_iters: TrackPose* handed to knn_rays -- the pose step of the previous iteration in the k-NN launch's prologue (psl_pose.h)
  float* trk_pref = nullptr; size_t trk_pref_cap = 0;   // camera-frame rays, sensor samples and depth masks of all iterations of a psl_track_iters call + the second pose / Adam buffer
  const void* ray_fuse = nullptr; const void* ray_wl = nullptr;   // ... RayFuse* / AdamWorklist* handed to launch_decode_bwd2
  float* d_small;        // 64 floats: dB_rel / exposure-affine accumulators
  float* d_expo;         // per-frame exposure scratch: affines [64][12], hidden activations [64][128], d(affine) [64][12]
  int* scan_flags;       // for add_points compaction
  int scan_flags_cap;
  // profiling: a ring of HIP event pairs per kernel class, recorded on the launch stream
  int prof_on;
  hipEvent_t* ev;                    // [PROF_N][PROF_RING][2]
  int prof_count[psl::PROF_N];       // launches bracketed with events since enable (may exceed the ring)
  int prof_seen[psl::PROF_N];        // launches since enable (prof_on = n > 1 brackets one in n, staggered by class)
  double prof_work[psl::PROF_N];     // algorithmic work (FLOP for MFMA classes, bytes for HBM classes) of those launches
  double prof_work_ring[psl::PROF_N];
};

namespace psl {

void set_error(const char* fmt, ...);

#define PSL_HIP(call)                                                                  \
  do {                                                                                 \
    hipError_t e__ = (call);                                                           \
    if (e__ != hipSuccess) {                                                           \
      psl::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
      return PSL_ERR_HIP;                                                              \
    }                                                                                  \
  } while (0)

void poison(void* p, size_t bytes);   // PSL_POISON=1: fill fresh allocations with 0x7F (uninitialised-read hunts)
bool debug_sync();   // PSL_DEBUG_SYNC=1: synchronise the device after every launch and report the failing one
#define PSL_LAUNCH_CHECK()                                                             \
  do {                                                                                 \
    hipError_t e__ = hipGetLastError();                                                \
    if (e__ == hipSuccess && psl::debug_sync()) {                                      \
      fprintf(stderr, "[psl sync] %s:%d\n", __FILE__, __LINE__);                       \
      e__ = hipDeviceSynchronize();                                                    \
    }                                                                                  \
    if (e__ != hipSuccess) {                                                           \
      psl::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
      return PSL_ERR_HIP;                                                              \
    }                                                                                  \
  } while (0)

// PSL_DEBUG_ADDRS=1: log (name, [begin, end)) of device buffers to stderr, to match a GPU memory-fault address
void dbg_range(const char* name, const void* p, size_t bytes);

// Timing of a kernel class (bench.py's roofline).  Two modes:
//  * ext (single-kernel scopes, all the hot classes): the scope ARMS an event pair and the one PSL_KLAUNCH inside it hands
//    the pair to hipExtLaunchKernelGGL, which stamps the events with the dispatch's own begin / end -- the timestamps
//    rocprofv3 reports, no barrier packet in the stream.  Marker pairs (hipEventRecord before / after) were measured to
//    cost ~24 us of stream time per pair and to read 51 us for a 50 us kernel but 35 us for the 18 us Adam launch.
//  * markers (scopes of several launches: repack, set-up): hipEventRecord on the launch stream before and after.
// prof_on = n > 1: one launch in n of a class is timed, classes staggered.
struct ProfArm { hipEvent_t start, stop; bool armed; };
extern thread_local ProfArm g_prof_arm;
struct ProfScope {
  psl_ctx* c; int slot; hipStream_t s; int k; bool on, ext;
  ProfScope(psl_ctx* c_, int slot_, hipStream_t s_, double work = 0.0, bool ext_ = false)
      : c(c_), slot(slot_), s(s_), k(0), on(false), ext(ext_) {
    if (c && c->prof_on) {
      const int n = c->prof_seen[slot]++;
      c->prof_work[slot] += work;
      on = c->prof_on == 1 || (n + slot) % c->prof_on == 0;
      if (on) {
        k = c->prof_count[slot] % PROF_RING;
        hipEvent_t e0 = c->ev[((size_t)slot * PROF_RING + k) * 2], e1 = c->ev[((size_t)slot * PROF_RING + k) * 2 + 1];
        if (ext) g_prof_arm = ProfArm{e0, e1, true};
        else (void)hipEventRecord(e0, s);
      }
    }
  }
  ~ProfScope() {
    if (!on) return;
    if (ext) {
      if (!g_prof_arm.armed) c->prof_count[slot]++;   // the launch took the pair
      g_prof_arm.armed = false;                       // (no launch in the scope: the sample is dropped)
    } else {
      (void)hipEventRecord(c->ev[((size_t)slot * PROF_RING + k) * 2 + 1], s);
      c->prof_count[slot]++;
    }
  }
};
// kernel launch that takes the armed event pair of the enclosing ext ProfScope, if there is one
#define PSL_KLAUNCH(kern, grid, block, lds, stream, ...)                                                          \
  do {                                                                                                            \
    if (psl::g_prof_arm.armed) {                                                                                  \
      psl::g_prof_arm.armed = false;                                                                              \
      hipExtLaunchKernelGGL(kern, grid, block, lds, stream, psl::g_prof_arm.start, psl::g_prof_arm.stop, 0, __VA_ARGS__); \
    } else {                                                                                                      \
      hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                                            \
    }                                                                                                             \
  } while (0)

// two launches under ONE armed event pair (split colour stage: k_nbr_* + k_trunk_*): the first launch takes the start event,
// the last one the stop event, so that the class time spans both dispatches and the gap between them
#define PSL_KLAUNCH2(kern, first, last, grid, block, lds, stream, ...)                                            \
  do {                                                                                                            \
    if (psl::g_prof_arm.armed) {                                                                                  \
      hipEvent_t e0__ = (first) ? psl::g_prof_arm.start : nullptr, e1__ = (last) ? psl::g_prof_arm.stop : nullptr; \
      if (last) psl::g_prof_arm.armed = false;                                                                    \
      hipExtLaunchKernelGGL(kern, grid, block, lds, stream, e0__, e1__, 0, __VA_ARGS__);                          \
    } else {                                                                                                      \
      hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                                            \
    }                                                                                                             \
  } while (0)

// ---- internal launchers (defined in the .hip files) -------------------------
int grid_build(psl_ctx* ctx, hipStream_t s);
int knn_rays(psl_ctx* ctx, const float* rays_o, const float* rays_d, const float* depth, const float* z_vals, const float* r_query,
             int n_rays, int* I_out, int* cnt_out, hipStream_t s, int max_blocks = 0);
struct AdamRowsSeg { float* feats; const int* rows; float4 *g, *m, *v; int n_rows; float lr_bc1, sqrt_bc2;
                     unsigned char* touched;   // [n_rows] set by the backward scatter when a row received a gradient
                     int* upto; };             // [n_rows] iterations of this call already applied to the row (-1: m = v = 0)
// Lazy, exact Adam of the mapper's feature rows (see k_map_adam): per-iteration constants of the whole call
constexpr int kAdamTabLds = 72;   // >= the k-NN prefetch block (64 iterations) + 1
constexpr int kAdamRowSlots = 256 * 8;   // rows_done is spread over 256 cache lines
constexpr int kLossSlots = 32;          // per-iteration loss sums of psl_map_iters: [slot][4] doubles, slot = workgroup & 31 (one address took ~10^3 atomics per launch)
constexpr int kKnnCandSlots = 256;       // the k-NN candidate counter likewise (one same-address atomic per query serialised a 25 000-query launch: 316 vs ~100 us)
struct AdamLazy { const float4* tab; const int* list; const int* count; long long list_cap; int it;
                  unsigned long long* rows_done; int base; };
struct AdamParSeg { float *p, *g, *m, *v; int n; float lr_bc1, sqrt_bc2;
                    const int* wf_index; float* wf; const int* wb_index; float* wb;
                    // slabs != null: g is not read; the gradient of element e is the ordered sum of its chunk partials
                    // (what k_dw_reduce would have written), taken inside the Adam launch
                    const float* slabs; const float* g_brel; DwReduceArgs ra; };
int launch_map_adam(AdamRowsSeg geo, int step_geo, float lr_geo, AdamRowsSeg col, int step_col, float lr_col, AdamParSeg par,
                    float lr_par, hipStream_t s, int step_par = -1, AdamLazy lazy = AdamLazy{nullptr, nullptr, nullptr, 0, 0, nullptr, 0});
int adam_lazy_row_blocks(const AdamLazy& lazy, int n_rows);
void adam_consts(int step, float lr, float b1, float b2, float& lr_bc1, float& sqrt_bc2);
// work list of the lazy Adam, built by extra workgroups of k_map_ray_fused (see adam_worklist_role)
struct AdamWorklist { const int* I_a; const int* I_b; int n4; const int* row_map; int* stamp_arr; int stamp; int* list; int* count; };
int launch_map_ray_fused(const float4* raw, const int* cnt, const float* gt_depth, const float* gt_color, const int* active,
                         float near_s, float far_s, int min_nn, int n_rays, float coef, float w_color, int color_stage,
                         float* depth, float* var, float* rgb, unsigned char* valid, float4* d_raw, double* loss_acc,
                         float* zero64, const float* frame_affine, int pix_per_frame, float* g_frame_affine, hipStream_t s,
                         const AdamWorklist* wl = nullptr);
int knn_queries(psl_ctx* ctx, const float* q, const float* r_per_query, float r_scalar, int nq, float* D_out,
                int64_t* I_out, int* cnt_out, hipStream_t s);
int repack_frags(psl_ctx* ctx, const float* master, hipStream_t s);
int build_frag_index(psl_ctx* ctx, hipStream_t s);

// workspace carving for render fwd/bwd (all sizes in floats, P = 5 * n_rays padded to TILE)
struct RenderWs {
  int P, Ppad;
  int* I;            // [Ppad][8]
  int* cnt;          // [Ppad]
  float* raw;        // [Ppad][4]  rgb (post sigmoid/affine), occ (masked)
  float* w;          // [Ppad][8]  normalised interpolation weights
  float* dcc;        // [Ppad][32]  dL/d(interpolated colour features), written by k_trunk_bwd, read by k_nbr_bwd
  float* cc;         // [Ppad][32]
  float* g_y;        // [Ppad][5][32]   geo post-activation
  float* c_y;        // [Ppad][5][128]  colour post-activation
  float* c_hin;      // [Ppad][5][128]  colour layer inputs h_1..h_5 (after +fc_c)
  float* c_emb;      // [Ppad][40]
  float* c_emb2;     // [Ppad][40]  the same values in k_trunk_fwd's lane order: [g][sin f = 4 ks + g, ks = 0..4 | cos ...] (split colour stage)
  float* out3;       // [Ppad][4] pre-affine colour logits
  float* n_x;        // [Ppad][8][52]
  float* n_h1;       // [Ppad][8][128]
  float* n_out;      // [Ppad][8][32]
  float* cw;         // [R][5] compositing weights
  float* ray_aux;    // [R][4] depth, W(sum+eps), var, -
  // backward-produced
  float* d_raw;      // [Ppad][4]
  float* c_dz;       // [Ppad][5][128]
  float* c_g;        // [Ppad][5][128]
  float* n_dz1;      // [Ppad][8][128]
  float* n_dnf;      // [Ppad][8][32]
  float* d_out3;     // [Ppad][4]
  float* dp;         // [Ppad][4]
  float* dp2;        // [Ppad][4]  geometry-branch share of dL/dp (register-chained backward: written by another workgroup)
  int64_t total;
};
RenderWs carve_ws(float* base, int n_rays, int flags);

}  // namespace psl
