/*
 * pointslam_hip.h -- C ABI of libpointslam_hip.so: the MI355X (gfx950) render /
 * optimise hot path of Point-SLAM.
 *
 * The reference (eriksandstroem/Point-SLAM, /root/reference) has no native
 * plugin API; its narrowest seams are Python method signatures.
 * Every entry point below names the reference interface it stands in for
 * (file:line relative to the reference tree).  The Python mirrors of those
 * interfaces live in point_slam_amd/{renderer,neural_point}.py and call ONLY
 * these functions (through ctypes).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; no exceptions cross the boundary.
 *  - every function returns 0 on success, a negative psl_status on failure;
 *    psl_last_error() returns a thread-local message.
 *  - all array arguments are BORROWED DEVICE pointers (owned by the caller,
 *    e.g. torch) valid until the work enqueued on `stream` has completed,
 *    unless the parameter is documented as host memory.
 *  - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and
 *    nothing synchronises the device except the functions documented to.
 *  - one psl_ctx per (process, device); a ctx is not thread-safe.
 *  - fixed architecture constants (asserted by psl_create): S = 5 samples per
 *    ray, K = 8 neighbours, C = 32 feature channels (configs/point_slam.yaml:10,95,107).
 */
#ifndef POINTSLAM_HIP_H
#define POINTSLAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct psl_ctx psl_ctx;

enum psl_status {
  PSL_OK = 0,
  PSL_ERR_ARG = -1,      /* bad argument */
  PSL_ERR_HIP = -2,      /* a HIP runtime call failed */
  PSL_ERR_CAPACITY = -3, /* point capacity exhausted */
  PSL_ERR_STATE = -4,    /* call order (e.g. render before index build) */
  PSL_ERR_UNSUPPORTED = -5
};

/* Hot-path subset of the reference YAML (configs/point_slam.yaml). */
typedef struct psl_config {
  int32_t n_surface;        /* rendering.N_surface (must be 5)           :95  */
  int32_t nn_num;           /* pointcloud.nn_num (must be 8)             :107 */
  int32_t c_dim;            /* model.c_dim (must be 32)                  :10  */
  int32_t min_nn_num;       /* pointcloud.min_nn_num                     :108 */
  float near_end_surface;   /* rendering.near_end_surface                :97  */
  float far_end_surface;    /* rendering.far_end_surface                 :98  */
  float radius_query;       /* pointcloud.radius_query (fixed-radius mode):112 */
  float max_query_radius;   /* largest radius any query may use; sets the grid cell size.
                               dynamic mode: radius_add_max*radius_query_ratio (:113,115) */
  int32_t encode_rel_pos;   /* model.encode_rel_pos_in_col               :13  */
  int32_t max_points;       /* position capacity of this ctx (points), <= 2^25 */
  int32_t nn_weighting;     /* pointcloud.nn_weighting :110 -- 0 'distance' 1/(D+1e-10), 1 'expo' exp(-20 sqrt(D))
                               (src/conv_onet/models/decoder.py:152-156,362-366); ABI v7 */
} psl_config;

/* ---- lifecycle ----------------------------------------------------------- */
int psl_create(int device, const psl_config* cfg, psl_ctx** out);
void psl_destroy(psl_ctx* ctx);
const char* psl_last_error(void);
int psl_abi_version(void);

/* ---- decoder parameter blob ----------------------------------------------
 * The decoders (src/conv_onet/models/decoder.py:452-475, POINT) are handed over
 * as ONE flat fp32 "master" blob: the tensors named by psl_param_entry(),
 * each in its torch layout ([out][in] row-major), concatenated in table
 * order.  Entries [0, psl_param_color_count()) form the colour-decoder
 * group (trainable in mapping), the rest the geometry-decoder group.
 * Gradients are returned in the same layout.  The fixed, non-persistent colour
 * Fourier matrix (decoder.py:305-306, `_B` [3][20]) is a separate argument. */
int psl_param_count(void);
int psl_param_color_count(void);
/* name: reference state_dict key; rows/cols: torch shape (cols=1 for biases); offset in floats */
int psl_param_entry(int i, char* name_out, int name_cap, int* rows, int* cols, int* offset);
int psl_param_master_floats(void);

/* ---- neural point cloud: positions + spatial index -------------------------
 * Replaces NeuralPointCloud._cloud_pos (a Python list, src/neural_point.py:29,147)
 * and the FAISS IVF index (src/neural_point.py:37-41).  Features stay with the
 * caller (torch tensors), exactly as in the reference. */
int psl_points_reset(psl_ctx* ctx);
/* raw append of n positions [n][3] f32 (no dedupe); index becomes stale */
int psl_points_append(psl_ctx* ctx, const float* pos, int n, void* stream);
/* drop every point with index >= n (multi-GPU merge re-appends blocks in rank order); index becomes stale */
int psl_points_truncate(psl_ctx* ctx, int n);
/* number of points (host value; exact after psl_sync or any *_sync call) */
int psl_points_count(psl_ctx* ctx);
/* copy positions [count][3] f32 to a device buffer (checkpoint compatibility, src/utils/Logger.py:22-40) */
int psl_points_download(psl_ctx* ctx, float* pos_out, int capacity_points, void* stream);
/* the same for points [first, first+count): what a rank sends in the multi-GPU exchange (O(new), not O(N)) */
int psl_points_download_range(psl_ctx* ctx, int first, int count, float* pos_out, void* stream);
/* (re)build the uniform-grid index over all points: index.train/index.add (src/neural_point.py:161-164) */
int psl_index_build(psl_ctx* ctx, void* stream);

/* find_neighbors_faiss (src/neural_point.py:169-215): EXACT search restricted to the query
 * radius.  D[nq][8] f32 squared distances ascending, I[nq][8] int64, cnt[nq] int32 = #(D < r^2).
 * Slots farther than the radius hold D=+inf, I=-1 (they carry weight 0 everywhere in the
 * reference, decoder.py:157,367).  r_per_query may be NULL -> r_scalar for all. */
int psl_knn(psl_ctx* ctx, const float* q, const float* r_per_query, float r_scalar, int nq,
            float* D_out, int64_t* I_out, int32_t* cnt_out, void* stream);

/* The admission test of add_neural_points (src/neural_point.py:116-121: a location is kept iff NO existing point lies
 * strictly inside its radius) for externally supplied surface points, restricted to the first idx_limit points of the
 * cloud: cnt_out[i] = #{ j < idx_limit : |x_j - q_i|^2 < r_i^2 } (an exact count, not capped at 8).  The multi-GPU merge
 * (point_slam_amd/dist.py) tests the other ranks' new locations against the BASE map while the index still covers this
 * rank's own tail -- no index rebuild before the test. */
int psl_dedupe_count(psl_ctx* ctx, const float* q, const float* r_per_query, float r_scalar, int nq, int idx_limit,
                     int32_t* cnt_out, void* stream);

/* The cross-rank half of the same admission test (point_slam_amd/dist.py, step 2): n_loc = block_first[n_blocks]
 * locations of three points each (N_add = 3, src/neural_point.py:126-143; the middle point is the surface point and
 * carries the radius) arrive in rank blocks, block b = locations [block_first[b], block_first[b+1]).  On entry keep[l] is 1
 * for every location that passed the test against the base map; a location of block b loses its flag when one of the
 * three points of a location of an EARLIER block that is still kept lies strictly inside its radius.  Blocks are
 * processed in order on `stream` (block 0 is admitted as it stands), so every rank derives the same flags.
 * pts / radius: device pointers to the first point's xyz and radius, strides in floats between consecutive points.
 * block_first: HOST array of n_blocks + 1 ascending offsets. */
int psl_dedupe_blocks(psl_ctx* ctx, const float* pts, int pts_stride, const float* radius, int radius_stride,
                      const int32_t* block_first, int n_blocks, uint8_t* keep, void* stream);

/* NeuralPointCloud.sample_near_pcl marching test (src/neural_point.py:232-249): hits[ray][step] = 1 when the point
 * rays_o + rays_d * z_steps[row][step] has >= 1 neural point strictly inside `radius` (cfg radius_query).
 * z_steps is [n_rows][n_steps]; row = step_row[ray], or 0 when step_row is NULL (render_img marches each of its
 * 3000-ray batches to that batch's own far bound, Renderer.py:108-112,254).  The caller turns the first two hits
 * of a ray into its sample interval (:251-277); rays with < 2 hits are masked. */
int psl_near_pcl_hits(psl_ctx* ctx, const float* rays_o, const float* rays_d, int32_t n_rays, const float* z_steps,
                      const int32_t* step_row, int32_t n_steps, float radius, uint8_t* hits, void* stream);

/* add_neural_points (src/neural_point.py:91-167): for rays with depth>0 compute the surface
 * point o+d*depth, keep locations with ZERO existing points closer than the radius
 * (dedupe against the index as built, not against this batch), append n_add=3 points per
 * kept location at linspace(near,far,3)*depth.  keep_out[n] uint8 (0 for depth<=0 rays).
 * Synchronises `stream` and returns the number of kept LOCATIONS in *n_kept_host.
 * The index is stale afterwards (call psl_index_build). */
int psl_add_points_sync(psl_ctx* ctx, const float* rays_o, const float* rays_d, const float* depth,
                        const float* radius_per_ray, float r_scalar, int n, float near_end, float far_end,
                        uint8_t* keep_out, int* n_kept_host, void* stream);

/* ---- render forward / backward ---------------------------------------------
 * Renderer.render_batch_ray (src/utils/Renderer.py:77-202) for rays with sensor depth > 0,
 * through POINT.forward (decoder.py:476-518) and raw2outputs_nerf_color (src/common.py:298-336). */
enum psl_render_flags {
  PSL_STAGE_COLOR = 1,      /* stage == 'color' (else 'geometry': rgb = 0)            */
  PSL_PTS_GRAD = 2,         /* is_tracker: d/d(rays) through D, embeddings, rel-pos   */
  PSL_PARAM_GRAD = 4,       /* produce colour-decoder parameter gradients             */
  PSL_FEAT_GRAD = 8,        /* produce feature gradients                              */
  PSL_NO_SIGMOID = 16,      /* encode_exposure without exposure_feat (decoder.py:439-446) */
  PSL_HAS_AFFINE = 32       /* exposure affine given (decoder.py:432-438)             */
};

/* floats of scratch the caller must provide to fwd and keep alive (unchanged) until bwd */
int64_t psl_render_ws_floats(int n_rays, int flags);

typedef struct psl_render_args {
  int32_t n_rays;
  int32_t flags;                 /* psl_render_flags */
  float sigmoid_coef;            /* Renderer.sigmoid_coefficient (Tracker.py:36, Mapper.py:45) */
  const float* rays_o;           /* [R][3] */
  const float* rays_d;           /* [R][3] unnormalised */
  const float* gt_depth;         /* [R] > 0 */
  const float* r_query;          /* [R] per-ray query radius or NULL (fixed radius) */
  const float* geo_feats;        /* [N][32] */
  const float* col_feats;        /* [N][32] */
  const float* params;           /* master blob, psl_param_master_floats() */
  const float* col_embed_B;      /* [3][20] fixed colour Fourier matrix */
  const float* fallback_geo;     /* [32] random vector for samples with < min_nn neighbours (decoder.py:170) */
  const float* fallback_col;     /* [32] (decoder.py:387) */
  const float* exposure_affine;  /* [12] (rot 3x3 row-major, trans) or NULL */
  float* ws;                     /* scratch, psl_render_ws_floats() floats */
  /* outputs */
  float* depth;                  /* [R] */
  float* var;                    /* [R] */
  float* rgb;                    /* [R][3] */
  uint8_t* valid_ray;            /* [R] valid_ray_mask (decoder.py:200-201) */
  /* optional input (ABI v2): explicit sample depths [R][5] replacing the 0.98..1.02*gt_depth rule, for batches
   * with depth-less pixels (Renderer.py:142-170: sample_near_pcl / uniform branch). NULL = derive from gt_depth. */
  const float* z_vals;
} psl_render_args;

int psl_render_fwd(psl_ctx* ctx, const psl_render_args* a, void* stream);

typedef struct psl_render_grads {
  const float* g_depth;          /* [R] cotangents */
  const float* g_var;            /* [R] or NULL */
  const float* g_rgb;            /* [R][3] or NULL */
  /* outputs (may be NULL when the matching flag is off). Feature gradients are ACCUMULATED
   * (+=) with atomics into caller-zeroed buffers: row i of g_*_feats is feature row
   * feat_row_map[i_point] (or i_point itself when feat_row_map is NULL; -1 = not trainable). */
  float* g_geo_feats;
  float* g_col_feats;
  const int32_t* feat_row_map;   /* [N] or NULL */
  float* g_params;               /* master layout, OVERWRITTEN (colour group; geo group zeroed) */
  float* g_rays_o;               /* [R][3] overwritten */
  float* g_rays_d;               /* [R][3] overwritten */
  float* g_exposure_affine;      /* [12] overwritten */
} psl_render_grads;

int psl_render_bwd(psl_ctx* ctx, const psl_render_args* a, const psl_render_grads* g, void* stream);

/* raw2outputs_nerf_color alone (src/common.py:298-336), for unit parity: raw [R][5][4], z [R][5] */
int psl_composite_fwd(const float* raw, const float* z, int n_rays, float coef, float* depth, float* var,
                      float* rgb, float* weights, void* stream);

/* ---- fused optimiser step (torch.optim.Adam defaults; Mapper.py:394-402,556) ------
 * p,g,m,v [n]; step is the 1-based step count of this group. g is zeroed afterwards when zero_grad != 0. */
int psl_adam_step(float* p, float* g, float* m, float* v, int64_t n, int step, float lr, float beta1,
                  float beta2, float eps, int zero_grad, void* stream);
/* same over rows of a [N][32] feature matrix selected by rows[n_rows]; g,m,v are compact [n_rows][32] */
int psl_adam_step_rows(float* feats, const int32_t* rows, float* g, float* m, float* v, int n_rows, int step,
                       float lr, float beta1, float beta2, float eps, int zero_grad, void* stream);

/* ---- fused optimisation loops (no host synchronisation inside) -------------------------------
 * The bodies of the reference's per-frame loops, run n_iters times back-to-back on `stream`:
 *   psl_track_iters : Tracker.optimize_cam_in_batch (src/Tracker.py:89-186) inside the loop of
 *                     Tracker.run (src/Tracker.py:332-350, incl. the lowest-loss candidate pose);
 *   psl_map_iters   : the joint_iter loop of Mapper.optimize_map (src/Mapper.py:408-568), without BA.
 * Random draws stay with the host (torch RNG): flat pixel indices as torch.randint would produce them in
 * select_uv (src/common.py:59-74) and the per-call fallback vectors (decoder.py:170,387), pre-drawn for
 * all iterations. */
typedef struct psl_cam_intr { int32_t H, W; float fx, fy, cx, cy; } psl_cam_intr;

typedef struct psl_frame_view {     /* one RGB-D (key)frame resident in device memory */
  const float* depth;               /* [H][W] f32, 0 = no reading */
  const float* color;               /* [H][W][3] f32 */
  const float* r_query;             /* [H][W] per-pixel query radius, or NULL (fixed radius) */
  float c2w[12];                    /* row-major 3x4 pose (mapping only) */
} psl_frame_view;

/* Per-frame exposure compensation (model.encode_exposure, configs/ScanNet/scannet.yaml:5): MLP_exposure
 * (decoder.py:243-258) maps a frame's 8-d latent to a 3x3 colour matrix + offset applied to the colour logits before the
 * sigmoid.  Tracker (Tracker.py:305-311): this frame's latent and the MLP are both optimised, lr 0.001.  Mapper
 * (Mapper.py:399-401,530-548): one latent per window frame, applied to that frame's slice of the COMPOSITED logits; only
 * the current frame's latent is optimised (lr 0.001), the MLP is part of color_decoder.parameters() (decoders_lr). */
#define PSL_EXPOSURE_DIM 8
#define PSL_EXPOSURE_MLP_FLOATS 2700   /* linear1.weight [128][8], linear1.bias [128], linear2.weight [12][128], linear2.bias [12] */
typedef struct psl_exposure_args {
  float* mlp;          /* [PSL_EXPOSURE_MLP_FLOATS], updated in place */
  float* feats;        /* tracker: [8]; mapper: [n_frames][8], the LAST row (current frame) is updated in place */
  float* adam;         /* [2][PSL_EXPOSURE_MLP_FLOATS + 8] exp_avg, exp_avg_sq; zero at frame start */
  float lr_mlp;        /* tracker: 0.001; mapper: mapping.stage.color.decoders_lr */
  float lr_feat;       /* 0.001 */
  int32_t step0;       /* Adam steps already taken on this state */
} psl_exposure_args;

typedef struct psl_track_args {
  psl_cam_intr cam;
  int32_t edge_h, edge_w;           /* tracking.ignore_edge_H/W */
  int32_t n_iters, n_pix;           /* tracking.iters, tracking.pixels (n_pix <= 16384) */
  const int32_t* pix_idx;           /* [n_iters][n_pix] indices into the cropped window, row-major */
  const float* fallback;            /* [n_iters][2][32] geometry / colour fallback vectors */
  psl_frame_view frame;
  float* cam_tensor;                /* [7] quaternion (w,x,y,z) + translation; updated in place */
  float* adam_state;                /* [14] exp_avg[7], exp_avg_sq[7]; zero at frame start */
  int32_t step0;                    /* Adam steps already taken on this state */
  float lr_T, lr_quat;              /* tracking.lr, 0.2*tracking.lr (separate_LR, Tracker.py:305-306) */
  float w_color;                    /* tracking.w_color_loss */
  int32_t handle_dynamic, use_color;   /* handle_dynamic 0: the median mask of Tracker.py:166-168 (round 5; ten-launch path) */
  float sigmoid_coef;
  const float *geo_feats, *col_feats, *params, *col_embed_B;
  float* ws;                        /* psl_track_ws_floats(n_pix) floats */
  float* loss_out;                  /* [n_iters][4] (loss, geo, colour, #active) or NULL */
  float* best_out;                  /* [8] lowest-loss pose [7] + its loss (candidate_cam_tensor, Tracker.py:347-350) */
  /* ABI v3 */
  int32_t pix_full_image;           /* 1: pix_idx are flat indices into the FULL image, no border crop and no depth filter --
                                       tracking.sample_with_color_grad (Tracker.py:115-128): the caller draws them from the
                                       top-gradient set of psl_topgrad_select_sync */
  const psl_exposure_args* exposure;/* NULL unless model.encode_exposure */
} psl_track_args;
int64_t psl_track_ws_floats(int n_pix);
int psl_track_iters(psl_ctx* ctx, const psl_track_args* t, void* stream);

typedef struct psl_map_args {
  psl_cam_intr cam;
  int32_t n_frames, pix_per_frame;  /* window size, mapping.pixels // window (n_frames*pix_per_frame <= 16384) */
  int32_t n_iters, n_geo_iters;     /* iterations; iterations 0..n_geo_iters run stage 'geometry' (Mapper.py:420-423) */
  const psl_frame_view* frames;     /* [n_frames] HOST array of device frame views */
  const int32_t* pix_idx;           /* [n_iters][n_frames][pix_per_frame] */
  const float* fallback;            /* [n_iters][2][32] */
  float* geo_feats;                 /* [N][32] updated in place at rows sel_rows */
  float* col_feats;
  float* params;                    /* master blob; colour group updated in place when train_decoder */
  const float* col_embed_B;
  const int32_t* sel_rows;          /* [n_sel] frustum-selected point indices (ascending) */
  const int32_t* row_map;           /* [N] point index -> compact row or -1 */
  int32_t n_sel;
  float *g_geo, *g_col;             /* [n_sel][32] compact gradient accumulators, zero on entry */
  float *adam_geo, *adam_col;       /* [2][n_sel][32] exp_avg, exp_avg_sq; zero at frame start */
  float* adam_params;               /* [2][colour floats] */
  int32_t step0_geo, step0_col;
  int32_t train_decoder;            /* !mapping.fix_color_decoder */
  float lr_geo_geo_stage, lr_geo_color_stage, lr_col, lr_decoder;   /* mapping.stage.* learning rates */
  float w_color, sigmoid_coef;
  float* ws;                        /* psl_map_ws_floats() floats */
  float* loss_out;                  /* [n_iters][4] or NULL */
  /* ABI v3 */
  const psl_exposure_args* exposure;/* NULL unless model.encode_exposure */
  int32_t step0_params;             /* Adam steps the colour-decoder group has already counted when its first gradient arrives.
                                       0 = torch >= 2.0 (zero_grad sets .grad to None: the group is skipped during the geometry
                                       stage); n_geo_iters + 1 reproduces torch 1.12 (env.yaml) from the second mapped frame on,
                                       where zero_grad leaves zero tensors and Adam counts the geometry-stage steps */
} psl_map_args;
int64_t psl_map_ws_floats(int n_rays, int n_frames);
int psl_map_iters(psl_ctx* ctx, const psl_map_args* m, void* stream);

/* Initial pose of the next frame (src/Tracker.py:283-290, const_speed_assumption): camera tensor [quat wxyz, T] of
 * delta @ pre_c2w with delta = pre_c2w @ inv(c2w[idx-2]), from the tracker's two previous camera tensors ON THE DEVICE
 * (cam_prev2 NULL: cam_prev itself, Tracker.py:288-289).  ABI 6.  One launch on `stream`, no synchronisation: a closed
 * track -> track loop never copies a pose to the host (the reference goes through numpy / scipy once per frame). */
int psl_pose_const_speed(const float* cam_prev, const float* cam_prev2, float* cam_out, void* stream);

/* Mapper.get_mask_from_c2w (src/Mapper.py:120-168): frustum feature selection.  c2w_host: [16] row-major 4x4
 * (host memory).  Writes the ascending index list sel_out[<=N] and row_map_out[N]; synchronises and returns
 * the count.  Points whose bilinear lookup is 0 take depth_max (:161-162: np.max over the PER-POINT lookups): pass a negative
 * depth_max to have that maximum taken on the device (the reference's rule), or a value >= 0 to impose one. */
int psl_frustum_select_sync(psl_ctx* ctx, const float* c2w_host, psl_cam_intr cam, const float* depth, float depth_max,
                            float edge, int32_t* sel_out, int32_t* row_map_out, int* n_sel_host, void* stream);

/* ---- per-frame image operators in front of the path ------------------------------------------
 * Dynamic radii (src/Tracker.py:235-250, src/Mapper.py:686-701): rgb2gray (0.2125, 0.7154, 0.0721), Sobel
 * [1,0,-1] x [1,2,1]/4 with scipy 'reflect' borders, magnitude, clip to [0, color_grad_threshold], piecewise-linear
 * map with knots [0, 0.01, threshold] -> radius_add_max .. radius_add_min (and ratio * that for the query radius).
 * float64 arithmetic as in numpy; grad_mag_out [H][W] f64 (or NULL), r_add_out / r_query_out [H][W] f32 (or NULL). */
int psl_frame_radii(const float* color, int32_t H, int32_t W, float color_grad_threshold, float radius_add_max,
                    float radius_add_min, float radius_query_ratio, double* grad_mag_out, float* r_add_out,
                    float* r_query_out, void* stream);

/* get_selected_index_with_grad (src/common.py:116-159): the k = ratio*n pixels with the largest gradient magnitude
 * over the WHOLE image (np.argpartition; which of several equal values at the threshold are taken is open there and
 * here), then masked by the region [H0,H1) x [W0,W1) and by depth > 0 (and <= depth_limit if > 0; depth may be NULL).
 * sel_out [<= k] flat pixel indices in arbitrary order; synchronises and returns the count. */
int psl_topgrad_select_sync(psl_ctx* ctx, const double* grad_mag, const float* depth, int32_t H, int32_t W, int32_t k,
                            int32_t H0, int32_t H1, int32_t W0, int32_t W1, float depth_limit, int32_t* sel_out,
                            int* n_sel_host, void* stream);

/* End-of-run image metrics of one rendered frame (src/Mapper.py:861-879): out3 = { PSNR over the pixels with sensor
 * depth, MS-SSIM (pytorch_msssim 0.2.x: data_range 1, 11-tap Gaussian sigma 1.5, 5 scales), mean depth L1 over the
 * pixels with sensor depth }.  Images [H][W][3] / [H][W] f32 on the device; min(H, W) > 160.  Synchronises.
 * (LPIPS needs a pretrained AlexNet and is out of scope.) */
int psl_image_metrics_sync(const float* gt_color, const float* gt_depth, const float* color, const float* depth,
                           int32_t H, int32_t W, double* out3_host, void* stream);

/* keyframe_selection_overlap (src/Mapper.py:170-235), the geometric part: for every keyframe pose the share of the
 * n_rays * n_samples frustum points (z from 0.8*depth to depth+0.5) of the current view that project inside its image
 * with an `edge`-pixel border and lie in front of it.  c2w_host: n_kf row-major 4x4 poses (host); percent_host [n_kf].
 * Sorting / thresholding / the random pick of k keyframes stay with the caller (host RNG). Synchronises. */
int psl_keyframe_overlap_sync(const float* rays_o, const float* rays_d, const float* depth, int32_t n_rays,
                              int32_t n_samples, const float* c2w_host, int32_t n_kf, psl_cam_intr cam, float edge,
                              float* percent_host, void* stream);

/* ---- multi-GPU exchange (BASELINE.json north_star: "periodic RCCL all-gather over xGMI of newly-added
 * neural points").  The reference has no distributed code (no NCCL / torch.distributed call anywhere in it); frames are
 * partitioned one per GPU and the replicas exchange record blocks.  librccl is resolved with dlopen at the first call.
 *   psl_comm_unique_id : rank 0 obtains the 128-byte ncclUniqueId; the host hands it to every rank (any side channel)
 *   psl_comm_reserve   : (ABI 6) everything psl_comm_init does that can fail on one rank alone (dlopen of librccl, the
 *                        device ints of the counts phase): a host calls it on every rank and lets the ranks agree on the
 *                        outcome before anybody enters the rendezvous of ncclCommInitRank
 *   psl_comm_init      : psl_comm_reserve, then ncclCommInitRank on the ctx's device; the communicator lives in the ctx
 *   psl_allgather_new_points : all-gather-v of per-rank record blocks [n_local][rec_floats] f32 (new points: xyz +
 *                        32 geometry + 32 colour features + add-radius = 68 floats; touched feature rows: row id + 64
 *                        changes) in rank order into rec_all [sum][rec_floats]; counts_host[world] receives the per-rank row
 *                        counts.  nccl_comm: an ncclComm_t the host already owns (then `world` is its size), or NULL for
 *                        the ctx's communicator.  Two ncclAllGather on `stream` ((rows, capacity) pairs, padded records);
 *                        synchronises `stream` once to learn the counts.  Returns the total number of rows (>= 0) or a
 *                        psl_status.  PSL_ERR_CAPACITY is decided on the gathered pairs (total > the SMALLEST capacity_rows of
 *                        any rank), i.e. by every rank alike and before the records collective: all ranks may grow their
 *                        buffers from counts_host (valid on that return) and call again. */
int psl_comm_unique_id(void* id_out_128_bytes);
int psl_comm_reserve(psl_ctx* ctx, int world);
int psl_comm_init(psl_ctx* ctx, const void* id_128_bytes, int rank, int world);
int psl_comm_destroy(psl_ctx* ctx);
int psl_allgather_new_points(psl_ctx* ctx, void* nccl_comm, int world, const float* rec_local, int n_local, int rec_floats,
                             float* rec_all, int capacity_rows, int32_t* counts_host, void* stream);
/* the counts-phase decision of psl_allgather_new_points by itself (host only, no GPU): pairs[world][2] = (rows, capacity) as
 * gathered; fills counts_out[world], *total_out, *n_max_out; PSL_ERR_CAPACITY iff total > min capacity. */
int psl_allgather_decide(const int32_t* pairs, int world, int32_t* counts_out, long long* total_out, int* n_max_out);

/* ---- unit harness of the fast-math device helpers (tests only; no reference counterpart: the reference calls torch's
 * sin/cos (decoder.py:33-36), nn.Softplus(beta=100) (decoder.py:124,231,335) and torch.optim.Adam, this library evaluates
 * them with hardware transcendentals).  kind: see psl_selftest_kind; in/out device arrays:
 *   SINCOS        in [n] angles            -> out [n][2] (sin, cos) of fast_sincosf
 *   SOFTPLUS(_NB) in [n]                   -> out [n]   softplus100 / its branch-free twin
 *   SOFTPLUS_GRAD in [n] softplus OUTPUTS  -> out [n]   d softplus / d input expressed through the output
 *   ADAM_REPLAY   in [n][6] (p, m, v, lr/bc1, sqrt(bc2), steps) -> out [n][6]: (p, m, v) after `steps` gradient-free steps
 *                 with the lazy Adam's replay arithmetic, then the same with the dense kernel's IEEE sequence */
typedef enum { PSL_SELFTEST_SINCOS = 0, PSL_SELFTEST_SOFTPLUS = 1, PSL_SELFTEST_SOFTPLUS_NB = 2, PSL_SELFTEST_SOFTPLUS_GRAD = 3,
               PSL_SELFTEST_ADAM_REPLAY = 4 } psl_selftest_kind;
int psl_selftest_math(int kind, const float* in, float* out, int n, void* stream);

/* Known-traffic kernels in the access patterns of the hot path, for calibrating what rocprofv3's FETCH_SIZE / WRITE_SIZE report
 * on gfx950 (tools/pmc_probe.py --calibrate; no reference counterpart).  kind 0: streaming read of n float4 from `table`
 * (16 B per lane); 1: streaming write of n float4; 2: gather of n 128-byte rows table[rows[i]][32] with two 16-byte loads per
 * lane, four lanes per row -- how the decode tiles gather feature rows; 3: float-atomic scatter of n 128-byte rows, 32 lanes
 * per row -- how the backward scatters row gradients.  `scratch`: >= 1024 floats (kinds 0 and 2).  ABI 6. */
int psl_selftest_traffic(int kind, float* table, const int32_t* rows, float* scratch, long long n, void* stream);

/* ---- timing helpers for the bench harness ---------------------------------- */
int psl_sync(psl_ctx* ctx, void* stream);
/* Kernel-class timing with HIP events recorded on the launch stream (for bench.py's roofline):
 * enable, run, then read per class: total ms, launch count, algorithmic work (FLOP for the MFMA-bound
 * classes decode_fwd/decode_bwd/dw_gemm, bytes for the HBM-bound ones). psl_profile_read synchronises.
 * on = 0: off; 1: every launch is bracketed; n > 1: one launch in n of each class, the classes staggered (a marker is a
 * barrier packet: bracketing every launch of a five-launch iteration perturbs what it measures); the totals
 * psl_profile_read returns are then mean-of-the-bracketed x launches. */
int psl_profile_enable(psl_ctx* ctx, int on);
/* candidates (16-byte sorted-position records) the ray k-NN examined since the previous call: the k-NN's roofline is
 * 16 B x candidates / kernel time against the L2 bandwidth (its traffic is index-dependent).
 * Synchronises the device and resets the counter. */
int64_t psl_knn_candidates(psl_ctx* ctx);
/* run-time A/B switches for tests and profiling: "knn" 0 = by launch size, 1 = one wavefront per sample, 2 = one per
 * ray; "lazy_adam" 0 = dense Adam sweep over every selected feature row, 1 (default) = lazy replay (psl_map_iters);
 * "track_fused" launch structure of psl_track_iters: 0 = ten launches per iteration; 1 = pre / mid launches for batches <= 1024 rays;
 * 2 = the ray stage inside the decode backward (four launches); 3 (default) = also the pose step inside the k-NN launch and the
 * pose-independent ray set-up of all iterations in one launch per call (three launches; eight above 1024 rays) -- bit-identical results;
 * "knn_start_hint" bits 0 / 1: ray k-NN launches below 5 000 queries start at the pass the row lengths suggest / carry the eighth-best
 * bound of a pass that did not close into the next one (default 3); bits 2 / 3 the same for larger launches (default off) -- same answers;
 * "color_split" launch structure of the colour-stage decode (decoder.py:341-449): 0 = fused 16-sample tiles, 2 = split F_theta /
 * trunk kernels, 1 (default) = split beyond 384 tiles; "wave_trunk" = tiles from which the split structure's trunk forward runs
 * one wavefront per tile (default 1024, 0 = never).  Results are the same to fp32 rounding of one sum order (the colour head) */
int psl_debug_option(const char* name, int value);
int psl_profile_classes(void);
const char* psl_profile_name(int i);
int psl_profile_read(psl_ctx* ctx, double* ms_out, int* count_out, double* work_out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* POINTSLAM_HIP_H */
