// Per-frame image operators that sit in front of the render/optimise path.  The reference runs
// them on the host with skimage / scipy / numpy once per tracked and per mapped frame:
//   * rgb2gray + Sobel magnitude + clip + piecewise-linear map -> per-pixel dynamic radii
//     (src/Tracker.py:235-250, src/Mapper.py:686-701);
//   * the top ratio*n pixels by colour-gradient magnitude, masked by region and sensor depth
//     (get_selected_index_with_grad, src/common.py:116-159);
//   * overlap of the current view with every keyframe (keyframe_selection_overlap, src/Mapper.py:170-235).
// All arithmetic that the reference does in float64 (numpy) is done in float64 here; the radii leave as float32.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#include <mutex>
#include "psl_common.h"

namespace psl {

// skimage.color.rgb2gray weights (ITU-R 601-2 luma as used by skimage: 0.2125, 0.7154, 0.0721)
__device__ __forceinline__ double gray_at(const float* __restrict__ color, int H, int W, int y, int x) {
  // scipy.ndimage 'reflect' boundary: (d c b a | a b c d | d c b a) -> index -1 -> 0, index n -> n-1
  y = y < 0 ? -y - 1 : (y >= H ? 2 * H - 1 - y : y);
  x = x < 0 ? -x - 1 : (x >= W ? 2 * W - 1 - x : x);
  const float* p = color + ((size_t)y * W + x) * 3;
  return 0.2125 * (double)p[0] + 0.7154 * (double)p[1] + 0.0721 * (double)p[2];
}

// sobel_h / sobel_v of skimage 0.19 = ndi.convolve with [1,0,-1] (x) [1,2,1]/4, mode='reflect'; magnitude;
// np.clip(., 0, thr); interp1d([0, 0.01, thr], [rmax, rmax, rmin]) and the same with ratio*r for the query radius.
__global__ __launch_bounds__(256) void k_frame_radii(const float* __restrict__ color, int H, int W, double thr, double rmax,
                                                     double rmin, double ratio, double* __restrict__ grad_mag,
                                                     float* __restrict__ r_add, float* __restrict__ r_query) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * W) return;
  const int y = i / W, x = i - y * W;
  double g[3][3];
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) g[dy + 1][dx + 1] = gray_at(color, H, W, y + dy, x + dx);
  // derivative along rows (axis 0), smoothed along columns -- and the transpose
  const double gy = ((g[0][0] - g[2][0]) + 2.0 * (g[0][1] - g[2][1]) + (g[0][2] - g[2][2])) * 0.25;
  const double gx = ((g[0][0] - g[0][2]) + 2.0 * (g[1][0] - g[1][2]) + (g[2][0] - g[2][2])) * 0.25;
  const double mag = sqrt(gx * gx + gy * gy);
  if (grad_mag) grad_mag[i] = mag;
  const double c = fmin(fmax(mag, 0.0), thr);
  double ra, rq;
  if (c <= 0.01) { ra = rmax; rq = ratio * rmax; }
  else {
    // scipy interp1d (linear): slope * (x - x_lo) + y_lo on the segment [0.01, thr]
    const double sa = (rmin - rmax) / (thr - 0.01), sq = (ratio * rmin - ratio * rmax) / (thr - 0.01);
    ra = sa * (c - 0.01) + rmax;
    rq = sq * (c - 0.01) + ratio * rmax;
  }
  if (r_add) r_add[i] = (float)ra;
  if (r_query) r_query[i] = (float)rq;
}

// ---- exact top-k by value: 4-pass radix select over the bit patterns of the (non-negative) doubles ----
__global__ void k_topk_init(unsigned long long* state, unsigned long long k) { state[0] = 0ull; state[1] = k; }

__global__ __launch_bounds__(256) void k_topk_hist(const double* __restrict__ v, int n, int pass,
                                                   const unsigned long long* __restrict__ state,
                                                   unsigned* __restrict__ hist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long b = (unsigned long long)__double_as_longlong(v[i]);
  const int shift = 48 - 16 * pass;
  if (pass > 0 && (b >> (shift + 16)) != state[0]) return;
  atomicAdd(&hist[(unsigned)((b >> shift) & 0xFFFFull)], 1u);
}

// one workgroup: walk the 65536 bins from the top until the k-th largest falls into a bin; state = {prefix, k_left}
__global__ __launch_bounds__(1024) void k_topk_pick(unsigned* __restrict__ hist, unsigned long long* __restrict__ state) {
  __shared__ unsigned part[1024];
  __shared__ unsigned long long s_prefix;
  __shared__ unsigned s_left;
  const int t = threadIdx.x;
  // thread t owns bins [64t, 64t+64), t counted from the TOP
  const int top = 65535 - 64 * t;
  unsigned sum = 0;
  for (int j = 0; j < 64; ++j) sum += hist[top - j];
  part[t] = sum;
  __syncthreads();
  if (t == 0) {
    unsigned left = (unsigned)state[1];
    int seg = 0;
    while (seg < 1023 && part[seg] < left) { left -= part[seg]; ++seg; }
    int bin = 65535 - 64 * seg;
    for (int j = 0; j < 63 && hist[bin] < left; ++j) { left -= hist[bin]; --bin; }
    s_prefix = (state[0] << 16) | (unsigned long long)bin;
    s_left = left;
  }
  __syncthreads();
  for (int j = t; j < 65536; j += 1024) hist[j] = 0;      // ready for the next pass
  if (t == 0) { state[0] = s_prefix; state[1] = s_left; }
}

// emit the pixels above the threshold, and `k_left` of those equal to it (np.argpartition leaves the choice among
// equal values open); region and sensor-depth masks are applied to what is emitted (common.py:144-155)
__global__ __launch_bounds__(256) void k_topk_emit(const double* __restrict__ v, const float* __restrict__ depth, int H, int W,
                                                   const unsigned long long* __restrict__ state, int H0, int H1, int W0,
                                                   int W1, float depth_limit, int* __restrict__ counters,
                                                   int* __restrict__ sel) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * W) return;
  const unsigned long long b = (unsigned long long)__double_as_longlong(v[i]);
  const unsigned long long T = state[0];
  bool take = b > T;
  if (b == T) take = atomicAdd(&counters[1], 1) < (int)state[1];
  if (!take) return;
  const int y = i / W, x = i - y * W;
  if (y < H0 || y >= H1 || x < W0 || x >= W1) return;
  if (depth) {
    const float d = depth[i];
    if (!(d > 0.f)) return;
    if (depth_limit > 0.f && !(d <= depth_limit)) return;
  }
  sel[atomicAdd(&counters[0], 1)] = i;
}

// keyframe_selection_overlap (Mapper.py:197-229): share of the current frame's frustum samples that project inside
// keyframe kf (20-pixel border, in front of the camera).  One workgroup per keyframe; the 8 samples per ray run
// from 0.8*d to d+0.5 (linspace).
__global__ __launch_bounds__(256) void k_keyframe_overlap(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                          const float* __restrict__ depth, int n_rays, int n_samples,
                                                          const float* __restrict__ w2c /*[n_kf][12]*/, psl_cam_intr cam,
                                                          float edge, float* __restrict__ percent) {
  __shared__ int cnt[4], cnt_all[4];
  const float* M = w2c + (size_t)blockIdx.x * 12;
  int c = 0, call = 0;
  const int total = n_rays * n_samples;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int r = e / n_samples, k = e - r * n_samples;
    const float d = depth[r];
    // get_samples(..., depth_filter=True) (Mapper.py:190-192, common.py:173-179): pixels without sensor depth carry no
    // samples -- skipped here, so that the caller need not compact the batch (a host synchronisation per mapped frame)
    if (!(d > 0.f)) continue;
    ++call;
    // torch.linspace(0,1,N) on near = 0.8 d, far = d + 0.5:  z = near*(1-t) + far*t
    const float t = (n_samples > 1) ? (float)k / (float)(n_samples - 1) : 0.f;
    const float z = __fadd_rn(__fmul_rn(__fmul_rn(d, 0.8f), 1.0f - t), __fmul_rn(__fadd_rn(d, 0.5f), t));
    const float px = __fadd_rn(rays_o[r * 3 + 0], __fmul_rn(rays_d[r * 3 + 0], z));
    const float py = __fadd_rn(rays_o[r * 3 + 1], __fmul_rn(rays_d[r * 3 + 1], z));
    const float pz = __fadd_rn(rays_o[r * 3 + 2], __fmul_rn(rays_d[r * 3 + 2], z));
    const float cx = M[0] * px + M[1] * py + M[2] * pz + M[3];
    const float cy = M[4] * px + M[5] * py + M[6] * pz + M[7];
    const float cz = M[8] * px + M[9] * py + M[10] * pz + M[11];
    // cam_cord[:,0] *= -1 ; uv = K @ cam_cord ; z = uv[2] + 1e-5 ; uv /= z   (float64 in the reference from here)
    const double zz = (double)cz + 1e-5;
    const float u = (float)(((double)cam.fx * (double)(-cx) + (double)cam.cx * (double)cz) / zz);
    const float v = (float)(((double)cam.fy * (double)cy + (double)cam.cy * (double)cz) / zz);
    const bool in = (u < (float)cam.W - edge) && (u > edge) && (v < (float)cam.H - edge) && (v > edge) && (zz < 0.0);
    c += in ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { c += __shfl_xor(c, o); call += __shfl_xor(call, o); }
  if ((threadIdx.x & 63) == 0) { cnt[threadIdx.x >> 6] = c; cnt_all[threadIdx.x >> 6] = call; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tot = cnt_all[0] + cnt_all[1] + cnt_all[2] + cnt_all[3];
    percent[blockIdx.x] = tot > 0 ? (float)((double)(cnt[0] + cnt[1] + cnt[2] + cnt[3]) / (double)tot) : 0.f;
  }
}


// ---- end-of-run image metrics (src/Mapper.py:861-879): PSNR and depth L1 over the pixels with sensor depth, and
// MS-SSIM as pytorch_msssim 0.2.x computes it (11-tap Gaussian sigma 1.5, 'valid' separable filtering, 5 scales,
// 2x2 average pooling with padding = size % 2).  float32 maps like the reference, float64 reductions.
struct GaussWin { float w[11]; };

__global__ __launch_bounds__(256) void k_metric_sums(const float* __restrict__ gt_color, const float* __restrict__ gt_depth,
                                                     const float* __restrict__ color, const float* __restrict__ depth,
                                                     int n, double* __restrict__ acc /*[3]: sq, count, |dd|*/) {
  __shared__ double red[3][4];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double sq = 0.0, cnt = 0.0, ad = 0.0;
  if (i < n && gt_depth[i] > 0.f) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { double d = (double)gt_color[i * 3 + c] - (double)color[i * 3 + c]; sq += d * d; }
    cnt = 1.0;
    ad = fabs((double)gt_depth[i] - (double)depth[i]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sq += __shfl_xor(sq, o); cnt += __shfl_xor(cnt, o); ad += __shfl_xor(ad, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sq; red[1][threadIdx.x >> 6] = cnt; red[2][threadIdx.x >> 6] = ad; }
  __syncthreads();
  if (threadIdx.x < 3) {
    double t = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    if (t != 0.0) atomicAdd(&acc[threadIdx.x], t);
  }
}

// [H][W][3] interleaved -> planar [3][H][W] for both images
__global__ __launch_bounds__(256) void k_planar(const float* __restrict__ a, const float* __restrict__ b, int n,
                                                float* __restrict__ X, float* __restrict__ Y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int c = 0; c < 3; ++c) { X[(size_t)c * n + i] = a[i * 3 + c]; Y[(size_t)c * n + i] = b[i * 3 + c]; }
}

// horizontal 11-tap pass of X, Y, XX, YY, XY: out [5][3][h][w-10]
__global__ __launch_bounds__(256) void k_ssim_rows(const float* __restrict__ X, const float* __restrict__ Y, int h, int w,
                                                   GaussWin g, float* __restrict__ out) {
  const int wo = w - 10;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3LL * h * wo) return;
  const int x = (int)(i % wo);
  const long long cy = i / wo;                       // c * h + y
  const float* xr = X + cy * w + x;
  const float* yr = Y + cy * w + x;
  float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    const float xv = xr[k], yv = yr[k], wk = g.w[k];
    a += wk * xv; b += wk * yv; aa += wk * (xv * xv); bb += wk * (yv * yv); ab += wk * (xv * yv);
  }
  const size_t plane = (size_t)3 * h * wo;
  out[i] = a; out[plane + i] = b; out[2 * plane + i] = aa; out[3 * plane + i] = bb; out[4 * plane + i] = ab;
}

// vertical pass + SSIM / contrast-structure maps + sums per channel: acc[c][0] += ssim, acc[c][1] += cs
__global__ __launch_bounds__(256) void k_ssim_cols(const float* __restrict__ t, int h, int wo, GaussWin g, float C1, float C2,
                                                   double* __restrict__ acc) {
  __shared__ double red[2][4];
  const int ho = h - 10;
  const int c = blockIdx.y;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  double s_ssim = 0.0, s_cs = 0.0;
  if (i < (long long)ho * wo) {
    const int x = (int)(i % wo), y = (int)(i / wo);
    const size_t plane = (size_t)3 * h * wo;
    const float* p = t + ((size_t)c * h + y) * wo + x;
    float m[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < 11; ++k) v += g.w[k] * p[q * plane + (size_t)k * wo];
      m[q] = v;
    }
    const float mu1 = m[0], mu2 = m[1];
    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
    const float s1 = m[2] - mu1_sq, s2 = m[3] - mu2_sq, s12 = m[4] - mu12;
    const float cs = (2.f * s12 + C2) / (s1 + s2 + C2);
    const float ss = ((2.f * mu12 + C1) / (mu1_sq + mu2_sq + C1)) * cs;
    s_ssim = (double)ss; s_cs = (double)cs;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s_ssim += __shfl_xor(s_ssim, o); s_cs += __shfl_xor(s_cs, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s_ssim; red[1][threadIdx.x >> 6] = s_cs; }
  __syncthreads();
  if (threadIdx.x < 2) atomicAdd(&acc[c * 2 + threadIdx.x], red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// F.avg_pool2d(kernel 2, padding = size % 2, zeros counted) on a planar [3][h][w] image
__global__ __launch_bounds__(256) void k_pool2(const float* __restrict__ in, int h, int w, float* __restrict__ out, int h2,
                                               int w2) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3LL * h2 * w2) return;
  const int x = (int)(i % w2), y = (int)((i / w2) % h2), c = (int)(i / ((long long)w2 * h2));
  const int y0 = 2 * y - (h & 1), x0 = 2 * x - (w & 1);
  float s = 0.f;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int yy = y0 + dy, xx = x0 + dx;
      if (yy >= 0 && yy < h && xx >= 0 && xx < w) s += in[((size_t)c * h + yy) * w + xx];
    }
  out[i] = s * 0.25f;
}

}  // namespace psl

using namespace psl;

extern "C" int psl_frame_radii(const float* color, int32_t H, int32_t W, float color_grad_threshold, float radius_add_max,
                               float radius_add_min, float radius_query_ratio, double* grad_mag_out, float* r_add_out,
                               float* r_query_out, void* stream) {
  if (!color || H <= 0 || W <= 0 || !(color_grad_threshold > 0.01f)) { set_error("psl_frame_radii: bad argument"); return PSL_ERR_ARG; }
  hipLaunchKernelGGL(k_frame_radii, dim3((H * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, color, H, W,
                     (double)color_grad_threshold, (double)radius_add_max, (double)radius_add_min,
                     (double)radius_query_ratio, grad_mag_out, r_add_out, r_query_out);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}

extern "C" int psl_topgrad_select_sync(psl_ctx* ctx, const double* grad_mag, const float* depth, int32_t H, int32_t W,
                                       int32_t k, int32_t H0, int32_t H1, int32_t W0, int32_t W1, float depth_limit,
                                       int32_t* sel_out, int* n_sel_host, void* stream) {
  if (!ctx || !grad_mag || !sel_out || !n_sel_host || H <= 0 || W <= 0 || k < 0) { set_error("psl_topgrad_select_sync: bad argument"); return PSL_ERR_ARG; }
  *n_sel_host = 0;
  const int n = H * W;
  if (k > n) k = n;
  if (k == 0) return PSL_OK;
  hipStream_t s = (hipStream_t)stream;
  if (!ctx->img_hist) {
    PSL_HIP(hipMalloc(&ctx->img_hist, sizeof(unsigned) * 65536 + 64)); psl::poison(ctx->img_hist, sizeof(unsigned) * 65536 + 64);
  }
  unsigned* hist = ctx->img_hist;
  unsigned long long* state = (unsigned long long*)(hist + 65536);     // {prefix, k_left} then 2 int counters
  int* counters = (int*)(state + 2);
  PSL_HIP(hipMemsetAsync(hist, 0, sizeof(unsigned) * 65536 + 64, s));
  {
    hipLaunchKernelGGL(k_topk_init, dim3(1), dim3(1), 0, s, state, (unsigned long long)k);
    for (int pass = 0; pass < 4; ++pass) {
      hipLaunchKernelGGL(k_topk_hist, dim3((n + 255) / 256), dim3(256), 0, s, grad_mag, n, pass, state, hist);
      hipLaunchKernelGGL(k_topk_pick, dim3(1), dim3(1024), 0, s, hist, state);
    }
    hipLaunchKernelGGL(k_topk_emit, dim3((n + 255) / 256), dim3(256), 0, s, grad_mag, depth, H, W, state, H0, H1, W0, W1,
                       depth_limit, counters, sel_out);
    PSL_LAUNCH_CHECK();
  }
  int tot = 0;
  PSL_HIP(hipMemcpyAsync(&tot, counters, sizeof(int), hipMemcpyDeviceToHost, s));
  PSL_HIP(hipStreamSynchronize(s));
  *n_sel_host = tot;
  return PSL_OK;
}

extern "C" int psl_keyframe_overlap_sync(const float* rays_o, const float* rays_d, const float* depth, int32_t n_rays,
                                         int32_t n_samples, const float* c2w_host /*[n_kf][16] row-major 4x4*/,
                                         int32_t n_kf, psl_cam_intr cam, float edge, float* percent_host, void* stream) {
  if (!rays_o || !rays_d || !depth || !c2w_host || !percent_host || n_rays < 0 || n_samples <= 0 || n_kf < 0) {
    set_error("psl_keyframe_overlap_sync: bad argument"); return PSL_ERR_ARG;
  }
  if (n_kf == 0) return PSL_OK;
  hipStream_t s = (hipStream_t)stream;
  std::vector<float> w2c((size_t)n_kf * 12);
  for (int f = 0; f < n_kf; ++f) {
    const float* c = c2w_host + (size_t)f * 16;
    // np.linalg.inv of the 4x4 pose (last row 0 0 0 1): inverse of the 3x3 block and -R^-1 T, in double
    double R[3][3], T[3];
    for (int a = 0; a < 3; ++a) { for (int b = 0; b < 3; ++b) R[a][b] = c[a * 4 + b]; T[a] = c[a * 4 + 3]; }
    double det = R[0][0] * (R[1][1] * R[2][2] - R[1][2] * R[2][1]) - R[0][1] * (R[1][0] * R[2][2] - R[1][2] * R[2][0]) +
                 R[0][2] * (R[1][0] * R[2][1] - R[1][1] * R[2][0]);
    double inv[3][3];
    inv[0][0] = (R[1][1] * R[2][2] - R[1][2] * R[2][1]) / det; inv[0][1] = (R[0][2] * R[2][1] - R[0][1] * R[2][2]) / det;
    inv[0][2] = (R[0][1] * R[1][2] - R[0][2] * R[1][1]) / det; inv[1][0] = (R[1][2] * R[2][0] - R[1][0] * R[2][2]) / det;
    inv[1][1] = (R[0][0] * R[2][2] - R[0][2] * R[2][0]) / det; inv[1][2] = (R[0][2] * R[1][0] - R[0][0] * R[1][2]) / det;
    inv[2][0] = (R[1][0] * R[2][1] - R[1][1] * R[2][0]) / det; inv[2][1] = (R[0][1] * R[2][0] - R[0][0] * R[2][1]) / det;
    inv[2][2] = (R[0][0] * R[1][1] - R[0][1] * R[1][0]) / det;
    for (int a = 0; a < 3; ++a) {
      for (int b = 0; b < 3; ++b) w2c[(size_t)f * 12 + a * 4 + b] = (float)inv[a][b];
      w2c[(size_t)f * 12 + a * 4 + 3] = (float)(-(inv[a][0] * T[0] + inv[a][1] * T[1] + inv[a][2] * T[2]));
    }
  }
  // scratch for the poses and the answers: kept per device and grown on demand (a hipMalloc + hipFree pair per call cost
  // two device-wide synchronisations per mapped frame).  Process-lifetime, shared by every context and host thread of the
  // process: the lock is held until this (synchronous) call has read its answers back, so a second thread can neither grow
  // the buffer under a kernel in flight nor overwrite its poses (advisor, round 5).
  static float* g_dev[64] = {nullptr};
  static size_t g_cap[64] = {0};
  static std::mutex g_mu;
  std::lock_guard<std::mutex> lock(g_mu);
  int devid = 0;
  PSL_HIP(hipGetDevice(&devid));
  if (devid < 0 || devid >= 64) { set_error("psl_keyframe_overlap_sync: device %d", devid); return PSL_ERR_ARG; }
  const size_t need = (size_t)n_kf * 13;
  if (g_cap[devid] < need) {
    if (g_dev[devid]) (void)hipFree(g_dev[devid]);
    g_dev[devid] = nullptr; g_cap[devid] = 0;
    PSL_HIP(hipMalloc(&g_dev[devid], sizeof(float) * (need + 13 * 64)));
    g_cap[devid] = need + 13 * 64;
  }
  float* dev = g_dev[devid];
  float* dpct = dev + (size_t)n_kf * 12;
  PSL_HIP(hipMemcpyAsync(dev, w2c.data(), sizeof(float) * (size_t)n_kf * 12, hipMemcpyHostToDevice, s));   // pageable source: staged before it returns
  hipLaunchKernelGGL(k_keyframe_overlap, dim3(n_kf), dim3(256), 0, s, rays_o, rays_d, depth, n_rays, n_samples, dev, cam,
                     edge, dpct);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(percent_host, dpct, sizeof(float) * n_kf, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) { set_error("psl_keyframe_overlap_sync: %s", hipGetErrorString(e)); return PSL_ERR_HIP; }
  return PSL_OK;
}

extern "C" int psl_image_metrics_sync(const float* gt_color, const float* gt_depth, const float* color, const float* depth,
                                      int32_t H, int32_t W, double* out3_host /*psnr, ms_ssim, depth_l1*/, void* stream) {
  if (!gt_color || !gt_depth || !color || !depth || !out3_host || H <= 0 || W <= 0) { set_error("psl_image_metrics_sync: bad argument"); return PSL_ERR_ARG; }
  if (std::min(H, W) <= 160) { set_error("psl_image_metrics_sync: MS-SSIM needs min(H, W) > 160 (5 scales, 11-tap window)"); return PSL_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  const int n = H * W;
  // scratch: two planar pyramids (ping-pong per level), the 5 row-filtered maps, 3 + 5*3*2 double accumulators
  float* buf = nullptr;
  const size_t plane3 = (size_t)3 * n;
  const size_t floats = (4 * plane3 + 5 * plane3 + 3) / 4 * 4;     // the double accumulators behind stay 16-B aligned
  PSL_HIP(hipMalloc(&buf, sizeof(float) * floats + sizeof(double) * 40));
  float *X = buf, *Y = buf + plane3, *X2 = buf + 2 * plane3, *Y2 = buf + 3 * plane3, *T = buf + 4 * plane3;
  double* acc = (double*)(buf + floats);
  hipError_t e = hipMemsetAsync(acc, 0, sizeof(double) * 40, s);
  GaussWin g;
  { // _fspecial_gauss_1d(11, 1.5) in float32 like torch
    float sum = 0.f;
    for (int k = 0; k < 11; ++k) { float c = (float)(k - 5); g.w[k] = expf(-(c * c) / (2.0f * 1.5f * 1.5f)); sum += g.w[k]; }
    for (int k = 0; k < 11; ++k) g.w[k] /= sum;
  }
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  hipLaunchKernelGGL(k_metric_sums, dim3((n + 255) / 256), dim3(256), 0, s, gt_color, gt_depth, color, depth, n, acc);
  hipLaunchKernelGGL(k_planar, dim3((n + 255) / 256), dim3(256), 0, s, gt_color, color, n, X, Y);
  int h = H, w = W;
  long long counts[5];
  for (int l = 0; l < 5; ++l) {
    const int wo = w - 10, ho = h - 10;
    counts[l] = (long long)wo * ho;
    hipLaunchKernelGGL(k_ssim_rows, dim3((unsigned)((3LL * h * wo + 255) / 256)), dim3(256), 0, s, X, Y, h, w, g, T);
    hipLaunchKernelGGL(k_ssim_cols, dim3((unsigned)(((long long)ho * wo + 255) / 256), 3), dim3(256), 0, s, T, h, wo, g, C1, C2,
                       acc + 3 + l * 6);
    if (l < 4) {
      const int h2 = (h + 1) / 2, w2 = (w + 1) / 2;
      hipLaunchKernelGGL(k_pool2, dim3((unsigned)((3LL * h2 * w2 + 255) / 256)), dim3(256), 0, s, X, h, w, X2, h2, w2);
      hipLaunchKernelGGL(k_pool2, dim3((unsigned)((3LL * h2 * w2 + 255) / 256)), dim3(256), 0, s, Y, h, w, Y2, h2, w2);
      std::swap(X, X2); std::swap(Y, Y2);
      h = h2; w = w2;
    }
  }
  double host[40];
  if (e == hipSuccess) e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(host, acc, sizeof(host), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(buf);
  if (e != hipSuccess) { set_error("psl_image_metrics_sync: %s", hipGetErrorString(e)); return PSL_ERR_HIP; }
  const double cnt = host[1];
  out3_host[0] = cnt > 0 ? -10.0 * log10(host[0] / (3.0 * cnt)) : 0.0;
  out3_host[2] = cnt > 0 ? host[2] / cnt : 0.0;
  static const double wts[5] = {0.0448, 0.2856, 0.3001, 0.2363, 0.1333};
  double ms = 0.0;
  for (int c = 0; c < 3; ++c) {
    double prod = 1.0;
    for (int l = 0; l < 5; ++l) {
      // float32 means like torch, relu, then the weighted power
      const float ssim = (float)(host[3 + l * 6 + c * 2] / (double)counts[l]);
      const float cs = (float)(host[3 + l * 6 + c * 2 + 1] / (double)counts[l]);
      const float v = std::max(l < 4 ? cs : ssim, 0.f);
      prod *= pow((double)v, wts[l]);
    }
    ms += prod;
  }
  out3_host[1] = ms / 3.0;
  return PSL_OK;
}
