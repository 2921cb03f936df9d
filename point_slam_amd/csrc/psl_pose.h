// Pose arithmetic of the tracker shared by psl_slam.hip (k_track_pre, k_pose_step) and psl_grid.hip (the tracker's k-NN launch, which
// since round 6 applies the pose step of the previous iteration in its prologue): quaternion -> rotation, the analytic quaternion chain
// of dL/dR and Adam on the seven pose parameters (Tracker.py:305-311,323,183; get_camera_from_tensor common.py:251-267).
#pragma once
#include "psl_common.h"

namespace psl {

// quad2rotation (src/common.py:225-248), same operation order
__device__ __forceinline__ void quat_to_rot(const float* q, float R[3][3]) {
  float qr = q[0], qi = q[1], qj = q[2], qk = q[3];
  float two_s = 2.0f / (((qr * qr + qi * qi) + qj * qj) + qk * qk);
  R[0][0] = 1.f - two_s * (qj * qj + qk * qk);
  R[0][1] = two_s * (qi * qj - qk * qr);
  R[0][2] = two_s * (qi * qk + qj * qr);
  R[1][0] = two_s * (qi * qj + qk * qr);
  R[1][1] = 1.f - two_s * (qi * qi + qk * qk);
  R[1][2] = two_s * (qj * qk - qi * qr);
  R[2][0] = two_s * (qi * qk - qj * qr);
  R[2][1] = two_s * (qj * qk + qi * qr);
  R[2][2] = 1.f - two_s * (qi * qi + qj * qj);
}

// bias corrections of Adam step `step`, in double like torch's Python scalars
struct AdamBias { double bc1; float sqrt_bc2; };
__device__ __forceinline__ AdamBias adam_bias(int step) {
  AdamBias c;
  c.bc1 = 1.0 - pow((double)0.9f, (double)step);
  c.sqrt_bc2 = (float)sqrt(1.0 - pow((double)0.999f, (double)step));
  return c;
}
__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float lr, const AdamBias& c) {
  const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
  m = m + (1.0f - b1) * (g - m);
  v = v * b2 + ((1.0f - b2) * g) * g;
  float denom = sqrtf(v) / c.sqrt_bc2 + eps;
  p = p + ((-(float)((double)lr / c.bc1)) * m) / denom;
}
__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float lr, int step) {
  adam1(p, g, m, v, lr, adam_bias(step));
}

// one thread: quaternion chain of dL/dR, then Adam on the 7 pose parameters
__device__ __forceinline__ void pose_adam(const float (&G)[3][3], const float (&gT)[3], float* cam_tensor, float* adam_mv,
                                          int step, float lr_T, float lr_q, const AdamBias* host_bias = nullptr) {
  float qr = cam_tensor[0], qi = cam_tensor[1], qj = cam_tensor[2], qk = cam_tensor[3];
  float nn = qr * qr + qi * qi + qj * qj + qk * qk;
  float s = 2.0f / nn;
  // R = I + s*M(q)
  float M[3][3] = {{-(qj * qj + qk * qk), qi * qj - qk * qr, qi * qk + qj * qr},
                   {qi * qj + qk * qr, -(qi * qi + qk * qk), qj * qk - qi * qr},
                   {qi * qk - qj * qr, qj * qk + qi * qr, -(qi * qi + qj * qj)}};
  float GM = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int k = 0; k < 3; ++k) GM += G[a][k] * M[a][k];
  float dMr = G[0][1] * (-qk) + G[0][2] * qj + G[1][0] * qk + G[1][2] * (-qi) + G[2][0] * (-qj) + G[2][1] * qi;
  float dMi = G[0][1] * qj + G[0][2] * qk + G[1][0] * qj + G[1][1] * (-2.f * qi) + G[1][2] * (-qr) + G[2][0] * qk +
              G[2][1] * qr + G[2][2] * (-2.f * qi);
  float dMj = G[0][0] * (-2.f * qj) + G[0][1] * qi + G[0][2] * qr + G[1][0] * qi + G[1][2] * qk + G[2][0] * (-qr) +
              G[2][1] * qk + G[2][2] * (-2.f * qj);
  float dMk = G[0][0] * (-2.f * qk) + G[0][1] * (-qr) + G[0][2] * qi + G[1][0] * qr + G[1][1] * (-2.f * qk) +
              G[1][2] * qj + G[2][0] * qi + G[2][1] * qj;
  float ds = -s * s;   // d s / d q_x = -s^2 q_x
  float gq[4] = {ds * qr * GM + s * dMr, ds * qi * GM + s * dMi, ds * qj * GM + s * dMj, ds * qk * GM + s * dMk};
  // once, not per parameter; k_track_pre gets the two double pow() from the host (measured: 13.0 -> 12.5 us)
  const AdamBias bias = host_bias ? *host_bias : adam_bias(step);
#pragma unroll
  for (int j = 0; j < 4; ++j) adam1(cam_tensor[j], gq[j], adam_mv[j], adam_mv[7 + j], lr_q, bias);
#pragma unroll
  for (int j = 0; j < 3; ++j) adam1(cam_tensor[4 + j], gT[j], adam_mv[4 + j], adam_mv[11 + j], lr_T, bias);
}

// The tracker's k-NN launch with the pose step in front (psl_track_iters, batches <= 1 024 rays): EVERY workgroup reduces the ray
// gradients of the previous iteration in k_track_pre's order, steps the pose locally (same arithmetic, so all workgroups hold the same
// bits), and turns the camera-frame directions of its own rays; workgroup 0 writes the stepped pose and Adam state to the OTHER of two
// buffers (the old ones are still being read by the rest of the grid).  Replaces one 12.5-us single-workgroup launch per iteration.
struct TrackPose {
  int rotate_only;                        // 1: no step at all, pose_in is the pose (batches above 1 024 rays; POSE = 2 instantiation)
  int do_step;                            // 0 in the first iteration: the pose is pose_in
  const float4* dp; const float4* dp2;    // [n S] d(loss)/d(sample point) of the previous backward (dp2: second accumulator or null)
  const float* dirs_prev; const float* gd_prev;   // camera-frame directions / sensor depths of the previous iteration's rays
  const float* pose_in; const float* adam_in;     // [7], [14]
  float* pose_out; float* adam_out;               // written by workgroup 0 when do_step
  int step; float lr_T, lr_q; AdamBias bias;
  const float* dirs;                      // camera-frame directions of this iteration's rays [n][3]
  float* rays_o; float* rays_d;           // world-frame rays of this iteration, written for the decode kernels
  int n;
  float near_s, far_s;
};

}  // namespace psl
