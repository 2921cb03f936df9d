// Device helpers shared by the kernels.  Everything parity-critical is written with explicit
// round-to-nearest mul/add intrinsics so that it reproduces torch's unfused fp32 op sequence
// regardless of -ffp-contract (the library is built with -ffp-contract=off as well).
#pragma once
#include "psl_common.h"

namespace psl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// squared distance exactly as torch.sum(torch.square(a-b), -1): ((dx*dx + dy*dy) + dz*dz)
// (decoder.py:146-147 recompute; oracle.sqdist)
__device__ __forceinline__ float dist2(float ax, float ay, float az, float bx, float by, float bz) {
  float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// z_s = near*d*(1-t_s) + far*d*t_s, t = linspace(0,1,5)  (Renderer.py:134-141)
__device__ __forceinline__ float sample_z(float depth, int s, float near_s, float far_s) {
  float t = 0.25f * (float)s;
  return __fadd_rn(__fmul_rn(__fmul_rn(near_s, depth), 1.0f - t), __fmul_rn(__fmul_rn(far_s, depth), t));
}

// pts = o + d*z : separate multiply then add (Renderer.py:172-173)
__device__ __forceinline__ void sample_point(float ox, float oy, float oz, float dx, float dy, float dz, float z,
                                             float& x, float& y, float& zz) {
  x = __fadd_rn(ox, __fmul_rn(dx, z));
  y = __fadd_rn(oy, __fmul_rn(dy, z));
  zz = __fadd_rn(oz, __fmul_rn(dz, z));
}

// torch.nn.Softplus(beta=100, threshold=20): x if 100x > 20 else log1p(exp(100x))/100.
// 104 of these per sample made the forward VALU-bound with ocml's expf/log1pf (~80 instructions each, PMC:
// 22 VALU per MFMA instruction), so they run on the hardware transcendentals instead: t = 2^(100x*log2 e)
// (v_exp_f32), log1p(t) = t - t^2/2 + t^3/3 for t < 2^-10 (truncation < 3e-13) else ln2*log2(1+t) (v_log_f32).
// Absolute error of the result <= ~2e-9, i.e. below one fp32 ulp of every activation > 0.02 and far inside the
// fp32 noise of the decoders' matrix products; the parity tests bound the end-to-end effect.
__device__ __forceinline__ float softplus100(float x) {
  float bx = 100.0f * x;
  if (bx > 20.0f) return x;
  float t = __builtin_amdgcn_exp2f(bx * 1.44269504088896341f);
  float l = (t < 9.765625e-4f) ? t * (1.0f - t * (0.5f - t * 0.33333333f))
                               : __builtin_amdgcn_logf(1.0f + t) * 0.69314718055994531f;
  return l * 0.01f;
}
// The same function without control flow (selects only): inside the register-chained kernels a branch per activation
// splits the instruction stream into small blocks and stops the scheduler from overlapping weight loads with MFMAs.
// Bit-identical to softplus100 (same expressions; the unused one is discarded by the select).
__device__ __forceinline__ float softplus100_nb(float x) {
  const float bx = 100.0f * x;
  const float t = __builtin_amdgcn_exp2f(fminf(bx, 20.0f) * 1.44269504088896341f);
  const float ls = t * (1.0f - t * (0.5f - t * 0.33333333f));
  const float lb = __builtin_amdgcn_logf(1.0f + t) * 0.69314718055994531f;
  const float l = (t < 9.765625e-4f) ? ls : lb;
  return (bx > 20.0f) ? x : l * 0.01f;
}
// derivative of softplus100 expressed through its OUTPUT y: sigmoid(100 z) = 1 - exp(-100 y)
__device__ __forceinline__ float softplus100_grad_from_out(float y) {
  return (100.0f * y > 20.0f) ? 1.0f : 1.0f - __builtin_amdgcn_exp2f(-144.269504088896341f * y);
}
// sin and cos of an fp32 angle to ~1.5 ulp (max abs error 9e-8, same as libm's sinf): three-term Cody-Waite
// reduction by pi/2 (exact through fma for |x| < 1e5 -- Fourier phases here are < 1e4) and the cephes minimax
// polynomials on [-pi/4, pi/4]; ~35 VALU instructions against ~250 for the library call, whose Payne-Hanek path is
// kept for huge or non-finite arguments.  The decode kernels are VALU-issue bound, and they evaluate ~3000 of
// these per 16-sample tile.
__device__ __forceinline__ void fast_sincosf(float x, float& sn, float& cs) {
  // huge or non-finite arguments (never produced by the Fourier phases of a room-scale scene): one double-precision
  // reduction by 2 pi first; NaN / inf come out as NaN.  Evaluated unconditionally and SELECTED (five full-rate fp64
  // instructions): as a branch -- rounds 2-3 -- every call site became its own basic block, the 24 sines of a geometry tile
  // turned into 24 serial [load three B entries, wait, test, branch] steps (one cache round trip each, nothing overlapped),
  // and the scheduler could not interleave the polynomials with anything.
  {
    const double xd = (double)x;
    const float xr = (float)fma(-rint(xd * 0.15915494309189535), 6.283185307179586, xd);
    x = (fabsf(x) < 1.0e5f) ? x : xr;
  }
  const float j = rintf(__fmul_rn(x, 0.636619772f));
  float r = fmaf(j, -1.57079601e+00f, x);
  r = fmaf(j, -3.13916473e-07f, r);
  r = fmaf(j, -5.39030253e-15f, r);
  const float s = __fmul_rn(r, r);
  float p = fmaf(s, -1.9515295891e-4f, 8.3321608736e-3f);
  p = fmaf(s, p, -1.6666654611e-1f);
  const float sr = fmaf(__fmul_rn(r, s), p, r);
  float q = fmaf(s, 2.443315711809948e-5f, -1.388731625493765e-3f);
  q = fmaf(s, q, 4.166664568298827e-2f);
  const float cr = fmaf(__fmul_rn(s, s), q, fmaf(s, -0.5f, 1.0f));
  const int qi = (int)j;
  const float a = (qi & 1) ? cr : sr, b = (qi & 1) ? sr : cr;
  sn = (qi & 2) ? -a : a;
  cs = ((qi + 1) & 2) ? -b : b;
}
__device__ __forceinline__ float fast_sinf(float x) { float s, c; fast_sincosf(x, s, c); return s; }
__device__ __forceinline__ float fast_cosf(float x) { float s, c; fast_sincosf(x, s, c); return c; }

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence: hipcc drains
// vmcnt(0) in front of it, i.e. every barrier would wait for the global STORES of saved activations (HBM write
// latency, ~1-2 us each) although no wave ever reads them back inside the kernel.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// LDS-DMA: 16 bytes per lane from global memory straight into LDS at (wave-uniform base) + 16 * lane; no VGPR in between.
// The compiler does not wait for these loads before later LDS reads (checked on the ROCm 7.2 hipcc: no vmcnt in front of
// the ds_read): the issuing wave drains them itself with lds_barrier_dma() / wait_dma().
__device__ __forceinline__ void glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_barrier_dma() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// forces a value to exist at this point of the program: without it the machine-sink pass moves the arithmetic that produces
// it into a later basic block (next to its first use), away from the loads that feed it, and the loads' registers stay live
// (or are spilled) across everything in between
__device__ __forceinline__ void pin(f32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }

// orders one wave's own LDS writes before its later LDS reads (wave-private scratch needs no workgroup barrier)
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// wave-level sums
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// a double of a given lane, wave-uniform
__device__ __forceinline__ double readlane_d(double v, int l) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), l), hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// hardware float atomic add (no CAS loop)
__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

}  // namespace psl
