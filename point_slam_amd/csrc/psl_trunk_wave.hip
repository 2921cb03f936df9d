// Colour trunk in the THROUGHPUT regime: one wavefront per 16-sample tile, all 128 output columns, registers only.
//
// Reference: MLP_color.forward (src/conv_onet/models/decoder.py:392-449) and autograd through it.
//
// The 8-wavefront tile of k_trunk_fwd / k_trunk_bwd (one wavefront per 16 output columns, hidden tile exchanged through LDS,
// one barrier per layer) is built for latency: a tile's five layers take ~12 us.  Its price is lockstep -- between two barriers
// a wavefront has 40 MFMAs (1.3 k cycles) of work, then all eight wait for the slowest one, activate, exchange, and the matrix
// pipe idles: measured 50 % MFMA-busy with two tiles per CU, and no better with a double tile (gpurun r06i, r06k).  When a
// launch has more tiles than the chip has SIMDs (25 000 samples = 1 563 tiles: the Replica / TUM / ScanNet yamls, the tracker
// of the TUM / ScanNet yamls) latency is not what is being bought.  Here a wavefront owns a whole tile, as the geometry role
// always has: the B operand of the next layer's k-group q IS accumulator tile q of this layer (psl_frag.h), so nothing is
// exchanged, there is no LDS and no barrier, and 3 independent wavefronts per SIMD keep the pipe busy across each other's
// epilogues.  Weights stream from L2 through a register ring (a fragment pair per step of 8 MFMAs, requested kAhead steps
// before it is consumed).  Same products, same order of every sum as trunk_tile_fwd / trunk_tile_bwd.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <utility>
#include <type_traits>
#include "psl_decode.h"
#include "psl_frag.h"
#include "psl_decode2.h"

namespace psl {

// compile-time loop: f(integral_constant<int, I>) for I in [B, E) -- a 200-step `#pragma unroll` loop over arrays of 200 register
// tuples is left rolled by hipcc (the arrays go to scratch); a fold over an index sequence cannot be
template <int B, class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, B + Is>{}), ...); }
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl<B>(f, std::make_integer_sequence<int, E - B>{}); }

constexpr int kTrunkLw[5] = {FL_C0, FL_C1, FL_C2, FL_C3, FL_C4};
constexpr int kTrunkFw[5] = {FL_CF0, FL_CF1, FL_CF2, FL_CF3, FL_CF4};

// ------------------------------------------------------------------------------------------------ forward
// One step = one k-group for a PAIR of output-column tiles (two independent accumulator chains, 8 MFMAs).
//   kind 0: fc_c product into u (B = interpolated colour features, two groups); kind 1: main product into z; kind 2: output layer
//   bsel: 0, 1 colour features; 2..5 embedding groups (sin 0..15, sin 16..19, cos 0..15, cos 16..19); 6..13 hidden k-groups
struct TStep { short f0, f1; signed char kind, bsel, nk, ep, layer, cp; };
struct TSteps { TStep s[216]; int n; };
constexpr TSteps make_trunk_steps() {
  TSteps t{};
  int n = 0;
  for (int i = 0; i < 5; ++i) {
    const int L = kTrunkLw[i], F = kTrunkFw[i], nq = kFLayers[L].ngroups;
    for (int cp = 0; cp < 4; ++cp) {
      const int c0 = 2 * cp, c1 = 2 * cp + 1;
      for (int q = 0; q < 2; ++q)
        t.s[n++] = TStep{(short)(ffirst(F) + c0 * 2 + q), (short)(ffirst(F) + c1 * 2 + q), 0, (signed char)q, 4, 0, (signed char)i, (signed char)cp};
      if (i == 0 || i == 3)
        for (int q = 0; q < 4; ++q)
          t.s[n++] = TStep{(short)(ffirst(L) + c0 * nq + q), (short)(ffirst(L) + c1 * nq + q), 1, (signed char)(2 + q), (signed char)((q & 1) ? 1 : 4), 0,
                           (signed char)i, (signed char)cp};
      if (i > 0) {
        const int off = (i == 3) ? 4 : 0;
        for (int q = 0; q < 8; ++q)
          t.s[n++] = TStep{(short)(ffirst(L) + c0 * nq + off + q), (short)(ffirst(L) + c1 * nq + off + q), 1, (signed char)(6 + q), 4, 0, (signed char)i,
                           (signed char)cp};
      }
      t.s[n - 1].ep = 1;
    }
  }
  for (int q = 0; q < 8; ++q) t.s[n++] = TStep{(short)(ffirst(FL_COUT) + q), -1, 2, (signed char)(6 + q), 4, 0, 5, 0};
  t.n = n;
  return t;
}
constexpr TSteps kTS = make_trunk_steps();
constexpr int kAhead = 3;      // steps between a fragment pair's request and its MFMAs (768 MFMA-cycles of this wave alone)

__device__ __forceinline__ void wave_trunk_fwd(const DecodeArgs& a, const float* __restrict__ WF, int p0) {
  const int lane = threadIdx.x & 63, rl = lane & 15, g = lane >> 4;
  const float* __restrict__ M = a.master;
  f32x4 W0[kTS.n], W1[kTS.n];
#pragma unroll
  for (int st = 0; st < kAhead; ++st) { W0[st] = ldfrag(WF, kTS.s[st].f0, lane); W1[st] = ldfrag(WF, kTS.s[st].f1, lane); }
  // B operands that do not change over the layers: interpolated colour features, Fourier features (k_nbr_fwd wrote both)
  const size_t row = (size_t)(p0 + rl);
  const f32x4 ccb0 = *reinterpret_cast<const f32x4*>(a.ws.cc + row * C + 4 * g);
  const f32x4 ccb1 = *reinterpret_cast<const f32x4*>(a.ws.cc + row * C + 16 + 4 * g);
  f32x4 esn, ecs;
  float sn4, cs4;
  {
    const float2* e2 = reinterpret_cast<const float2*>(a.ws.c_emb2 + row * EC + g * 10);
    const float2 v0 = e2[0], v1 = e2[1], v2 = e2[2], v3 = e2[3], v4 = e2[4];
    esn = f32x4{v0.x, v0.y, v1.x, v1.y}; sn4 = v2.x;
    ecs = f32x4{v2.y, v3.x, v3.y, v4.x}; cs4 = v4.y;
  }
  const float ob0 = M[MO(PI_C_OUT + 1) + 0], ob1 = M[MO(PI_C_OUT + 1) + 1], ob2 = M[MO(PI_C_OUT + 1) + 2];
  f32x4 h[8], hn[8];
  f32x4 z0, z1, u0, u1, oa = {0.f, 0.f, 0.f, 0.f}, ob = {0.f, 0.f, 0.f, 0.f};
  // biases of the first pair; those of pair cp + 1 are requested while pair cp computes
  f32x4 zb0 = ldbias(WF, fbias(FL_C0), 0, g), zb1 = ldbias(WF, fbias(FL_C0), 1, g);
  f32x4 ub0 = ldbias(WF, fbias(FL_CF0), 0, g), ub1 = ldbias(WF, fbias(FL_CF0), 1, g);
  static_for<0, kTS.n>([&](auto ST_) {
    constexpr int st = decltype(ST_)::value;
    constexpr TStep S = kTS.s[st];
    sched_fence();
    if constexpr (st + kAhead < kTS.n) {
      W0[st + kAhead] = ldfrag(WF, kTS.s[st + kAhead].f0, lane);
      if constexpr (kTS.s[st + kAhead].f1 >= 0) W1[st + kAhead] = ldfrag(WF, kTS.s[st + kAhead].f1, lane);
    }
    if (S.kind == 0 && S.bsel == 0) {        // first step of a pair: accumulators <- biases, next pair's biases requested
      z0 = zb0; z1 = zb1; u0 = ub0; u1 = ub1;
      const int i = S.layer, cp = S.cp;
      const int ni = cp == 3 ? i + 1 : i, ncp = cp == 3 ? 0 : cp + 1;
      if (ni < 5) {
        zb0 = ldbias(WF, fbias(kTrunkLw[ni < 5 ? ni : 4]), 2 * ncp, g); zb1 = ldbias(WF, fbias(kTrunkLw[ni < 5 ? ni : 4]), 2 * ncp + 1, g);
        ub0 = ldbias(WF, fbias(kTrunkFw[ni < 5 ? ni : 4]), 2 * ncp, g); ub1 = ldbias(WF, fbias(kTrunkFw[ni < 5 ? ni : 4]), 2 * ncp + 1, g);
      }
    }
    const int bs = S.bsel;
    const f32x4 b = bs == 0 ? ccb0 : bs == 1 ? ccb1 : bs == 2 ? esn : bs == 3 ? f32x4{sn4, 0.f, 0.f, 0.f} : bs == 4 ? ecs
                  : bs == 5 ? f32x4{cs4, 0.f, 0.f, 0.f} : h[bs >= 6 ? bs - 6 : 0];
    if (S.kind == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { u0 = mfma16(W0[st][r], b[r], u0); u1 = mfma16(W1[st][r], b[r], u1); }
    } else if (S.kind == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (r < S.nk) { z0 = mfma16(W0[st][r], b[r], z0); z1 = mfma16(W1[st][r], b[r], z1); }
    } else {
      if (st & 1) mma4(ob, W0[st], b); else mma4(oa, W0[st], b);
    }
    if (S.ep) {      // the pair's activation: h = softplus(z) + u  (decoder.py:421-428)
      const int i = S.layer, c0 = 2 * S.cp;
      f32x4 y0, y1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        y0[r] = softplus100_nb(z0[r]); hn[c0][r] = y0[r] + u0[r];
        y1[r] = softplus100_nb(z1[r]); hn[c0 + 1][r] = y1[r] + u1[r];
      }
      if (a.ws.c_y) {
        const size_t o = ((size_t)i * a.ws.Ppad + p0 + rl) * HC + c0 * 16 + 4 * g;
        *reinterpret_cast<f32x4*>(a.ws.c_y + o) = y0;
        *reinterpret_cast<f32x4*>(a.ws.c_y + o + 16) = y1;
        if (a.ws.c_hin) {
          *reinterpret_cast<f32x4*>(a.ws.c_hin + o) = hn[c0];
          *reinterpret_cast<f32x4*>(a.ws.c_hin + o + 16) = hn[c0 + 1];
        }
      }
      if (S.cp == 3) {
#pragma unroll
        for (int q = 0; q < 8; ++q) h[q] = hn[q];
      }
    }
  });
  // ---- colour head (decoder.py:430-448): row = sample rl, outputs 0..2 in the lanes of g = 0
  if (g == 0 && p0 + rl < a.P) {
    const int pp = p0 + rl;
    float r0 = (oa[0] + ob[0]) + ob0, r1 = (oa[1] + ob[1]) + ob1, r2 = (oa[2] + ob[2]) + ob2;
    a.ws.out3[(size_t)pp * 4 + 0] = r0; a.ws.out3[(size_t)pp * 4 + 1] = r1; a.ws.out3[(size_t)pp * 4 + 2] = r2;
    if (a.flags & PSL_HAS_AFFINE) {  // out @ rot + trans (decoder.py:433-436)
      const float* A = a.affine;
      const float q0 = r0 * A[0] + r1 * A[3] + r2 * A[6] + A[9];
      const float q1 = r0 * A[1] + r1 * A[4] + r2 * A[7] + A[10];
      const float q2 = r0 * A[2] + r1 * A[5] + r2 * A[8] + A[11];
      r0 = q0; r1 = q1; r2 = q2;
    }
    if (!(a.flags & PSL_NO_SIGMOID)) { r0 = sigmoidf(r0); r1 = sigmoidf(r1); r2 = sigmoidf(r2); }
    a.ws.raw[(size_t)pp * 4 + 0] = r0; a.ws.raw[(size_t)pp * 4 + 1] = r1; a.ws.raw[(size_t)pp * 4 + 2] = r2;
  }
}

__global__ __launch_bounds__(256, 3) void k_trunk_fwd_w(DecodeArgs a, const float* __restrict__ WF, int tiles) {
  __builtin_amdgcn_s_setprio(1);      // above the side-stream k-NN prefetch (see k_decode_fwd2)
  BlkTrace bt(a);
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int tile = (int)blockIdx.x * 4 + wave;
  if (tile < tiles) wave_trunk_fwd(a, WF, tile * TILE);
  bt.done(a);
}

int launch_trunk_fwd_w(psl_ctx* ctx, const DecodeArgs& a, int tiles, bool last, hipStream_t s) {
  PSL_KLAUNCH2(k_trunk_fwd_w, false, last, dim3((tiles + 3) / 4), dim3(256), 0, s, a, (const float*)ctx->wf, tiles);
  PSL_LAUNCH_CHECK();
  return PSL_OK;
}

}  // namespace psl
