// Fragment-major weight layouts for the register-chained ("transposed") decode kernels.
//
// The decoders are evaluated as  H^T[out][sample] = W[out][in] . X^T[in][sample]  with v_mfma_f32_16x16x4_f32:
//   A operand = the weights      : lane l supplies W[16*nt + (l & 15)][k(l >> 4)]
//   B operand = the activations  : lane l supplies X[sample = l & 15][k(l >> 4)]
//   D         = H^T tile         : lane l receives H[sample = l & 15][16*nt + 4*(l >> 4) + r], r = 0..3
// A dot product does not care in which order its k index is walked as long as A and B agree, so the k-steps of the
// NEXT layer are numbered (q, r) -> input channel 16*q + 4*g + r (g = l >> 4): the B operand of k-step (q, r) is then
// exactly register r of accumulator tile q of the previous layer.  Activations never leave the register file inside
// a wavefront, and a wavefront's A operand for four consecutive k-steps is ONE 16-byte load from a buffer that holds
// every (out-tile, k-group) fragment as 64 lanes x 4 floats = 1 KiB, contiguous ("fragment-major").
//
// Fourier inputs use their own k numbering so that a lane evaluates each sin/cos pair once:
//   colour embedding  [sin(20) | cos(20)] (decoder.py:302-306,411): lane (sample, g) evaluates frequencies f = 4 s + g,
//                     s = 0..4; k-step (h, s) carries channel 20 h + 4 s + g  (h = 0 sin, 1 cos)
//   rel-pos embedding [sin(10) | cos(10)] (decoder.py:371-378): lane (row, g) evaluates f = 2 s + (g >> 1), s = 0..4,
//                     and keeps sin (g even) or cos (g odd); k-step s carries channel 10 (g & 1) + 2 s + (g >> 1)
#pragma once
#include "psl_common.h"

namespace psl {

constexpr int FK_STD = 0, FK_CEMB = 1, FK_REL = 2;
struct FGroup { int kind, base, lim; };   // STD: channel = base + 4 g + r, valid below lim; CEMB: base = 2 h + sgrp; REL: base = sgrp

// input channel of k-step r of group `grp` for lane group g (-1: zero weight, the k-step is not issued or is padding)
__host__ __device__ constexpr int frag_chan(FGroup grp, int g, int r) {
  if (grp.kind == FK_STD) { const int c = grp.base + 4 * g + r; return c < grp.lim ? c : -1; }
  if (grp.kind == FK_CEMB) {
    const int h = grp.base >> 1, sg = grp.base & 1;
    if (sg == 0) return h * ECF + 4 * r + g;
    return r == 0 ? h * ECF + 16 + g : -1;
  }
  const int s = grp.base == 0 ? r : (r == 0 ? 4 : -1);
  return s < 0 ? -1 : (g & 1) * ERF + 2 * s + (g >> 1);
}

// ---------------------------------------------------------------- forward fragments (A = W[out][in])
struct FLayer { int pi, N, K, ntiles, ngroups, g0; };   // parameter index (weight; bias = pi + 1), torch shape [N][K]
constexpr FGroup kFGroups[] = {
    // 0: F_theta linear1 [128][52] = [sin10 cos10 | feat32]: two feature groups, rel-pos s = 0..3, rel-pos s = 4
    {FK_STD, ER, ER + 16}, {FK_STD, ER + 16, ER + 32}, {FK_REL, 0, 0}, {FK_REL, 1, 0},
    // 4: eight standard groups over 128 inputs
    {FK_STD, 0, 16}, {FK_STD, 16, 32}, {FK_STD, 32, 48}, {FK_STD, 48, 64}, {FK_STD, 64, 80}, {FK_STD, 80, 96}, {FK_STD, 96, 112},
    {FK_STD, 112, 128},
    // 12: colour embedding (40)
    {FK_CEMB, 0, 0}, {FK_CEMB, 1, 0}, {FK_CEMB, 2, 0}, {FK_CEMB, 3, 0},
    // 16: colour skip layer [emb 40 | h 128]
    {FK_CEMB, 0, 0}, {FK_CEMB, 1, 0}, {FK_CEMB, 2, 0}, {FK_CEMB, 3, 0},
    {FK_STD, EC, EC + 16}, {FK_STD, EC + 16, EC + 32}, {FK_STD, EC + 32, EC + 48}, {FK_STD, EC + 48, EC + 64}, {FK_STD, EC + 64, EC + 80},
    {FK_STD, EC + 80, EC + 96}, {FK_STD, EC + 96, EC + 112}, {FK_STD, EC + 112, EC + 128},
    // 28: two standard groups over 32 inputs (fc_c, geometry hidden layers)
    {FK_STD, 0, 16}, {FK_STD, 16, 32},
    // 30: geometry embedding (93 sin values, padded to 96)
    {FK_STD, 0, EG}, {FK_STD, 16, EG}, {FK_STD, 32, EG}, {FK_STD, 48, EG}, {FK_STD, 64, EG}, {FK_STD, 80, EG},
    // 36: geometry skip layer [emb 93 | h 32]
    {FK_STD, 0, EG}, {FK_STD, 16, EG}, {FK_STD, 32, EG}, {FK_STD, 48, EG}, {FK_STD, 64, EG}, {FK_STD, 80, EG},
    {FK_STD, EG, EG + 16}, {FK_STD, EG + 16, EG + 32},
};
constexpr int FG_N1 = 0, FG_H128 = 4, FG_CEMB = 12, FG_CSKIP = 16, FG_H32 = 28, FG_GEMB = 30, FG_GSKIP = 36;

enum FLayerId {
  FL_N1 = 0, FL_N2, FL_C0, FL_C1, FL_C2, FL_C3, FL_C4, FL_CF0, FL_CF1, FL_CF2, FL_CF3, FL_CF4, FL_COUT,
  FL_G0, FL_G1, FL_G2, FL_G3, FL_G4, FL_GF0, FL_GF1, FL_GF2, FL_GF3, FL_GF4, FL_GOUT, FL_COUNT
};
constexpr FLayer kFLayers[FL_COUNT] = {
    {PI_C_N1, HC, NX, 8, 4, FG_N1}, {PI_C_N2, C, HC, 2, 8, FG_H128},
    {PI_C_L + 0, HC, EC, 8, 4, FG_CEMB}, {PI_C_L + 2, HC, HC, 8, 8, FG_H128}, {PI_C_L + 4, HC, HC, 8, 8, FG_H128},
    {PI_C_L + 6, HC, EC + HC, 8, 12, FG_CSKIP}, {PI_C_L + 8, HC, HC, 8, 8, FG_H128},
    {PI_C_FCC + 0, HC, C, 8, 2, FG_H32}, {PI_C_FCC + 2, HC, C, 8, 2, FG_H32}, {PI_C_FCC + 4, HC, C, 8, 2, FG_H32},
    {PI_C_FCC + 6, HC, C, 8, 2, FG_H32}, {PI_C_FCC + 8, HC, C, 8, 2, FG_H32},
    {PI_C_OUT, 3, HC, 1, 8, FG_H128},
    {PI_G_L + 0, HG, EG, 2, 6, FG_GEMB}, {PI_G_L + 2, HG, HG, 2, 2, FG_H32}, {PI_G_L + 4, HG, HG, 2, 2, FG_H32},
    {PI_G_L + 6, HG, EG + HG, 2, 8, FG_GSKIP}, {PI_G_L + 8, HG, HG, 2, 2, FG_H32},
    {PI_G_FCC + 0, HG, C, 2, 2, FG_H32}, {PI_G_FCC + 2, HG, C, 2, 2, FG_H32}, {PI_G_FCC + 4, HG, C, 2, 2, FG_H32},
    {PI_G_FCC + 6, HG, C, 2, 2, FG_H32}, {PI_G_FCC + 8, HG, C, 2, 2, FG_H32},
    {PI_G_OUT, 1, HG, 1, 2, FG_H32},
};
constexpr int FRAG = 256;   // floats per fragment (64 lanes x 4)
// first fragment of layer i (fragment (nt, q) of a layer sits at  first + nt * ngroups + q)
constexpr int ffirst(int i) {
  int o = 0;
  for (int j = 0; j < i; ++j) o += kFLayers[j].ntiles * kFLayers[j].ngroups;
  return o;
}
constexpr int kFFrags = ffirst(FL_COUNT);
constexpr int kFColorFrags = ffirst(FL_G0);
// biases follow the fragments, every layer padded to 16 * ntiles floats (16-byte aligned float4 reads of 4 g .. 4 g + 3)
constexpr int fbias(int i) {
  int o = kFFrags * FRAG;
  for (int j = 0; j < i; ++j) o += kFLayers[j].ntiles * 16;
  return o;
}
constexpr int kFFloats = fbias(FL_COUNT);

// ---------------------------------------------------------------- backward fragments (A = W^T: dX^T = W^T . dZ^T)
// Lane l supplies W[out = 16 q + 4 g + r][in = inmap(tile, l & 15)], r = 0..3: k-steps walk the OUTPUT channels of the
// layer in accumulator order (q, r), the tile index walks its INPUT channels.  `in0` lists, per tile, the first input
// channel (inputs of a tile are in0 .. in0 + 15, valid below `lim`).
struct BLayer { int pi, N, K, ntiles, ngroups, t0; };
struct BTile { int in0, lim; };
constexpr BTile kBTiles[] = {
    // 0: 128 inputs in order
    {0, 128}, {16, 128}, {32, 128}, {48, 128}, {64, 128}, {80, 128}, {96, 128}, {112, 128},
    // 8: F_theta linear1 inputs [52]: features first (x index 20..51), then the 20 rel-pos channels
    {ER, NX}, {ER + 16, NX}, {0, ER}, {16, ER},
    // 12: colour skip layer [emb 40 | h 128]: hidden part first, embedding part (pose gradient only) after
    {EC, EC + HC}, {EC + 16, EC + HC}, {EC + 32, EC + HC}, {EC + 48, EC + HC}, {EC + 64, EC + HC}, {EC + 80, EC + HC},
    {EC + 96, EC + HC}, {EC + 112, EC + HC}, {0, EC}, {16, EC}, {32, EC},
    // 23: colour embedding layer (40 inputs)
    {0, EC}, {16, EC}, {32, EC},
    // 26: 32 inputs in order
    {0, 32}, {16, 32},
    // 28: geometry skip layer [emb 93 | h 32]: hidden part first
    {EG, EG + HG}, {EG + 16, EG + HG}, {0, EG}, {16, EG}, {32, EG}, {48, EG}, {64, EG}, {80, EG},
    // 36: geometry embedding layer (93 inputs)
    {0, EG}, {16, EG}, {32, EG}, {48, EG}, {64, EG}, {80, EG},
};
constexpr int BT_H128 = 0, BT_N1 = 8, BT_CSKIP = 12, BT_CEMB = 23, BT_H32 = 26, BT_GSKIP = 28, BT_GEMB = 36;
enum BLayerId {
  BL_N2 = 0, BL_N1, BL_C1, BL_C2, BL_C3, BL_C4, BL_C0, BL_CF0, BL_CF1, BL_CF2, BL_CF3, BL_CF4, BL_COUT,
  BL_G1, BL_G2, BL_G3, BL_G4, BL_G0, BL_GF0, BL_GF1, BL_GF2, BL_GF3, BL_GF4, BL_COUNT
};
constexpr BLayer kBLayers[BL_COUNT] = {
    {PI_C_N2, C, HC, 8, 2, BT_H128}, {PI_C_N1, HC, NX, 4, 8, BT_N1},
    {PI_C_L + 2, HC, HC, 8, 8, BT_H128}, {PI_C_L + 4, HC, HC, 8, 8, BT_H128}, {PI_C_L + 6, HC, EC + HC, 11, 8, BT_CSKIP},
    {PI_C_L + 8, HC, HC, 8, 8, BT_H128}, {PI_C_L + 0, HC, EC, 3, 8, BT_CEMB},
    {PI_C_FCC + 0, HC, C, 2, 8, BT_H32}, {PI_C_FCC + 2, HC, C, 2, 8, BT_H32}, {PI_C_FCC + 4, HC, C, 2, 8, BT_H32},
    {PI_C_FCC + 6, HC, C, 2, 8, BT_H32}, {PI_C_FCC + 8, HC, C, 2, 8, BT_H32},
    {PI_C_OUT, 3, HC, 8, 1, BT_H128},
    {PI_G_L + 2, HG, HG, 2, 2, BT_H32}, {PI_G_L + 4, HG, HG, 2, 2, BT_H32}, {PI_G_L + 6, HG, EG + HG, 8, 2, BT_GSKIP},
    {PI_G_L + 8, HG, HG, 2, 2, BT_H32}, {PI_G_L + 0, HG, EG, 6, 2, BT_GEMB},
    {PI_G_FCC + 0, HG, C, 2, 2, BT_H32}, {PI_G_FCC + 2, HG, C, 2, 2, BT_H32}, {PI_G_FCC + 4, HG, C, 2, 2, BT_H32},
    {PI_G_FCC + 6, HG, C, 2, 2, BT_H32}, {PI_G_FCC + 8, HG, C, 2, 2, BT_H32},
};
constexpr int bfirst(int i) {
  int o = 0;
  for (int j = 0; j < i; ++j) o += kBLayers[j].ntiles * kBLayers[j].ngroups;
  return o;
}
constexpr int kBFrags = bfirst(BL_COUNT);
constexpr int kBFloats = kBFrags * FRAG;

}  // namespace psl
