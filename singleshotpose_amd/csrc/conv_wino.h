// Host-side interface of the Winograd transform kernels (conv_wino.hip), shared by the conv / filter-gradient launchers.
#pragma once
#include "ssp_common.h"

// Plan codes of the conv entry points: 9000000 + c = F(2x2, 3x3), 8000000 + c = F(4x4, 3x3); c = bm * 100 + 10 + slots is
// the tile / ring-depth choice of the batched GEMM launch (conv_igemm_dma.hip).
// 7000000 + v = F(2x2, 3x3) with the transform domain kept on the chip (conv_wino_fused.hip): ONE persistent launch, no
// workspace; same filter operand as a 9xxxxxx plan (tile 2), statistics in the counted format with one group per patch block.
#define SSP_WINO2_PLAN 9000000
#define SSP_WINO4_PLAN 8000000
#define SSP_WINOF_PLAN 7000000
static inline int ssp_wino_plan_fused(int plan) { return plan >= SSP_WINOF_PLAN && plan < SSP_WINOF_PLAN + 1000000; }
static inline int ssp_wino_plan_tile(int plan) {      // 0: not a Winograd plan
  if (plan >= SSP_WINO2_PLAN && plan < SSP_WINO2_PLAN + 1000000) return 2;
  if (plan >= SSP_WINO4_PLAN && plan < SSP_WINO4_PLAN + 1000000) return 4;
  if (ssp_wino_plan_fused(plan)) return 2;
  return 0;
}

#define SSP_WINO_TG 16      // tiles per finishing workgroup = per BatchNorm-statistics group of a Winograd launch

struct WinoOutArgs {
  const float* Mw;     // [P][T][Cout]: the batched GEMM's output planes
  float* out;          // [B*H*W][ldout]
  const float* bias;
  const float* escale;
  float act_slope;
  float* stats;        // [groups][Cout][2] (mean, M2) | [groups] pixel counts, or nullptr
  int Cout, ldout, accumulate;
  const float* bn_raw; // fused BatchNorm-backward reductions (ConvArgs::bn_*)
  const float* bn_scale;
  const float* bn_shift;
  const float* bn_mean;
  const float* bn_invstd;
  float* bn_partial;
  int bn_nslot, bn_ld;
  float bn_slope;
};

int64_t ssp_wino_tiles(int B, int H, int W, int tile);
int ssp_wino_planes(int tile);
int64_t ssp_wino_stat_groups(int B, int H, int W, int tile);
// prof_kind: SSP_PROF_WINO_FWD / _DGRAD / _WGRAD - the family of the launch the pass belongs to
int ssp_wino_input_launch(const float* in, int ldin, float* V, int B, int H, int W, int C, int tile, int prof_kind, hipStream_t stream);
int ssp_wino_filter_launch(const float* w, float* U, int rows, int K, int tile, hipStream_t stream);
int ssp_wino_outgrad_launch(const float* dy, int lddy, float* dM, int B, int H, int W, int C, int tile, hipStream_t stream);
int ssp_wino_wgrad_finish_launch(const float* dU, float* dw, int rows, int K, int tile, hipStream_t stream);
int ssp_wino_output_launch(const WinoOutArgs& a, int B, int H, int W, int tile, int prof_kind, hipStream_t stream);

// on-chip F(2x2) (conv_wino_fused.hip); `a` = the conv launch's ConvArgs (struct ConvArgs, conv_igemm_common.h)
struct ConvArgs;
int ssp_wino_fused_launch(const ConvArgs& a, int B, int H, int W, int prof_kind, hipStream_t stream);
int ssp_wino_fused_stat_groups(int B, int H, int W, int Cout);
bool ssp_wino_fused_fits(int B, int H, int W, int Cin, int Cout, int R);

// filter gradient with both Winograd transforms on the chip (conv_wino_wgrad_fused.hip): the `tile` value of
// ssp_conv_wgrad_wino_t / ssp_conv_wgrad_wino_workspace_floats_t that selects it
#define SSP_WINO_WGRAD_FUSED 12
int ssp_wino_wgrad_fused_launch(const float* dy, const float* x, float* dw, int B, int H, int W, int Cin, int Cout, int lddy,
                                int ldx, float* ws, int64_t ws_floats, hipStream_t stream);
int64_t ssp_wino_wgrad_fused_ws_floats(int Cin, int Cout);
bool ssp_wino_wgrad_fused_fits(int B, int H, int W, int Cin, int Cout);
