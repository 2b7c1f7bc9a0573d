// extern "C" surface of libssp_hip.so (include/ssp_hip.h): argument marshalling, thread-local error text and the
// HIP-event launch timer used by bench.py.  No kernels here.
#include <stdarg.h>
#include <mutex>
#include <vector>

#include "../../include/ssp_hip.h"
#include "conv_wino.h"

// ---- kernels' host launchers (defined next to the kernels) ----
int ssp_conv_tile_m(int M, int Cin, int Cout, int R, int plan);
int64_t ssp_conv_ws_floats(int M, int Cin, int Cout, int R, int plan);
int ssp_conv_igemm_launch(const float* in, const float* wt, float* out, const float* bias, float* stats, int B, int H,
                          int W, int Cin, int Cout, int ldin, int ldout, int R, int accumulate, float* ws,
                          int64_t ws_floats, int plan, int prof_kind, hipStream_t stream,
                          const float* escale = nullptr, float act_slope = 1.f, const SspBnBwdFuse* bnb = nullptr);
int64_t ssp_wino_ws_floats(int B, int H, int W, int Cin, int Cout, int tile);
int ssp_conv_wgrad_launch(const float* dy, const float* x, float* dw, int B, int H, int W, int Cin, int Cout, int lddy,
                          int ldx, int R, hipStream_t stream);
int64_t ssp_conv_wgrad_wino_ws_floats(int B, int H, int W, int Cin, int Cout, int tile);
int ssp_conv_wgrad_wino_launch(const float* dy, const float* x, float* dw, int B, int H, int W, int Cin, int Cout, int lddy,
                               int ldx, int tile, float* ws, int64_t ws_floats, hipStream_t stream);
int ssp_bn_fwd_finalize_launch(const float* stats, int ntile, int BM, int M, int C, const float* gamma,
                               const float* beta, float* rmean, float* rvar, float momentum, float eps, float* mean,
                               float* invstd, float* scale, float* shift, hipStream_t stream);
int ssp_bn_eval_prepare_launch(int C, const float* gamma, const float* beta, const float* rmean, const float* rvar,
                               float eps, float* mean, float* invstd, float* scale, float* shift, hipStream_t stream);
int ssp_bn_act_fwd_launch(const float* x, int ldx, float* out, int ldo, const float* scale, const float* shift, int C,
                          int B, int H, int W, int pool, float slope, hipStream_t stream);
int ssp_bn_bwd_blocks_impl(void);
int ssp_bn_act_bwd_launch(const float* x, int ldx, const float* g, int ldg, float* dx, int lddx, const float* scale,
                          const float* shift, const float* mean, const float* invstd, int C, int B, int H, int W,
                          int pool, float slope, int training, float* partial, float* dgamma, float* dbeta, float* c1,
                          float* c2, hipStream_t stream);
int ssp_bn_act_bwd_partials_launch(const float* x, int ldx, const float* g, int ldg, float* dx, int lddx,
                                   const float* scale, const float* shift, const float* mean, const float* invstd,
                                   int C, int B, int H, int W, float slope, int training, float* partial,
                                   int npartial, int zero_after, float* dgamma, float* dbeta, float* c1, float* c2,
                                   hipStream_t stream);
int ssp_bn_bwd_finalize_launch(float* partial, int npartial, int C, int64_t npix, int training, int zero_after,
                               float* dgamma, float* dbeta, float* c1, float* c2, hipStream_t stream);
int ssp_first_tile_pixels_impl(void);
int ssp_first_groups_impl(int B, int H, int W);
int ssp_first_fwd_stats_launch(const float* x, const float* wt, float* stats, int B, int H, int W, hipStream_t stream);
int ssp_first_fwd_apply_launch(const float* x, const float* wt, const float* scale, const float* shift, float slope,
                               float* out, int ldo, int B, int H, int W, hipStream_t stream);
int ssp_first_conv_raw_launch(const float* x, const float* wt, float* raw, int ldraw, int B, int H, int W,
                              hipStream_t stream);
int ssp_first_bwd_reduce_launch(const float* x, const float* wt, const float* g, int ldg, const float* scale,
                                const float* shift, const float* mean, const float* invstd, float slope, float* partial,
                                int B, int H, int W, hipStream_t stream);
int ssp_first_bwd_wgrad_launch(const float* x, const float* wt, const float* g, int ldg, const float* scale,
                               const float* shift, const float* mean, const float* invstd, const float* c1,
                               const float* c2, float slope, float* dw, float* workspace, int64_t workspace_floats, int B,
                               int H, int W, hipStream_t stream);
int64_t ssp_first_wgrad_workspace_floats_impl(int B, int H, int W);
int ssp_colsum_launch(const float* g, int ldg, int64_t M, int C, float* out, hipStream_t stream);
int ssp_sgd_step_launch(float* p, const float* g, float* m, int64_t n, float lr, float momentum, float dampening,
                        float weight_decay, int nesterov, int first_step, hipStream_t stream);
int ssp_nchw_to_nhwc_launch(const float* src, float* dst, int B, int C, int H, int W, int Cp, int ld, hipStream_t stream);
int ssp_nhwc_to_nchw_launch(const float* src, float* dst, int B, int C, int H, int W, int ld, hipStream_t stream);
int ssp_pose_errors_launch(const double* verts, int N, const double* Rt_gt, const double* Rt_pr, const double* K,
                           int k_per_pose, int n, double* out, hipStream_t stream);
int ssp_pts_diameter_launch(const double* pts, int N, double* out, double* scratch, hipStream_t stream);
int ssp_resample_u8_launch(const SspResampleDesc* descs, int count, int pass, int epilogue, int max_dst_pixels, hipStream_t stream);
int ssp_distort_u8_launch(const unsigned char* rgb, unsigned char* out, int64_t npix, const unsigned char* lut, int mode, hipStream_t stream);
int ssp_u8hwc_to_nhwc_launch(const unsigned char* src, float* dst, int B, int H, int W, int C, int Cp, int ld,
                             hipStream_t stream);
int ssp_repack_fwd_launch(const float* w, float* out, int Cout, int Cin, int Cinp, int R, hipStream_t stream);
int ssp_unpack_grad_launch(const float* dwp, float* grad, int Cout, int Cin, int Cinp, int R, hipStream_t stream);
int ssp_repack_dgrad_launch(const float* w, float* out, int Cout, int Cin, int Coutp, int R, hipStream_t stream);
int ssp_repack_dgrad_packed_launch(const float* wp, float* out, int Cout, int Cin, int Coutp, int R, hipStream_t stream);
int ssp_reorg_launch(const float* src, int lds_, float* dst, int ldd, int C, int B, int H, int W, int backward,
                     int accumulate, hipStream_t stream);
int ssp_copy_channels_launch(const float* src, int lds_, float* dst, int ldd, int C, int64_t M, int accumulate,
                             hipStream_t stream);
int ssp_maxpool_fwd_launch(const float* x, int ldx, float* out, int ldo, int C, int B, int H, int W, hipStream_t stream);
int ssp_maxpool_bwd_launch(const float* x, int ldx, const float* g, int ldg, float* dx, int lddx, int C, int B, int H,
                           int W, int accumulate, hipStream_t stream);
int ssp_region_loss_launch(const float* out, const void* target, int target_is_f64, float* grad, float* partials,
                           float* stats, int nB, int nA, int nC, int nH, int nW, int num_keypoints,
                           float noobject_scale, float object_scale, float coord_scale, float class_scale, float thresh,
                           int conf_on, int multi, const float* anchors, int anchor_step, hipStream_t stream);
int ssp_region_decode_argmax_launch(const float* out, float* boxes, int nB, int nA, int nC, int nH, int nW,
                                    int num_keypoints, int only_objectness, hipStream_t stream);
int ssp_region_decode_all_launch(const float* out, float* rows, int nB, int nA, int nC, int nH, int nW,
                                 int num_keypoints, hipStream_t stream);
int ssp_pnp_batched_launch(const double* pts3d, const double* pts2d, const double* K, double* Rt, int n, int N,
                           int max_iter, hipStream_t stream);

// ---- error text ----
static thread_local char g_err[512] = "";
void ssp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- tuning knobs ----
static int g_options[SSP_OPT_COUNT] = {1, 0, 0, 0, 0, 0, 1};
static const char* g_option_names[SSP_OPT_COUNT] = {"igemm_xcd", "igemm_variant", "wgrad_variant", "igemm_plan", "wgrad_split", "wino_variant", "acc_chunk"};
int ssp_option(int which) { return g_options[which]; }

// ---- per-device kernel configuration cache ----
static std::mutex g_kernel_mu;
int ssp_kernel_prepare(const void* kern, int lds_bytes, int threads, SspKernelCache* cache, int* slots, const char* name) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SSP_MAX_DEVICES) {
    ssp_set_error("%s: no current HIP device (or device index >= %d)", name, SSP_MAX_DEVICES);
    return SSP_ERR_HIP;
  }
  std::lock_guard<std::mutex> lk(g_kernel_mu);
  if (lds_bytes > cache->configured[dev]) {
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) {
      ssp_set_error("%s: cannot reserve %d bytes of LDS", name, lds_bytes);
      return SSP_ERR_HIP;
    }
    int per_cu = 0, ncu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds_bytes) != hipSuccess || per_cu < 1)
      per_cu = 2;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 1) ncu = 256;
    cache->slots[dev] = per_cu * ncu;
    cache->configured[dev] = lds_bytes;
  }
  if (slots != nullptr) *slots = cache->slots[dev];
  return SSP_OK;
}

// ---- launch timer ----
namespace {
struct ProfRec {
  int kind;
  hipEvent_t start, stop;
  double work;
};
std::mutex g_prof_mu;
unsigned g_prof_mask = 0;   // bit k: launches of family k are bracketed by HIP events
std::vector<ProfRec> g_recs;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_free_events;
}  // namespace

SspProfScope::SspProfScope(int kind, hipStream_t s, double work) : slot(-1), stream(s) {
  if (!((g_prof_mask >> kind) & 1u)) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r;
  r.kind = kind;
  r.work = work;
  if (!g_free_events.empty()) {
    r.start = g_free_events.back().first;
    r.stop = g_free_events.back().second;
    g_free_events.pop_back();
  } else {
    if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
  }
  (void)hipEventRecord(r.start, s);
  g_recs.push_back(r);
  slot = (int)g_recs.size() - 1;
}
SspProfScope::~SspProfScope() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  (void)hipEventRecord(g_recs[slot].stop, stream);
}

extern "C" {

const char* ssp_last_error(void) { return g_err; }
int ssp_abi_version(void) { return 5; }
int ssp_set_option(const char* name, int value) {
  for (int i = 0; i < SSP_OPT_COUNT; ++i)
    if (name != nullptr && strcmp(name, g_option_names[i]) == 0) {
      g_options[i] = value;
      return SSP_OK;
    }
  ssp_set_error("set_option: unknown option '%s'", name ? name : "(null)");
  return SSP_ERR_ARG;
}

int ssp_conv_fwd(const float* in, const float* wt, float* out, const float* bias, float* stats, int B, int H, int W,
                 int Cin, int Cout, int ldin, int ldout, int R, int accumulate, int plan, float* workspace,
                 int64_t workspace_floats, void* stream) {
  return ssp_conv_igemm_launch(in, wt, out, bias, stats, B, H, W, Cin, Cout, ldin, ldout, R, accumulate, workspace,
                               workspace_floats, plan, SSP_PROF_CONV_FWD, (hipStream_t)stream);
}
int ssp_conv_fwd_affine(const float* in, const float* wt, float* out, const float* scale, const float* shift,
                        float slope, int B, int H, int W, int Cin, int Cout, int ldin, int ldout, int R, int plan,
                        float* workspace, int64_t workspace_floats, void* stream) {
  return ssp_conv_igemm_launch(in, wt, out, shift, nullptr, B, H, W, Cin, Cout, ldin, ldout, R, 0, workspace,
                               workspace_floats, plan, SSP_PROF_CONV_FWD, (hipStream_t)stream, scale, slope);
}
int ssp_conv_stats_tile_m(int B, int H, int W, int Cin, int Cout, int R, int plan) {
  return ssp_conv_tile_m(B * H * W, Cin, Cout, R, plan);
}
int ssp_conv_plan_wino_tile(int plan) { return ssp_wino_plan_tile(plan); }
int64_t ssp_conv_wino_tiles(int B, int H, int W, int tile) {
  if (tile != 2 && tile != 4) return 0;
  return ssp_wino_tiles(B, H, W, tile);
}
int ssp_conv_stats_tiles(int B, int H, int W, int Cin, int Cout, int R, int plan) {
  if (ssp_wino_plan_fused(plan)) return ssp_wino_fused_stat_groups(B, H, W, Cout);
  if (const int tile = ssp_wino_plan_tile(plan)) return (int)ssp_wino_stat_groups(B, H, W, tile);
  return ssp_cdiv((int64_t)B * H * W, ssp_conv_tile_m(B * H * W, Cin, Cout, R, plan));
}
int64_t ssp_conv_stats_floats(int B, int H, int W, int Cin, int Cout, int R, int plan) {
  const int64_t nt = ssp_conv_stats_tiles(B, H, W, Cin, Cout, R, plan);
  return nt * Cout * 2 + (ssp_wino_plan_tile(plan) ? nt : 0);
}
int64_t ssp_conv_workspace_floats(int B, int H, int W, int Cin, int Cout, int R, int plan) {
  if (ssp_wino_plan_fused(plan)) return 0;      // V and M stay on the chip
  if (const int tile = ssp_wino_plan_tile(plan)) return ssp_wino_ws_floats(B, H, W, Cin, Cout, tile);      // V + M planes
  return ssp_conv_ws_floats(B * H * W, Cin, Cout, R, plan);
}
int ssp_wino_filter_transform_t(const float* w9, float* U, int rows, int K, int tile, void* stream) {
  return ssp_wino_filter_launch(w9, U, rows, K, tile, (hipStream_t)stream);
}
int ssp_wino_filter_transform(const float* w9, float* U, int rows, int K, void* stream) {
  return ssp_wino_filter_launch(w9, U, rows, K, 2, (hipStream_t)stream);
}

int ssp_conv_dgrad(const float* dy, const float* wt, float* dx, int B, int H, int W, int Cout_dy, int Cin_dx, int lddy,
                   int lddx, int R, int accumulate, int plan, float* workspace, int64_t workspace_floats,
                   void* stream) {
  return ssp_conv_igemm_launch(dy, wt, dx, nullptr, nullptr, B, H, W, Cout_dy, Cin_dx, lddy, lddx, R, accumulate,
                               workspace, workspace_floats, plan, SSP_PROF_CONV_DGRAD, (hipStream_t)stream);
}

int ssp_conv_dgrad_bnbwd(const float* dy, const float* wt, float* dx, int B, int H, int W, int Cout_dy, int Cin_dx,
                         int lddy, int lddx, int R, int plan, float* workspace, int64_t workspace_floats,
                         const float* raw, int ldraw, const float* scale, const float* shift, const float* mean,
                         const float* invstd, float slope, float* partial, int partial_rows, void* stream) {
  SspBnBwdFuse f = {raw, ldraw, scale, shift, mean, invstd, slope, partial, partial_rows};
  if (partial == nullptr) {
    ssp_set_error("conv_dgrad_bnbwd: partial must not be NULL (use ssp_conv_dgrad)");
    return SSP_ERR_ARG;
  }
  return ssp_conv_igemm_launch(dy, wt, dx, nullptr, nullptr, B, H, W, Cout_dy, Cin_dx, lddy, lddx, R, 0, workspace,
                               workspace_floats, plan, SSP_PROF_CONV_DGRAD, (hipStream_t)stream, nullptr, 1.f, &f);
}
int ssp_bn_act_bwd_partials(const float* x, int ldx, const float* g, int ldg, float* dx, int lddx, const float* scale,
                            const float* shift, const float* mean, const float* invstd, int C, int B, int H, int W,
                            float slope, int training, float* partial, int npartial, int zero_after, float* dgamma,
                            float* dbeta, float* c1, float* c2, void* stream) {
  return ssp_bn_act_bwd_partials_launch(x, ldx, g, ldg, dx, lddx, scale, shift, mean, invstd, C, B, H, W, slope, training,
                                        partial, npartial, zero_after, dgamma, dbeta, c1, c2, (hipStream_t)stream);
}

int ssp_bn_bwd_finalize(float* partial, int npartial, int C, int64_t npix, int training, int zero_after, float* dgamma,
                        float* dbeta, float* c1, float* c2, void* stream) {
  return ssp_bn_bwd_finalize_launch(partial, npartial, C, npix, training, zero_after, dgamma, dbeta, c1, c2,
                                    (hipStream_t)stream);
}
int ssp_first_tile_pixels(void) { return ssp_first_tile_pixels_impl(); }
int ssp_first_groups(int B, int H, int W) { return ssp_first_groups_impl(B, H, W); }
int ssp_first_fwd_stats(const float* x, const float* wt, float* stats, int B, int H, int W, void* stream) {
  return ssp_first_fwd_stats_launch(x, wt, stats, B, H, W, (hipStream_t)stream);
}
int ssp_first_fwd_apply(const float* x, const float* wt, const float* scale, const float* shift, float slope, float* out,
                        int ldo, int B, int H, int W, void* stream) {
  return ssp_first_fwd_apply_launch(x, wt, scale, shift, slope, out, ldo, B, H, W, (hipStream_t)stream);
}
int ssp_first_conv_raw(const float* x, const float* wt, float* raw, int ldraw, int B, int H, int W, void* stream) {
  return ssp_first_conv_raw_launch(x, wt, raw, ldraw, B, H, W, (hipStream_t)stream);
}
int ssp_first_bwd_reduce(const float* x, const float* wt, const float* g, int ldg, const float* scale, const float* shift,
                         const float* mean, const float* invstd, float slope, float* partial, int B, int H, int W,
                         void* stream) {
  return ssp_first_bwd_reduce_launch(x, wt, g, ldg, scale, shift, mean, invstd, slope, partial, B, H, W,
                                     (hipStream_t)stream);
}
int ssp_first_bwd_wgrad(const float* x, const float* wt, const float* g, int ldg, const float* scale, const float* shift,
                        const float* mean, const float* invstd, const float* c1, const float* c2, float slope, float* dw,
                        float* workspace, int64_t workspace_floats, int B, int H, int W, void* stream) {
  return ssp_first_bwd_wgrad_launch(x, wt, g, ldg, scale, shift, mean, invstd, c1, c2, slope, dw, workspace,
                                    workspace_floats, B, H, W, (hipStream_t)stream);
}
int64_t ssp_first_wgrad_workspace_floats(int B, int H, int W) { return ssp_first_wgrad_workspace_floats_impl(B, H, W); }

int ssp_conv_wgrad(const float* dy, const float* x, float* dw, int B, int H, int W, int Cin, int Cout, int lddy,
                   int ldx, int R, void* stream) {
  return ssp_conv_wgrad_launch(dy, x, dw, B, H, W, Cin, Cout, lddy, ldx, R, (hipStream_t)stream);
}

int64_t ssp_conv_wgrad_wino_workspace_floats_t(int B, int H, int W, int Cin, int Cout, int tile) {
  return ssp_conv_wgrad_wino_ws_floats(B, H, W, Cin, Cout, tile);
}
int ssp_conv_wgrad_wino_t(const float* dy, const float* x, float* dw, int B, int H, int W, int Cin, int Cout, int lddy,
                          int ldx, int tile, float* workspace, int64_t workspace_floats, void* stream) {
  return ssp_conv_wgrad_wino_launch(dy, x, dw, B, H, W, Cin, Cout, lddy, ldx, tile, workspace, workspace_floats, (hipStream_t)stream);
}
int ssp_wino_input_transform_t(const float* x, int ldx, float* V, int B, int H, int W, int C, int tile, void* stream) {
  return ssp_wino_input_launch(x, ldx, V, B, H, W, C, tile, SSP_PROF_WINO_WGRAD, (hipStream_t)stream);
}
int64_t ssp_conv_wgrad_wino_workspace_floats(int B, int H, int W, int Cin, int Cout) {
  return ssp_conv_wgrad_wino_ws_floats(B, H, W, Cin, Cout, 2);
}
int ssp_conv_wgrad_wino(const float* dy, const float* x, float* dw, int B, int H, int W, int Cin, int Cout, int lddy,
                        int ldx, float* workspace, int64_t workspace_floats, void* stream) {
  return ssp_conv_wgrad_wino_launch(dy, x, dw, B, H, W, Cin, Cout, lddy, ldx, 2, workspace, workspace_floats, (hipStream_t)stream);
}

int ssp_bn_fwd_finalize(const float* stats, int ntile, int tile_m, int M, int C, const float* gamma,
                        const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                        float* mean, float* invstd, float* scale, float* shift, void* stream) {
  return ssp_bn_fwd_finalize_launch(stats, ntile, tile_m, M, C, gamma, beta, running_mean, running_var, momentum, eps,
                                    mean, invstd, scale, shift, (hipStream_t)stream);
}
int ssp_bn_eval_prepare(int C, const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float eps, float* mean, float* invstd, float* scale, float* shift,
                        void* stream) {
  return ssp_bn_eval_prepare_launch(C, gamma, beta, running_mean, running_var, eps, mean, invstd, scale, shift,
                                    (hipStream_t)stream);
}
int ssp_bn_act_fwd(const float* x, int ldx, float* out, int ldo, const float* scale, const float* shift, int C, int B,
                   int H, int W, int pool, float slope, void* stream) {
  return ssp_bn_act_fwd_launch(x, ldx, out, ldo, scale, shift, C, B, H, W, pool, slope, (hipStream_t)stream);
}
int ssp_bn_act_bwd(const float* x, int ldx, const float* g, int ldg, float* dx, int lddx, const float* scale,
                   const float* shift, const float* mean, const float* invstd, int C, int B, int H, int W, int pool,
                   float slope, int training, float* partial, float* dgamma, float* dbeta, float* c1, float* c2,
                   void* stream) {
  return ssp_bn_act_bwd_launch(x, ldx, g, ldg, dx, lddx, scale, shift, mean, invstd, C, B, H, W, pool, slope, training,
                               partial, dgamma, dbeta, c1, c2, (hipStream_t)stream);
}
int ssp_bn_bwd_blocks(void) { return ssp_bn_bwd_blocks_impl(); }
int ssp_colsum(const float* g, int ldg, int64_t M, int C, float* out, void* stream) {
  return ssp_colsum_launch(g, ldg, M, C, out, (hipStream_t)stream);
}
int ssp_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum,
                 float dampening, float weight_decay, int nesterov, int first_step, void* stream) {
  return ssp_sgd_step_launch(param, grad, momentum_buf, n, lr, momentum, dampening, weight_decay, nesterov, first_step,
                             (hipStream_t)stream);
}

int ssp_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, int Cpad, int ld, void* stream) {
  return ssp_nchw_to_nhwc_launch(src, dst, B, C, H, W, Cpad, ld, (hipStream_t)stream);
}
int ssp_u8hwc_to_nhwc(const unsigned char* src, float* dst, int B, int H, int W, int C, int Cpad, int ld, void* stream) {
  return ssp_u8hwc_to_nhwc_launch(src, dst, B, H, W, C, Cpad, ld, (hipStream_t)stream);
}
int ssp_resample_u8(const SspResampleDesc* descs_dev, int count, int pass, int epilogue, int max_dst_pixels, void* stream) {
  return ssp_resample_u8_launch(descs_dev, count, pass, epilogue, max_dst_pixels, (hipStream_t)stream);
}
int ssp_distort_u8(const unsigned char* rgb, unsigned char* out, int64_t npix, const unsigned char* lut, int mode, void* stream) {
  return ssp_distort_u8_launch(rgb, out, npix, lut, mode, (hipStream_t)stream);
}
int ssp_nhwc_to_nchw(const float* src, float* dst, int B, int C, int H, int W, int ld, void* stream) {
  return ssp_nhwc_to_nchw_launch(src, dst, B, C, H, W, ld, (hipStream_t)stream);
}
int ssp_repack_fwd(const float* w, float* out, int Cout, int Cin, int Cinp, int R, void* stream) {
  return ssp_repack_fwd_launch(w, out, Cout, Cin, Cinp, R, (hipStream_t)stream);
}
int ssp_repack_dgrad_packed(const float* wp, float* out, int Cout, int Cin, int Coutp, int R, void* stream) {
  return ssp_repack_dgrad_packed_launch(wp, out, Cout, Cin, Coutp, R, (hipStream_t)stream);
}
int ssp_repack_dgrad(const float* w, float* out, int Cout, int Cin, int Coutp, int R, void* stream) {
  return ssp_repack_dgrad_launch(w, out, Cout, Cin, Coutp, R, (hipStream_t)stream);
}
int ssp_unpack_grad(const float* dwp, float* grad, int Cout, int Cin, int Cinp, int R, void* stream) {
  return ssp_unpack_grad_launch(dwp, grad, Cout, Cin, Cinp, R, (hipStream_t)stream);
}
int ssp_reorg(const float* src, int lds, float* dst, int ldd, int C, int B, int H, int W, int backward, int accumulate,
              void* stream) {
  return ssp_reorg_launch(src, lds, dst, ldd, C, B, H, W, backward, accumulate, (hipStream_t)stream);
}
int ssp_copy_channels(const float* src, int lds, float* dst, int ldd, int C, int64_t M, int accumulate, void* stream) {
  return ssp_copy_channels_launch(src, lds, dst, ldd, C, M, accumulate, (hipStream_t)stream);
}
int ssp_maxpool_fwd(const float* x, int ldx, float* out, int ldo, int C, int B, int H, int W, void* stream) {
  return ssp_maxpool_fwd_launch(x, ldx, out, ldo, C, B, H, W, (hipStream_t)stream);
}
int ssp_maxpool_bwd(const float* x, int ldx, const float* g, int ldg, float* dx, int lddx, int C, int B, int H, int W,
                    int accumulate, void* stream) {
  return ssp_maxpool_bwd_launch(x, ldx, g, ldg, dx, lddx, C, B, H, W, accumulate, (hipStream_t)stream);
}

int ssp_region_loss(const float* out, const void* target, int target_is_f64, float* grad, float* partials,
                    float* stats, int nB, int nA, int nC, int nH, int nW, int num_keypoints, float noobject_scale,
                    float object_scale, float coord_scale, float class_scale, float thresh, int conf_on, int multi,
                    const float* anchors, int anchor_step, void* stream) {
  return ssp_region_loss_launch(out, target, target_is_f64, grad, partials, stats, nB, nA, nC, nH, nW, num_keypoints,
                                noobject_scale, object_scale, coord_scale, class_scale, thresh, conf_on, multi, anchors,
                                anchor_step, (hipStream_t)stream);
}
int ssp_region_decode_argmax(const float* out, float* boxes, int nB, int nA, int nC, int nH, int nW,
                             int num_keypoints, int only_objectness, void* stream) {
  return ssp_region_decode_argmax_launch(out, boxes, nB, nA, nC, nH, nW, num_keypoints, only_objectness,
                                         (hipStream_t)stream);
}

int ssp_region_decode_all(const float* out, float* rows, int nB, int nA, int nC, int nH, int nW, int num_keypoints,
                          void* stream) {
  return ssp_region_decode_all_launch(out, rows, nB, nA, nC, nH, nW, num_keypoints, (hipStream_t)stream);
}

int ssp_pnp_batched(const double* pts3d, const double* pts2d, const double* K, double* Rt, int n, int N, int max_iter,
                    void* stream) {
  return ssp_pnp_batched_launch(pts3d, pts2d, K, Rt, n, N, max_iter, (hipStream_t)stream);
}

int ssp_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_mask = (unsigned)on;     // bit k = family k (SSP_PROF_*); -1 = every family, 0 = off
  return SSP_OK;
}
int ssp_pose_errors(const double* vertices, int N, const double* Rt_gt, const double* Rt_pr, const double* K,
                    int k_per_pose, int n, double* out, void* stream) {
  return ssp_pose_errors_launch(vertices, N, Rt_gt, Rt_pr, K, k_per_pose, n, out, (hipStream_t)stream);
}
int ssp_pts_diameter(const double* pts, int N, double* out, double* scratch, void* stream) {
  return ssp_pts_diameter_launch(pts, N, out, scratch, (hipStream_t)stream);
}
int ssp_prof_nkinds(void) { return SSP_PROF_NKINDS; }
int ssp_prof_collect(double* ms, double* work, int64_t* count) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int k = 0; k < SSP_PROF_NKINDS; ++k) { ms[k] = 0.0; work[k] = 0.0; count[k] = 0; }
  for (auto& r : g_recs) {
    float t = 0.f;
    if (hipEventSynchronize(r.stop) != hipSuccess || hipEventElapsedTime(&t, r.start, r.stop) != hipSuccess) {
      ssp_set_error("prof_collect: event query failed");
      g_recs.clear();
      return SSP_ERR_HIP;
    }
    ms[r.kind] += (double)t;
    work[r.kind] += r.work;
    count[r.kind] += 1;
    g_free_events.emplace_back(r.start, r.stop);
  }
  g_recs.clear();
  return SSP_OK;
}

}  // extern "C"
