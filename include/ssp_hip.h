/* ssp_hip.h - C ABI of libssp_hip.so, the MI355X (gfx950) kernels behind the singleshotpose hot path.
 *
 * The reference (microsoft/singleshotpose) is pure Python on PyTorch; it has no native interface.  The device
 * work it performs happens inside ATen/cuDNN ops called from darknet.py / region_loss.py / utils.py.  Each entry
 * point below names the reference call site (file:line under /root/reference) whose device work it replaces.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (never throws); ssp_last_error() gives the thread-local text.
 *   - all pointers are DEVICE pointers owned by the caller (PyTorch); nothing is allocated or freed here, workspaces
 *     are passed in.  Launches are asynchronous on `stream` (a hipStream_t passed as void*); no hidden sync.
 *   - activations are fp32 NHWC: pixel p = (b*H + y)*W + x, channel c at base[p*ld + c]; `ld` (floats per pixel)
 *     may exceed the channel count so a call can read/write a channel slice of a wider buffer (route/concat).
 *   - packed filters are fp32 [rows][R*R][cols] with cols contiguous (see ssp_repack_*).
 */
#ifndef SSP_HIP_H
#define SSP_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* ssp_last_error(void);
int ssp_abi_version(void);
/* EXPERIMENT knobs (kernel variant / tile-order A-B switches used by bench.py --opt and tools/): process-wide, not
 * synchronised, never written by the product path (engine.Plan passes its per-launch choices as explicit `plan`
 * arguments below).  Unknown names are an error.  Never changes results beyond fp32 summation order. */
int ssp_set_option(const char* name, int value);

/* ---- convolution (stride 1, "same" padding, R = 1 or 3): nn.Conv2d at darknet.py:156,160 ------------------- */

/* out[p][co] (+)= sum_{tap,ci} in[p + tap][ci] * wt[co][tap][ci]  (+ bias[co]);  wt from ssp_repack_fwd.
 * stats (nullable): [ceil(B*H*W / ssp_conv_stats_tile_m(...))][Cout][2] per-tile (mean, M2) of the raw output,
 * input of ssp_bn_fwd_finalize (training-mode BatchNorm statistics, darknet.py:157).
 * workspace: ssp_conv_workspace_floats(...) floats (0 for most shapes; the 13x13 layers split their K loop over
 * several workgroups per tile and sum the partial tiles from it; so do thin layers - Cout <= 64 - whose single tile
 * column cannot fill the chip, e.g. the 1024 -> 20 head conv: those fall back to the un-split walk when no workspace
 * is passed).  The shape arguments of the two queries are those
 * of the launch (for ssp_conv_dgrad: Cin = channels of dy, Cout = channels of dx).
 * plan: tile / split choice of THIS launch, 0 = the library's shape heuristic, else
 *   tail*100000 + tile_rows*100 + ksplit*10 + ring_slots   (tile_rows 64|128, ksplit 1..9, ring_slots 3|4, or 8 =
 *   the latency form for grids of about one workgroup per CU: seven K chunks in flight, one workgroup per CU;
 *   tail 0 or 2..9 = hybrid launch: whole resident waves un-split, the last partial wave's tiles split `tail` ways);
 * a code that does not fit the shape falls back to the heuristic.  It is an argument, not state: two threads (or two
 * models) may run different plans concurrently.  The same code must be passed to the two queries.
 *   9000000 + tile_rows*100 + 10 + ring_slots (3|4|8) = Winograd F(2x2, 3x3) evaluation of a 3x3 layer (Cin % 16 == 0, Cout >= 64
 *   and % 4 == 0; conv_wino.hip): same result to ~1e-6 of its range with 16/36 of the multiplies;
 *   8000000 + the same = Winograd F(4x4, 3x3) (points 0, 1, -1, 1/2, -2): 36/144 of the multiplies, result within ~5e-6 of its
 *   range (about twice the direct fp32 kernel's own rounding error).  `wt` must then be the TRANSFORMED filter from
 *   ssp_wino_filter_transform_t with the plan's tile size (of the ssp_repack_fwd / ssp_repack_dgrad layout) and the
 *   workspace (ssp_conv_workspace_floats: (tile+2)^2 * tiles * (Cin + Cout) floats, tiles = ssp_conv_wino_tiles(B, H, W, tile))
 *   is mandatory; a Winograd code on a shape it does not fit is an error, not a fallback (the filter operand differs).
 *   7000001 = Winograd F(2x2, 3x3) with the transform domain kept ON THE CHIP (conv_wino_fused.hip; ABI 5): ONE persistent
 *   launch, no workspace (ssp_conv_workspace_floats = 0), `wt` = the same transformed filter as a 9xxxxxx plan
 *   (ssp_wino_filter_transform_t, tile 2; ssp_conv_plan_wino_tile returns 2); Cin % 32 == 0 and Cout % 32 == 0 (no
 *   lower bound of 64), 16-byte aligned in / out / bias / scale, ldin % 4 == 0, ldout % 4 == 0, every operand below
 *   2 GiB; statistics in the counted format with one group per block of four 8 x 16-pixel patches
 *   (ssp_conv_stats_tiles); accumulate, bias / affine and the fused BatchNorm-backward sums as for the other plans
 *   (bias / affine may be combined with accumulate, the BatchNorm-backward sums with neither).
 *   Valid for ssp_conv_fwd, ssp_conv_fwd_affine, ssp_conv_dgrad and ssp_conv_dgrad_bnbwd.
 * stats of a Winograd plan are in the COUNTED format: ssp_conv_stats_tile_m returns 0, the buffer holds
 *   [ssp_conv_stats_tiles(...)][Cout][2] (mean, M2) pairs followed by [ssp_conv_stats_tiles(...)] pixel counts
 *   (ssp_conv_stats_floats floats in all) and ssp_bn_fwd_finalize takes tile_m = 0. */
int ssp_conv_fwd(const float* in, const float* wt, float* out, const float* bias, float* stats, int B, int H, int W,
                 int Cin, int Cout, int ldin, int ldout, int R, int accumulate, int plan, float* workspace,
                 int64_t workspace_floats, void* stream);
/* eval-mode block in one launch (darknet.py:154-167 with BatchNorm in inference mode):
 * out[p][co] = leaky(scale[co] * conv(in, wt)[p][co] + shift[co], slope); scale / shift from ssp_bn_eval_prepare
 * (either may be NULL = 1 / 0), slope = 1 for a linear block.  Same shapes, workspace and limits as ssp_conv_fwd. */
int ssp_conv_fwd_affine(const float* in, const float* wt, float* out, const float* scale, const float* shift,
                        float slope, int B, int H, int W, int Cin, int Cout, int ldin, int ldout, int R, int plan,
                        float* workspace, int64_t workspace_floats, void* stream);
int ssp_conv_stats_tile_m(int B, int H, int W, int Cin, int Cout, int R, int plan);
/* number of statistics tiles of the launch (= rows of its `stats` / of ssp_conv_dgrad_bnbwd's `partial`), and the size of
 * its `stats` buffer in floats */
int ssp_conv_stats_tiles(int B, int H, int W, int Cin, int Cout, int R, int plan);
int64_t ssp_conv_stats_floats(int B, int H, int W, int Cin, int Cout, int R, int plan);
/* tile size of a plan code: 2 or 4 for a Winograd plan, 0 otherwise */
int ssp_conv_plan_wino_tile(int plan);
/* tiles (rows of each of the (tile+2)^2 batched GEMMs) a Winograd launch of that tile size cuts B maps of H x W into:
 * B * ceil(H/tile) * ceil(W/tile), or - when that needs fewer - ceil(B/4) * ceil((2H+1)/tile) * ceil((2W+1)/tile): four
 * images tiled as one 2 x 2 mosaic with a zero row / column between them (13 x 13 at tile 4: 49 tiles per four images
 * instead of 64).  What the workspace and statistics queries are built on. */
int64_t ssp_conv_wino_tiles(int B, int H, int W, int tile);
/* U[xi][row][k] = (G g G^T)[xi], xi = 0..(tile+2)^2-1, of the 3x3 filters g[tap] = w9[row][tap][k] (rows x 9 x K floats,
 * K % 4 == 0): the filter operand of a Winograd plan of that tile size (2 or 4).  rows = Cout, K = Cin for the forward
 * layout; rows = Cin_dx, K = Cout_dy for the data-gradient layout.  U: (tile+2)^2 * rows * K floats.
 * ssp_wino_filter_transform = tile 2. */
int ssp_wino_filter_transform_t(const float* w9, float* U, int rows, int K, int tile, void* stream);
int ssp_wino_filter_transform(const float* w9, float* U, int rows, int K, void* stream);
int64_t ssp_conv_workspace_floats(int B, int H, int W, int Cin, int Cout, int R, int plan);

/* data gradient (autograd of nn.Conv2d, train.py:103): dx[p][ci] (+)= sum dy[p - tap][co] * w[co][ci][tap];
 * same contraction as ssp_conv_fwd with `wt` from ssp_repack_dgrad; Cout_dy = channels of dy (multiple of 4). */
int ssp_conv_dgrad(const float* dy, const float* wt, float* dx, int B, int H, int W, int Cout_dy, int Cin_dx, int lddy,
                   int lddx, int R, int accumulate, int plan, float* workspace, int64_t workspace_floats,
                   void* stream);

/* ssp_conv_dgrad (accumulate = 0) that ALSO performs the two BatchNorm-backward reductions of the block whose
 * activation gradient it writes (darknet.py:157,162 under autograd): dx is g = dL/d leaky(BN(raw)); with that block's
 * raw conv output and forward BN vectors the finishing pass of the launch leaves
 *   partial[tile][c] = (sum over the tile's pixels of dy, of dy * xhat),  dy = g * leaky'(scale*raw + shift),
 *   xhat = (raw - mean) * invstd;   tile = pixel / ssp_conv_stats_tile_m(B,H,W,Cout_dy,Cin_dx,R,plan) (Winograd plans: groups of 16 tiles),
 * i.e. what ssp_bn_act_bwd's reduce pass would re-read g and raw for; ssp_bn_act_bwd_partials finishes the block.
 * partial: partial_rows * Cin_dx * 2 floats.  A launch with ntile = ssp_conv_stats_tiles(...) <= partial_rows tiles stores row
 * `tile` plainly (deterministic); a launch with more tiles folds tile t into row t % partial_rows with fp32 atomic adds,
 * and the buffer must then be ZERO on entry (ssp_bn_act_bwd_partials with zero_after = 1 leaves it so).  Cin_dx % 4 == 0. */
int ssp_conv_dgrad_bnbwd(const float* dy, const float* wt, float* dx, int B, int H, int W, int Cout_dy, int Cin_dx,
                         int lddy, int lddx, int R, int plan, float* workspace, int64_t workspace_floats,
                         const float* raw, int ldraw, const float* scale, const float* shift, const float* mean,
                         const float* invstd, float slope, float* partial, int partial_rows, void* stream);

/* filter gradient: dw[co][tap][ci] += sum_p dy[p][co] * x[p + tap][ci]; dw is [Cout][R*R][Cin] packed and must be
 * zeroed by the caller (split reduction uses fp32 atomics). */
int ssp_conv_wgrad(const float* dy, const float* x, float* dw, int B, int H, int W, int Cin, int Cout, int lddy,
                   int ldx, int R, void* stream);
/* The same filter gradient of a 3x3 layer (Cin, Cout >= 64 and % 16 == 0) evaluated in the Winograd F(tile x tile, 3x3)
 * domain, tile = 2 or 4 (csrc/conv_wino.hip): dw += the direct result to ~1e-6 (tile 2) / ~4e-6 (tile 4: the direct
 * kernel's own rounding level) of its range with 16/36 / 36/144 of the multiplies.  dw must be 16-byte
 * aligned, [Cout][3][3][Cin] floats (the ssp_repack_fwd layout), written by this launch alone while it runs;
 * workspace: ssp_conv_wgrad_wino_workspace_floats_t(...) floats.  x == NULL: the transformed input V is already at the head
 * of the workspace - left there by this layer's ssp_conv_fwd with a Winograd plan of the SAME tile size and the SAME
 * workspace buffer (both layouts start with V [(tile+2)^2][tiles][Cin]) - so the layer input is transformed once per
 * training step, not twice.  The un-suffixed names = tile 2.
 * tile = 12 (ABI 5): F(2x2) with BOTH transforms on the chip (conv_wino_wgrad_fused.hip) - each lane loads the 2 x 2
 * output-gradient pixels / the 4 x 4 input window of its (channel, tile) and transforms them in registers, waves own a
 * 32 x 32 channel block of all 16 planes over a chunk of tile rows, back-transform their own sums (G^T dU G is linear) and
 * add nine taps per channel pair into dw with fp32 atomics; Cin % 32 == 0 and Cout % 32 == 0 (from 32 channels up), x must
 * be given (no shared transformed input), NO workspace (the query returns 0; a workspace argument is ignored). */
int64_t ssp_conv_wgrad_wino_workspace_floats_t(int B, int H, int W, int Cin, int Cout, int tile);
/* The input transform alone: V[(tile+2)^2][tiles][C] = B^T d B of x [B*H*W][ldx] - for a layer whose FORWARD does not run in
 * the Winograd domain (or runs another tile size) while its filter gradient does: the engine queues it on the second stream
 * during the forward pass, into the head of that layer's filter-gradient workspace, and calls ssp_conv_wgrad_wino_t with
 * x == NULL (an HBM-bound pass under the MFMA-bound forward launches instead of inside the backward pass, where both
 * streams are busy). */
int ssp_wino_input_transform_t(const float* x, int ldx, float* V, int B, int H, int W, int C, int tile, void* stream);
int ssp_conv_wgrad_wino_t(const float* dy, const float* x, float* dw, int B, int H, int W, int Cin, int Cout, int lddy,
                          int ldx, int tile, float* workspace, int64_t workspace_floats, void* stream);
int64_t ssp_conv_wgrad_wino_workspace_floats(int B, int H, int W, int Cin, int Cout);
int ssp_conv_wgrad_wino(const float* dy, const float* x, float* dw, int B, int H, int W, int Cin, int Cout, int lddy,
                        int ldx, float* workspace, int64_t workspace_floats, void* stream);

/* ---- BatchNorm2d(eps) + LeakyReLU(slope) (+ 2x2/2 max-pool): darknet.py:157,162,172 ------------------------- */
/* stats: `ntile` per-tile (mean, M2) pairs per channel from a conv launch; tile_m = its ssp_conv_stats_tile_m (rows per
 * tile; the last tile holds M - (ntile-1)*tile_m), or 0 = counted format (the tiles' pixel counts follow the pairs) */
int ssp_bn_fwd_finalize(const float* stats, int ntile, int tile_m, int M, int C, const float* gamma,
                        const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                        float* mean, float* invstd, float* scale, float* shift, void* stream);
int ssp_bn_eval_prepare(int C, const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float eps, float* mean, float* invstd, float* scale, float* shift,
                        void* stream);
/* out = [maxpool2x2](leaky(scale*x + shift)); H, W are the un-pooled sizes */
int ssp_bn_act_fwd(const float* x, int ldx, float* out, int ldo, const float* scale, const float* shift, int C, int B,
                   int H, int W, int pool, float slope, void* stream);
/* g = gradient wrt the (pooled) activation; dx (may alias x) = gradient wrt the raw conv output;
 * partial: workspace of ssp_bn_bwd_blocks()*C*2 floats; c1, c2: C floats each (reduce -> fp64 finalize -> apply).
 * partial == NULL selects the single-pass form: dgamma / dbeta must be ZERO on entry and 16-byte aligned, the reduce
 * pass accumulates them with fp32 atomics and the apply pass reads them (no finalize launch; c1 / c2 unused). */
int ssp_bn_bwd_blocks(void);
int ssp_bn_act_bwd(const float* x, int ldx, const float* g, int ldg, float* dx, int lddx, const float* scale,
                   const float* shift, const float* mean, const float* invstd, int C, int B, int H, int W, int pool,
                   float slope, int training, float* partial, float* dgamma, float* dbeta, float* c1, float* c2,
                   void* stream);
/* un-pooled block whose reductions came out of ssp_conv_dgrad_bnbwd: fp64 sum of the `npartial` per-tile pairs ->
 * dgamma / dbeta (and c1 / c2), then dx (may alias x) = scale * (dy - c1 - xhat * c2) */
int ssp_bn_act_bwd_partials(const float* x, int ldx, const float* g, int ldg, float* dx, int lddx, const float* scale,
                            const float* shift, const float* mean, const float* invstd, int C, int B, int H, int W,
                            float slope, int training, float* partial, int npartial, int zero_after, float* dgamma,
                            float* dbeta, float* c1, float* c2, void* stream);
/* just the fp64 finalize of `npartial` rows of (sum dy, sum dy * xhat) pairs -> dgamma, dbeta, c1 = mean(dy),
 * c2 = mean(dy * xhat) over npix pixels (c1 = c2 = 0 when training == 0) */
int ssp_bn_bwd_finalize(float* partial, int npartial, int C, int64_t npix, int training, int zero_after, float* dgamma,
                        float* dbeta, float* c1, float* c2, void* stream);
/* out[c] = sum_p g[p][c]  (bias gradient of the linear head conv) */
int ssp_colsum(const float* g, int ldg, int64_t M, int C, float* out, void* stream);

/* ---- first block: conv 3x3 (image, Cin padded to 4 -> 32) + BatchNorm + leaky + 2x2/2 max-pool, training mode, with
 * the convolution RECOMPUTED by every pass instead of stored (darknet.py:154-176 on the 416 x 416 input and its autograd
 * backward; SURVEY.md section 7: the 22 MB-per-image map is the largest tensor of the net, its conv has K = 27).
 * x: [B*H*W][4] (channel 3 zero), wt: ssp_repack_fwd(conv.weight, Cinp = 4) = [32][9][4]; H even, W % 16 == 0.
 *   fwd_stats : stats[ssp_first_groups(B,H,W)][32][2] = per-group (mean, M2) of the raw conv output; feed
 *               ssp_bn_fwd_finalize(stats, groups, ssp_first_tile_pixels(), B*H*W, 32, ...)
 *   fwd_apply : out[pooled pixel][0..32) = maxpool2x2(leaky(scale * conv + shift))
 *   bwd_reduce: partial[groups][32][2] = (sum dy, sum dy * xhat) from g = dL/d out (pool arg-max = first maximum in
 *               window scan order, as ATen); feed ssp_bn_bwd_finalize
 *   bwd_wgrad : dw[32][9][4] = filter gradient (dx formed in registers from g, the recomputed conv and c1 / c2; the padding
 *               channel written as zero).  Two launches: every workgroup leaves its partial gradient in `workspace`
 *               (ssp_first_wgrad_workspace_floats(B,H,W) floats), a second kernel sums the workgroups in float64 - the
 *               gradient is a sum of terms that cancel ~1e3 : 1, and deterministic this way (ABI 4; ABI 3 accumulated
 *               with fp32 atomics into a zeroed dw)  */
int ssp_first_tile_pixels(void);
int ssp_first_groups(int B, int H, int W);
int ssp_first_fwd_stats(const float* x, const float* wt, float* stats, int B, int H, int W, void* stream);
int ssp_first_fwd_apply(const float* x, const float* wt, const float* scale, const float* shift, float slope, float* out,
                        int ldo, int B, int H, int W, void* stream);
/* the raw conv output itself, [B*H*W][ldraw], by the same instruction sequence the four passes use (bit-identical
 * values): for checkers that need the pool / leaky decisions of the fused path - the training path never calls it */
int ssp_first_conv_raw(const float* x, const float* wt, float* raw, int ldraw, int B, int H, int W, void* stream);
int ssp_first_bwd_reduce(const float* x, const float* wt, const float* g, int ldg, const float* scale, const float* shift,
                         const float* mean, const float* invstd, float slope, float* partial, int B, int H, int W,
                         void* stream);
int ssp_first_bwd_wgrad(const float* x, const float* wt, const float* g, int ldg, const float* scale, const float* shift,
                        const float* mean, const float* invstd, const float* c1, const float* c2, float slope, float* dw,
                        float* workspace, int64_t workspace_floats, int B, int H, int W, void* stream);
int64_t ssp_first_wgrad_workspace_floats(int B, int H, int W);

/* ---- optimizer ------------------------------------------------------------------ */
/* One torch.optim.SGD step (train.py:388,106) over a contiguous fp32 range of n values, in place:
 *   d = grad + weight_decay*param;  buf = first_step ? d : momentum*buf + (1-dampening)*d;
 *   param -= lr * (nesterov ? d + momentum*buf : buf)        (momentum == 0: param -= lr*d, momentum_buf may be NULL)
 * All three pointers 16-byte aligned. */
int ssp_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum,
                 float dampening, float weight_decay, int nesterov, int first_step, void* stream);

/* ---- layout / index kernels -------------------------------------------------------------------------------- */
int ssp_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, int Cpad, int ld, void* stream);
int ssp_nhwc_to_nchw(const float* src, float* dst, int B, int C, int H, int W, int ld, void* stream);
/* uint8 (B,H,W,C) image bytes -> fp32 [B*H*W][ld], value/255 as transforms.ToTensor (dataset.py:113-131); channels
 * [C,Cpad) = 0.  SURVEY.md section 8(f) row 3: the byte image is uploaded instead of the fp32 NCHW tensor. */
int ssp_u8hwc_to_nhwc(const unsigned char* src, float* dst, int B, int H, int W, int C, int Cpad, int ld, void* stream);
/* ---- training-time augmentation (second half): image.py:14-31,46-76,111-128 ----------
 * What the reference does per sample with Pillow on the host - bg.resize + mask composite, jitter crop (zero fill) +
 * resize to the network shape, HSV jitter through Image.point tables - as batched launches, byte-exact with Pillow.
 * A resize is two passes (horizontal over the rows the vertical pass needs, 8-bit intermediate, then vertical) of a
 * per-output-index FIR with Pillow's 22-bit fixed-point bicubic coefficients; the coefficient rows come from the caller
 * (singleshotpose_amd/image.py computes them in double, as ImagingResample's precompute_coeffs does).
 * One descriptor per sample, an array of them in DEVICE memory; u8 images are (rows, cols, 3) with a byte pitch. */
typedef struct SspResampleDesc {
  const void* src;     /* pass 0: source image (src_h x src_w); pass 1: the horizontal pass's output (src_h rows) */
  void* dst;           /* dst_h x dst_w */
  const void* bounds;  /* int32 [n_out][2] = (first input index, taps) of this pass's axis */
  const void* kk;      /* int32 [n_out][ksize] fixed-point coefficients */
  const void* img;     /* epilogue 1: foreground image, dst-sized */
  const void* mask;    /* epilogue 1: mask, dst-sized: channel value >= 128 keeps the foreground (image.py:121-125) */
  const void* lut;     /* epilogue 2: 768 bytes = H, S, V tables of image.py:14-31 */
  int src_w, src_h, src_pitch;
  int x0, y0;          /* pass 0: logical input pixel (Y, X) is source pixel (Y + y0, X + x0); outside the source = 0 (crop) */
  int row0;            /* first logical input row the vertical pass needs (pass 0 writes rows row0 ...; pass 1 reads them) */
  int dst_w, dst_h, dst_pitch;
  int ksize, img_pitch, reserved;
} SspResampleDesc;
/* pass 0 = horizontal (epilogue must be 0), 1 = vertical; epilogue 0 = store, 1 = composite with img / mask,
 * 2 = distort_image (RGB -> HSV -> tables -> RGB).  max_dst_pixels = the largest dst_w * dst_h of the batch. */
int ssp_resample_u8(const SspResampleDesc* descs_dev, int count, int pass, int epilogue, int max_dst_pixels, void* stream);
/* distort_image alone on npix packed RGB pixels (mode 0, lut = 768 bytes); modes 1 / 2 = RGB -> HSV / HSV -> RGB only
 * (Image.convert), which is how the tests pin both conversions over all 2^24 inputs. */
int ssp_distort_u8(const unsigned char* rgb, unsigned char* out, int64_t npix, const unsigned char* lut, int mode, void* stream);
/* conv.weight (Cout,Cin,R,R) (cfg.py:157,175) -> [Cout][R*R][Cinp] */
int ssp_repack_fwd(const float* w, float* out, int Cout, int Cin, int Cinp, int R, void* stream);
/* conv.weight -> [Cin][R*R][Coutp], taps flipped */
int ssp_repack_dgrad(const float* w, float* out, int Cout, int Cin, int Coutp, int R, void* stream);
/* same operand from filters stored channels-last, [Cout][R*R][Cin] (a torch.channels_last conv.weight): that layout IS
 * the forward operand and the layout ssp_conv_wgrad accumulates, so such parameters need no other repack */
int ssp_repack_dgrad_packed(const float* wp, float* out, int Cout, int Cin, int Coutp, int R, void* stream);
/* packed gradient [Cout][R*R][Cinp] -> (Cout,Cin,R,R) */
int ssp_unpack_grad(const float* dwp, float* grad, int Cout, int Cin, int Cinp, int R, void* stream);
/* Reorg(2) (darknet.py:16-35) on NHWC: dst[b,y/2,x/2,((y&1)*2+(x&1))*C+c] = src[b,y,x,c]; backward = inverse */
int ssp_reorg(const float* src, int lds, float* dst, int ldd, int C, int B, int H, int W, int backward,
              int accumulate, void* stream);
/* route (darknet.py:96-106): dst[p][0..C) (+)= src[p][0..C) */
int ssp_copy_channels(const float* src, int lds, float* dst, int ldd, int C, int64_t M, int accumulate, void* stream);
/* nn.MaxPool2d(2,2) (darknet.py:172) standalone */
int ssp_maxpool_fwd(const float* x, int ldx, float* out, int ldo, int C, int B, int H, int W, void* stream);
int ssp_maxpool_bwd(const float* x, int ldx, const float* g, int ldg, float* dx, int lddx, int C, int B, int H, int W,
                    int accumulate, void* stream);

/* ---- RegionLoss (region_loss.py:9-175, region_loss_multi.py:9-189, utils.py:138-187) ----------------------- */
/* out, grad: (nB, nA*(2K+1+nC), nH, nW) NCHW contiguous; target: (nB, 50*(2K+3)) float or double on the device;
 * partials: nB*8 floats; stats[8] = {loss_x, loss_y, loss_conf, loss_cls, total, nGT, nCorrect, nProposals}. */
int ssp_region_loss(const float* out, const void* target, int target_is_f64, float* grad, float* partials,
                    float* stats, int nB, int nA, int nC, int nH, int nW, int num_keypoints, float noobject_scale,
                    float object_scale, float coord_scale, float class_scale, float thresh, int conf_on, int multi,
                    const float* anchors, int anchor_step, void* stream);
/* get_region_boxes (utils.py:216-296): boxes[b][2K+4] = {2K coords, det_conf, cls_max_conf, cls_max_id, conf} */
int ssp_region_decode_argmax(const float* out, float* boxes, int nB, int nA, int nC, int nH, int nW,
                             int num_keypoints, int only_objectness, void* stream);

/* get_multi_region_boxes (multi_obj_pose_estimation/utils_multi.py:266-382): dense decode, scan order
 * key = (cy*nW + cx)*nA + anchor: rows[b][key][2K+3+nC] = {2K coords, det_conf, cls_max_conf, cls_max_id, softmax[nC]} */
int ssp_region_decode_all(const float* out, float* rows, int nB, int nA, int nC, int nH, int nW, int num_keypoints,
                          void* stream);

/* ---- PnP (utils.py:86-100: cv2.solvePnP ITERATIVE + cv2.Rodrigues) ----------------------------------------- */
/* batched: pts3d [n][N][3], pts2d [n][N][2], K [n][9] (row-major) doubles on the device -> Rt [n][12] = R (9) | t (3) */
int ssp_pnp_batched(const double* pts3d, const double* pts2d, const double* K, double* Rt, int n, int N, int max_iter,
                    void* stream);

/* ---- evaluation metrics (valid.py:148-172, utils.py:31-58) ------------------------ */
/* n pose pairs over one mesh: vertices [N][3], Rt_gt / Rt_pr [n][12] = R (9, row-major) | t (3) (ssp_pnp_batched's
 * layout), K [9] (k_per_pose = 0) or [n][9], all doubles on the device ->
 * out [n][4] = {mean 2D reprojection distance (float32 projections, valid.py:160-165), mean 3D vertex distance
 * (valid.py:168-172), translation distance (valid.py:148), angular distance in degrees (utils.py:31-35)} */
int ssp_pose_errors(const double* vertices, int N, const double* Rt_gt, const double* Rt_pr, const double* K,
                    int k_per_pose, int n, double* out, void* stream);
/* calc_pts_diameter (utils.py:50-58): out[0] = largest pairwise distance of pts [N][3]; scratch: 8 bytes */
int ssp_pts_diameter(const double* pts, int N, double* out, double* scratch, void* stream);

/* ---- timed-launch bookkeeping (bench.py roofline): HIP events around every launch of a kernel family -------- */
/* mask: bit k = kernel family k (0 conv fwd, 1 conv dgrad, 2 conv wgrad, 3 BN/activation, 4 layout, 5 region/pnp/eval,
 * 6 optimizer, 7 first block forward passes, 8 first block backward passes); -1 = all, 0 = off.  Every timed launch costs two event packets on its stream. */
int ssp_prof_enable(int mask);
/* ms[k], work[k] (FLOPs or bytes), count[k] for k in 0..ssp_prof_nkinds()-1; synchronises on the recorded events */
int ssp_prof_nkinds(void);
int ssp_prof_collect(double* ms, double* work, int64_t* count);

#ifdef __cplusplus
}
#endif
#endif /* SSP_HIP_H */
