// Shared device/host helpers for the singleshotpose hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define SSP_OK 0
#define SSP_ERR_ARG -1
#define SSP_ERR_HIP -2
#define SSP_ERR_UNSUPPORTED -3

// thread-local last-error string (include/ssp_hip.h: ssp_last_error)
void ssp_set_error(const char* fmt, ...);

#define SSP_CHECK_ARG(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      ssp_set_error(__VA_ARGS__);           \
      return SSP_ERR_ARG;                   \
    }                                       \
  } while (0)

#define SSP_CHECK_LAUNCH(name)                                                        \
  do {                                                                                \
    hipError_t e__ = hipGetLastError();                                               \
    if (e__ != hipSuccess) {                                                          \
      ssp_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));           \
      return SSP_ERR_HIP;                                                             \
    }                                                                                 \
  } while (0)

// Timed-launch bookkeeping (ssp_prof_*): kernel families we report rooflines for.
enum SspProfKind {
  SSP_PROF_CONV_FWD = 0,   // implicit-GEMM conv forward launches
  SSP_PROF_CONV_DGRAD = 1, // same kernel, flipped/transposed filters
  SSP_PROF_CONV_WGRAD = 2,
  SSP_PROF_BN_ACT = 3,     // BN finalize/apply/leaky/pool fwd+bwd elementwise family
  SSP_PROF_LAYOUT = 4,     // repack / transpose / reorg / maxpool
  SSP_PROF_REGION = 5,     // region loss / decode / pnp
  SSP_PROF_OPTIM = 6,      // fused SGD step
  SSP_PROF_FIRST_FWD = 7,  // first block, fused with recompute (conv_first.hip): statistics pass + apply pass
  SSP_PROF_FIRST_BWD = 8,  // first block: BatchNorm-backward reduce pass + filter-gradient pass
  // Winograd transform / finishing passes (conv_wino.hip; HBM-bound, work = algorithmic bytes), by the conv family of the
  // launch they belong to; nested INSIDE that family's scope (its time includes them)
  SSP_PROF_WINO_FWD = 9,
  SSP_PROF_WINO_DGRAD = 10,
  SSP_PROF_WINO_WGRAD = 11,
  // on-chip Winograd launches (conv_wino_fused.hip, conv_wino_wgrad_fused.hip): families of their own, NOT nested in 0 - 2, so
  // that the implicit-GEMM kernel's launches (the dominant kernel of the roofline reports) stay a clean set
  SSP_PROF_ONCHIP_FWD = 12,
  SSP_PROF_ONCHIP_DGRAD = 13,
  SSP_PROF_ONCHIP_WGRAD = 14,
  SSP_PROF_NKINDS = 15
};

struct SspProfScope {
  SspProfScope(int kind, hipStream_t s, double work);
  ~SspProfScope();
  int slot;
  hipStream_t stream;
};

// tuning knobs set through ssp_set_option (ssp_api.hip)
enum SspOption { SSP_OPT_IGEMM_XCD = 0, SSP_OPT_IGEMM_VARIANT = 1, SSP_OPT_WGRAD_VARIANT = 2, SSP_OPT_IGEMM_PLAN = 3, SSP_OPT_WGRAD_SPLIT = 4, SSP_OPT_WINO_VARIANT = 5, SSP_OPT_ACC_CHUNK = 6, SSP_OPT_COUNT = 7 };
int ssp_option(int which);

// Per-device cache of one kernel instantiation's dynamic-LDS reservation and chip-wide resident-workgroup count
// (the only mutable state the library keeps besides the experiment knobs): filled under a mutex, one entry per HIP
// device so that several devices driven from one process each get their hipFuncSetAttribute call.
#define SSP_MAX_DEVICES 16
struct SspKernelCache {
  int configured[SSP_MAX_DEVICES];   // bytes of dynamic LDS the kernel was last configured for on that device
  int slots[SSP_MAX_DEVICES];        // workgroups resident on the whole chip at that size
};
// Returns SSP_OK and the resident-workgroup count in *slots (may be nullptr); configures the kernel on first use.
int ssp_kernel_prepare(const void* kern, int lds_bytes, int threads, SspKernelCache* cache, int* slots, const char* name);

// BatchNorm-backward reductions fused into a data-gradient launch (ConvArgs::bn_*, include/ssp_hip.h ssp_conv_dgrad_bnbwd)
struct SspBnBwdFuse {
  const float* raw;      // raw conv output of the block that produced dx's activation, [pixels][ldraw]
  int ldraw;
  const float* scale;    // that block's forward BN vectors (ssp_bn_fwd_finalize / ssp_bn_eval_prepare)
  const float* shift;
  const float* mean;
  const float* invstd;
  float slope;
  float* partial;        // out: [min(ntile, nslot)][C][2] = (sum dy, sum dy * xhat) per M tile (folded modulo nslot)
  int nslot;
};

// Division of a non-negative 32-bit value by a launch-invariant divisor in ~5 VALU instructions (the compiler's
// sequence for a run-time divisor is ~25; the conv kernels divide per tile row in their prologue, on the same lanes the
// fp32 MFMA uses).  Round-up method: q = (t + ((n - t) >> 1)) >> (shr - 1), t = umulhi(n, mul); exact for 0 <= n < 2^32.
struct SspFastDiv {
  unsigned mul, shr, d;
};
static inline SspFastDiv ssp_fastdiv(unsigned d) {
  SspFastDiv f;
  f.d = d;
  unsigned l = 0;
  while ((1ull << l) < d) ++l;      // ceil(log2 d)
  f.shr = l;
  f.mul = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  return f;
}
__device__ __forceinline__ unsigned ssp_div(unsigned n, const SspFastDiv& f) {
  if (f.shr == 0) return n;         // d == 1 (wave-uniform branch)
  const unsigned t = __umulhi(n, f.mul);
  return (t + ((n - t) >> 1)) >> (f.shr - 1);
}

static inline int ssp_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Observed dispatch places workgroup b on XCD b%8 (MI355X_MICROARCH.md, "Workgroup dispatch").
// Remap so that each XCD walks a contiguous chunk of the logical tile order (L2 locality only,
// never correctness).  Bijective for any nwg (cdna_hip_programming.md §5 "XCD swizzle must be bijective").
__device__ __forceinline__ int ssp_xcd_remap(int bid, int nwg) {
  const int nx = 8;
  int xcd = bid % nx, idx = bid / nx;
  int q = nwg / nx, r = nwg % nx;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
