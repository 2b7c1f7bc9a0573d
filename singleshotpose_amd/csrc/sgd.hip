// SGD with momentum, dampening, L2 weight decay and Nesterov over one contiguous fp32 range: the update the reference
// runs through torch.optim.SGD (train.py:388 `optim.SGD(model.parameters(), lr=learning_rate/batch_size,
// momentum=momentum, dampening=0, weight_decay=decay*batch_size)`, train.py:106 `optimizer.step()`).
//
//   d = g + weight_decay * p
//   first step with momentum:  buf = d          later steps:  buf = momentum * buf + (1 - dampening) * d
//   d = nesterov ? d + momentum * buf : buf     (momentum == 0: d stays g + weight_decay * p)
//   p = p - lr * d
//
// HBM-bound: 3 reads + 2 writes of 4 B per parameter (202 MB each for yolo-pose.cfg => ~1 GB, ~0.13 ms at 8 TB/s).
// The host side (singleshotpose_amd/optim.py) keeps parameters, momentum and the backward's gradients in three flat
// buffers with one layout, so a whole step is ONE launch instead of torch's ~70 x 3 foreach segments.
#include "ssp_common.h"

struct SgdArgs {
  float* p;
  const float* g;
  float* m;
  int64_t n;
  float lr, momentum, dampening, wd;
  int nesterov, first;
};

__device__ __forceinline__ float sgd_one(float p, float g, float& buf, const SgdArgs& a) {
  // one fused multiply-add per torch foreach pass (add(alpha) -> mul, add(alpha) -> add(alpha)): the same operation
  // order as torch.optim.SGD, each pass rounded once (torch's CPU build may round the product separately: <= 1 ulp)
  float d = (a.wd != 0.f) ? __fmaf_rn(a.wd, p, g) : g;
  if (a.momentum != 0.f) {
    if (a.first) buf = d;
    else buf = __fmaf_rn(1.f - a.dampening, d, __fmul_rn(a.momentum, buf));
    d = a.nesterov ? __fmaf_rn(a.momentum, buf, d) : buf;
  }
  return __fmaf_rn(-a.lr, d, p);
}

__global__ void __launch_bounds__(256) sgd_kernel(SgdArgs a) {
  const int64_t n4 = a.n >> 2;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const bool mom = a.momentum != 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 p = reinterpret_cast<const float4*>(a.p)[i];
    const float4 g = reinterpret_cast<const float4*>(a.g)[i];
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mom && !a.first) b = reinterpret_cast<const float4*>(a.m)[i];
    p.x = sgd_one(p.x, g.x, b.x, a);
    p.y = sgd_one(p.y, g.y, b.y, a);
    p.z = sgd_one(p.z, g.z, b.z, a);
    p.w = sgd_one(p.w, g.w, b.w, a);
    reinterpret_cast<float4*>(a.p)[i] = p;
    if (mom) reinterpret_cast<float4*>(a.m)[i] = b;
  }
  // tail (n not a multiple of 4): the first workgroup finishes it
  if (blockIdx.x == 0) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    if (i < a.n) {
      float b = (mom && !a.first) ? a.m[i] : 0.f;
      a.p[i] = sgd_one(a.p[i], a.g[i], b, a);
      if (mom) a.m[i] = b;
    }
  }
}

int ssp_sgd_step_launch(float* p, const float* g, float* m, int64_t n, float lr, float momentum, float dampening,
                        float weight_decay, int nesterov, int first_step, hipStream_t stream) {
  SSP_CHECK_ARG(p != nullptr && g != nullptr && n > 0, "sgd_step: null buffer or empty range");
  SSP_CHECK_ARG(momentum == 0.f || m != nullptr, "sgd_step: momentum needs a momentum buffer");
  SSP_CHECK_ARG(!nesterov || (momentum > 0.f && dampening == 0.f), "sgd_step: nesterov needs momentum > 0 and dampening == 0");
  SSP_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m) & 15) == 0, "sgd_step: buffers must be 16-byte aligned");
  SspProfScope prof(SSP_PROF_OPTIM, stream, 0.0);
  SgdArgs a{p, g, m, n, lr, momentum, dampening, weight_decay, nesterov, first_step};
  int64_t blocks = ((n >> 2) + 255) / 256;
  const int64_t cap = 256 * 8;
  if (blocks < 1) blocks = 1;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
  SSP_CHECK_LAUNCH("sgd_step");
  return SSP_OK;
}
