// Activations of the training path and their derivatives as autograd differentiates them (F.hardswish, F.relu, F.gelu, torch.sigmoid), shared
// by the elementwise kernels (kernels_train.hip: act_kernel) and the BatchNorm kernels that apply them in the same pass (kernels_stage1.hip).
#pragma once
#include "esam3_common.h"

__device__ __forceinline__ float act_fwd(float x, int act) {
  switch (act) {
    case ACT_RELU: return x > 0.f ? x : 0.f;
    case ACT_GELU: return gelu_fast(x);
    case ACT_HSWISH: return x * fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f);
    case ACT_SIGMOID: return 1.f / (1.f + __expf(-x));
    default: return x;
  }
}
__device__ __forceinline__ float act_grad(float x, int act) {
  switch (act) {
    case ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case ACT_GELU: {  // Phi(x) + x phi(x)
      const float phi = 0.3989422804014327f * __expf(-0.5f * x * x);
      const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
      return cdf + x * phi;
    }
    // torch (ATen cpu/Activation.cpp hardswish_backward): 0 for x <= -3, x / 3 + 1 / 2 inside, 1 for x >= 3 (the CUDA kernel puts
    // the two boundary points on the other side; a set of measure zero)
    case ACT_HSWISH: return x <= -3.f ? 0.f : (x >= 3.f ? 1.f : (2.f * x + 3.f) * (1.f / 6.f));
    case ACT_SIGMOID: {
      const float sg = 1.f / (1.f + __expf(-x));
      return sg * (1.f - sg);
    }
    default: return 1.f;
  }
}

