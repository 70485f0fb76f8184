#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned* in, float* out, float scale) {
  unsigned v = in[0];
  for (int i = 0; i < 4; ++i) {
    bf16x2 r;
    f32x2 f;
    switch (i) {
      case 0: r = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(v, scale, 0); f = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(v, scale, 0); break;
      case 1: r = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(v, scale, 1); f = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(v, scale, 1); break;
      case 2: r = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(v, scale, 2); f = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(v, scale, 2); break;
      default: r = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(v, scale, 3); f = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(v, scale, 3); break;
    }
    out[i * 4 + 0] = (float)r[0]; out[i * 4 + 1] = (float)r[1];
    out[i * 4 + 2] = f[0]; out[i * 4 + 3] = f[1];
  }
}
int main() {
  unsigned h = 0x76543210u, h2 = 0xFEDCBA98u; unsigned* d; float* o; float ho[16];
  hipMalloc(&d, 4); hipMalloc(&o, 64);
  const float scales[] = {1.0f, 1.5f, 2.0f, 0.75f, 3.0f};
  for (unsigned w : {h, h2}) for (float s : scales) {
    hipMemcpy(d, &w, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d, o, s);
    hipMemcpy(ho, o, 64, hipMemcpyDeviceToHost);
    printf("word %08x scale %.2f:", w, s);
    for (int i = 0; i < 4; ++i) printf("  sel%d bf16(%g,%g) f32(%g,%g)", i, ho[i*4], ho[i*4+1], ho[i*4+2], ho[i*4+3]);
    printf("\n");
  }
  return 0;
}
