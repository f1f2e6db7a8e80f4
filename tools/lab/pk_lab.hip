// Lab: semantics of the packed-f32 instructions the generated backward loops use (v_pk_mul_f32; v_pk_add_f32 with a broadcast, negated second source)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 x = {1.5f + threadIdx.x, 10.f}, y = {3.f, 7.f}, r0, r1, r2;
  asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r0) : "v"(x), "v"(y));
  asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r1) : "v"(x), "v"(y));
  asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r2) : "v"(x), "v"(y));
  if (threadIdx.x == 0) { out[0] = r0[0]; out[1] = r0[1]; out[2] = r1[0]; out[3] = r1[1]; out[4] = r2[0]; out[5] = r2[1]; }
}
int main() {
  float* d; hipMalloc(&d, 64); k<<<1, 64>>>(d); float h[6]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  printf("pk_mul (1.5,10)*(3,7) = (%g, %g) want (4.5, 70)\n", h[0], h[1]);
  printf("pk_sub bcast lo: (1.5,10) - 3 = (%g, %g) want (-1.5, 7)\n", h[2], h[3]);
  printf("pk_sub bcast hi: (1.5,10) - 7 = (%g, %g) want (-5.5, 3)\n", h[4], h[5]);
  return 0;
}
