// What do op_sel / neg_hi do on v_pk_add_f32 (gfx950)?  build: hipcc --offload-arch=gfx950 -O3 -o pk_probe pk_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(float* o) {
  f32x2 a = {1.0f, 10.0f}, b = {100.0f, 1000.0f};
  if (threadIdx.x == 99) { a[0] = o[0]; b[1] = o[1]; }
  f32x2 r0, r1, r2, r3;
  asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(r0) : "v"(a), "v"(b));
  asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0]" : "=v"(r1) : "v"(a), "v"(b));
  asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r2) : "v"(a), "v"(b));
  asm volatile("v_pk_add_f32 %0, %1, %2 neg_hi:[1,0]" : "=v"(r3) : "v"(a), "v"(b));
  if (threadIdx.x == 0) { o[0] = r0[0]; o[1] = r0[1]; o[2] = r1[0]; o[3] = r1[1]; o[4] = r2[0]; o[5] = r2[1]; o[6] = r3[0]; o[7] = r3[1]; }
}
int main() {
  float* d; hipMalloc(&d, 64); k<<<1, 64>>>(d); float h[8]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
  printf("a=(1,10) b=(100,1000)\nop_sel[1,0] op_sel_hi[1,0] neg_hi[1,0]: %g %g\nop_sel[1,0] op_sel_hi[1,0]: %g %g\nop_sel[1,0]: %g %g\nneg_hi[1,0]: %g %g\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
}
