// v_cndmask_b32 issue cost on gfx950 in the forms hipcc emits (VCC and SGPR-pair masks).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
constexpr int ITER = 512;
#define K(NAME, SETUP, ASMSTR)                                                    \
  __global__ void __launch_bounds__(256) k_##NAME(uint32_t* out, uint32_t s) {    \
    uint32_t a[8], b = s * 2654435761u + threadIdx.x;                             \
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 747796405u + i + s;          \
    SETUP;                                                                        \
    for (int it = 0; it < ITER; it++) {                                           \
      _Pragma("unroll") for (int u = 0; u < 4; u++) {                             \
        _Pragma("unroll") for (int i = 0; i < 8; i++) {                           \
          asm volatile(ASMSTR : "+v"(a[i]) : "v"(b));                             \
        }                                                                         \
      }                                                                           \
    }                                                                             \
    uint32_t r = 0;                                                               \
    for (int i = 0; i < 8; i++) r ^= a[i];                                        \
    if (r == 0x12345678u) out[0] = r;                                             \
  }
K(cnd_vcc_e32, asm volatile("v_cmp_gt_u32 vcc, %0, %1" ::"v"(b), "v"(a[0]) : "vcc"), "v_cndmask_b32 %0, %0, %1, vcc")
K(cnd_sgpr_e64, asm volatile("v_cmp_gt_u32 s[20:21], %0, %1" ::"v"(b), "v"(a[0]) : "s20", "s21"), "v_cndmask_b32 %0, %0, %1, s[20:21]")
K(xor3, , "v_xor_b32 %0, %0, %1")
K(bfi, , "v_bfi_b32 %0, %1, %0, %1")
K(add, , "v_add_u32 %0, %0, %1")
typedef void (*kern_t)(uint32_t*, uint32_t);
int main() {
  uint32_t* d; hipMalloc(&d, 64);
  struct { const char* n; kern_t k; } es[] = {{"cnd_vcc_e32", k_cnd_vcc_e32}, {"cnd_sgpr_e64", k_cnd_sgpr_e64}, {"xor", k_xor3}, {"bfi", k_bfi}, {"add", k_add}};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (auto& e : es) {
    int blocks = 256 * 2 * 8;
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, 1u); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    double insts = (double)blocks * 4 * ITER * 32;
    printf("%-14s %.4f ms  %.2f cyc/inst (2.4GHz nominal)\n", e.n, ms, 256.0 * 4 * 2.4e9 * ms * 1e-3 / insts);
  }
  return 0;
}
