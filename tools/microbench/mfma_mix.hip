// Does an i8 MFMA stream hide behind a v_mad_u64_u32 stream on gfx950?
//
// Question behind it (HISTORY.md §6): the linear layers of Hades multiply the state by CONSTANT
// field elements -- over a batch that is a constant (Toeplitz-of-bytes) matrix times a matrix of
// state bytes, i.e. MFMA-shaped work, while everything else in the engine is 64-bit integer MADs
// on the VALU.  It only pays if the matrix pipe runs beside the VALU instead of taking its issue
// slots, and if the power budget has room.  Three kernels, same grid (2 waves per SIMD):
//   mad   : ITER x 32 v_mad_u64_u32 (8 independent chains)
//   mfma  : ITER x M  v_mfma_i32_32x32x32_i8 (2 independent accumulators)
//   mix   : both in one loop body
// Output: ms per kernel, cycles per MAD, cycles per MFMA, and mix / mad.
//
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_mix mfma_mix.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                        \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                          \
    }                                                                                   \
  } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int ITER = 2048;

template <int NMAD_GROUPS, int NMFMA>
__global__ void __launch_bounds__(256) k_mix(uint32_t* out, uint32_t s) {
  uint64_t acc[8];
  uint32_t b = s * 2654435761u + threadIdx.x, c = s ^ 0x9e3779b9u;
  for (int i = 0; i < 8; i++) acc[i] = threadIdx.x * 747796405u + i + s;
  v4i ma = {(int)b, (int)c, (int)(b ^ c), (int)(b + c)};
  v4i mb = {(int)(b * 3), (int)(c * 5), (int)(b * 7), (int)(c * 11)};
  v16i c0 = {0}, c1 = {0};
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int g = 0; g < NMAD_GROUPS; g++) {
#pragma unroll
      for (int i = 0; i < 8; i++)
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(b), "v"(c) : "vcc");
      if (NMFMA > 0 && g < NMFMA) {
        if (g & 1)
          c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ma, mb, c1, 0, 0, 0);
        else
          c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ma, mb, c0, 0, 0, 0);
      }
    }
    if (NMAD_GROUPS == 0) {
#pragma unroll
      for (int g = 0; g < NMFMA; g++) {
        if (g & 1)
          c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ma, mb, c1, 0, 0, 0);
        else
          c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ma, mb, c0, 0, 0, 0);
      }
    }
  }
  uint32_t r = 0;
  for (int i = 0; i < 8; i++) r ^= (uint32_t)acc[i] ^ (uint32_t)(acc[i] >> 32);
  for (int i = 0; i < 16; i++) r ^= (uint32_t)c0[i] ^ (uint32_t)c1[i];
  if (r == 0x12345678u) out[0] = r;
}

// r05: WAVE-SPECIALISED mix.  One workgroup of 12 waves per CU = three per SIMD (waves of a workgroup
// go round-robin over the four SIMDs): waves 0..3 issue ONLY MFMAs (M per iteration), waves 4..11 ONLY
// MADs (32 per iteration) — the question behind VERDICT r04 item 5: if the MFMAs of the hash moved to a
// wave of their own, would the S-box waves keep their whole MAD issue rate?  roles: bit 0 = the MAD
// waves run, bit 1 = the MFMA waves run (the others leave at once).
template <int NMFMA>
__global__ void __launch_bounds__(768) k_spec(uint32_t* out, uint32_t s, int roles) {
  const int wave = threadIdx.x >> 6;
  uint32_t r = 0;
  if (wave < 4) {
    if (!(roles & 2)) return;
    uint32_t b = s * 2654435761u + threadIdx.x, c = s ^ 0x9e3779b9u;
    v4i ma = {(int)b, (int)c, (int)(b ^ c), (int)(b + c)};
    v4i mb = {(int)(b * 3), (int)(c * 5), (int)(b * 7), (int)(c * 11)};
    v16i c0 = {0}, c1 = {0};
    for (int it = 0; it < ITER; it++) {
#pragma unroll
      for (int g = 0; g < NMFMA; g++) {
        if (g & 1)
          c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ma, mb, c1, 0, 0, 0);
        else
          c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ma, mb, c0, 0, 0, 0);
      }
    }
    for (int i = 0; i < 16; i++) r ^= (uint32_t)c0[i] ^ (uint32_t)c1[i];
  } else {
    if (!(roles & 1)) return;
    uint64_t acc[8];
    uint32_t b = s * 2654435761u + threadIdx.x, c = s ^ 0x9e3779b9u;
    for (int i = 0; i < 8; i++) acc[i] = threadIdx.x * 747796405u + i + s;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
      for (int g = 0; g < 4; g++)
#pragma unroll
        for (int i = 0; i < 8; i++)
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(b), "v"(c) : "vcc");
    }
    for (int i = 0; i < 8; i++) r ^= (uint32_t)acc[i] ^ (uint32_t)(acc[i] >> 32);
  }
  if (r == 0x12345678u) out[0] = r;
}
template <int NMFMA>
static float time_spec(int blocks, uint32_t* out, int roles) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  k_spec<NMFMA><<<blocks, 768>>>(out, 1, roles);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    CHECK(hipEventRecord(e0));
    for (int k = 0; k < 4; k++) k_spec<NMFMA><<<blocks, 768>>>(out, 2 + rep, roles);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms / 4 < best) best = ms / 4;
  }
  return best;
}
template <int NMFMA>
static void report_spec(int cus, uint32_t* out) {
  const float t_mad = time_spec<NMFMA>(cus, out, 1), t_mfma = time_spec<NMFMA>(cus, out, 2),
              t_both = time_spec<NMFMA>(cus, out, 3);
  printf("  2 MAD waves (32/iter) + 1 MFMA wave (%d/iter) per SIMD: MAD waves alone %.3f ms, MFMA wave alone %.3f ms, "
         "together %.3f ms = x%.3f of the longer one (x1.000 = the pipes run side by side)\n",
         NMFMA, t_mad, t_mfma, t_both, t_both / (t_mad > t_mfma ? t_mad : t_mfma));
}

template <typename K>
static float time_kernel(K kern, int blocks, uint32_t* out) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  kern<<<blocks, 256>>>(out, 1);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    CHECK(hipEventRecord(e0));
    for (int k = 0; k < 4; k++) kern<<<blocks, 256>>>(out, 2 + rep);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms / 4 < best) best = ms / 4;
  }
  return best;
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const double ghz = p.clockRate * 1e-6;
  uint32_t* out;
  CHECK(hipMalloc(&out, 64));
  printf("device %s, %d CUs, nominal %.2f GHz; ITER %d; cycles are per wave instruction per SIMD at the NOMINAL clock\n",
         p.name, cus, ghz, ITER);
  for (int wps = 1; wps <= 2; wps++) {
    const int blocks = cus * wps;  // 256 threads = 4 waves = one per SIMD
    const float t_mad = time_kernel(k_mix<4, 0>, blocks, out);
    const float t_m1 = time_kernel(k_mix<0, 1>, blocks, out);
    const float t_m2 = time_kernel(k_mix<0, 2>, blocks, out);
    const float t_m4 = time_kernel(k_mix<0, 4>, blocks, out);
    const float t_x1 = time_kernel(k_mix<4, 1>, blocks, out);
    const float t_x2 = time_kernel(k_mix<4, 2>, blocks, out);
    const float t_x4 = time_kernel(k_mix<4, 4>, blocks, out);
    auto cyc = [&](float ms, int n_per_iter) { return ms * 1e-3 * ghz * 1e9 / ((double)ITER * n_per_iter * wps); };
    printf("waves/SIMD %d\n", wps);
    printf("  mad only (32/iter)       %8.3f ms  %.2f cycles/MAD\n", t_mad, cyc(t_mad, 32));
    printf("  mfma only 1/iter         %8.3f ms  %.1f cycles/MFMA\n", t_m1, cyc(t_m1, 1));
    printf("  mfma only 2/iter         %8.3f ms  %.1f cycles/MFMA\n", t_m2, cyc(t_m2, 2));
    printf("  mfma only 4/iter         %8.3f ms  %.1f cycles/MFMA\n", t_m4, cyc(t_m4, 4));
    printf("  mix 32 MAD + 1 MFMA      %8.3f ms  x%.3f of mad only\n", t_x1, t_x1 / t_mad);
    printf("  mix 32 MAD + 2 MFMA      %8.3f ms  x%.3f of mad only\n", t_x2, t_x2 / t_mad);
    printf("  mix 32 MAD + 4 MFMA      %8.3f ms  x%.3f of mad only\n", t_x4, t_x4 / t_mad);
  }
  printf("wave-specialised, 3 waves/SIMD (one workgroup of 12 waves per CU)\n");
  report_spec<2>(cus, out);
  report_spec<4>(cus, out);
  report_spec<6>(cus, out);
  report_spec<8>(cus, out);
  return 0;
}
