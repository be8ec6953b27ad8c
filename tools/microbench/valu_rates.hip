// VALU instruction-rate microbenchmark for gfx950 (MI355X).
//
// Decides the limb strategy of the Fq multiplier: how many
// SIMD cycles does one wave64 v_mad_u64_u32 / v_mul_lo_u32 / v_mul_hi_u32 /
// v_fma_f64 / 24-bit multiply cost relative to a full-rate v_add_u32?
//
// Every kernel runs ITER iterations of UNROLL x 8 independent chains of the
// instruction under test (inline asm, so nothing is folded), on a grid that
// fills every SIMD with `waves_per_simd` waves.  Output: wave-instructions per
// second chip-wide and the implied cycles per wave-instruction per SIMD at the
// clock measured by s_memtime / s_memrealtime.
//
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                               \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_),        \
              __FILE__, __LINE__);                                             \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

constexpr int ITER = 512;

// 8 independent chains, 4x unrolled => 32 instructions per loop body.
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define DEF_KERNEL_U32(NAME, ASMSTR)                                           \
  __global__ void __launch_bounds__(256) k_##NAME(uint32_t* out, uint32_t s) { \
    uint32_t a[8], b = s * 2654435761u + threadIdx.x, c = s ^ 0x9e3779b9u;     \
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 747796405u + i + s;       \
    for (int it = 0; it < ITER; it++) {                                        \
      _Pragma("unroll") for (int u = 0; u < 4; u++) {                          \
        _Pragma("unroll") for (int i = 0; i < 8; i++) {                        \
          asm volatile(ASMSTR : "+v"(a[i]) : "v"(b), "v"(c));                  \
        }                                                                      \
      }                                                                        \
    }                                                                          \
    uint32_t r = 0;                                                            \
    for (int i = 0; i < 8; i++) r ^= a[i];                                     \
    if (r == 0x12345678u) out[0] = r;                                          \
  }

DEF_KERNEL_U32(add_u32, "v_add_u32 %0, %0, %1")
DEF_KERNEL_U32(add_co, "v_add_co_u32 %0, vcc, %0, %1")
DEF_KERNEL_U32(addc_co, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
DEF_KERNEL_U32(add3, "v_add3_u32 %0, %0, %1, %2")
DEF_KERNEL_U32(mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
DEF_KERNEL_U32(mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
DEF_KERNEL_U32(mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
DEF_KERNEL_U32(mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %1")
DEF_KERNEL_U32(mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
DEF_KERNEL_U32(mad_u32_u16, "v_mad_u32_u16 %0, %0, %1, %2")
DEF_KERNEL_U32(fma_f32, "v_fma_f32 %0, %0, %1, %2")
DEF_KERNEL_U32(alignbit, "v_alignbit_b32 %0, %0, %1, 13")
DEF_KERNEL_U32(lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
DEF_KERNEL_U32(and_or, "v_and_or_b32 %0, %0, %1, %2")
DEF_KERNEL_U32(xad, "v_xad_u32 %0, %0, %1, %2")
DEF_KERNEL_U32(dot2_u32_u16, "v_dot2_u32_u16 %0, %1, %2, %0")
DEF_KERNEL_U32(dot4_u32_u8, "v_dot4_u32_u8 %0, %1, %2, %0")
DEF_KERNEL_U32(pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
DEF_KERNEL_U32(pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
DEF_KERNEL_U32(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
// r02: the cheap VOP1/VOP2 forms (is everything without a carry double rate, or only v_add_u32?)
DEF_KERNEL_U32(and_b32, "v_and_b32 %0, %0, %1")
DEF_KERNEL_U32(or_b32, "v_or_b32 %0, %0, %1")
DEF_KERNEL_U32(xor_b32, "v_xor_b32 %0, %0, %1")
DEF_KERNEL_U32(sub_u32, "v_sub_u32 %0, %0, %1")
DEF_KERNEL_U32(lshrrev_b32, "v_lshrrev_b32 %0, 3, %0")
DEF_KERNEL_U32(lshlrev_b32, "v_lshlrev_b32 %0, 1, %0")
DEF_KERNEL_U32(not_b32, "v_not_b32 %0, %0")
DEF_KERNEL_U32(mov_b32, "v_mov_b32 %0, %1")
DEF_KERNEL_U32(bitop3, "v_bitop3_b32 %0, %0, %1, %0 bitop3:0xc")
DEF_KERNEL_U32(add_u32_nop, "v_add_u32 %0, %0, %1\n s_nop 0")
// selects and quad exchanges (costing a 4-lanes-per-signature group law, HISTORY.md §6)
DEF_KERNEL_U32(cndmask_sgpr, "v_cndmask_b32 %0, %0, %1, s[10:11]")
DEF_KERNEL_U32(mov_dpp_quad, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")

#define DEF_KERNEL_U64(NAME, ASMSTR)                                           \
  __global__ void __launch_bounds__(256) k_##NAME(uint32_t* out, uint32_t s) { \
    uint64_t a[8];                                                             \
    uint32_t b = s * 2654435761u + threadIdx.x, c = s ^ 0x9e3779b9u;           \
    uint64_t d = ((uint64_t)b << 32) | c;                                      \
    for (int i = 0; i < 8; i++)                                                \
      a[i] = (uint64_t)(threadIdx.x * 747796405u + i + s) << 11;               \
    for (int it = 0; it < ITER; it++) {                                        \
      _Pragma("unroll") for (int u = 0; u < 4; u++) {                          \
        _Pragma("unroll") for (int i = 0; i < 8; i++) {                        \
          asm volatile(ASMSTR : "+v"(a[i]) : "v"(b), "v"(c), "v"(d));          \
        }                                                                      \
      }                                                                        \
    }                                                                          \
    uint64_t r = 0;                                                            \
    for (int i = 0; i < 8; i++) r ^= a[i];                                     \
    if (r == 0x12345678u) out[0] = (uint32_t)r;                                \
  }

DEF_KERNEL_U64(mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
DEF_KERNEL_U64(mad_u64_u32_sgprcarry, "v_mad_u64_u32 %0, s[10:11], %1, %2, %0")
DEF_KERNEL_U64(lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %3")
DEF_KERNEL_U64(lshlrev_b64, "v_lshlrev_b64 %0, 1, %0")
DEF_KERNEL_U64(lshrrev_b64, "v_lshrrev_b64 %0, 29, %0")
// what a hazard no-op between two dependent MADs costs (inline-asm MAD chains get one each)
DEF_KERNEL_U64(mad_u64_u32_nop, "v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n s_nop 0")
DEF_KERNEL_U64(fma_f64, "v_fma_f64 %0, %0, %3, %3")
DEF_KERNEL_U64(add_f64, "v_add_f64 %0, %0, %3")
DEF_KERNEL_U64(mul_f64, "v_mul_f64 %0, %0, %3")
DEF_KERNEL_U64(pk_fma_f32, "v_pk_fma_f32 %0, %0, %3, %3")
DEF_KERNEL_U64(pk_add_f32, "v_pk_add_f32 %0, %0, %3")
DEF_KERNEL_U64(cvt_f64_u32, "v_cvt_f64_u32 %0, %1")
DEF_KERNEL_U64(pk_mul_f32, "v_pk_mul_f32 %0, %0, %3")

// mad_u64_u32 followed by a dependent addc (the Comba column pattern)
__global__ void __launch_bounds__(256) k_mad_addc_pair(uint32_t* out, uint32_t s) {
  uint64_t a[8];
  uint32_t hi[8];
  uint32_t b = s * 2654435761u + threadIdx.x, c = s ^ 0x9e3779b9u;
  for (int i = 0; i < 8; i++) {
    a[i] = (uint64_t)(threadIdx.x * 747796405u + i + s) << 11;
    hi[i] = i;
  }
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
                     "v_addc_co_u32 %1, vcc, 0, %1, vcc"
                     : "+v"(a[i]), "+v"(hi[i])
                     : "v"(b), "v"(c)
                     : "vcc");
      }
    }
  }
  uint64_t r = 0;
  for (int i = 0; i < 8; i++) r ^= a[i] + hi[i];
  if (r == 0x12345678u) out[0] = (uint32_t)r;
}

// single dependent chain: latency of v_mad_u64_u32 (one wave per SIMD)
__global__ void __launch_bounds__(256) k_mad_u64_u32_chain(uint32_t* out, uint32_t s) {
  uint64_t a = (uint64_t)(threadIdx.x * 747796405u + s) << 11;
  uint32_t b = s * 2654435761u + threadIdx.x, c = s ^ 0x9e3779b9u;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int u = 0; u < 32; u++)
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c) : "vcc");
  }
  if (a == 0x12345678u) out[0] = (uint32_t)a;
}

__global__ void k_clock(unsigned long long* out) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  // ~ spin
  uint32_t x = threadIdx.x;
  for (int i = 0; i < 2000000; i++) asm volatile("v_add_u32 %0, %0, %0" : "+v"(x));
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[0] = t1 - t0;
    out[1] = r1 - r0;
  }
  if (x == 0x1234567) out[2] = x;
}

typedef void (*kern_t)(uint32_t*, uint32_t);

struct Entry {
  const char* name;
  kern_t k;
  int insts_per_body;  // wave-instructions of interest per loop body
};

int main(int argc, char** argv) {
  int dev = 0;
  CHECK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, dev));
  int cus = prop.multiProcessorCount;
  printf("# device %s, %d CUs, clockRate %d kHz\n", prop.name, cus, prop.clockRate);

  uint32_t* d_out;
  CHECK(hipMalloc(&d_out, 4096));
  unsigned long long* d_clk;
  CHECK(hipMalloc(&d_clk, 64));

  // measure shader clock
  hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, d_clk);
  CHECK(hipDeviceSynchronize());
  unsigned long long h_clk[2];
  CHECK(hipMemcpy(h_clk, d_clk, 16, hipMemcpyDeviceToHost));
  double ghz = (double)h_clk[0] / (double)h_clk[1] * 0.1;  // realtime = 100 MHz
  printf("# in-kernel shader clock (idle chip, 1 wave): %.3f GHz\n", ghz);

  std::vector<Entry> es = {
#define E(n) {#n, k_##n, 32}
      E(add_u32), E(add_co), E(addc_co), E(add3), E(mul_lo_u32), E(mul_hi_u32),
      E(mul_u32_u24), E(mul_hi_u32_u24), E(mad_u32_u24), E(mad_u32_u16),
      E(fma_f32), E(alignbit), E(lshl_add), E(and_or), E(xad), E(dot2_u32_u16),
      E(dot4_u32_u8), E(pk_mul_lo_u16), E(pk_mad_u16), E(cndmask),
      E(and_b32), E(or_b32), E(xor_b32), E(sub_u32), E(lshrrev_b32), E(lshlrev_b32), E(not_b32),
      E(mov_b32), E(bitop3), E(add_u32_nop), E(cndmask_sgpr), E(mov_dpp_quad),
      E(mad_u64_u32), E(mad_u64_u32_sgprcarry), E(lshl_add_u64), E(lshlrev_b64), E(lshrrev_b64),
      E(mad_u64_u32_nop),
      E(fma_f64), E(add_f64), E(mul_f64), E(pk_fma_f32), E(pk_add_f32),
      E(cvt_f64_u32), E(pk_mul_f32),
      {"mad_addc_pair", k_mad_addc_pair, 32},
      {"mad_u64_u32_chain", k_mad_u64_u32_chain, 32},
  };

  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));

  printf("%-24s %6s %12s %14s %10s\n", "instr", "w/SIMD", "ms", "Gwaveinst/s",
         "cyc/inst");
  for (int wps : {1, 2, 4}) {
    for (auto& e : es) {
      // 256-thread blocks = 4 waves = one per SIMD; wps blocks per CU; x8 rounds
      int blocks = cus * wps * 8;
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d_out, 1u);  // warm
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      for (int r = 0; r < 3; r++)
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      ms /= 3;
      double waveinsts = (double)blocks * 4 * ITER * e.insts_per_body;
      double rate = waveinsts / (ms * 1e-3);
      // cycles per wave-instruction per SIMD at 2.4 GHz nominal
      double simd_cycles = (double)cus * 4 * 2.4e9 * (ms * 1e-3);
      printf("%-24s %6d %12.4f %14.2f %10.2f\n", e.name, wps, ms, rate * 1e-9,
             simd_cycles / waveinsts);
    }
  }
  return 0;
}
