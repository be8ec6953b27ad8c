// Wave-cooperative vs lane-per-element Fq multiplication on gfx950 (SURVEY.md §7 "prototype both
// on the modmul microbenchmark and keep the faster").
//
//   per-lane : the product build's fe_mul (schnorr_amd/csrc/fe29.h) — one element per lane, 9 limbs
//              in 9 VGPRs, 153 v_mad_u64_u32 + 36 other instructions, no cross-lane traffic.
//   coop     : one element per 16-lane DPP row, limb j of every operand in lane j of the row
//              ("limbs across lanes").  Lane k accumulates column k of a*b + M*q in one 64-bit
//              register; operands reach it through ds_bpermute broadcasts / rotations
//              (__shfl width 16), the Montgomery digits are broadcast one after the other, the
//              carries travel lane to lane.  Same R = 2^261 representation, results checked
//              against the per-lane multiplier bit for bit (mod q).
//
// Each kernel runs a dependent chain x <- x * y of CHAIN multiplications, so the figure is both a
// throughput (multiplications / s, chip-wide at 1 and 2 waves per SIMD) and, with ONE wave on the
// chip, a latency per multiplication.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o coop_mul coop_mul.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../schnorr_amd/csrc/fe29.h"

using namespace dsv;

#define CHECK(x)                                                                                  \
  do {                                                                                            \
    hipError_t e_ = (x);                                                                          \
    if (e_ != hipSuccess) {                                                                       \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);      \
      exit(1);                                                                                    \
    }                                                                                             \
  } while (0)

constexpr int CHAIN = 256;

__global__ void __launch_bounds__(256) k_lane(const Fe* __restrict__ in, Fe* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  Fe x = in[2 * i], y = in[2 * i + 1];
#pragma unroll 1
  for (int k = 0; k < CHAIN; k++) x = fe_mul(x, y);
  out[i] = x;
}

// ---- cooperative multiplication: 16 lanes per element -----------------------------------------
// in : a, b = this lane's limb of the operands (lanes 9..15 of a row hold 0), limbs < 2^29 + 8
// out: this lane's limb of a*b*2^-261 (mod q, lazily reduced: limbs < 2^31), lanes 9..15: 0
__device__ __forceinline__ u32 coop_mul(u32 a, u32 b, const u32 (&qrot)[NL]) {
  const int k = threadIdx.x & 15;  // column of this lane (and limb index of its operands)
  // columns 0..15 live in lanes 0..15; column 16 = a8*b8 + (digit products) is kept by lane 0 in
  // a second accumulator
  u64 col = 0, col16 = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const u32 ai = __shfl(a, i, 16);                    // broadcast limb i of a over the row
    const int j = k - i;                                // limb of b that meets a_i in column k
    u32 bj = __shfl(b, j & 15, 16);
    bj = (j >= 0 && j < NL) ? bj : 0u;
    col += (u64)ai * bj;
    if (i == NL - 1) col16 = (u64)ai * __shfl(b, NL - 1, 16);  // a8 * b8 (used by lane 0 only)
  }
  // Montgomery reduction, digit by digit: column i (lane i) is cleared, its carry moves to lane
  // i + 1, every lane adds m_i * q[k - i] (qrot[i] = q[k - i] or 0, constant per lane)
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const u32 lo = __shfl((u32)col, i, 16);
    const u32 m = (0u - lo) & M29;                      // -col_i mod 2^29  (q = 1 mod 2^29)
    col += (u64)m * qrot[i];                            // lane i: += m (q[0] = 1) -> low 29 bits 0
    if (i == NL - 1) col16 += (u64)m * kQ29[NL - 1];    // m_8 * q[8] belongs to column 16
    // carry of column i into column i + 1
    const u32 clo = __shfl((u32)(col >> 29), i, 16);
    const u32 chi = __shfl((u32)(col >> 61), i, 16);
    if (k == i + 1) col += ((u64)chi << 32) | clo;
    if (k == i) col = 0;
  }
  // digit products for columns >= 9 were added with the loop above (qrot covers k - i in 0..8);
  // column 16 receives only a8*b8 plus carries.  Split every column in 29-bit pieces and hand the
  // upper pieces to the next two lanes (carry-save): limb_k = lo(col_k) + mid(col_k-1) + hi(col_k-2)
  const u32 p0 = (u32)col & M29, p1 = (u32)(col >> 29) & M29, p2 = (u32)(col >> 58);
  u32 up1 = __shfl_up(p1, 1, 16), up2 = __shfl_up(p2, 2, 16);
  if (k < 1) up1 = 0;
  if (k < 2) up2 = 0;
  u32 limb = p0 + up1 + up2;                            // columns 9..15 -> result limbs 0..6
  // result limb j sits in lane j + 9; limbs 7, 8 come from column 16 and the tail pieces
  const u32 t15p1 = __shfl(p1, 15, 16), t15p2 = __shfl(p2, 15, 16), t14p2 = __shfl(p2, 14, 16);
  const u32 c16 = __shfl((u32)col16 & M29, 0, 16), c16h = __shfl((u32)(col16 >> 29), 0, 16);
  u32 r = __shfl(limb, (k + 9) & 15, 16);
  if (k == 7) r = c16 + t15p1 + t14p2;
  if (k == 8) r = c16h + t15p2;
  return k < NL ? r : 0u;
}

__global__ void __launch_bounds__(256) k_coop(const u32* __restrict__ in, u32* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int k = threadIdx.x & 15;
  u32 qrot[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) qrot[i] = (k - i >= 0 && k - i < NL) ? kQ29[k - i] : 0u;
  u32 x = in[2 * t], y = in[2 * t + 1];
#pragma unroll 1
  for (int c = 0; c < CHAIN; c++) {
    x = coop_mul(x, y, qrot);
    // keep limbs below 2^29 + 8 for the next product: one carry step across lanes
    const u32 up = __shfl_up(x >> 29, 1, 16);
    x = (k == 8 ? x : (x & M29)) + ((k >= 1 && k < NL) ? up : 0u);
  }
  out[t] = x;
}

static unsigned __int128 val(const u32* l) {  // low 128 bits are enough to compare two lazily
  unsigned __int128 v = 0;                   // reduced values after canonical reduction below
  for (int i = 4; i >= 0; i--) v = (v << 29) + l[i];
  return v;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("# %s, %d CUs; chain of %d dependent multiplications per element\n", prop.name, cus, CHAIN);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("%-10s %7s %12s %10s %14s %16s\n", "kernel", "waves", "elements", "ms", "Gmul/s", "ns per mul (1 elem)");
  for (int mode = 0; mode < 2; mode++) {
    for (int wps : {0, 1, 2}) {  // 0: ONE wave on the whole chip (latency)
      const int blocks = wps == 0 ? 1 : cus * wps;
      const int threads = wps == 0 ? 64 : 256;
      const size_t lanes = (size_t)blocks * threads;
      const size_t elems = mode == 0 ? lanes : lanes / 16;
      std::vector<u32> h(2 * lanes * (mode == 0 ? NL : 1));
      srand(7);
      if (mode == 0) {
        for (size_t i = 0; i < 2 * lanes; i++)
          for (int j = 0; j < NL; j++) h[i * NL + j] = (u32)rand() & (j == NL - 1 ? 0x3fffffu : M29);
      } else {
        for (size_t t = 0; t < lanes; t++)
          for (int w = 0; w < 2; w++) {
            const int j = t & 15;
            h[2 * t + w] = j < NL ? ((u32)rand() & (j == NL - 1 ? 0x3fffffu : M29)) : 0u;
          }
      }
      u32 *din, *dout;
      CHECK(hipMalloc(&din, h.size() * 4));
      CHECK(hipMalloc(&dout, lanes * (mode == 0 ? sizeof(Fe) : 4)));
      CHECK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice));
      auto launch = [&] {
        if (mode == 0)
          hipLaunchKernelGGL(k_lane, dim3(blocks), dim3(threads), 0, 0, (const Fe*)din, (Fe*)dout);
        else
          hipLaunchKernelGGL(k_coop, dim3(blocks), dim3(threads), 0, 0, (const u32*)din, dout);
      };
      launch();
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      for (int r = 0; r < 5; r++) launch();
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      ms /= 5;
      printf("%-10s %7s %12zu %10.4f %14.3f %16.1f\n", mode == 0 ? "per-lane" : "coop16",
             wps == 0 ? "1 wave" : (wps == 1 ? "1/SIMD" : "2/SIMD"), elems, ms,
             (double)elems * CHAIN / (ms * 1e-3) * 1e-9, ms * 1e6 / CHAIN);
      CHECK(hipFree(din));
      CHECK(hipFree(dout));
    }
  }
  // ---- correctness of the cooperative multiplier: same inputs through both, compare mod q
  {
    const int n = 64;  // 64 elements: one wave per-lane, 16 waves cooperative
    std::vector<u32> a(n * NL), b(n * NL);
    srand(11);
    for (auto& v : a) v = (u32)rand() & M29;
    for (auto& v : b) v = (u32)rand() & M29;
    for (int i = 0; i < n; i++) a[i * NL + 8] &= 0x3fffff, b[i * NL + 8] &= 0x3fffff;
    std::vector<u32> hl(2 * n * NL), hc(2 * n * 16);
    for (int i = 0; i < n; i++)
      for (int j = 0; j < NL; j++) {
        hl[(2 * i) * NL + j] = a[i * NL + j];
        hl[(2 * i + 1) * NL + j] = b[i * NL + j];
      }
    for (int i = 0; i < n; i++)
      for (int j = 0; j < 16; j++) {
        hc[2 * (i * 16 + j)] = j < NL ? a[i * NL + j] : 0;
        hc[2 * (i * 16 + j) + 1] = j < NL ? b[i * NL + j] : 0;
      }
    u32 *d1, *o1, *d2, *o2;
    CHECK(hipMalloc(&d1, hl.size() * 4));
    CHECK(hipMalloc(&o1, n * sizeof(Fe)));
    CHECK(hipMalloc(&d2, hc.size() * 4));
    CHECK(hipMalloc(&o2, n * 16 * 4));
    CHECK(hipMemcpy(d1, hl.data(), hl.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d2, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_lane, dim3(1), dim3(64), 0, 0, (const Fe*)d1, (Fe*)o1);
    hipLaunchKernelGGL(k_coop, dim3(4), dim3(256), 0, 0, (const u32*)d2, o2);
    CHECK(hipDeviceSynchronize());
    std::vector<u32> r1(n * NL), r2(n * 16);
    CHECK(hipMemcpy(r1.data(), o1, r1.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(r2.data(), o2, r2.size() * 4, hipMemcpyDeviceToHost));
    // compare as integers mod q using Python-free arithmetic: reduce both to canonical via repeated
    // subtraction is overkill here — the chain ends in values < 2 q, so compare v and v +- q
    static const u32 q[NL] = DSV_Q29;
    int bad = 0;
    for (int i = 0; i < n; i++) {
      // signed limb-wise difference, then check it is 0, +q or -q after carry propagation
      long long d[NL];
      for (int j = 0; j < NL; j++) d[j] = (long long)r1[i * NL + j] - (long long)r2[i * 16 + j];
      bool ok = false;
      for (int s = -3; s <= 3 && !ok; s++) {
        long long c = 0;
        bool zero = true;
        for (int j = 0; j < NL; j++) {
          long long t = d[j] - (long long)s * q[j] + c;
          c = t >> 29;
          if (j < NL - 1 ? (t & M29) != 0 : t != 0) zero = false;
          if (j == NL - 1) c = 0;
        }
        ok = zero;
      }
      if (!ok) bad++;
    }
    printf("# cooperative result == per-lane result (mod q) on %d / %d elements\n", n - bad, n);
    (void)val;
  }
  return 0;
}
