// Operand layout probe for v_mfma_i32_32x32x32_i8 on gfx950: runs one MFMA on random signed bytes
// and checks the result against a host matrix product under the candidate (lane, byte) -> (row, k)
// maps.  Prints which hypothesis matches.
//
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_layout mfma_layout.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void k_one(const int* a, const int* b, int* c) {
  const int l = threadIdx.x;
  v4i va = {a[4 * l], a[4 * l + 1], a[4 * l + 2], a[4 * l + 3]};
  v4i vb = {b[4 * l], b[4 * l + 1], b[4 * l + 2], b[4 * l + 3]};
  v16i acc = {0};
  acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(va, vb, acc, 0, 0, 0);
  for (int i = 0; i < 16; i++) c[16 * l + i] = acc[i];
}

static int kmap(int hyp, int lane, int byte) {
  if (hyp == 0) return 16 * (lane / 32) + byte;                      // 16 consecutive k per lane half
  return 8 * (lane / 32) + (byte % 8) + 16 * (byte / 8);             // two 32x32x16 steps back to back
}

int main() {
  int8_t A[64][16], B[64][16];
  srand(12345);
  for (int l = 0; l < 64; l++)
    for (int j = 0; j < 16; j++) {
      A[l][j] = (int8_t)(rand() % 256 - 128);
      B[l][j] = (int8_t)(rand() % 256 - 128);
    }
  int *da, *db, *dc;
  hipMalloc(&da, 1024);
  hipMalloc(&db, 1024);
  hipMalloc(&dc, 4096);
  hipMemcpy(da, A, 1024, hipMemcpyHostToDevice);
  hipMemcpy(db, B, 1024, hipMemcpyHostToDevice);
  k_one<<<1, 64>>>(da, db, dc);
  int C[64][16];
  if (hipMemcpy(C, dc, 4096, hipMemcpyDeviceToHost) != hipSuccess) {
    printf("HIP error\n");
    return 1;
  }
  for (int hyp = 0; hyp < 2; hyp++) {
    // dense matrices under the hypothesis
    int Am[32][32], Bm[32][32];
    for (int l = 0; l < 64; l++)
      for (int j = 0; j < 16; j++) {
        Am[l % 32][kmap(hyp, l, j)] = A[l][j];
        Bm[kmap(hyp, l, j)][l % 32] = B[l][j];
      }
    int bad = 0;
    for (int l = 0; l < 64; l++)
      for (int i = 0; i < 16; i++) {
        const int row = 8 * (i / 4) + 4 * (l / 32) + (i % 4), col = l % 32;
        int s = 0;
        for (int k = 0; k < 32; k++) s += Am[row][k] * Bm[k][col];
        if (s != C[l][i]) bad++;
      }
    printf("hypothesis %d (k = %s): %d mismatches of 1024\n", hyp,
           hyp == 0 ? "16*(lane/32)+byte" : "8*(lane/32)+byte%8+16*(byte/8)", bad);
  }
  return 0;
}
