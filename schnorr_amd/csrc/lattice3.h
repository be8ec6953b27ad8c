// lattice3.h — three short scalars for the var-generator equation, exact on the WHOLE curve group.
//
// `PublicKeyVarGen::verify` (/root/reference/src/keys/public.rs:401-415) checks
//        Q := u*Gen + c*PK - R == O
// with a 252-bit u and a 250-bit c, both on variable bases: 252 doublings.  For any integers
// (x, y, z) with  x = z*u (mod 8r),  y = z*c (mod 8r),  z odd,  0 < |z| < r:
//        z*Q = x*Gen + y*PK - z*R          and          z*Q == O  <=>  Q == O.
// Proof.  Every point of E(Fq) = Z_r x E[8] is killed by 8r, so (z*u)*Gen = x*Gen and (z*c)*PK =
// y*PK whenever the scalars agree modulo 8r — with or without a small-order component in Gen, PK
// (the reference's types can hold such points).  z*R uses the integer z itself.  Multiplication by
// z is injective: on Z_r because 0 < |z| < r, on the 2-group E[8] because z is odd (same argument
// as halfgcd.h, one dimension up).
// Such triples form the lattice spanned by (8r, 0, 0), (0, 8r, 0), (u, c, 1), of determinant
// (8r)^2 = 2^510: its short vectors have components of ~170 bits, so the two-base chain of 63
// windows becomes a three-base chain of ~43: 80 doublings fewer, 3 additions more, one table more.
//
// Reduction: greedy pairwise (Lagrange) reduction of the three basis vectors, Lehmer style.  The
// basis lives as exact 288-bit two's-complement integers; a BATCH works on double-precision images
// only: passes over the six ordered pairs (i, j), b_i -= round(<b_i, b_j> / <b_j, b_j>) * b_j, the
// accumulated unimodular transformation T kept alongside (entries < 2^31, exact in a double); when
// an entry of T would pass 2^31 — the images have lost ~30 of their 53 bits by then — or nothing
// changes any more, T is applied to the exact basis (27 products of a 32-bit by a 288-bit integer)
// and new images are taken.  ~6 batches, ~63 passes per signature (tests/pymodel.py: lattice3 is
// the same algorithm on Python integers and floats).
// Exactness does not depend on any of the floating-point decisions: every vector ever held is an
// INTEGER combination of lattice vectors, so the congruences hold by construction; the kernel only
// needs SOME lattice vector with an odd z that fits its chain, and falls back to (u, c, 1) — the
// reference's own equation — if the reduction did not produce one below 2^251.
#pragma once
#include "fe29.h"
#include "halfgcd.h"

namespace dsv {

constexpr int kLatWords = 9;          // 288-bit two's complement
// -DDSV_LAT_MAX_BATCHES=0 builds the kernel WITHOUT any reduction: every item then takes the fallback
// row (u, c, 1), i.e. the reference equation as a 63-window chain — the test build
// schnorr_amd/libdsv_lat0.so (build.py: VARIANTS), which is how that branch gets exercised at all:
// no hash output steers a challenge there
#ifndef DSV_LAT_MAX_BATCHES
#define DSV_LAT_MAX_BATCHES 24
#endif
constexpr int kLatMaxBatches = DSV_LAT_MAX_BATCHES;
constexpr int kLatMaxPasses = 40;     // per batch

// -x
DSV_DEV void lat_neg(u32 (&o)[kLatWords], const u32 (&x)[kLatWords]) {
  u32 borrow = 0;
#pragma unroll
  for (int i = 0; i < kLatWords; i++) {
    const u64 t = (u64)0 - x[i] - borrow;
    o[i] = (u32)t;
    borrow = (u32)(t >> 63);
  }
}
// signed 288-bit -> double (relative error < 2^-50).  Two's complement = signed top word * 2^256 +
// the unsigned words below it: Horner from the top needs no negation, and stays exact while the
// partial value is small (a small negative number is -1, -1 * 2^32 + 0xffffffff = -1, ...).
DSV_DEV double lat_to_double(const u32 (&x)[kLatWords]) {
  double d = (double)(int)x[kLatWords - 1];
#pragma unroll
  for (int i = kLatWords - 2; i >= 0; i--) d = __builtin_fma(d, 4294967296.0, (double)x[i]);
  return d;
}
// acc += t * x (mod 2^288), t a signed 32-bit integer given as magnitude and sign; nx = -x
DSV_DEV void lat_mul_acc(u32 (&acc)[kLatWords], u32 mag, bool neg, const u32 (&x)[kLatWords],
                         const u32 (&nx)[kLatWords]) {
  u64 carry = 0;
#pragma unroll
  for (int i = 0; i < kLatWords; i++) {
    const u32 xi = neg ? nx[i] : x[i];
    // mag * xi + acc[i] + carry <= (2^32-1)^2 + 2 (2^32-1) = 2^64 - 1
    const u64 t = (u64)mag * xi + acc[i] + carry;
    acc[i] = (u32)t;
    carry = t >> 32;
  }
}
DSV_DEV int lat_bitlen_mag(u32 (&mag)[8], bool& neg, const u32 (&x)[kLatWords], bool& fits) {
  neg = (x[kLatWords - 1] >> 31) != 0;
  u32 n[kLatWords];
  lat_neg(n, x);
#pragma unroll
  for (int i = 0; i < 8; i++) mag[i] = neg ? n[i] : x[i];
  fits = (neg ? n[8] : x[8]) == 0;
  return bitlen8(mag);
}

// out: magnitudes (8 words each) and signs of (x, y, z); z odd, z != 0.  A reduced row is only taken
// when all three components are below 2^251; the fallback row (u, c, 1) has x = u < 2^252 — the bound
// the chain's recoding (recode_signed4: exact below 2^252) and its 63-window top digit are built for.
// u < 2^252 (callers mask a non-canonical u), c < 2^250.
DSV_DEV void lattice3_scalars(u32 (&mx)[8], u32 (&my)[8], u32 (&mz)[8], bool& nx, bool& ny, bool& nz,
                              const u32 (&u)[8], const u32 (&c)[8]) {
  // B[row][component][word]
  u32 B[3][3][kLatWords];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
      for (int w = 0; w < kLatWords; w++) B[r][m][w] = 0;
#pragma unroll
  for (int w = 0; w < 8; w++) {
    B[0][0][w] = kN8R[w];
    B[1][1][w] = kN8R[w];
    B[2][0][w] = u[w];
    B[2][1][w] = c[w];
  }
  B[2][2][0] = 1;
  bool done = false;
#pragma unroll 1
  for (int batch = 0; batch < kLatMaxBatches && !done; batch++) {
    double D[3][3], T[3][3];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int m = 0; m < 3; m++) {
        D[r][m] = lat_to_double(B[r][m]);
        T[r][m] = r == m ? 1.0 : 0.0;
      }
    bool any = false, stop = false, more = true, converged = false;
#pragma unroll 1
    for (int pass = 0; pass < kLatMaxPasses && more; pass++) {
      bool changed = false;
#pragma unroll
      for (int p = 0; p < 6; p++) {
        constexpr int PI[6] = {0, 0, 1, 1, 2, 2}, PJ[6] = {1, 2, 2, 0, 0, 1};
        const int i = PI[p], j = PJ[p];
        const double njj = D[j][0] * D[j][0] + D[j][1] * D[j][1] + D[j][2] * D[j][2];
        const double dij = D[i][0] * D[j][0] + D[i][1] * D[j][1] + D[i][2] * D[j][2];
        // v_rcp_f64 (relative error ~2^-26, one instruction) instead of an IEEE division (~12, five of
        // them quarter-rate): a quotient that is off by one only makes the step a little less greedy
        double q = njj > 0.0 ? __builtin_rint(dij * __builtin_amdgcn_rcp(njj)) : 0.0;
        const double t0 = T[i][0] - q * T[j][0], t1 = T[i][1] - q * T[j][1], t2 = T[i][2] - q * T[j][2];
        const double tm = __builtin_fmax(__builtin_fabs(t0), __builtin_fmax(__builtin_fabs(t1), __builtin_fabs(t2)));
        // (a NaN / infinite q — images of a degenerate basis — fails this comparison and ends the batch)
        const bool ok = (tm < 2147483648.0) & !stop;
        const bool want = q != 0.0;
        stop |= want & !ok;
        if (want & ok) {
          T[i][0] = t0;
          T[i][1] = t1;
          T[i][2] = t2;
          D[i][0] -= q * D[j][0];
          D[i][1] -= q * D[j][1];
          D[i][2] -= q * D[j][2];
          changed = true;
        }
      }
      any |= changed;
      more = changed & !stop;
      converged = !changed & !stop;  // a whole pass without a step: pairwise reduced (to the images' precision)
    }
    if (!any) {
      done = true;
    } else {
      // B <- T * B, one component (column) at a time: new[i] = sum_k T[i][k] * old[k]
      int ti[3][3];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int k = 0; k < 3; k++) ti[r][k] = (int)T[r][k];
#pragma unroll
      for (int m = 0; m < 3; m++) {
        u32 old[3][kLatWords], nold[3][kLatWords];
#pragma unroll
        for (int k = 0; k < 3; k++) {
#pragma unroll
          for (int w = 0; w < kLatWords; w++) old[k][w] = B[k][m][w];
          lat_neg(nold[k], old[k]);
        }
#pragma unroll
        for (int r = 0; r < 3; r++) {
          u32 acc[kLatWords];
#pragma unroll
          for (int w = 0; w < kLatWords; w++) acc[w] = 0;
#pragma unroll
          for (int k = 0; k < 3; k++) {
            const int t = ti[r][k];
            lat_mul_acc(acc, (u32)(t < 0 ? -t : t), t < 0, old[k], nold[k]);
          }
#pragma unroll
          for (int w = 0; w < kLatWords; w++) B[r][m][w] = acc[w];
        }
      }
      done = converged;  // (no further batch just to find that nothing changes)
    }
  }
  // the shortest row (by its longest component) with an odd z that fits the chain; else (u, c, 1)
  int best = 252;
#pragma unroll
  for (int w = 0; w < 8; w++) {
    mx[w] = u[w];
    my[w] = c[w];
    mz[w] = w == 0 ? 1u : 0u;
  }
  nx = ny = nz = false;
#pragma unroll
  for (int r = 0; r < 3; r++) {
    u32 a[8], b[8], d[8];
    bool na, nb, nd, fa, fb, fd;
    const int la = lat_bitlen_mag(a, na, B[r][0], fa);
    const int lb = lat_bitlen_mag(b, nb, B[r][1], fb);
    const int ld = lat_bitlen_mag(d, nd, B[r][2], fd);
    const int len = la > lb ? (la > ld ? la : ld) : (lb > ld ? lb : ld);
    const bool take = fa & fb & fd & ((d[0] & 1u) != 0) & (len < best);
    if (take) {
      best = len;
#pragma unroll
      for (int w = 0; w < 8; w++) {
        mx[w] = a[w];
        my[w] = b[w];
        mz[w] = d[w];
      }
      nx = na;
      ny = nb;
      nz = nd;
    }
  }
}

}  // namespace dsv
