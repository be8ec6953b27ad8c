// k_misc.hip — everything around the two hot kernels: fixed-base table construction (init), on-device
// `to_hash_inputs`, signing / key derivation, wire-format decoding, the reference harness's input
// generator, and the device-side split of a mixed batch by kind.
#include "common.h"
#include "decode29.h"
#include "inv29.h"
#include "stdrng.h"

namespace dsv {

// ------------------------------------------------------------------------------------------
// init: fixed-base tables.  table[w][d] = affine niels of (d * 2^(kFixedBits*w)) * Gen, d = 0 ..
// 2^(kFixedBits-1), canonical limbs (+ the negated 2d*uv).
// ------------------------------------------------------------------------------------------
// generic double-and-add over a 256-bit LE scalar (init-time table construction only)
DSV_DEV Ext ext_mul_words(const Ext& p, const u32 (&s)[8]) {
  Niels n = ext_to_niels(p);
  Niels id = niels_identity();
  Ext acc = ext_identity();
#pragma unroll 1
  for (int bit = 255; bit >= 0; bit--) {
    acc = ext_double(acc);
    bool b = (s[bit >> 5] >> (bit & 31)) & 1;
    Niels sel;
    sel.vpu = fe_select(b, n.vpu, id.vpu);
    sel.vmu = fe_select(b, n.vmu, id.vmu);
    sel.z = fe_select(b, n.z, id.z);
    sel.t2d = fe_select(b, n.t2d, id.t2d);
    acc = ext_add_niels(acc, sel);
  }
  return acc;
}
__global__ void __launch_bounds__(64) k_build_fixed_table(u32* __restrict__ table, int which) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= kFixedWindows * kFixedEntries) return;
  const int w = idx / kFixedEntries, d = idx % kFixedEntries;
  const u32 gu[NL] = DSV_GEN_U, gv[NL] = DSV_GEN_V, nu[NL] = DSV_GENN_U, nv[NL] = DSV_GENN_V;
  Ext g = which == 0 ? ext_from_affine(fe_const(gu), fe_const(gv))
                     : ext_from_affine(fe_const(nu), fe_const(nv));
  // scalar = d << (bits * w), d <= 2^(bits-1) <= 2^15: at most two words
  u32 s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // (entries whose scalar would not fit 256 bits are never looked up: a digit there is 0 or 1
  //  and 1 << pos < 2^253)
  const int pos = kFixedBits * w, wi = pos >> 5, sh = pos & 31;
  const u64 v = (u64)(u32)d << sh;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    if (i == wi) s[i] = (u32)v;
    if (i == wi + 1) s[i] = (u32)(v >> 32);
  }
  Ext p = ext_mul_words(g, s);
  Fe zi = fe_invert(p.z);
  Fe u = fe_mul(p.u, zi), v2 = fe_mul(p.v, zi);
  Fe vpu = fe_canon(fe_add(v2, u));
  Fe vmu = fe_canon(fe_sub2(v2, u));
  Fe t2d = fe_mul(fe_mul(u, v2), fe_const(kD2));
  Fe nt2d = fe_canon(fe_neg2(t2d));
  t2d = fe_canon(t2d);
  u32* e = table + (size_t)idx * kEntryWords;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    e[i] = vpu.l[i];
    e[NL + i] = vmu.l[i];
    e[2 * NL + i] = t2d.l[i];
    e[3 * NL + i] = nt2d.l[i];
  }
}

// ------------------------------------------------------------------------------------------
// `JubJubExtended::to_hash_inputs` (call sites /root/reference/src/signatures.rs:131, :280-281) for
// callers that hold un-normalised points: (u, v, z) -> (u/z, v/z), canonical bytes.  Montgomery's
// trick over ALL the z's a lane sees — the NP points of an item and the `per_lane` items of a lane
// (items t, t + T, t + 2T, ...: coalesced) — so the Fermat inversion (255 squarings + 91
// multiplications) is paid once per lane, not once per point: 7 multiplications per point +
// 346 / per_lane per item.  The running products on the way up are parked in `prefix`
// ([slot][lane][9 words]); the z's are simply read again on the way down.
// valid[i] = every coordinate of item i canonical (< q) and every z != 0; an offending z is
// replaced by 1 in the product so that it cannot poison the other items of its lane.
// ------------------------------------------------------------------------------------------
DSV_DEV Fe load_z_for_product(const uint8_t* in, size_t i, bool& ok) {
  u32 w[8];
  load_words8(w, in, 3 * i + 2);
  const bool usable = words_lt(w, kQ32) & ((w[0] | w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7]) != 0);
  ok &= usable;
  const Fe z = fe_to_mont(fe_from_words_plain(w));
  return fe_select(usable, z, fe_one());
}
// ------------------------------------------------------------------------------------------
// The reference's in-memory scalars.  `BlsScalar` (dusk-bls12_381 0.13) and `JubJubScalar`
// (dusk-jubjub 0.14; /root/reference/Cargo.toml:25-26) hold `[u64; 4]` MONTGOMERY limbs, R = 2^256:
// the `u` of a Signature (/root/reference/src/signatures.rs:58-61) is u * 2^256 mod r, the message
// handed to verify (/root/reference/src/keys/public.rs:121) is m * 2^256 mod q.  The *_mont entry
// points take those limbs as they lie in memory — the caller runs no `to_bytes()`, i.e. no Montgomery
// reduction, on the host — and the device performs the two reductions: u by one CIOS pass against 1
// (fr.h), m as the fe29 product M * 2^5 * 2^-261.  Limbs that are not below the modulus (the Rust types
// cannot hold them) are poisoned, so the verify kernels give the item verdict 0.  The POINTS of such a
// caller need no conversion: (u R, v R, z R) normalises to the same (u/z, v/z), so they go into
// k_normalize_uvz unchanged.  r04 ran this as a kernel of its own (k_scalars_from_mont: 1024 trivially
// short waves per 2^16 items); r05 calls it from k_normalize_uvz — the host pipeline's preprocessing is
// one launch of few waves.
// ------------------------------------------------------------------------------------------
__device__ constexpr u32 kTwo5[NL] = {32, 0, 0, 0, 0, 0, 0, 0, 0};
DSV_DEV void scalars_from_mont_item(const uint8_t* __restrict__ u_mont, const uint8_t* __restrict__ m_mont, size_t i,
                                    uint8_t* __restrict__ u_out, uint8_t* __restrict__ m_out) {
  u32 w[8], o[8];
  load_words8(w, u_mont, i);
  if (words_lt(w, kR32)) {
    const u32 one[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    fr_mont_mul(o, w, one);
    store_words8(u_out, i, o);
  } else {
    store_poison(u_out, i);
  }
  load_words8(w, m_mont, i);
  if (words_lt(w, kQ32)) {
    fe_to_words_plain(o, fe_canon(fe_mul(fe_from_words_plain(w), fe_const(kTwo5))));
    store_words8(m_out, i, o);
  } else {
    store_poison(m_out, i);
  }
}
template <int NP>
__global__ void __launch_bounds__(256)
k_normalize_uvz(NormalizeArgs a, size_t n, size_t lanes, int per_lane, uint8_t* __restrict__ valid,
                u32* __restrict__ prefix) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= lanes) return;
  Fe acc = fe_one();
  u32 okmask = 0;
#pragma unroll 1
  for (int k = 0; k < per_lane; k++) {
    const size_t i = (size_t)k * lanes + t;
    if (i >= n) break;
    bool ok = true;
    auto up = [&](int p) {
      const Fe z = load_z_for_product(a.in[p], i, ok);
      store_fe_words(prefix + ((size_t)(k * NP + p) * lanes + t) * NL, acc);
      acc = fe_mul(acc, z);
    };
    up(0);
    if (NP > 1) up(1);
    if (NP > 2) up(2);
    if (NP > 3) up(3);
    okmask |= (ok ? 1u : 0u) << k;
    if (a.u_mont) scalars_from_mont_item(a.u_mont, a.m_mont, i, a.u_out, a.m_out);  // (wave-uniform branch)
  }
  Fe inv = fe_invert_euclid(acc);  // the lane's one dependent chain: Euclid, ~4x shorter than Fermat (inv29.h)
#pragma unroll 1
  for (int k = per_lane - 1; k >= 0; k--) {
    const size_t i = (size_t)k * lanes + t;
    if (i >= n) continue;
    bool ok = (okmask >> k) & 1;
    auto down = [&](int p) {
      bool dummy = true;
      const Fe z = load_z_for_product(a.in[p], i, dummy);
      const Fe pre = load_fe_words(prefix + ((size_t)(k * NP + p) * lanes + t) * NL);
      const Fe zinv = fe_mul(inv, pre);  // 1/z in Montgomery form
      inv = fe_mul(inv, z);
      // plain coordinate x Montgomery 1/z = plain quotient: no conversion in either direction
      u32 w[8], o[8];
      load_words8(w, a.in[p], 3 * i);
      ok &= words_lt(w, kQ32);
      fe_to_words_plain(o, fe_canon(fe_mul(fe_from_words_plain(w), zinv)));
      store_words8(a.out[p], 2 * i, o);
      load_words8(w, a.in[p], 3 * i + 1);
      ok &= words_lt(w, kQ32);
      fe_to_words_plain(o, fe_canon(fe_mul(fe_from_words_plain(w), zinv)));
      store_words8(a.out[p], 2 * i + 1, o);
    };
    if (NP > 3) down(3);
    if (NP > 2) down(2);
    if (NP > 1) down(1);
    down(0);
    valid[i] = ok ? 1 : 0;
  }
}
// ------------------------------------------------------------------------------------------
// signing / key derivation ("next" row of the scope table: the step that precedes verify)
// ------------------------------------------------------------------------------------------
// out = scalar * Gen (fixed-base table), affine.  R = r*G, PK = sk*G
// (/root/reference/src/keys/secret.rs:159, public.rs:61-67)
__global__ void __launch_bounds__(256, kWavesVerify)
k_fixed_base_points(const uint8_t* __restrict__ scalar, const u32* __restrict__ table, size_t n,
                    uint8_t* __restrict__ out_uv) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 s[8];
  load_words8(s, scalar, i);
  if (!words_lt(s, kR32)) {  // not a JubJubScalar: poison (no canonical point has 0xff.. coordinates)
    store_poison(out_uv, 2 * i);
    store_poison(out_uv, 2 * i + 1);
    return;
  }
  Ext acc = fixed_base_accumulate(ext_identity(), s, table);
  Fe zi = fe_invert_euclid(acc.z);  // (Fermat was two thirds of this kernel's instructions)
  store_fq(out_uv, 2 * i, fe_mul(acc.u, zi));
  store_fq(out_uv, 2 * i + 1, fe_mul(acc.v, zi));
}
// u = r - c * sk  in Fr  (secret.rs:165)
__global__ void __launch_bounds__(256)
k_sign_finish(const uint8_t* __restrict__ r, const uint8_t* c,  // c may alias u_out
              const uint8_t* __restrict__ sk, size_t n, uint8_t* u_out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 rs[8], cs[8], ks[8], t[8], u[8];
  load_words8(rs, r, i);
  load_words8(cs, c, i);
  load_words8(ks, sk, i);
  if (!words_lt(rs, kR32) || !words_lt(ks, kR32)) {  // nonce or key not a JubJubScalar
    store_poison(u_out, i);
    return;
  }
  fr_mul(t, cs, ks);
  fr_sub(u, rs, t);
  store_words8(u_out, i, u);
}

// ------------------------------------------------------------------------------------------
// wire formats: point decompression (JubJubAffine::from_bytes) and field gathering
// ------------------------------------------------------------------------------------------
// in: one 32-byte compressed point per item at in + i*in_stride (16-byte aligned);
// out_uv: affine u || v canonical; ok[i] = (accumulate ? ok[i] : 1) & decodable
__global__ void __launch_bounds__(256, kWavesHash)
k_decompress(const uint8_t* __restrict__ in, size_t in_stride, size_t n,
             uint8_t* __restrict__ out_uv, uint8_t* __restrict__ ok, int accumulate, TsTables ts) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 w[8];
  {
    const uint4* p = reinterpret_cast<const uint4*>(in + i * in_stride);
    uint4 a = p[0], b = p[1];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
    w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
  }
  const u32 sign = w[7] >> 31;
  w[7] &= 0x7fffffffu;
  bool good = words_lt(w, kQ32);
  const Fe v = fe_to_mont(fe_from_words_plain(w));
  const Fe v2 = fe_sqr(v);
  const Fe num = fe_sub2(v2, fe_one());                          // v^2 - 1
  const Fe den = fe_add(fe_mul(v2, fe_const(kD)), fe_one());     // 1 + d v^2  (never 0: -1/d is a non-square)
  // u = n * (n d)^(-1/2); accept iff u^2 d == n  (rejects non-squares; n == 0 gives u == 0)
  Fe u = fe_mul(num, fe_inv_sqrt(fe_mul(num, den), ts));
  good &= fe_equal(fe_mul(fe_sqr(u), den), num);
  u32 uw[8];
  fe_to_words_plain(uw, fe_from_mont(u));
  if ((uw[0] & 1u) != sign) {                                    // take the other root
    u = fe_neg2(u);
    fe_to_words_plain(uw, fe_from_mont(u));
  }
  store_words8(out_uv, 2 * i, uw);
  store_words8(out_uv, 2 * i + 1, w);
  if (accumulate) good &= ok[i] != 0;
  ok[i] = good ? 1 : 0;
}
// out[i] = 32 bytes at in + i*stride   (AoS wire records -> SoA scalar array)
__global__ void k_gather32(const uint8_t* __restrict__ in, size_t stride, size_t n,
                           uint8_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4* p = reinterpret_cast<const uint4*>(in + i * stride);
  uint4* o = reinterpret_cast<uint4*>(out + i * 32);
  o[0] = p[0];
  o[1] = p[1];
}

// ------------------------------------------------------------------------------------------
// the reference harness's inputs: item i = (sk, message, nonce) from StdRng keystream blocks
// 3i .. 3i+2 (stdrng.h)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_stdrng_triples(ChaChaKey key, size_t first_item, size_t n, uint8_t* __restrict__ sk,
                 uint8_t* __restrict__ m, uint8_t* __restrict__ r) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 blk = 3 * (u64)(first_item + i);
  u32 ks[16], o[8];
  chacha12_block(ks, key.w, blk);
  fr_from_wide(o, ks);
  store_words8(sk, i, o);
  chacha12_block(ks, key.w, blk + 1);
  fq_from_wide(o, ks);
  store_words8(m, i, o);
  chacha12_block(ks, key.w, blk + 2);
  fr_from_wide(o, ks);
  store_words8(r, i, o);
}
// var-generator harness (tests/schnorr_var_generator.rs:16-22, benches/signature_var_generator.rs:
// 50-63): SecretKeyVarGen::random draws sk then the generator scalar (src/keys/secret.rs:371-373),
// then the message, then (inside sign) the nonce: item i = keystream blocks 4i .. 4i+3
__global__ void __launch_bounds__(256)
k_stdrng_quads(ChaChaKey key, size_t first_item, size_t n, uint8_t* __restrict__ sk,
               uint8_t* __restrict__ g, uint8_t* __restrict__ m, uint8_t* __restrict__ r) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 blk = 4 * (u64)(first_item + i);
  u32 ks[16], o[8];
  chacha12_block(ks, key.w, blk);
  fr_from_wide(o, ks);
  store_words8(sk, i, o);
  chacha12_block(ks, key.w, blk + 1);
  fr_from_wide(o, ks);
  store_words8(g, i, o);
  chacha12_block(ks, key.w, blk + 2);
  fq_from_wide(o, ks);
  store_words8(m, i, o);
  chacha12_block(ks, key.w, blk + 3);
  fr_from_wide(o, ks);
  store_words8(r, i, o);
}

__global__ void __launch_bounds__(256)
k_debug_fq_mul(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, size_t n,
               uint8_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fe x, y;
  load_fq(x, a, i);
  load_fq(y, b, i);
  // exercise mul, sqr, add, sub paths: out = a*b  (computed as ((a+b)^2 - a^2 - b^2) / 2 cross-checked)
  Fe p = fe_mul(x, y);
  Fe s = fe_sqr(fe_add(x, y));
  Fe t = fe_sub4(fe_sub4(s, fe_sqr(x)), fe_sqr(y));  // 2ab, < 9.2 q
  t = fe_mul(t, fe_one());                           // back to < 1.2 q before the comparison
  bool same = fe_equal(t, fe_dbl(p));
  u32 w[8];
  fe_to_words_plain(w, fe_from_mont(p));
  if (!same) w[7] |= 0x80000000u;  // poison: can never be canonical
  store_words8(out, i, w);
}

// ------------------------------------------------------------------------------------------
// mixed batches (BASELINE.json configs[4]): split a batch by kind ON THE DEVICE
// kinds[i] = 0 (single signature) / 1 (double signature); anything else is an invalid item that
// lands in neither list (its verdict stays 0).  Stable compaction in three small kernels:
// per-tile counts, one-block exclusive scan of the tile counts, per-tile write-out.  HBM-bound
// byte work (n bytes in, 4n bytes out); against the ~600 k VALU instructions per verdict it is
// noise — written for coalescing, not tuned further.
// ------------------------------------------------------------------------------------------
// counts of kind 0 and kind 1 among the 16 items of this thread, packed (kind1 << 16 | kind0)
DSV_DEV u32 split_thread_counts(const uint8_t* __restrict__ kinds, size_t n, size_t first,
                                uint8_t (&k)[kSplitPerThread]) {
  u32 cnt = 0;
  if (first + kSplitPerThread <= n) {
    const uint4 v = *reinterpret_cast<const uint4*>(kinds + first);
    const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < kSplitPerThread; j++) k[j] = (uint8_t)(w[j >> 2] >> (8 * (j & 3)));
  } else {
#pragma unroll
    for (int j = 0; j < kSplitPerThread; j++) k[j] = first + j < n ? kinds[first + j] : (uint8_t)0xff;
  }
#pragma unroll
  for (int j = 0; j < kSplitPerThread; j++) cnt += (k[j] == 0 ? 1u : 0u) + (k[j] == 1 ? 0x10000u : 0u);
  return cnt;
}
// exclusive scan over the workgroup of one packed counter per thread; returns the block total
DSV_DEV u32 split_block_scan(u32 v, u32& exclusive) {
  __shared__ u32 wave_tot[kSplitThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32 inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u32 t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  u32 before = 0, total = 0;
#pragma unroll
  for (int w2 = 0; w2 < kSplitThreads / 64; w2++) {
    const u32 t = wave_tot[w2];
    if (w2 < wave) before += t;
    total += t;
  }
  exclusive = before + inc - v;
  __syncthreads();
  return total;
}
__global__ void __launch_bounds__(kSplitThreads)
k_kind_count(const uint8_t* __restrict__ kinds, size_t n, u32* __restrict__ tile_counts) {
  uint8_t k[kSplitPerThread];
  const size_t first = ((size_t)blockIdx.x * kSplitThreads + threadIdx.x) * kSplitPerThread;
  u32 ex;
  const u32 total = split_block_scan(split_thread_counts(kinds, n, first, k), ex);
  if (threadIdx.x == 0) {
    tile_counts[2 * blockIdx.x] = total & 0xffffu;
    tile_counts[2 * blockIdx.x + 1] = total >> 16;
  }
}
// in place: tile_counts[2t + k] -> number of kind-k items in tiles before t; totals[k] = all of them
__global__ void __launch_bounds__(1024)
k_kind_scan(u32* __restrict__ tile_counts, size_t ntiles, u32* __restrict__ totals) {
  __shared__ u32 part[2][1024];
  const size_t per = (ntiles + 1023) / 1024;
  const size_t lo = (size_t)threadIdx.x * per, hi = lo + per < ntiles ? lo + per : ntiles;
  u32 s0 = 0, s1 = 0;
  for (size_t t = lo; t < hi; t++) {
    s0 += tile_counts[2 * t];
    s1 += tile_counts[2 * t + 1];
  }
  part[0][threadIdx.x] = s0;
  part[1][threadIdx.x] = s1;
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 a = 0, b = 0;
    for (int t = 0; t < 1024; t++) {
      const u32 x = part[0][t], y = part[1][t];
      part[0][t] = a;
      part[1][t] = b;
      a += x;
      b += y;
    }
    totals[0] = a;
    totals[1] = b;
  }
  __syncthreads();
  u32 a = part[0][threadIdx.x], b = part[1][threadIdx.x];
  for (size_t t = lo; t < hi; t++) {
    const u32 x = tile_counts[2 * t], y = tile_counts[2 * t + 1];
    tile_counts[2 * t] = a;
    tile_counts[2 * t + 1] = b;
    a += x;
    b += y;
  }
}
// idx_k[j] = batch position of the j-th item of kind k (j < cap_k: a caller that understated a
// count loses the surplus instead of overrunning its buffer; totals[] tell)
__global__ void __launch_bounds__(kSplitThreads)
k_kind_write(const uint8_t* __restrict__ kinds, size_t n, const u32* __restrict__ tile_offsets,
             u32* __restrict__ idx0, size_t cap0, u32* __restrict__ idx1, size_t cap1) {
  uint8_t k[kSplitPerThread];
  const size_t first = ((size_t)blockIdx.x * kSplitThreads + threadIdx.x) * kSplitPerThread;
  u32 ex;
  split_block_scan(split_thread_counts(kinds, n, first, k), ex);
  size_t p0 = (size_t)tile_offsets[2 * blockIdx.x] + (ex & 0xffffu);
  size_t p1 = (size_t)tile_offsets[2 * blockIdx.x + 1] + (ex >> 16);
#pragma unroll
  for (int j = 0; j < kSplitPerThread; j++) {
    if (k[j] == 0) {
      if (p0 < cap0) idx0[p0] = (u32)(first + j);
      p0++;
    } else if (k[j] == 1) {
      if (p1 < cap1) idx1[p1] = (u32)(first + j);
      p1++;
    }
  }
}
// Index vectors are only dereferenced where the split wrote them: entries j < min(count, *limit)
// (limit = the split's own total for that kind, device memory), and an index >= rows is skipped.
// A caller whose declared counts disagree with the kind vector therefore never makes these kernels
// read or write out of bounds (r02 did: ADVICE r02, dsv_verify_mixed_dev / MixedShardedVerifier).
// dst row j = src row idx[j]; rows of row16 * 16 bytes, one thread per 16-byte piece
__global__ void __launch_bounds__(256)
k_gather_rows(const uint4* __restrict__ src, size_t src_rows, u32 row16, const u32* __restrict__ idx,
              size_t count, const u32* __restrict__ limit, uint4* __restrict__ dst) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (limit && *limit < count) count = *limit;
  if (g >= count * row16) return;
  const size_t j = g / row16;
  const u32 part = (u32)(g - j * row16);
  const size_t row = idx[j];
  if (row < src_rows) dst[g] = src[row * row16 + part];
}
__global__ void __launch_bounds__(256)
k_scatter_bytes(const uint8_t* __restrict__ src, const u32* __restrict__ idx, size_t count,
                const u32* __restrict__ limit, uint8_t* __restrict__ dst, size_t dst_len) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (limit && *limit < count) count = *limit;
  if (j >= count) return;
  const size_t pos = idx[j];
  if (pos < dst_len) dst[pos] = src[j];
}
// a mixed call whose declared kind counts disagree with the kind vector has no usable verdicts
__global__ void __launch_bounds__(256)
k_mixed_check(const u32* __restrict__ totals, u32 want0, u32 want1, uint8_t* __restrict__ ok, size_t n) {
  if (totals[0] == want0 && totals[1] == want1) return;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    ok[i] = 0;
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
void launch_build_fixed_table(uint32_t* table, int which, hipStream_t s) {
  const int total = kFixedWindows * kFixedEntries;
  hipLaunchKernelGGL(k_build_fixed_table, dim3((total + 63) / 64), dim3(64), 0, s, table, which);
}
void launch_fixed_base_points(const uint8_t* scalar, const uint32_t* table, size_t n, uint8_t* out_uv,
                              hipStream_t s) {
  hipLaunchKernelGGL(k_fixed_base_points, dim3(grid_for(n)), dim3(256), 0, s, scalar, table, n, out_uv);
}
void launch_normalize_uvz(const NormalizeArgs& a, int npoints, size_t n, uint8_t* valid,
                          uint32_t* prefix, hipStream_t s, int want_per_lane, int block_threads) {
  int per_lane;
  const size_t lanes = normalize_lanes(n, per_lane, want_per_lane);
  const unsigned bt = block_threads > 0 ? (unsigned)block_threads : 256u;
  const dim3 grid(grid_for(lanes, bt)), block(bt);
  switch (npoints) {
    case 1: hipLaunchKernelGGL(k_normalize_uvz<1>, grid, block, 0, s, a, n, lanes, per_lane, valid, prefix); break;
    case 2: hipLaunchKernelGGL(k_normalize_uvz<2>, grid, block, 0, s, a, n, lanes, per_lane, valid, prefix); break;
    case 3: hipLaunchKernelGGL(k_normalize_uvz<3>, grid, block, 0, s, a, n, lanes, per_lane, valid, prefix); break;
    default: hipLaunchKernelGGL(k_normalize_uvz<4>, grid, block, 0, s, a, n, lanes, per_lane, valid, prefix); break;
  }
}
void launch_sign_finish(const uint8_t* r, const uint8_t* c, const uint8_t* sk, size_t n, uint8_t* u_out,
                        hipStream_t s) {
  hipLaunchKernelGGL(k_sign_finish, dim3(grid_for(n)), dim3(256), 0, s, r, c, sk, n, u_out);
}
void launch_decompress(const uint8_t* in, size_t in_stride, size_t n, uint8_t* out_uv, uint8_t* ok,
                       int accumulate, const uint32_t* ts_cancel, const uint8_t* ts_hash, hipStream_t s) {
  hipLaunchKernelGGL(k_decompress, dim3(grid_for(n)), dim3(256), 0, s, in, in_stride, n, out_uv, ok,
                     accumulate, TsTables{ts_cancel, ts_hash});
}
void launch_gather32(const uint8_t* in, size_t stride, size_t n, uint8_t* out, hipStream_t s) {
  hipLaunchKernelGGL(k_gather32, dim3(grid_for(n)), dim3(256), 0, s, in, stride, n, out);
}
void launch_stdrng_triples(ChaChaKey key, size_t first_item, size_t n, uint8_t* sk, uint8_t* m,
                           uint8_t* r, hipStream_t s) {
  hipLaunchKernelGGL(k_stdrng_triples, dim3(grid_for(n)), dim3(256), 0, s, key, first_item, n, sk, m, r);
}
void launch_stdrng_quads(ChaChaKey key, size_t first_item, size_t n, uint8_t* sk, uint8_t* g,
                         uint8_t* m, uint8_t* r, hipStream_t s) {
  hipLaunchKernelGGL(k_stdrng_quads, dim3(grid_for(n)), dim3(256), 0, s, key, first_item, n, sk, g, m, r);
}
void launch_debug_fq_mul(const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out, hipStream_t s) {
  hipLaunchKernelGGL(k_debug_fq_mul, dim3(grid_for(n)), dim3(256), 0, s, a, b, n, out);
}
void launch_split_kinds(const uint8_t* kinds, size_t n, uint32_t* tile_counts, uint32_t* totals,
                        uint32_t* idx0, size_t cap0, uint32_t* idx1, size_t cap1, hipStream_t s) {
  const size_t tiles = (n + kSplitTile - 1) / kSplitTile;
  hipLaunchKernelGGL(k_kind_count, dim3((unsigned)tiles), dim3(kSplitThreads), 0, s, kinds, n, tile_counts);
  hipLaunchKernelGGL(k_kind_scan, dim3(1), dim3(1024), 0, s, tile_counts, tiles, totals);
  hipLaunchKernelGGL(k_kind_write, dim3((unsigned)tiles), dim3(kSplitThreads), 0, s, kinds, n,
                     (const u32*)tile_counts, idx0, cap0, idx1, cap1);
}
void launch_gather_rows(const void* src, size_t src_rows, uint32_t row16, const uint32_t* idx,
                        size_t count, const uint32_t* limit, void* dst, hipStream_t s) {
  hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(count * row16)), dim3(256), 0, s, (const uint4*)src,
                     src_rows, row16, idx, count, limit, (uint4*)dst);
}
void launch_scatter_bytes(const uint8_t* src, const uint32_t* idx, size_t count, const uint32_t* limit,
                          uint8_t* dst, size_t dst_len, hipStream_t s) {
  hipLaunchKernelGGL(k_scatter_bytes, dim3(grid_for(count)), dim3(256), 0, s, src, idx, count, limit, dst,
                     dst_len);
}
void launch_mixed_check(const uint32_t* totals, uint32_t want0, uint32_t want1, uint8_t* ok, size_t n,
                        hipStream_t s) {
  hipLaunchKernelGGL(k_mixed_check, dim3(256), dim3(256), 0, s, totals, want0, want1, ok, n);
}

}  // namespace dsv
