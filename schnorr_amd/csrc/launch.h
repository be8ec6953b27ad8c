// launch.h — what the host units (dsv_*.hip) know about the kernel translation units: table / grid
// geometry and one launcher per kernel.  The kernels themselves are compiled separately
// (k_hash.hip, k_verify.hip, k_quad.hip, k_vargen.hip, k_misc.hip; one hipcc job each, no
// relocatable device code: nothing on the device crosses a translation unit).
// Launchers enqueue on the given stream and never synchronise; errors surface through
// hipGetLastError() in the caller.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace dsv {

constexpr int kLimbs = 9;  // fe29: 9 limbs of 29 bits (fe29.h: NL)

// ---- fixed-base tables (G, G'): signed kFixedBits-bit windows ------------------------------
constexpr int kFixedBits = 16;
constexpr int kFixedWindows = (253 + kFixedBits - 1) / kFixedBits;  // +1 bit: recoding carry
constexpr int kFixedEntries = (1 << (kFixedBits - 1)) + 1;          // |digit| = 0 .. 2^(bits-1)
constexpr int kEntryWords = 4 * kLimbs;  // v+u, v-u, 2d*uv, -(2d*uv): 144 B, 16-byte aligned
constexpr size_t kTableBytes = (size_t)kFixedWindows * kFixedEntries * kEntryWords * 4;
static_assert(kFixedWindows * kFixedBits <= 288 && kFixedBits >= 4 && kFixedBits <= 16, "window");

// ---- per-lane window tables of a variable base (common.h) -----------------------------------
constexpr int kVarEntries = 9;                                // |digit| = 0 .. 8 (slot 0 unused)
constexpr int kVarEntryWords = 4 * kLimbs;                    // v+u, v-u, z, 2d*t
constexpr int kVarLaneWords = kVarEntries * kVarEntryWords;   // 324 words = 1296 B
constexpr int kVerifyBlock = 64;        // ONE wave per workgroup: a finished wave's slot is refilled
                                        // at once instead of waiting for its workgroup mates
constexpr unsigned kMaxVerifyGrid = 4096;                     // 16 single-wave workgroups per CU
constexpr int kWavesVerify = 2, kWavesHash = 2;               // resident waves per SIMD (launch bounds)

// ---- small batches: eight lanes per signature (k_quad.hip) ---------------------------------
constexpr int kQuadBlock = 256;              // 32 signatures per workgroup
constexpr size_t kQuadMaxItems = (size_t)1 << 14;
// the var-generator scheme's sixteen-lanes-per-signature kernel (k_vargen.hip: k_verify_var_hex): launches of
// at most this many items (2^13 items are 2048 waves: one per SIMD and a half)
constexpr size_t kVarHexMaxItems = (size_t)1 << 13;

// ---- mixed-batch split (k_misc.hip) ---------------------------------------------------------
constexpr int kSplitThreads = 256;
constexpr int kSplitPerThread = 16;                          // one 16-byte load per thread
constexpr int kSplitTile = kSplitThreads * kSplitPerThread;  // 4096 items per workgroup

inline unsigned grid_for(size_t n, unsigned block = 256) { return (unsigned)((n + block - 1) / block); }
inline unsigned verify_grid(size_t n) {
  const unsigned g = grid_for(n, kVerifyBlock);
  return g < kMaxVerifyGrid ? g : kMaxVerifyGrid;
}

// one equation's operands: the (PK, R) pair and the fixed-base table of the generator that goes
// with it
struct ChainOperands {
  const uint8_t* PK_uv;
  const uint8_t* R_uv;
  const uint32_t* table;
};
struct ChaChaKey {
  uint32_t w[8];
};

// ---- k_hash.hip ------------------------------------------------------------------------------
hipError_t hash_upload_constants();  // the selected device's __constant__ round constants
// valid_in (may be null): per-item bytes of an earlier stage, AND-ed into `valid`
void launch_challenge(bool dbl, const uint8_t* R_uv, const uint8_t* Rp_uv, const uint8_t* m, size_t n,
                      uint8_t* c, uint8_t* valid, hipStream_t s, const uint8_t* valid_in = nullptr);
// ---- k_verify.hip: ok[i] = (accumulate ? ok[i] : valid[i]) & [every chain's equation holds] ----
// gate (device memory, may be null; all three verify kernels): two words written earlier on the stream by
// the batch fast accept (rlc.h).  gate[0] == 0 and gate[1] == 1 — "these items were decided by their
// aggregate" — makes every workgroup return at once and leaves ok[] as it is: the launch is enqueued
// unconditionally behind the aggregate and costs a few microseconds when it is not needed.
inline __device__ bool gate_says_done(const uint32_t* gate) { return gate && gate[0] == 0u && gate[1] == 1u; }
void launch_verify_half(int nchain, bool accumulate, const uint8_t* u, const uint8_t* c,
                        ChainOperands op0, ChainOperands op1, const uint8_t* valid, size_t n,
                        uint8_t* ok, uint32_t* var_tables, hipStream_t s, const uint32_t* gate = nullptr);
// ---- k_quad.hip: the same for n <= kQuadMaxItems, eight lanes per signature; tables_ready: the
// window tables of the FIRST equation's (PK, R) were built by launch_prep_var_tables already ----
void launch_verify_half_quad(int nchain, bool accumulate, bool tables_ready, const uint8_t* u,
                             const uint8_t* c, ChainOperands op0, ChainOperands op1,
                             const uint8_t* valid, size_t n, uint8_t* ok, uint32_t* var_tables,
                             hipStream_t s, const uint32_t* gate = nullptr);
void launch_prep_var_tables(const uint8_t* PK_uv, const uint8_t* R_uv, size_t n, uint32_t* var_tables,
                            hipStream_t s);
// ---- k_vargen.hip ----------------------------------------------------------------------------
void launch_verify_var(const uint8_t* u, const uint8_t* c, const uint8_t* PK_uv, const uint8_t* Gen_uv,
                       const uint8_t* R_uv, const uint8_t* valid, size_t n, uint8_t* ok,
                       uint32_t* var_tables, hipStream_t s, const uint32_t* gate = nullptr);
void launch_debug_lattice3(const uint8_t* u, const uint8_t* c, size_t n, uint8_t* out, hipStream_t s);
void launch_debug_half_scalars(const uint8_t* c, size_t n, uint8_t* out, hipStream_t s);
void launch_var_base_points(const uint8_t* scalar, const uint8_t* P_uv, size_t n, uint8_t* out_uv,
                            uint32_t* var_tables, hipStream_t s);
// ---- k_misc.hip ------------------------------------------------------------------------------
void launch_build_fixed_table(uint32_t* table, int which, hipStream_t s);
void launch_fixed_base_points(const uint8_t* scalar, const uint32_t* table, size_t n, uint8_t* out_uv,
                              hipStream_t s);
// to_hash_inputs on the device: `npoints` projective points per item, each (u, v, z) 96 B in its
// own array in[k], normalised into out[k] (u/z, v/z; 64 B) with at most ONE inversion per item;
// valid[i] = every coordinate canonical and every z != 0
// (Montgomery's trick over the item's z's AND over the 8 or 16 items a lane handles once the batch
// is large); prefix: scratch of normalize_prefix_bytes(n, npoints) device bytes
struct NormalizeArgs {
  const uint8_t* in[4];
  uint8_t* out[4];
  // r05, optional (all four or none): the item's two scalars in the reference's in-memory form
  // (Montgomery limbs, R = 2^256) converted to canonical bytes BY THE SAME KERNEL — the host pipeline's
  // preprocessing is then one launch of few waves instead of two (k_scalars_from_mont alone is 1024
  // trivially short waves per 2^16 items, each waiting for a wave slot next to the resident verify waves:
  // 0.16 - 0.37 ms of a chunk's staging chain, profiles/r05/host_timeline_e2e.txt)
  const uint8_t* u_mont = nullptr;
  const uint8_t* m_mont = nullptr;
  uint8_t* u_out = nullptr;
  uint8_t* m_out = nullptr;
};
constexpr int kNormalizePerLane = 32;  // at most (the kernel keeps one validity bit per item in a u32)
inline size_t normalize_lanes(size_t n, int& per_lane, int want_per_lane = 0) {
  // items that share one inversion: more of them = less work, fewer lanes = longer latency
  per_lane = n >= ((size_t)1 << 19) ? 16 : (n >= ((size_t)1 << 15) ? 8 : 1);
  if (want_per_lane > 0) per_lane = want_per_lane > kNormalizePerLane ? kNormalizePerLane : want_per_lane;
  return (n + per_lane - 1) / per_lane;
}
inline size_t normalize_prefix_bytes(size_t n, int npoints) {
  // per_lane * lanes < n + per_lane slots of npoints running products (any per_lane up to the maximum)
  return (size_t)npoints * (n + kNormalizePerLane) * kLimbs * 4;
}
// want_per_lane / block: 0 = the defaults above / 256 threads (the host pipeline asks for few,
// single-wave workgroups: DESIGN.md §4)
void launch_normalize_uvz(const NormalizeArgs& a, int npoints, size_t n, uint8_t* valid,
                          uint32_t* prefix, hipStream_t s, int want_per_lane = 0, int block = 0);
void launch_sign_finish(const uint8_t* r, const uint8_t* c, const uint8_t* sk, size_t n, uint8_t* u_out,
                        hipStream_t s);
void launch_decompress(const uint8_t* in, size_t in_stride, size_t n, uint8_t* out_uv, uint8_t* ok,
                       int accumulate, const uint32_t* ts_cancel, const uint8_t* ts_hash, hipStream_t s);
void launch_gather32(const uint8_t* in, size_t stride, size_t n, uint8_t* out, hipStream_t s);
void launch_stdrng_triples(ChaChaKey key, size_t first_item, size_t n, uint8_t* sk, uint8_t* m,
                           uint8_t* r, hipStream_t s);
void launch_stdrng_quads(ChaChaKey key, size_t first_item, size_t n, uint8_t* sk, uint8_t* g,
                         uint8_t* m, uint8_t* r, hipStream_t s);
void launch_debug_fq_mul(const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out, hipStream_t s);
// mixed batches: stable split of the index vector by kind, bounded row gather / verdict scatter.
// `limit` (device, may be null): the gather / scatter touches min(count, *limit) entries, and an
// index >= rows is skipped — the index vectors are only ever dereferenced where the split wrote.
void launch_split_kinds(const uint8_t* kinds, size_t n, uint32_t* tile_counts, uint32_t* totals,
                        uint32_t* idx0, size_t cap0, uint32_t* idx1, size_t cap1, hipStream_t s);
void launch_gather_rows(const void* src, size_t src_rows, uint32_t row16, const uint32_t* idx,
                        size_t count, const uint32_t* limit, void* dst, hipStream_t s);
void launch_scatter_bytes(const uint8_t* src, const uint32_t* idx, size_t count, const uint32_t* limit,
                          uint8_t* dst, size_t dst_len, hipStream_t s);
void launch_mixed_check(const uint32_t* totals, uint32_t want0, uint32_t want1, uint8_t* ok, size_t n,
                        hipStream_t s);

}  // namespace dsv
