// dsv_host.hip — the host-pointer entry points of include/dsv.h: every verify scheme over affine bytes,
// projective points (`to_hash_inputs` on the device), the reference's in-memory Montgomery limbs and
// columns of typed objects (what the Rust `verify_batch(&[Signature], &[PublicKey], &[BlsScalar])`
// binds), their *_multi forms over all initialised devices, and the submit / wait form.  All of them
// run the chunked pipeline of dsv_pipeline.h.
#include "dsv_pipeline.h"

namespace dsvh {
std::atomic<int> g_host_threads{0};
}
using namespace dsvh;

extern "C" {

namespace {
// one sub-batch of a staged chunk through the affine path of its scheme (0 single, 1 double, 2 vargen)
int part_verify(Context& ctx, int kind, const Staged& g, size_t off, size_t cnt, void* dok, void* ws,
                hipStream_t st) {
  auto at = [&](int k) { return g.p[k] + off * g.bytes[k]; };
  int rc;
  const uint8_t* vin = g.valid ? g.valid + off : nullptr;  // (the chunk-level preprocessing's verdict on the items)
  if (kind == 0) rc = verify_single_on(ctx, at(0), at(1), at(2), at(3), cnt, dok, ws, st, vin);
  else if (kind == 1) rc = verify_double_on(ctx, at(0), at(1), at(2), at(3), at(4), at(5), cnt, dok, ws, st, vin);
  else rc = verify_vargen_on(ctx, at(0), at(1), at(2), at(3), at(4), cnt, dok, ws, st, vin);
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
#define DSV_PART(kind_)                                                                                     \
  [=](const Staged& g, size_t off, size_t cnt, void* dok, void* ws, Stager&, hipStream_t st) {              \
    return part_verify(*cp, kind_, g, off, cnt, dok, ws, st);                                               \
  }
int verify_single_host(Context& ctx, const uint8_t* u, const uint8_t* R_uv, const uint8_t* PK_uv,
                       const uint8_t* m, size_t n, uint8_t* ok) {
  const HostIn ins[4] = {{u, 32}, {R_uv, 64}, {PK_uv, 64}, {m, 32}};
  Context* cp = &ctx;
  return run_pipelined(ctx, ins, ok, n, 0, 0, NoPrep{}, DSV_PART(0));
}
int verify_double_host(Context& ctx, const uint8_t* u, const uint8_t* R_uv, const uint8_t* Rp_uv,
                       const uint8_t* PK_uv, const uint8_t* PKp_uv, const uint8_t* m, size_t n,
                       uint8_t* ok) {
  const HostIn ins[6] = {{u, 32}, {R_uv, 64}, {Rp_uv, 64}, {PK_uv, 64}, {PKp_uv, 64}, {m, 32}};
  Context* cp = &ctx;
  return run_pipelined(ctx, ins, ok, n, 0, 0, NoPrep{}, DSV_PART(1), kPipeHeavy);
}
int verify_vargen_host(Context& ctx, const uint8_t* u, const uint8_t* R_uv, const uint8_t* PK_uv,
                       const uint8_t* Gen_uv, const uint8_t* m, size_t n, uint8_t* ok) {
  const HostIn ins[5] = {{u, 32}, {R_uv, 64}, {PK_uv, 64}, {Gen_uv, 64}, {m, 32}};
  Context* cp = &ctx;
  return run_pipelined(ctx, ins, ok, n, 0, 0, NoPrep{}, DSV_PART(2), kPipeHeavy);
}
}  // namespace

int dsv_verify_single(const uint8_t* u, const uint8_t* R_uv, const uint8_t* PK_uv,
                      const uint8_t* m, size_t n, uint8_t* ok) {
  if (n && (!u || !R_uv || !PK_uv || !m || !ok)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  return verify_single_host(ctx, u, R_uv, PK_uv, m, n, ok);
}
int dsv_verify_double(const uint8_t* u, const uint8_t* R_uv, const uint8_t* Rp_uv,
                      const uint8_t* PK_uv, const uint8_t* PKp_uv, const uint8_t* m, size_t n,
                      uint8_t* ok) {
  if (n && (!u || !R_uv || !Rp_uv || !PK_uv || !PKp_uv || !m || !ok))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  return verify_double_host(ctx, u, R_uv, Rp_uv, PK_uv, PKp_uv, m, n, ok);
}
int dsv_verify_vargen(const uint8_t* u, const uint8_t* R_uv, const uint8_t* PK_uv,
                      const uint8_t* Gen_uv, const uint8_t* m, size_t n, uint8_t* ok) {
  if (n && (!u || !R_uv || !PK_uv || !Gen_uv || !m || !ok))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  return verify_vargen_host(ctx, u, R_uv, PK_uv, Gen_uv, m, n, ok);
}

// ---- the same over ALL initialised devices (what a Rust verify_batch on an 8-GPU node calls) ----
int dsv_verify_single_multi(const uint8_t* u, const uint8_t* R_uv, const uint8_t* PK_uv,
                            const uint8_t* m, size_t n, uint8_t* ok) {
  if (int r = check_n(n)) return r;
  if (n == 0) return DSV_OK;
  if (!u || !R_uv || !PK_uv || !m || !ok) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  return run_multi(n, [=](Context& ctx, size_t off, size_t cnt) {
    return verify_single_host(ctx, u + 32 * off, R_uv + 64 * off, PK_uv + 64 * off, m + 32 * off, cnt,
                              ok + off);
  });
}
int dsv_verify_double_multi(const uint8_t* u, const uint8_t* R_uv, const uint8_t* Rp_uv,
                            const uint8_t* PK_uv, const uint8_t* PKp_uv, const uint8_t* m, size_t n,
                            uint8_t* ok) {
  if (int r = check_n(n)) return r;
  if (n == 0) return DSV_OK;
  if (!u || !R_uv || !Rp_uv || !PK_uv || !PKp_uv || !m || !ok)
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  return run_multi(n, [=](Context& ctx, size_t off, size_t cnt) {
    return verify_double_host(ctx, u + 32 * off, R_uv + 64 * off, Rp_uv + 64 * off, PK_uv + 64 * off,
                              PKp_uv + 64 * off, m + 32 * off, cnt, ok + off);
  });
}
int dsv_verify_vargen_multi(const uint8_t* u, const uint8_t* R_uv, const uint8_t* PK_uv,
                            const uint8_t* Gen_uv, const uint8_t* m, size_t n, uint8_t* ok) {
  if (int r = check_n(n)) return r;
  if (n == 0) return DSV_OK;
  if (!u || !R_uv || !PK_uv || !Gen_uv || !m || !ok)
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  return run_multi(n, [=](Context& ctx, size_t off, size_t cnt) {
    return verify_vargen_host(ctx, u + 32 * off, R_uv + 64 * off, PK_uv + 64 * off, Gen_uv + 64 * off,
                              m + 32 * off, cnt, ok + off);
  });
}

// ---- projective inputs: the reference's in-memory types ------------------------------------
// `PublicKey::from(&sk)` = GENERATOR_EXTENDED * sk and R = GENERATOR_EXTENDED * r are JubJubExtended
// values with z != 1 (/root/reference/src/keys/public.rs:61-67, src/keys/secret.rs:159), and the
// reference's verify starts with `to_hash_inputs` (src/signatures.rs:131, :280-281): one field
// inversion per point.  The *_ext entry points take (u, v, z) and do that step on the device —
// Montgomery's trick over all points of an item and over the items of a lane (k_normalize_uvz) —
// so a caller (the Rust verify_batch) does no field arithmetic on the host at all.
extern "C++" {
namespace {
struct ExtWs {
  uint8_t* pts[4];
  uint8_t* valid;
  u32* prefix;
  void* vws;
};
size_t ext_workspace_bytes(size_t n) {
  return 4 * align_up(n * 64, 256) + align_up(n, 256) + align_up(normalize_prefix_bytes(n, 4), 256) +
         align_up(dsv_workspace_bytes(n), 256) + 256;
}
ExtWs carve_ext(void* ws, size_t n) {
  Stager st(static_cast<uint8_t*>(ws));
  ExtWs w;
  for (int k = 0; k < 4; k++) w.pts[k] = st.take(n * 64);
  w.valid = st.take(n);
  w.prefix = reinterpret_cast<u32*>(st.take(normalize_prefix_bytes(n, 4)));
  w.vws = st.take(dsv_workspace_bytes(n));
  return w;
}
// kind 0: pts = {R, PK}; 1: {R, R', PK, PK'}; 2: {R, PK, Gen} — each n x 96 B (u || v || z)
// u_mont / m_mont (both or neither): the scalars as Montgomery limbs, converted INTO u / m by the
// normalisation kernel (dsv_verify_*_mont_dev)
int verify_ext_on(Context& ctx, int kind, const void* u, const void* const* pts_uvz, const void* m,
                  size_t n, void* ok, void* workspace, hipStream_t s, const uint8_t* u_mont = nullptr,
                  const uint8_t* m_mont = nullptr) {
  const int np = kind == 0 ? 2 : (kind == 1 ? 4 : 3);
  const ExtWs w = carve_ext(workspace, n);
  NormalizeArgs a = {};
  for (int k = 0; k < np; k++) {
    a.in[k] = (const uint8_t*)pts_uvz[k];
    a.out[k] = w.pts[k];
  }
  if (u_mont) {
    a.u_mont = u_mont;
    a.m_mont = m_mont;
    a.u_out = (uint8_t*)const_cast<void*>(u);
    a.m_out = (uint8_t*)const_cast<void*>(m);
  }
  launch_normalize_uvz(a, np, n, w.valid, w.prefix, s);
  int rc;
  if (kind == 0) rc = verify_single_on(ctx, u, w.pts[0], w.pts[1], m, n, ok, w.vws, s, w.valid);
  else if (kind == 1) rc = verify_double_on(ctx, u, w.pts[0], w.pts[1], w.pts[2], w.pts[3], m, n, ok, w.vws, s, w.valid);
  else rc = verify_vargen_on(ctx, u, w.pts[0], w.pts[1], w.pts[2], m, n, ok, w.vws, s, w.valid);
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
// host buffers: the same through the chunked pipeline; the ext workspace of a chunk sits in the
// slot's `extra` area and the pipeline's own verify workspace is not used
// whole-chunk `to_hash_inputs` of the scheme's points (d: u, points..., m; [u_alt, m_alt]: converted
// scalars to use instead of d's); fills the staged view of the affine path
// mont: the scalars d[0] / d[1 + np] are Montgomery limbs — the same kernel converts them into (u_alt, m_alt)
int prep_normalize(const Context& ctx, int kind, const void* const* d, size_t cnt, Stager& x, hipStream_t st, Staged& g,
                   uint8_t* u_alt = nullptr, uint8_t* m_alt = nullptr, bool mont = false) {
  const int np = kind == 0 ? 2 : (kind == 1 ? 4 : 3);
  NormalizeArgs a = {};
  for (int k = 0; k < np; k++) {
    a.in[k] = (const uint8_t*)d[1 + k];
    a.out[k] = x.take(cnt * 64);
    g.p[1 + k] = a.out[k];
    g.bytes[1 + k] = 64;
  }
  uint8_t* valid = x.take(cnt);
  u32* prefix = reinterpret_cast<u32*>(x.take(normalize_prefix_bytes(cnt, np)));
  if (mont) {
    a.u_mont = (const uint8_t*)d[0];
    a.m_mont = (const uint8_t*)d[1 + np];
    a.u_out = u_alt;
    a.m_out = m_alt;
  }
  launch_normalize_uvz(a, np, cnt, valid, prefix, st, ctx.norm_per_lane, ctx.norm_block);
  HIP_TRY(hipGetLastError());
  g.p[0] = u_alt ? u_alt : (const uint8_t*)d[0];
  g.p[1 + np] = m_alt ? m_alt : (const uint8_t*)d[1 + np];
  g.bytes[0] = g.bytes[1 + np] = 32;
  g.valid = valid;
  return DSV_OK;
}
template <size_t NIN>
int verify_ext_host(Context& ctx, int kind, const HostIn (&ins)[NIN], size_t n, uint8_t* ok) {
  Context* cp = &ctx;
  return run_pipelined(ctx, ins, ok, n, kExtItemBytes, 0,
                       [kind, cp](const void* const* d, size_t cnt, Stager& x, hipStream_t st, Staged& g) {
                         return prep_normalize(*cp, kind, d, cnt, x, st, g);
                       },
                       DSV_PART(kind), kind != 0 ? kPipeHeavy : 0u);
}
int verify_single_ext_host(Context& ctx, const uint8_t* u, const uint8_t* R, const uint8_t* PK,
                           const uint8_t* m, size_t n, uint8_t* ok) {
  const HostIn ins[4] = {{u, 32}, {R, 96}, {PK, 96}, {m, 32}};
  return verify_ext_host(ctx, 0, ins, n, ok);
}
int verify_double_ext_host(Context& ctx, const uint8_t* u, const uint8_t* R, const uint8_t* Rp,
                           const uint8_t* PK, const uint8_t* PKp, const uint8_t* m, size_t n, uint8_t* ok) {
  const HostIn ins[6] = {{u, 32}, {R, 96}, {Rp, 96}, {PK, 96}, {PKp, 96}, {m, 32}};
  return verify_ext_host(ctx, 1, ins, n, ok);
}
int verify_vargen_ext_host(Context& ctx, const uint8_t* u, const uint8_t* R, const uint8_t* PK,
                           const uint8_t* Gen, const uint8_t* m, size_t n, uint8_t* ok) {
  const HostIn ins[5] = {{u, 32}, {R, 96}, {PK, 96}, {Gen, 96}, {m, 32}};
  return verify_ext_host(ctx, 2, ins, n, ok);
}
}  // namespace
}  // extern "C++"

size_t dsv_ext_workspace_bytes(size_t n) { return ext_workspace_bytes(n); }

// JubJubExtended::to_hash_inputs for n points: (u, v, z) -> (u/z, v/z); ok[i] = 0 for z = 0 or a
// non-canonical coordinate (the reference would panic / cannot hold such a value)
int dsv_to_hash_inputs(const uint8_t* in_uvz, size_t n, uint8_t* out_uv, uint8_t* ok) {
  if (n && (!in_uvz || !out_uv || !ok)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  DSV_HOST_LOCK();
  const size_t pre = normalize_prefix_bytes(n, 1);
  if (int r = ensure_stage(ctx, align_up(n * 96, 256) + align_up(n * 64, 256) + align_up(n, 256) +
                                    align_up(pre, 256)))
    return r;
  Stager st(ctx.stage);
  uint8_t *din = st.take(n * 96), *dout = st.take(n * 64), *dok = st.take(n);
  u32* dpre = reinterpret_cast<u32*>(st.take(pre));
  H2D(din, in_uvz, n * 96);
  NormalizeArgs a = {};
  a.in[0] = din;
  a.out[0] = dout;
  launch_normalize_uvz(a, 1, n, dok, dpre, 0);
  HIP_TRY(hipGetLastError());
  D2H(out_uv, dout, n * 64);
  D2H(ok, dok, n);
  HIP_TRY(hipStreamSynchronize(0));
  return DSV_OK;
}

int dsv_verify_single_ext(const uint8_t* u, const uint8_t* R_uvz, const uint8_t* PK_uvz,
                          const uint8_t* m, size_t n, uint8_t* ok) {
  if (n && (!u || !R_uvz || !PK_uvz || !m || !ok)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  return verify_single_ext_host(ctx, u, R_uvz, PK_uvz, m, n, ok);
}
int dsv_verify_double_ext(const uint8_t* u, const uint8_t* R_uvz, const uint8_t* Rp_uvz,
                          const uint8_t* PK_uvz, const uint8_t* PKp_uvz, const uint8_t* m, size_t n,
                          uint8_t* ok) {
  if (n && (!u || !R_uvz || !Rp_uvz || !PK_uvz || !PKp_uvz || !m || !ok))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  return verify_double_ext_host(ctx, u, R_uvz, Rp_uvz, PK_uvz, PKp_uvz, m, n, ok);
}
int dsv_verify_vargen_ext(const uint8_t* u, const uint8_t* R_uvz, const uint8_t* PK_uvz,
                          const uint8_t* Gen_uvz, const uint8_t* m, size_t n, uint8_t* ok) {
  if (n && (!u || !R_uvz || !PK_uvz || !Gen_uvz || !m || !ok))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  return verify_vargen_ext_host(ctx, u, R_uvz, PK_uvz, Gen_uvz, m, n, ok);
}
int dsv_verify_single_ext_multi(const uint8_t* u, const uint8_t* R_uvz, const uint8_t* PK_uvz,
                                const uint8_t* m, size_t n, uint8_t* ok) {
  if (int r = check_n(n)) return r;
  if (n == 0) return DSV_OK;
  if (!u || !R_uvz || !PK_uvz || !m || !ok) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  return run_multi(n, [=](Context& ctx, size_t off, size_t cnt) {
    return verify_single_ext_host(ctx, u + 32 * off, R_uvz + 96 * off, PK_uvz + 96 * off, m + 32 * off,
                                  cnt, ok + off);
  });
}
int dsv_verify_double_ext_multi(const uint8_t* u, const uint8_t* R_uvz, const uint8_t* Rp_uvz,
                                const uint8_t* PK_uvz, const uint8_t* PKp_uvz, const uint8_t* m,
                                size_t n, uint8_t* ok) {
  if (int r = check_n(n)) return r;
  if (n == 0) return DSV_OK;
  if (!u || !R_uvz || !Rp_uvz || !PK_uvz || !PKp_uvz || !m || !ok)
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  return run_multi(n, [=](Context& ctx, size_t off, size_t cnt) {
    return verify_double_ext_host(ctx, u + 32 * off, R_uvz + 96 * off, Rp_uvz + 96 * off,
                                  PK_uvz + 96 * off, PKp_uvz + 96 * off, m + 32 * off, cnt, ok + off);
  });
}
int dsv_verify_vargen_ext_multi(const uint8_t* u, const uint8_t* R_uvz, const uint8_t* PK_uvz,
                                const uint8_t* Gen_uvz, const uint8_t* m, size_t n, uint8_t* ok) {
  if (int r = check_n(n)) return r;
  if (n == 0) return DSV_OK;
  if (!u || !R_uvz || !PK_uvz || !Gen_uvz || !m || !ok)
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  return run_multi(n, [=](Context& ctx, size_t off, size_t cnt) {
    return verify_vargen_ext_host(ctx, u + 32 * off, R_uvz + 96 * off, PK_uvz + 96 * off,
                                  Gen_uvz + 96 * off, m + 32 * off, cnt, ok + off);
  });
}
int dsv_verify_single_ext_dev(const void* u, const void* R_uvz, const void* PK_uvz, const void* m,
                              size_t n, void* ok, void* workspace, void* stream) {
  if (n && (!u || !R_uvz || !PK_uvz || !m || !ok || !workspace))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, ok);
  const void* pts[2] = {R_uvz, PK_uvz};
  return verify_ext_on(ctx, 0, u, pts, m, n, ok, workspace, (hipStream_t)stream);
}
int dsv_verify_double_ext_dev(const void* u, const void* R_uvz, const void* Rp_uvz, const void* PK_uvz,
                              const void* PKp_uvz, const void* m, size_t n, void* ok, void* workspace,
                              void* stream) {
  if (n && (!u || !R_uvz || !Rp_uvz || !PK_uvz || !PKp_uvz || !m || !ok || !workspace))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, ok);
  const void* pts[4] = {R_uvz, Rp_uvz, PK_uvz, PKp_uvz};
  return verify_ext_on(ctx, 1, u, pts, m, n, ok, workspace, (hipStream_t)stream);
}
int dsv_verify_vargen_ext_dev(const void* u, const void* R_uvz, const void* PK_uvz, const void* Gen_uvz,
                              const void* m, size_t n, void* ok, void* workspace, void* stream) {
  if (n && (!u || !R_uvz || !PK_uvz || !Gen_uvz || !m || !ok || !workspace))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, ok);
  const void* pts[3] = {R_uvz, PK_uvz, Gen_uvz};
  return verify_ext_on(ctx, 2, u, pts, m, n, ok, workspace, (hipStream_t)stream);
}

// ---- the reference's in-memory representation: Montgomery limbs --------------------------------
// The Rust types hold every field element as `[u64; 4]` Montgomery limbs with R = 2^256
// (`BlsScalar(pub [u64; 4])`, dusk-bls12_381 0.13; `JubJubScalar`, the coordinates of
// `JubJubExtended`, dusk-jubjub 0.14 — /root/reference/Cargo.toml:25-26; the fields:
// src/signatures.rs:58-61, src/keys/public.rs:59).  `to_bytes()` is one Montgomery reduction per
// element — eight per single signature, fourteen per double one — on ONE host thread: ~30x below the
// engine.  The *_mont entry points take the limbs as they lie in memory:
//   points  (u R, v R, z R) : straight into k_normalize_uvz — a quotient does not see the common factor
//   u, m                    : two reductions per signature inside the same kernel (scalars_from_mont_item)
// so a binding copies bytes and nothing else; the *_mont_cols forms even take the typed objects
// where they lie (one strided column per field) and gather them into the pinned staging with the
// pipeline's copy threads — no intermediate structure of arrays on the host.
extern "C++" {
namespace {
size_t mont_workspace_bytes(size_t n) { return 2 * align_up(n * 32, 256) + ext_workspace_bytes(n); }
int verify_mont_on(Context& ctx, int kind, const void* u, const void* const* pts_uvz, const void* m,
                   size_t n, void* ok, void* workspace, hipStream_t s) {
  Stager st(static_cast<uint8_t*>(workspace));
  uint8_t *cu = st.take(n * 32), *cm = st.take(n * 32);
  return verify_ext_on(ctx, kind, cu, pts_uvz, cm, n, ok, st.take(0), s, (const uint8_t*)u, (const uint8_t*)m);
}
// ins: u, points..., m — dense arrays or strided columns of typed objects
template <size_t NIN>
int verify_mont_host(Context& ctx, int kind, const HostIn (&ins)[NIN], size_t n, uint8_t* ok) {
  Context* cp = &ctx;
  return run_pipelined(ctx, ins, ok, n, kMontItemBytes, 0,
                       [kind, cp](const void* const* d, size_t cnt, Stager& x, hipStream_t st, Staged& g) {
                         const int np = kind == 0 ? 2 : (kind == 1 ? 4 : 3);
                         uint8_t *cu = x.take(cnt * 32), *cm = x.take(cnt * 32);
                         (void)np;
                         return prep_normalize(*cp, kind, d, cnt, x, st, g, cu, cm, true);  // ONE launch
                       },
                       DSV_PART(kind), kind != 0 ? kPipeHeavy : 0u);
}
}  // namespace
namespace dsvh {
// columns of one scheme: u (32 B), its points (96 B each), m (32 B)
constexpr int kMontCols[3] = {4, 6, 5};
int check_cols(int kind, const dsv_column* cols, size_t n, const uint8_t* ok) {
  if (int r = check_n(n)) return r;
  if (n == 0) return DSV_OK;
  if (!cols || !ok) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  const int nc = kMontCols[kind];
  for (int k = 0; k < nc; k++) {
    const size_t width = (k == 0 || k == nc - 1) ? 32 : 96;
    if (!cols[k].base) return fail(DSV_ERR_INVALID_ARGUMENT, "column %d: null pointer", k);
    if (cols[k].stride < width) return fail(DSV_ERR_INVALID_ARGUMENT, "column %d: stride %zu < %zu", k, cols[k].stride, width);
  }
  return DSV_OK;
}
}  // namespace dsvh
namespace {
// one shard [off, off + cnt) of a column batch on one device
int verify_mont_cols_shard(Context& ctx, int kind, const dsv_column* cols, size_t off, size_t cnt, uint8_t* ok) {
  auto in = [&](int k, size_t width) {
    return HostIn{static_cast<const uint8_t*>(cols[k].base) + off * cols[k].stride, width, cols[k].stride};
  };
  if (kind == 0) {
    const HostIn ins[4] = {in(0, 32), in(1, 96), in(2, 96), in(3, 32)};
    return verify_mont_host(ctx, 0, ins, cnt, ok + off);
  }
  if (kind == 1) {
    const HostIn ins[6] = {in(0, 32), in(1, 96), in(2, 96), in(3, 96), in(4, 96), in(5, 32)};
    return verify_mont_host(ctx, 1, ins, cnt, ok + off);
  }
  const HostIn ins[5] = {in(0, 32), in(1, 96), in(2, 96), in(3, 96), in(4, 32)};
  return verify_mont_host(ctx, 2, ins, cnt, ok + off);
}
}  // namespace
namespace dsvh {
int verify_mont_cols(int kind, const dsv_column* cols, size_t n, uint8_t* ok, bool multi) {
  if (int r = check_cols(kind, cols, n, ok)) return r;
  if (n == 0) return DSV_OK;
  if (multi)
    return run_multi(n, [=](Context& ctx, size_t off, size_t cnt) {
      return verify_mont_cols_shard(ctx, kind, cols, off, cnt, ok);
    });
  Context* ctxp = nullptr;
  if (int r = host_context(ctxp)) return r;
  return verify_mont_cols_shard(*ctxp, kind, cols, 0, n, ok);
}
}  // namespace dsvh
namespace {
}  // namespace
}  // extern "C++"

size_t dsv_mont_workspace_bytes(size_t n) { return mont_workspace_bytes(n); }

// copy threads of the host entry points (per process): n >= 1 sets, 0 restores the default
// ($DSV_HOST_THREADS, else 4); returns the value now in force
int dsv_set_host_threads(int n) {
  g_host_threads.store(n > 0 ? clamp_host_threads(n) : 0, std::memory_order_relaxed);
  return host_copy_threads();
}

// ---- asynchronous form: submit returns at once, the batch runs on a library-owned driver thread ----
// What a caller with a stream of batches uses to keep the GPU busy across calls: while batch k's last
// chunks are on the GPU, batch k + 1's driver already gathers, transfers and enqueues its first ones
// (each call in flight owns a Pipe; the compute lanes are shared, so the GPU sees one FIFO of
// sub-batches).  A third submit simply waits for a pipe inside its driver thread.
struct dsv_job {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  bool started = false;
  std::atomic<bool> finished{false};
  int rc = DSV_OK;
  std::string err;
  int kind = 0;
  dsv_column cols[6] = {};
  size_t n = 0;
  uint8_t* ok = nullptr;
};
namespace {
int submit_mont_cols(int kind, const dsv_column* cols, size_t n, uint8_t* ok, dsv_job** out) {
  if (!out) return fail(DSV_ERR_INVALID_ARGUMENT, "null job pointer");
  *out = nullptr;
  if (int r = check_cols(kind, cols, n, ok)) return r;
  if (g_primary.load(std::memory_order_acquire) < 0)
    return fail(DSV_ERR_NOT_INITIALIZED, "dsv_init() has not been called");
  dsv_job* j = new (std::nothrow) dsv_job;
  if (!j) return fail(DSV_ERR_HIP, "out of host memory");
  j->kind = kind;
  j->n = n;
  j->ok = ok;
  for (int k = 0; n && k < kMontCols[kind]; k++) j->cols[k] = cols[k];
  auto job_count = [](int d) {
    std::lock_guard<std::mutex> lk(g_jobs_mu);
    g_jobs += d;
    if (g_jobs == 0) g_jobs_cv.notify_all();
  };
  job_count(+1);
  try {
    j->th = std::thread([j, job_count] {
      {
        std::lock_guard<std::mutex> lk(j->m);
        j->started = true;
      }
      j->cv.notify_all();
      {  // the driver gathers the first shard itself: next to the device that takes it
        const int d = t_device >= 0 ? t_device : g_primary.load(std::memory_order_acquire);
        if (d >= 0 && d < kMaxDevices) (void)pin_this_thread(g_ctx[d].numa_cpus);
      }
      j->rc = verify_mont_cols(j->kind, j->cols, j->n, j->ok, true);
      if (j->rc) j->err = g_err;  // the text lives in this thread's thread-local
      j->finished.store(true, std::memory_order_release);
      job_count(-1);
    });
  } catch (...) {
    job_count(-1);
    delete j;
    return fail(DSV_ERR_HIP, "could not start the driver thread of the batch");
  }
  {
    // jobs take their place in the device's queue in submission order: return once the driver runs
    // (it queues for its pipe within microseconds; the next submit has a thread to start first)
    std::unique_lock<std::mutex> lk(j->m);
    j->cv.wait(lk, [j] { return j->started; });
  }
  *out = j;
  return DSV_OK;
}
}  // namespace
int dsv_verify_single_mont_cols_submit(const dsv_column* cols, size_t n, uint8_t* ok, dsv_job** job) { return submit_mont_cols(0, cols, n, ok, job); }
int dsv_verify_double_mont_cols_submit(const dsv_column* cols, size_t n, uint8_t* ok, dsv_job** job) { return submit_mont_cols(1, cols, n, ok, job); }
int dsv_verify_vargen_mont_cols_submit(const dsv_column* cols, size_t n, uint8_t* ok, dsv_job** job) { return submit_mont_cols(2, cols, n, ok, job); }
int dsv_job_done(const dsv_job* job) {
  if (!job) return fail(DSV_ERR_INVALID_ARGUMENT, "null job");
  return job->finished.load(std::memory_order_acquire) ? 1 : 0;
}
int dsv_job_wait(dsv_job* job) {
  if (!job) return fail(DSV_ERR_INVALID_ARGUMENT, "null job");
  if (job->th.joinable()) job->th.join();
  const int rc = job->rc;
  if (rc) g_err = job->err;
  delete job;
  return rc;
}
int dsv_max_in_flight(void) { return kPipes; }

int dsv_verify_single_mont_cols(const dsv_column* cols, size_t n, uint8_t* ok) { return verify_mont_cols(0, cols, n, ok, true); }
int dsv_verify_double_mont_cols(const dsv_column* cols, size_t n, uint8_t* ok) { return verify_mont_cols(1, cols, n, ok, true); }
int dsv_verify_vargen_mont_cols(const dsv_column* cols, size_t n, uint8_t* ok) { return verify_mont_cols(2, cols, n, ok, true); }

#define DSV_DENSE_COLS_SINGLE {{u, 32}, {R_uvz, 96}, {PK_uvz, 96}, {m, 32}}
#define DSV_DENSE_COLS_DOUBLE {{u, 32}, {R_uvz, 96}, {Rp_uvz, 96}, {PK_uvz, 96}, {PKp_uvz, 96}, {m, 32}}
#define DSV_DENSE_COLS_VARGEN {{u, 32}, {R_uvz, 96}, {PK_uvz, 96}, {Gen_uvz, 96}, {m, 32}}
int dsv_verify_single_mont(const uint8_t* u, const uint8_t* R_uvz, const uint8_t* PK_uvz, const uint8_t* m,
                           size_t n, uint8_t* ok) {
  const dsv_column cols[4] = DSV_DENSE_COLS_SINGLE;
  return verify_mont_cols(0, cols, n, ok, false);
}
int dsv_verify_double_mont(const uint8_t* u, const uint8_t* R_uvz, const uint8_t* Rp_uvz,
                           const uint8_t* PK_uvz, const uint8_t* PKp_uvz, const uint8_t* m, size_t n,
                           uint8_t* ok) {
  const dsv_column cols[6] = DSV_DENSE_COLS_DOUBLE;
  return verify_mont_cols(1, cols, n, ok, false);
}
int dsv_verify_vargen_mont(const uint8_t* u, const uint8_t* R_uvz, const uint8_t* PK_uvz,
                           const uint8_t* Gen_uvz, const uint8_t* m, size_t n, uint8_t* ok) {
  const dsv_column cols[5] = DSV_DENSE_COLS_VARGEN;
  return verify_mont_cols(2, cols, n, ok, false);
}
int dsv_verify_single_mont_multi(const uint8_t* u, const uint8_t* R_uvz, const uint8_t* PK_uvz,
                                 const uint8_t* m, size_t n, uint8_t* ok) {
  const dsv_column cols[4] = DSV_DENSE_COLS_SINGLE;
  return verify_mont_cols(0, cols, n, ok, true);
}
int dsv_verify_double_mont_multi(const uint8_t* u, const uint8_t* R_uvz, const uint8_t* Rp_uvz,
                                 const uint8_t* PK_uvz, const uint8_t* PKp_uvz, const uint8_t* m, size_t n,
                                 uint8_t* ok) {
  const dsv_column cols[6] = DSV_DENSE_COLS_DOUBLE;
  return verify_mont_cols(1, cols, n, ok, true);
}
int dsv_verify_vargen_mont_multi(const uint8_t* u, const uint8_t* R_uvz, const uint8_t* PK_uvz,
                                 const uint8_t* Gen_uvz, const uint8_t* m, size_t n, uint8_t* ok) {
  const dsv_column cols[5] = DSV_DENSE_COLS_VARGEN;
  return verify_mont_cols(2, cols, n, ok, true);
}
int dsv_verify_single_mont_dev(const void* u, const void* R_uvz, const void* PK_uvz, const void* m,
                               size_t n, void* ok, void* workspace, void* stream) {
  if (n && (!u || !R_uvz || !PK_uvz || !m || !ok || !workspace))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, ok);
  const void* pts[2] = {R_uvz, PK_uvz};
  return verify_mont_on(ctx, 0, u, pts, m, n, ok, workspace, (hipStream_t)stream);
}
int dsv_verify_double_mont_dev(const void* u, const void* R_uvz, const void* Rp_uvz, const void* PK_uvz,
                               const void* PKp_uvz, const void* m, size_t n, void* ok, void* workspace,
                               void* stream) {
  if (n && (!u || !R_uvz || !Rp_uvz || !PK_uvz || !PKp_uvz || !m || !ok || !workspace))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, ok);
  const void* pts[4] = {R_uvz, Rp_uvz, PK_uvz, PKp_uvz};
  return verify_mont_on(ctx, 1, u, pts, m, n, ok, workspace, (hipStream_t)stream);
}
int dsv_verify_vargen_mont_dev(const void* u, const void* R_uvz, const void* PK_uvz, const void* Gen_uvz,
                               const void* m, size_t n, void* ok, void* workspace, void* stream) {
  if (n && (!u || !R_uvz || !PK_uvz || !Gen_uvz || !m || !ok || !workspace))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, ok);
  const void* pts[3] = {R_uvz, PK_uvz, Gen_uvz};
  return verify_mont_on(ctx, 2, u, pts, m, n, ok, workspace, (hipStream_t)stream);
}

int dsv_challenge_single(const uint8_t* R_uv, const uint8_t* m, size_t n, uint8_t* c) {
  if (n && (!R_uv || !m || !c)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  DSV_HOST_LOCK();
  if (int r = ensure_stage(ctx, align_up(n * 64, 256) + 2 * align_up(n * 32, 256))) return r;
  Stager st(ctx.stage);
  uint8_t *dR = st.take(n * 64), *dm = st.take(n * 32), *dc = st.take(n * 32);
  H2D(dR, R_uv, n * 64);
  H2D(dm, m, n * 32);
  launch_challenge(false, (const uint8_t*)dR, (const uint8_t*)nullptr, (const uint8_t*)dm, n, dc, (uint8_t*)nullptr, 0);
  HIP_TRY(hipGetLastError());
  D2H(c, dc, n * 32);
  HIP_TRY(hipStreamSynchronize(0));
  return DSV_OK;
}
int dsv_challenge_double(const uint8_t* R_uv, const uint8_t* Rp_uv, const uint8_t* m, size_t n,
                         uint8_t* c) {
  if (n && (!R_uv || !Rp_uv || !m || !c)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  DSV_HOST_LOCK();
  if (int r = ensure_stage(ctx, 2 * align_up(n * 64, 256) + 2 * align_up(n * 32, 256))) return r;
  Stager st(ctx.stage);
  uint8_t *dR = st.take(n * 64), *dRp = st.take(n * 64), *dm = st.take(n * 32),
          *dc = st.take(n * 32);
  H2D(dR, R_uv, n * 64);
  H2D(dRp, Rp_uv, n * 64);
  H2D(dm, m, n * 32);
  launch_challenge(true, (const uint8_t*)dR, (const uint8_t*)dRp, (const uint8_t*)dm, n, dc, (uint8_t*)nullptr, 0);
  HIP_TRY(hipGetLastError());
  D2H(c, dc, n * 32);
  HIP_TRY(hipStreamSynchronize(0));
  return DSV_OK;
}


}  // extern "C"
