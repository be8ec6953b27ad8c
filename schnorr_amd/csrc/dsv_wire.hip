// dsv_wire.hip — serialized records (`Signature::from_bytes`, `PublicKey::from_bytes`:
// /root/reference/src/signatures.rs:117-122, :261-269, :398-403; src/keys/public.rs:94-100, :294-298,
// :361-371): point decompression on the device (k_decompress), then the per-signature kernels.
#include "dsv_pipeline.h"

using namespace dsvh;

extern "C" {

// ---- wire formats ---------------------------------------------------------------------------
int dsv_decompress_points_dev(const void* in, size_t in_stride, size_t n, void* out_uv, void* ok,
                              int accumulate, void* stream) {
  DSV_DEV_PROLOGUE(n, out_uv);
  return decompress_on(ctx, in, in_stride, n, out_uv, ok, accumulate, (hipStream_t)stream);
}
// JubJubAffine::to_bytes: canonical v with bit 255 = lowest bit of canonical u.  Pure byte
// shuffling on affine input, so it runs on the host.
int dsv_compress_points(const uint8_t* in_uv, size_t n, uint8_t* out32) {
  if (int r = check_n(n)) return r;
  if (n == 0) return DSV_OK;
  if (!in_uv || !out32) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  for (size_t i = 0; i < n; i++) {
    memcpy(out32 + 32 * i, in_uv + 64 * i + 32, 32);
    out32[32 * i + 31] |= (uint8_t)((in_uv[64 * i] & 1) << 7);
  }
  return DSV_OK;
}

int dsv_decompress_points(const uint8_t* in32, size_t n, uint8_t* out_uv, uint8_t* ok) {
  if (n && (!in32 || !out_uv || !ok)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  DSV_HOST_LOCK();
  if (int r = ensure_stage(ctx, align_up(n * 32, 256) + align_up(n * 64, 256) + align_up(n, 256))) return r;
  Stager st(ctx.stage);
  uint8_t *din = st.take(n * 32), *dout = st.take(n * 64), *dok = st.take(n);
  H2D(din, in32, n * 32);
  if (int r = decompress_on(ctx, din, 32, n, dout, dok, 0, 0)) return r;
  D2H(out_uv, dout, n * 64);
  D2H(ok, dok, n);
  HIP_TRY(hipStreamSynchronize(0));
  return DSV_OK;
}

namespace {
// shared body of the *_wire entry points.  sig: n records of sig_bytes = 32 (u) + 32*n_sig_points;
// pk: n records of 32*n_pk_points compressed points.  kind: 0 single, 1 double, 2 vargen.
constexpr size_t kWireItemBytes = 32 + 4 * 64 + 1;
int verify_wire_on(Context& ctx, int kind, const uint8_t* dsig, const uint8_t* dpk, const void* dm,
                   size_t cnt, void* dok, const WireWs& x, void* vws, hipStream_t st) {
  const size_t sig_bytes = kind == 1 ? 96 : 64, pk_bytes = kind == 0 ? 32 : 64;
  launch_gather32(dsig, sig_bytes, cnt, x.u, st);
  if (int r = decompress_on(ctx, dsig + 32, sig_bytes, cnt, x.R, x.valid, 0, st)) return r;
  if (kind == 1)
    if (int r = decompress_on(ctx, dsig + 64, sig_bytes, cnt, x.Rp, x.valid, 1, st)) return r;
  if (int r = decompress_on(ctx, dpk, pk_bytes, cnt, x.P0, x.valid, 1, st)) return r;
  if (kind != 0)
    if (int r = decompress_on(ctx, dpk + 32, pk_bytes, cnt, x.P1, x.valid, 1, st)) return r;
  int rc;
  if (kind == 0) rc = verify_single_on(ctx, x.u, x.R, x.P0, dm, cnt, dok, vws, st, x.valid);
  else if (kind == 1) rc = verify_double_on(ctx, x.u, x.R, x.Rp, x.P0, x.P1, dm, cnt, dok, vws, st, x.valid);
  else rc = verify_vargen_on(ctx, x.u, x.R, x.P0, x.P1, dm, cnt, dok, vws, st, x.valid);
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
// device-pointer form: serialized records already resident in HBM
int verify_wire_dev(int kind, const void* sig, const void* pk, const void* m, size_t n, void* ok,
                    void* workspace, void* stream) {
  if (n && (!sig || !pk || !m || !ok || !workspace)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  if (((uintptr_t)sig | (uintptr_t)pk) & 15) return fail(DSV_ERR_INVALID_ARGUMENT, "records must be 16-byte aligned");
  DSV_DEV_PROLOGUE(n, ok);
  Stager x(static_cast<uint8_t*>(workspace));
  WireWs w;
  w.u = x.take(n * 32);
  w.R = x.take(n * 64);
  w.Rp = x.take(n * 64);
  w.P0 = x.take(n * 64);
  w.P1 = x.take(n * 64);
  w.valid = x.take(n);
  void* vws = x.take(dsv_workspace_bytes(n));
  return verify_wire_on(ctx, kind, (const uint8_t*)sig, (const uint8_t*)pk, m, n, ok, w, vws,
                        (hipStream_t)stream);
}
}  // namespace

extern "C++" {
namespace dsvh {
int verify_wire(Context& ctx, int kind, const uint8_t* sig, const uint8_t* pk, const uint8_t* m,
                size_t n, uint8_t* ok) {
  const size_t sig_bytes = kind == 1 ? 96 : 64, pk_bytes = kind == 0 ? 32 : 64;
  const HostIn ins[3] = {{sig, sig_bytes}, {pk, pk_bytes}, {m, 32}};
  Context* cp = &ctx;
  return run_pipelined(ctx, ins, ok, n, 0, kWireItemBytes, NoPrep{},
                       [=](const Staged& g, size_t off, size_t cnt, void* dok, void* ws, Stager& x, hipStream_t st) {
    WireWs w;
    w.u = x.take(cnt * 32);
    w.R = x.take(cnt * 64);
    w.Rp = x.take(cnt * 64);
    w.P0 = x.take(cnt * 64);
    w.P1 = x.take(cnt * 64);
    w.valid = x.take(cnt);
    return verify_wire_on(*cp, kind, g.p[0] + off * g.bytes[0], g.p[1] + off * g.bytes[1],
                          g.p[2] + off * g.bytes[2], cnt, dok, w, ws, st);
  });
}
}  // namespace dsvh
}  // extern "C++"

size_t dsv_wire_workspace_bytes(size_t n) {
  return align_up(n * 32, 256) + 4 * align_up(n * 64, 256) + align_up(n, 256) +
         align_up(dsv_workspace_bytes(n), 256) + 256;
}
int dsv_verify_single_wire_dev(const void* sig64, const void* pk32, const void* m, size_t n, void* ok,
                               void* workspace, void* stream) {
  return verify_wire_dev(0, sig64, pk32, m, n, ok, workspace, stream);
}
int dsv_verify_double_wire_dev(const void* sig96, const void* pk64, const void* m, size_t n, void* ok,
                               void* workspace, void* stream) {
  return verify_wire_dev(1, sig96, pk64, m, n, ok, workspace, stream);
}
int dsv_verify_vargen_wire_dev(const void* sig64, const void* pk64, const void* m, size_t n, void* ok,
                               void* workspace, void* stream) {
  return verify_wire_dev(2, sig64, pk64, m, n, ok, workspace, stream);
}

int dsv_verify_single_wire(const uint8_t* sig64, const uint8_t* pk32, const uint8_t* m, size_t n,
                           uint8_t* ok) {
  if (n && (!sig64 || !pk32 || !m || !ok)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  return verify_wire(ctx, 0, sig64, pk32, m, n, ok);
}
int dsv_verify_double_wire(const uint8_t* sig96, const uint8_t* pk64, const uint8_t* m, size_t n,
                           uint8_t* ok) {
  if (n && (!sig96 || !pk64 || !m || !ok)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  return verify_wire(ctx, 1, sig96, pk64, m, n, ok);
}
int dsv_verify_vargen_wire(const uint8_t* sig64, const uint8_t* pk64, const uint8_t* m, size_t n,
                           uint8_t* ok) {
  if (n && (!sig64 || !pk64 || !m || !ok)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  return verify_wire(ctx, 2, sig64, pk64, m, n, ok);
}


}  // extern "C"
