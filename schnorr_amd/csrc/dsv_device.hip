// dsv_device.hip — the device-pointer entry points of include/dsv.h: challenge hash, the three verify
// schemes over affine inputs resident in HBM (`PublicKey::verify`, `PublicKeyDouble::verify`,
// `PublicKeyVarGen::verify`, /root/reference/src/keys/public.rs:121-130, :222-244, :401-415), the second
// stage alone, and mixed single + double batches (device-side split by kind).  The bodies
// verify_*_on are what the host pipeline and the other units call per sub-batch.
#include "dsv_host.h"

namespace dsvh {
// bodies shared by the device-pointer entry points and the host pipeline (context resolved)
// Small batches (eight-lane kernel): the window tables of (PK, R) do not depend on the challenge, so
// they are built on one of the lane's internal streams WHILE k_challenge runs on the caller's: fork
// by an event here; the caller enqueues the hash, then waits for the returned join event on its own
// stream in front of the verify kernel.  nullptr: not a small batch (or the overlap is off, or no
// lane could be had) — the verify kernel then builds its tables itself.
thread_local bool t_pipeline_part = false;
thread_local size_t t_chunk_first = 0;
hipEvent_t prep_tables_beside_hash(Context& ctx, const void* PK_uv, const void* R_uv, size_t n,
                                   u32* tables, hipStream_t user) {
  if (!(ctx.quad && ctx.small_overlap && n <= kQuadMaxItems) || t_pipeline_part) return nullptr;
  SplitLane* lane = nullptr;
  if (acquire_lane(ctx, user, lane) != DSV_OK) return nullptr;
  std::lock_guard<std::mutex> lk(ctx.lane_mu);
  if (hipEventRecord(lane->fork, user) != hipSuccess ||
      hipStreamWaitEvent(lane->side, lane->fork, 0) != hipSuccess)
    return nullptr;
  launch_prep_var_tables((const uint8_t*)PK_uv, (const uint8_t*)R_uv, n, tables, lane->side);
  if (hipEventRecord(lane->side_join, lane->side) != hipSuccess) {
    // the prep kernel is already writing this call's table slots: it must have finished before the
    // verify kernel, told to build its tables itself, writes the same slots from the caller's stream
    (void)hipStreamSynchronize(lane->side);
    return nullptr;
  }
  return lane->side_join;
}
// valid_in (may be null): per-item validity found by an earlier stage (normalisation, decompression);
// the hash kernel folds it into the validity the verify kernel starts from
int verify_single_on(Context& ctx, const void* u, const void* R_uv, const void* PK_uv, const void* m,
                     size_t n, void* ok, void* workspace, hipStream_t stream, const uint8_t* valid_in) {
  const uint8_t *pu = (const uint8_t*)u, *pR = (const uint8_t*)R_uv, *pPK = (const uint8_t*)PK_uv,
                *pm = (const uint8_t*)m;
  uint8_t* pok = (uint8_t*)ok;
  Context* cp = &ctx;
  return run_split(ctx, n, workspace, stream,
                   [=](size_t off, size_t cnt, const Workspace& w, hipStream_t s) {
    // (only for an unsplit call: inside run_split's loop the lane lock is held and cnt > 2^14 anyway,
    //  except for a short last part, which simply builds its tables in the kernel)
    hipEvent_t ready = cnt == n ? prep_tables_beside_hash(*cp, pPK + 64 * off, pR + 64 * off, cnt, w.tables, s)
                                : nullptr;
    launch_challenge(false, pR + 64 * off, (const uint8_t*)nullptr, pm + 32 * off, cnt, w.c, w.valid, s,
                     valid_in ? valid_in + off : nullptr);
    if (ready && hipStreamWaitEvent(s, ready, 0) != hipSuccess) return;  // (surfaces through hipGetLastError)
    launch_verify_fixed(*cp, false, pu + 32 * off, w.c, pPK + 64 * off, pR + 64 * off, 0, w.valid,
                        cnt, pok + off, w.tables, s, ready != nullptr);
  });
}
int verify_double_on(Context& ctx, const void* u, const void* R_uv, const void* Rp_uv,
                     const void* PK_uv, const void* PKp_uv, const void* m, size_t n, void* ok,
                     void* workspace, hipStream_t stream, const uint8_t* valid_in) {
  const uint8_t *pu = (const uint8_t*)u, *pR = (const uint8_t*)R_uv, *pRp = (const uint8_t*)Rp_uv,
                *pPK = (const uint8_t*)PK_uv, *pPKp = (const uint8_t*)PKp_uv, *pm = (const uint8_t*)m;
  uint8_t* pok = (uint8_t*)ok;
  Context* cp = &ctx;
  return run_split(ctx, n, workspace, stream,
                   [=](size_t off, size_t cnt, const Workspace& w, hipStream_t s) {
    // (only for an unsplit call: inside run_split's loop the lane lock is held and cnt > 2^14 anyway,
    //  except for a short last part, which simply builds its tables in the kernel)
    hipEvent_t ready = cnt == n ? prep_tables_beside_hash(*cp, pPK + 64 * off, pR + 64 * off, cnt, w.tables, s)
                                : nullptr;
    launch_challenge(true, pR + 64 * off, pRp + 64 * off, pm + 32 * off, cnt, w.c, w.valid, s,
                     valid_in ? valid_in + off : nullptr);
    if (ready && hipStreamWaitEvent(s, ready, 0) != hipSuccess) return;
    launch_verify_fixed_double(*cp, pu + 32 * off, w.c, pPK + 64 * off, pR + 64 * off,
                               pPKp + 64 * off, pRp + 64 * off, w.valid, cnt, pok + off, w.tables, s,
                               ready != nullptr);
  });
}
int verify_vargen_on(Context& ctx, const void* u, const void* R_uv, const void* PK_uv,
                     const void* Gen_uv, const void* m, size_t n, void* ok, void* workspace,
                     hipStream_t stream, const uint8_t* valid_in) {
  const uint8_t *pu = (const uint8_t*)u, *pR = (const uint8_t*)R_uv, *pPK = (const uint8_t*)PK_uv,
                *pG = (const uint8_t*)Gen_uv, *pm = (const uint8_t*)m;
  uint8_t* pok = (uint8_t*)ok;
  return run_split(ctx, n, workspace, stream,
                   [=](size_t off, size_t cnt, const Workspace& w, hipStream_t s) {
    launch_challenge(false, pR + 64 * off, (const uint8_t*)nullptr, pm + 32 * off, cnt, w.c, w.valid, s,
                     valid_in ? valid_in + off : nullptr);
    launch_verify_var(pu + 32 * off, (const uint8_t*)w.c, pPK + 64 * off, pG + 64 * off, pR + 64 * off, (const uint8_t*)w.valid, cnt, pok + off, w.tables, s);
  });
}
int decompress_on(Context& ctx, const void* in, size_t in_stride, size_t n, void* out_uv, void* ok,
                  int accumulate, hipStream_t stream) {
  if (!in || !out_uv || !ok || in_stride < 32 || (in_stride & 15) || ((uintptr_t)in & 15))
    return fail(DSV_ERR_INVALID_ARGUMENT, "bad pointer / stride (need 16-byte alignment)");
  launch_decompress((const uint8_t*)in, in_stride, n, (uint8_t*)out_uv, (uint8_t*)ok, accumulate, ctx.ts_cancel, ctx.ts_hash, stream);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
}  // namespace dsvh

using namespace dsvh;

extern "C" {

int dsv_challenge_single_dev(const void* R_uv, const void* m, size_t n, void* c, void* valid,
                             void* stream) {
  if (n && (!R_uv || !m || !c)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, c);
  launch_challenge(false, (const uint8_t*)R_uv, (const uint8_t*)nullptr, (const uint8_t*)m, n, (uint8_t*)c, (uint8_t*)valid, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
int dsv_challenge_double_dev(const void* R_uv, const void* Rp_uv, const void* m, size_t n, void* c,
                             void* valid, void* stream) {
  if (n && (!R_uv || !Rp_uv || !m || !c)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, c);
  launch_challenge(true, (const uint8_t*)R_uv, (const uint8_t*)Rp_uv, (const uint8_t*)m, n, (uint8_t*)c, (uint8_t*)valid, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}


int dsv_verify_single_dev(const void* u, const void* R_uv, const void* PK_uv, const void* m,
                          size_t n, void* ok, void* workspace, void* stream) {
  if (n && (!u || !R_uv || !PK_uv || !m || !ok || !workspace))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, ok);
  return verify_single_on(ctx, u, R_uv, PK_uv, m, n, ok, workspace, (hipStream_t)stream);
}

// second stage alone (c and valid already computed): lets callers time / profile the dominant
// kernel separately, and re-use one challenge for several key pairs
int dsv_verify_core_dev(const void* u, const void* c, const void* valid, const void* PK_uv,
                        const void* R_uv, int which, int accumulate, size_t n, void* ok,
                        void* workspace, void* stream) {
  if (n && (!u || !c || !valid || !PK_uv || !R_uv || !ok || !workspace || which < 0 || which > 1))
    return fail(DSV_ERR_INVALID_ARGUMENT, "bad argument");
  DSV_DEV_PROLOGUE(n, ok);
  Workspace w = carve(workspace, n);
  launch_verify_fixed(ctx, accumulate != 0, u, c, PK_uv, R_uv, which, valid, n, ok, w.tables,
                      (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
// both equations of a double signature from a precomputed challenge, one launch
int dsv_verify_core_double_dev(const void* u, const void* c, const void* valid, const void* PK_uv,
                               const void* R_uv, const void* PKp_uv, const void* Rp_uv, size_t n,
                               void* ok, void* workspace, void* stream) {
  if (n && (!u || !c || !valid || !PK_uv || !R_uv || !PKp_uv || !Rp_uv || !ok || !workspace))
    return fail(DSV_ERR_INVALID_ARGUMENT, "bad argument");
  DSV_DEV_PROLOGUE(n, ok);
  Workspace w = carve(workspace, n);
  launch_verify_fixed_double(ctx, u, c, PK_uv, R_uv, PKp_uv, Rp_uv, valid, n, ok, w.tables,
                             (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}

int dsv_verify_double_dev(const void* u, const void* R_uv, const void* Rp_uv, const void* PK_uv,
                          const void* PKp_uv, const void* m, size_t n, void* ok, void* workspace,
                          void* stream) {
  if (n && (!u || !R_uv || !Rp_uv || !PK_uv || !PKp_uv || !m || !ok || !workspace))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, ok);
  return verify_double_on(ctx, u, R_uv, Rp_uv, PK_uv, PKp_uv, m, n, ok, workspace, (hipStream_t)stream);
}

int dsv_verify_vargen_dev(const void* u, const void* R_uv, const void* PK_uv, const void* Gen_uv,
                          const void* m, size_t n, void* ok, void* workspace, void* stream) {
  if (n && (!u || !R_uv || !PK_uv || !Gen_uv || !m || !ok || !workspace))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, ok);
  return verify_vargen_on(ctx, u, R_uv, PK_uv, Gen_uv, m, n, ok, workspace, (hipStream_t)stream);
}

// ---- mixed batches: device-side split by kind ---------------------------------------------
namespace {
struct SplitScratch {
  u32* tile_counts;
  u32* totals;
};
size_t split_scratch_bytes(size_t n) {
  const size_t tiles = (n + kSplitTile - 1) / kSplitTile;
  return align_up(tiles * 8, 256) + 256;
}
SplitScratch carve_split(void* p, size_t n) {
  const size_t tiles = (n + kSplitTile - 1) / kSplitTile;
  SplitScratch s;
  s.tile_counts = static_cast<u32*>(p);
  s.totals = reinterpret_cast<u32*>(static_cast<uint8_t*>(p) + align_up(tiles * 8, 256));
  return s;
}
int split_on(const void* kinds, size_t n, void* idx_single, size_t cap_single, void* idx_double,
             size_t cap_double, void* scratch, hipStream_t s) {
  if ((uintptr_t)kinds & 15) return fail(DSV_ERR_INVALID_ARGUMENT, "kinds must be 16-byte aligned");
  SplitScratch sc = carve_split(scratch, n);
  launch_split_kinds((const uint8_t*)kinds, n, sc.tile_counts, sc.totals, (u32*)idx_single, cap_single,
                     (u32*)idx_double, cap_double, s);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
// limit (device u32, may be null): only min(count, *limit) index entries are dereferenced
int gather_on(const void* src, size_t src_rows, size_t row_bytes, const void* idx, size_t count,
              const void* limit, void* dst, hipStream_t s) {
  if (row_bytes == 0 || (row_bytes & 15) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15))
    return fail(DSV_ERR_INVALID_ARGUMENT, "rows must be multiples of 16 bytes, 16-byte aligned");
  if (count == 0) return DSV_OK;
  launch_gather_rows(src, src_rows, (u32)(row_bytes / 16), (const u32*)idx, count, (const u32*)limit, dst, s);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
}  // namespace

size_t dsv_split_scratch_bytes(size_t n) { return split_scratch_bytes(n); }

int dsv_split_kinds_dev(const void* kinds, size_t n, void* idx_single, size_t cap_single,
                        void* idx_double, size_t cap_double, void* scratch, void* stream) {
  if (n && (!kinds || !idx_single || !idx_double || !scratch))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, scratch);
  return split_on(kinds, n, idx_single, cap_single, idx_double, cap_double, scratch, (hipStream_t)stream);
}
int dsv_gather_rows_dev(const void* src, size_t src_rows, size_t row_bytes, const void* idx,
                        size_t count, const void* count_limit, void* dst, void* stream) {
  if (count && (!src || !idx || !dst)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(count, dst);
  return gather_on(src, src_rows, row_bytes, idx, count, count_limit, dst, (hipStream_t)stream);
}
int dsv_scatter_verdicts_dev(const void* src, const void* idx, size_t count, const void* count_limit,
                             void* dst, size_t dst_len, void* stream) {
  if (count && (!src || !idx || !dst)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(count, dst);
  launch_scatter_bytes((const uint8_t*)src, (const u32*)idx, count, (const u32*)count_limit, (uint8_t*)dst,
                       dst_len, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}

size_t dsv_mixed_workspace_bytes(size_t n) {
  // split scratch | idx_single[n] | idx_double[n] | compacted rows (<= 320 B per item) |
  // per-kind verdicts | verify workspace
  return split_scratch_bytes(n) + 2 * align_up(n * 4, 256) + 6 * align_up(n * 64, 256) +
         2 * align_up(n, 256) + align_up(dsv_workspace_bytes(n), 256) + 256;
}

// One batch holding single (kind 0) and double (kind 1) signatures in any interleaving, as a
// structure of arrays over ALL n items (Rp_uv / PKp_uv rows of single items are ignored).
// n_double = number of kind-1 items (the caller knows its batch); every other item must be kind 0.
extern "C++" {
namespace {
// fast: both kinds' groups through the batch fast accept (verify_rlc_on: enqueue-only); *accepted =
// every group of both kinds was decided by its aggregates
int verify_mixed_dev(const void* kinds, const void* u, const void* R_uv, const void* Rp_uv, const void* PK_uv,
                     const void* PKp_uv, const void* m, size_t n, size_t n_double, void* ok, void* workspace,
                     void* stream, bool fast, int* accepted) {
  if (n == 0 || !fast) {
    if (int r = rlc_clear_accepted(accepted)) return r;
  }
  if (n && (!kinds || !u || !R_uv || !Rp_uv || !PK_uv || !PKp_uv || !m || !ok || !workspace))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  if (n_double > n) return fail(DSV_ERR_INVALID_ARGUMENT, "n_double exceeds n");
  DSV_DEV_PROLOGUE(n, ok);
  RlcVerdictTarget vt;
  if (fast)
    if (int r = rlc_verdict_target(ctx, accepted, vt)) return r;
  hipStream_t s = (hipStream_t)stream;
  const size_t ns = n - n_double, nd = n_double;
  Stager st(static_cast<uint8_t*>(workspace));
  void* scratch = st.take(split_scratch_bytes(n));
  u32* idx_s = reinterpret_cast<u32*>(st.take(n * 4));
  u32* idx_d = reinterpret_cast<u32*>(st.take(n * 4));
  uint8_t *cu = st.take(n * 32), *cm = st.take(n * 32);   // singles first, doubles behind them
  uint8_t *cR = st.take(n * 64), *cPK = st.take(n * 64);
  uint8_t *cRp = st.take(n * 64), *cPKp = st.take(n * 64);  // doubles only
  uint8_t *oks = st.take(n), *okd = st.take(n);
  void* vws = st.take(fast ? dsv_rlc_workspace_bytes(n, 0) : dsv_workspace_bytes(n));
  HIP_TRY(hipMemsetAsync(ok, 0, n, s));  // items of an invalid kind keep verdict 0
  if (int r = split_on(kinds, n, idx_s, ns, idx_d, nd, scratch, s)) return r;
  // Only index entries the split really wrote are dereferenced: every gather / scatter is bounded
  // on the device by the split's own totals (and skips an index >= n), whatever the caller
  // declared.  With a wrong n_double the compacted rows are partly stale workspace bytes — the
  // verify kernels take any bytes (out-of-contract inputs never fault) and k_mixed_check zeroes
  // the whole verdict vector at the end.
  const u32* totals = carve_split(scratch, n).totals;
  struct Col { const void* src; size_t bytes; uint8_t* dst; };
  const Col single_cols[4] = {{u, 32, cu}, {m, 32, cm}, {R_uv, 64, cR}, {PK_uv, 64, cPK}};
  for (const Col& c : single_cols)
    if (int r = gather_on(c.src, n, c.bytes, idx_s, ns, totals, c.dst, s)) return r;
  const Col double_cols[6] = {{u, 32, cu + ns * 32},      {m, 32, cm + ns * 32},
                              {R_uv, 64, cR + ns * 64},   {PK_uv, 64, cPK + ns * 64},
                              {Rp_uv, 64, cRp},           {PKp_uv, 64, cPKp}};
  for (const Col& c : double_cols)
    if (int r = gather_on(c.src, n, c.bytes, idx_d, nd, totals + 1, c.dst, s)) return r;
  if (ns) {
    if (int r = fast ? verify_rlc_on(ctx, 0, cu, cR, nullptr, cPK, nullptr, nullptr, cm, ns, oks, vws, s, 0, vt.dev)
                     : verify_single_on(ctx, cu, cR, cPK, cm, ns, oks, vws, s))
      return r;
    launch_scatter_bytes(oks, idx_s, ns, totals, (uint8_t*)ok, n, s);
  }
  if (nd) {
    if (int r = fast ? verify_rlc_on(ctx, 1, cu + ns * 32, cR + ns * 64, cRp, cPK + ns * 64, cPKp, nullptr,
                                     cm + ns * 32, nd, okd, vws, s, 0, vt.dev, /*and_into*/ ns != 0)
                     : verify_double_on(ctx, cu + ns * 32, cR + ns * 64, cRp, cPK + ns * 64, cPKp, cm + ns * 32, nd,
                                        okd, vws, s))
      return r;
    launch_scatter_bytes(okd, idx_d, nd, totals + 1, (uint8_t*)ok, n, s);
  }
  launch_mixed_check(totals, (u32)ns, (u32)nd, (uint8_t*)ok, n, s);
  HIP_TRY(hipGetLastError());
  return fast ? rlc_verdict_wait(vt, s) : (int)DSV_OK;
}
}  // namespace
}  // extern "C++"
int dsv_verify_mixed_dev(const void* kinds, const void* u, const void* R_uv, const void* Rp_uv,
                         const void* PK_uv, const void* PKp_uv, const void* m, size_t n,
                         size_t n_double, void* ok, void* workspace, void* stream) {
  return verify_mixed_dev(kinds, u, R_uv, Rp_uv, PK_uv, PKp_uv, m, n, n_double, ok, workspace, stream, false, nullptr);
}
// the same with each kind's items through the batch fast accept (a wrong n_double: every verdict 0, as above)
size_t dsv_mixed_rlc_workspace_bytes(size_t n) {
  return dsv_mixed_workspace_bytes(n) - align_up(dsv_workspace_bytes(n), 256) + align_up(dsv_rlc_workspace_bytes(n, 0), 256);
}
int dsv_verify_mixed_rlc_dev(const void* kinds, const void* u, const void* R_uv, const void* Rp_uv,
                             const void* PK_uv, const void* PKp_uv, const void* m, size_t n, size_t n_double,
                             void* ok, void* workspace, void* stream, int* accepted) {
  return verify_mixed_dev(kinds, u, R_uv, Rp_uv, PK_uv, PKp_uv, m, n, n_double, ok, workspace, stream, true, accepted);
}

}  // extern "C"
