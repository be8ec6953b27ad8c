/* dsv.h — C ABI of the MI355X batch Schnorr-verify engine (libdsv.so).
 *
 * Drop-in boundary for the native verification path of dusk-schnorr 0.18.  The reference has
 * no FFI of its own; the entry points below are what a Rust `extern "C"` block in that crate
 * would bind to offload its three verify functions (binding shown in INTEGRATION.md):
 *
 *   dsv_verify_single   replaces  PublicKey::verify        /root/reference/src/keys/public.rs:121-130
 *   dsv_verify_double   replaces  PublicKeyDouble::verify  /root/reference/src/keys/public.rs:222-244
 *   dsv_verify_vargen   replaces  PublicKeyVarGen::verify  /root/reference/src/keys/public.rs:401-415
 *   dsv_challenge_*     exposes   challenge_hash{,_double} /root/reference/src/signatures.rs:127-134, 275-290
 *   dsv_sign_*          computes what SecretKey::sign / sign_double / SecretKeyVarGen::sign compute
 *                                                          /root/reference/src/keys/secret.rs:150-168, 217-240, 433-451
 *   dsv_public_keys     computes  PublicKey::from(&SecretKey)  /root/reference/src/keys/public.rs:61-67, 265-272
 *                       — as INPUT GENERATORS for tests and benchmarks, not as a production
 *                       signer: see the note at the signing section.
 *
 * Data layout (all entry points): structure-of-arrays, caller-owned, one batch = n items.
 *   scalar  (JubJubScalar u, sk, c; BlsScalar message m) : 32 bytes, canonical little-endian
 *                                                           (what `to_bytes()` yields)
 *   point   (R, R', PK, PK', Gen)                         : 64 bytes = affine u || v, each a
 *                                                           canonical LE BlsScalar, i.e. the pair
 *                                                           `JubJubExtended::to_hash_inputs()` returns
 *   ext point (the *_ext entry points)                     : 96 bytes = u || v || z of a
 *                                                           JubJubExtended with arbitrary z != 0,
 *                                                           each coordinate canonical LE (what
 *                                                           get_u/get_v/get_z().to_bytes() yield);
 *                                                           z = 0 or a coordinate >= q: ok[i] = 0
 *   verdict ok[i]                                          : one byte, 1 = verify() true, 0 = false
 * Verdicts are those of the reference's `verify` for every input its types can hold (any
 * on-curve point incl. identity / small order, any scalar) — bit-exact against this repository's
 * CPU restatement of the reference algorithm (oracle/); the challenge hash's constants are
 * recipe-derived and NOT pinned by any upstream vector ("parity unpinned").
 * Encodings the Rust types
 * cannot hold (scalar >= r, coordinate or message >= q) give ok[i] = 0; off-curve
 * coordinates are out of contract (result unspecified, never a fault).
 *
 * Host entry points take HOST pointers (pageable is fine), stage through library-owned pinned
 * and device buffers in chunks of 2^15 .. 2^18 items (DSV_HOST_THREADS / dsv_set_host_threads copy
 * threads per call, default 4; six library-owned streams per device, the two that carry the kernels
 * on different priority levels so that they never share a hardware queue) and block until the
 * verdicts are in `ok`.  The *_dev entry points take DEVICE pointers (hipMalloc'd,
 * 16-byte aligned) plus a hipStream_t passed as void*, enqueue only, and never synchronise —
 * they are what the bench times with inputs resident in HBM.
 *
 * Devices and threads.  dsv_init(d) may be called for several devices of one process; each gets
 * its own context (tables, streams, staging) and nothing is shared between contexts.
 *   - *_dev entry points run on the device that OWNS their output buffer (the stream must belong
 *     to the same device); they may be called concurrently from any number of threads, on the
 *     same or on different streams, and leave the calling thread's current device unchanged.
 *     Callers on different streams get different internal sub-batch streams (up to 8 per device,
 *     shared beyond that), so they overlap on the GPU.
 *   - host entry points run on the calling thread's device: dsv_set_device(d) (thread-local), else
 *     the first device initialised.  Up to dsv_max_in_flight() (2) verify calls run concurrently on
 *     ONE device, each with its own staging, sharing the device's compute streams (a further call
 *     waits its turn, first come first served); host calls on different devices run in parallel.
 *     The small utility entry points (challenge, sign, decompress, debug) serialise on a lock.
 *   - dsv_verify_*_multi shard one host batch over ALL initialised devices (contiguous shards, one
 *     host thread per device, no collective).
 *   - dsv_shutdown[_device] waits for the host calls (and jobs) in flight on that device, synchronises the
 *     device and releases everything; the caller must not have *_dev work of its own still
 *     enqueued whose workspace it frees, and must not start new calls on that device until a
 *     later dsv_init.  Calls that arrive after shutdown return DSV_ERR_NOT_INITIALIZED.
 *
 * Return value: DSV_OK (0) or a negative dsv_status; dsv_last_error() gives the text for
 * the calling thread.  The library keeps no pointer after a call returns.  dsv_init is
 * idempotent and thread-safe.
 */
#ifndef DSV_H
#define DSV_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  DSV_OK = 0,
  DSV_ERR_NOT_INITIALIZED = -1,
  DSV_ERR_INVALID_ARGUMENT = -2,
  DSV_ERR_HIP = -3,
  DSV_ERR_NO_DEVICE = -4,
  DSV_ERR_TOO_LARGE = -5
} dsv_status;

#define DSV_MAX_BATCH ((size_t)1 << 28)

/* ---- lifecycle ---- */
int dsv_init(int device);             /* create this GPU's context: fixed-base tables for G and G'
                                         (2 x 75.5 MB, ~50 ms of device time) */
int dsv_init_visible(void);           /* dsv_init for every device listed in the environment variable
                                         DSV_DEVICES ("0,2,3"), else for every visible device;
                                         returns how many are initialised or a negative status */
int dsv_shutdown(void);               /* every initialised device */
int dsv_shutdown_device(int device);
int dsv_set_device(int device);       /* device of THIS thread's host entry points (must be initialised) */
int dsv_get_device(void);             /* ... the current choice; -1 before any dsv_init */
int dsv_initialized_devices(int *out, int cap);
/* NUMA placement (an 8-GPU host has two sockets).  dsv_init reads the device's PCI address and, from
 * /sys/bus/pci/devices/<address>/numa_node and /sys/devices/system/node/node<N>/cpulist, the cpus of the
 * socket its PCIe root hangs off; the copy threads of that device's host calls and the library's own
 * per-device worker threads (*_multi shards beyond the first, the drivers of submitted jobs) are bound to
 * them (threads of the application never are).  Pinned staging needs nothing: hipHostMalloc allocates on
 * the node nearest the current device.  DSV_NUMA=0 in the environment switches the binding off;
 * DSV_SYSFS_ROOT points the lookup at another tree.  dsv_device_numa: *node (-1: unknown), up to `cap`
 * cpus and the PCI address (bdf: NULL or >= 32 bytes) of an initialised device; returns the cpu count or a
 * negative dsv_status.  dsv_debug_numa_lookup: the lookup alone, no device needed (tests). */
int dsv_device_numa(int device, int *node, int *cpus, int cap, char *bdf);
int dsv_debug_numa_lookup(const char *sysfs_root, const char *bdf, int *node, int *cpus, int cap); /* returns how many; fills out[0 .. min(cap, count)) */
const char *dsv_version(void);
const char *dsv_last_error(void);
int dsv_device_count(void);
int dsv_set_host_threads(int n);      /* copy threads the host entry points gather the caller's arrays with
                                         (per process, 1..16): n >= 1 sets, 0 restores the default (the
                                         environment variable DSV_HOST_THREADS, else 4); returns the value
                                         in force */

/* ---- verify, host buffers ---- */
int dsv_verify_single(const uint8_t *u, const uint8_t *R_uv, const uint8_t *PK_uv,
                      const uint8_t *m, size_t n, uint8_t *ok);
int dsv_verify_double(const uint8_t *u, const uint8_t *R_uv, const uint8_t *Rp_uv,
                      const uint8_t *PK_uv, const uint8_t *PKp_uv, const uint8_t *m, size_t n,
                      uint8_t *ok);
int dsv_verify_vargen(const uint8_t *u, const uint8_t *R_uv, const uint8_t *PK_uv,
                      const uint8_t *Gen_uv, const uint8_t *m, size_t n, uint8_t *ok);
/* ---- verify, projective inputs ---------------------------------------------------------------
 * Every point as (u, v, z), 96 B: what the reference's in-memory types hold (`PublicKey::from(&sk)`
 * = GENERATOR_EXTENDED * sk and R = GENERATOR_EXTENDED * r are JubJubExtended with z != 1,
 * /root/reference/src/keys/public.rs:61-67, src/keys/secret.rs:159).  The device performs the
 * `to_hash_inputs` normalisation the reference's verify starts with
 * (/root/reference/src/signatures.rs:131, :280-281) — at most one field inversion per signature,
 * shared by all its points and, in large batches, by eight signatures — so the caller does no
 * field arithmetic on the host.  Same verdicts as the affine entry points on the normalised
 * points. */
int dsv_to_hash_inputs(const uint8_t *in_uvz, size_t n, uint8_t *out_uv, uint8_t *ok); /* the step alone */
int dsv_verify_single_ext(const uint8_t *u, const uint8_t *R_uvz, const uint8_t *PK_uvz,
                          const uint8_t *m, size_t n, uint8_t *ok);
int dsv_verify_double_ext(const uint8_t *u, const uint8_t *R_uvz, const uint8_t *Rp_uvz,
                          const uint8_t *PK_uvz, const uint8_t *PKp_uvz, const uint8_t *m, size_t n,
                          uint8_t *ok);
int dsv_verify_vargen_ext(const uint8_t *u, const uint8_t *R_uvz, const uint8_t *PK_uvz,
                          const uint8_t *Gen_uvz, const uint8_t *m, size_t n, uint8_t *ok);
/* ... sharded over every initialised device (what the Rust / C++ verify_batch* bind) */
int dsv_verify_single_ext_multi(const uint8_t *u, const uint8_t *R_uvz, const uint8_t *PK_uvz,
                                const uint8_t *m, size_t n, uint8_t *ok);
int dsv_verify_double_ext_multi(const uint8_t *u, const uint8_t *R_uvz, const uint8_t *Rp_uvz,
                                const uint8_t *PK_uvz, const uint8_t *PKp_uvz, const uint8_t *m,
                                size_t n, uint8_t *ok);
int dsv_verify_vargen_ext_multi(const uint8_t *u, const uint8_t *R_uvz, const uint8_t *PK_uvz,
                                const uint8_t *Gen_uvz, const uint8_t *m, size_t n, uint8_t *ok);
/* ... device buffers (enqueue only); workspace: dsv_ext_workspace_bytes(n) device bytes, 256-byte
 * aligned */
size_t dsv_ext_workspace_bytes(size_t n);
int dsv_verify_single_ext_dev(const void *u, const void *R_uvz, const void *PK_uvz, const void *m,
                              size_t n, void *ok, void *workspace, void *stream);
int dsv_verify_double_ext_dev(const void *u, const void *R_uvz, const void *Rp_uvz,
                              const void *PK_uvz, const void *PKp_uvz, const void *m, size_t n,
                              void *ok, void *workspace, void *stream);
int dsv_verify_vargen_ext_dev(const void *u, const void *R_uvz, const void *PK_uvz,
                              const void *Gen_uvz, const void *m, size_t n, void *ok, void *workspace,
                              void *stream);

/* ---- verify, the reference's IN-MEMORY representation (Montgomery limbs) -----------------------
 * The Rust types hold every field element as `[u64; 4]` little-endian limbs in Montgomery form,
 * R = 2^256: `BlsScalar(pub [u64; 4])` (dusk-bls12_381 0.13), `JubJubScalar` and the coordinates
 * of `JubJubExtended` (dusk-jubjub 0.14) — /root/reference/Cargo.toml:25-26; the fields that verify
 * reads: /root/reference/src/signatures.rs:58-61, 180-184, 337-340 (u, R, R'),
 * src/keys/public.rs:59, 189, 331-334 (pk, pk', generator), the `BlsScalar` message of
 * src/keys/public.rs:121, 222-226, 401-405.  Every `to_bytes()` is a Montgomery reduction on the
 * host (8 per single signature, 14 per double one).  These entry points take the limbs as they
 * lie in memory, so a binding copies bytes and does no arithmetic at all:
 *   scalar (u, m)              : 32 B = the four u64 limbs of x * 2^256 mod (r | q)
 *   point  (R, R', PK, PK', Gen): 96 B = limbs of u || v || z of the JubJubExtended (t1, t2 are not
 *                                read; a common factor of the three coordinates does not change
 *                                the point, so the device normalises them exactly like *_ext input)
 * Limbs that are not below their modulus (the Rust types cannot hold them) or z = 0: ok[i] = 0.
 * Same verdicts as dsv_verify_*_ext on the `to_bytes()` of the same values.
 *
 * The *_mont_cols forms read the typed objects WHERE THEY LIE: one dsv_column per field, item i at
 * base + i * stride (stride = sizeof of the struct the field lives in).  The pipeline's copy threads
 * gather the fields straight into pinned staging while the GPU works on the previous chunk — no
 * intermediate structure of arrays, no second pass over host memory.  Column order:
 *   single: u, R, PK, m            double: u, R, R', PK, PK', m            vargen: u, R, PK, Gen, m
 * They shard over every initialised device like the *_multi forms (what verify_batch* bind). */
typedef struct dsv_column {
  const void *base;   /* field of item 0 */
  size_t stride;      /* bytes from one item's field to the next (>= the field's width) */
} dsv_column;
int dsv_verify_single_mont_cols(const dsv_column *cols /*[4]*/, size_t n, uint8_t *ok);
int dsv_verify_double_mont_cols(const dsv_column *cols /*[6]*/, size_t n, uint8_t *ok);
int dsv_verify_vargen_mont_cols(const dsv_column *cols /*[5]*/, size_t n, uint8_t *ok);
/* ... asynchronous: `submit` validates the arguments, hands the batch to a library-owned driver
 * thread and returns at once with a job; `dsv_job_wait` blocks until the verdicts are in `ok`, returns
 * the status the blocking form would have returned (dsv_last_error() then holds its text) and
 * releases the job — every submitted job must be waited for exactly once.  cols[] is copied; the
 * objects the columns point into and `ok` must stay valid until the wait returns.
 * Why: a single call pays a ramp (the GPU idles until the first chunk has been gathered and
 * transferred, then runs small first chunks at low occupancy) and a tail.  With two batches in
 * flight the second one's ramp runs under the first one's tail: each call in flight owns its own
 * staging (dsv_max_in_flight() per device; a further job waits inside its driver thread), the
 * compute streams are shared, so the GPU sees one queue of sub-batches.  The blocking host entry
 * points may equally be called from several threads at once — same mechanism.
 * Jobs start in submission order.  dsv_job_done: 1 finished / 0 running (does not release).
 * dsv_shutdown* first lets every job submitted so far (and every blocking call that already owns
 * its staging) run to its verdicts; a submit racing with the shutdown may fail with
 * DSV_ERR_NOT_INITIALIZED at its wait. */
/* ... with the batch fast accept (dsv_verify_*_rlc_dev below: same verdicts; ONE aggregate test decides
 * "all true", anything else is decided by the per-signature kernels): the pipeline only fills a
 * per-device arena — gather, transfer, normalisation and the challenge hash chunk by chunk while the
 * transfers run — then the aggregate runs over the resident group.  With several devices initialised
 * and >= 2^17 items per device the batch is sharded like the *_multi forms, one group (one aggregate)
 * per device; else one group (2^17 <= n <= 2^22) on the calling thread's device; larger and smaller
 * batches take the ordinary column path.  One such call at a time per device.  *accepted (may be NULL): 1 = every
 * group's aggregate decided. */
int dsv_verify_single_mont_cols_rlc(const dsv_column *cols /*[4]*/, size_t n, uint8_t *ok, int *accepted);
int dsv_verify_double_mont_cols_rlc(const dsv_column *cols /*[6]*/, size_t n, uint8_t *ok, int *accepted);
int dsv_verify_vargen_mont_cols_rlc(const dsv_column *cols /*[5]*/, size_t n, uint8_t *ok, int *accepted);
typedef struct dsv_job dsv_job;
int dsv_verify_single_mont_cols_submit(const dsv_column *cols /*[4]*/, size_t n, uint8_t *ok, dsv_job **job);
int dsv_verify_double_mont_cols_submit(const dsv_column *cols /*[6]*/, size_t n, uint8_t *ok, dsv_job **job);
int dsv_verify_vargen_mont_cols_submit(const dsv_column *cols /*[5]*/, size_t n, uint8_t *ok, dsv_job **job);
int dsv_job_wait(dsv_job *job);
int dsv_job_done(const dsv_job *job);
int dsv_max_in_flight(void);
/* ... dense arrays (structure of arrays), this thread's device */
int dsv_verify_single_mont(const uint8_t *u, const uint8_t *R_uvz, const uint8_t *PK_uvz,
                           const uint8_t *m, size_t n, uint8_t *ok);
int dsv_verify_double_mont(const uint8_t *u, const uint8_t *R_uvz, const uint8_t *Rp_uvz,
                           const uint8_t *PK_uvz, const uint8_t *PKp_uvz, const uint8_t *m, size_t n,
                           uint8_t *ok);
int dsv_verify_vargen_mont(const uint8_t *u, const uint8_t *R_uvz, const uint8_t *PK_uvz,
                           const uint8_t *Gen_uvz, const uint8_t *m, size_t n, uint8_t *ok);
/* ... dense arrays, sharded over every initialised device */
int dsv_verify_single_mont_multi(const uint8_t *u, const uint8_t *R_uvz, const uint8_t *PK_uvz,
                                 const uint8_t *m, size_t n, uint8_t *ok);
int dsv_verify_double_mont_multi(const uint8_t *u, const uint8_t *R_uvz, const uint8_t *Rp_uvz,
                                 const uint8_t *PK_uvz, const uint8_t *PKp_uvz, const uint8_t *m,
                                 size_t n, uint8_t *ok);
int dsv_verify_vargen_mont_multi(const uint8_t *u, const uint8_t *R_uvz, const uint8_t *PK_uvz,
                                 const uint8_t *Gen_uvz, const uint8_t *m, size_t n, uint8_t *ok);
/* ... device buffers (enqueue only); workspace: dsv_mont_workspace_bytes(n) device bytes, 256-byte
 * aligned */
size_t dsv_mont_workspace_bytes(size_t n);
int dsv_verify_single_mont_dev(const void *u, const void *R_uvz, const void *PK_uvz, const void *m,
                               size_t n, void *ok, void *workspace, void *stream);
int dsv_verify_double_mont_dev(const void *u, const void *R_uvz, const void *Rp_uvz,
                               const void *PK_uvz, const void *PKp_uvz, const void *m, size_t n,
                               void *ok, void *workspace, void *stream);
int dsv_verify_vargen_mont_dev(const void *u, const void *R_uvz, const void *PK_uvz,
                               const void *Gen_uvz, const void *m, size_t n, void *ok, void *workspace,
                               void *stream);

/* ---- verify, host buffers, sharded over every initialised device (see "Devices and threads") ---- */
int dsv_verify_single_multi(const uint8_t *u, const uint8_t *R_uv, const uint8_t *PK_uv,
                            const uint8_t *m, size_t n, uint8_t *ok);
int dsv_verify_double_multi(const uint8_t *u, const uint8_t *R_uv, const uint8_t *Rp_uv,
                            const uint8_t *PK_uv, const uint8_t *PKp_uv, const uint8_t *m,
                            size_t n, uint8_t *ok);
int dsv_verify_vargen_multi(const uint8_t *u, const uint8_t *R_uv, const uint8_t *PK_uv,
                            const uint8_t *Gen_uv, const uint8_t *m, size_t n, uint8_t *ok);

/* ---- verify, device buffers (enqueue only) ----
 * workspace: device scratch of dsv_workspace_bytes(n) bytes, 256-byte aligned, owned by the call
 * until its work on `stream` has completed (two calls in flight need two workspaces).
 * Ordering: everything the call enqueues happens after the work already on `stream` and before
 * anything enqueued on `stream` afterwards.  Batches of >= 2^17 items are cut into 2^16-item
 * parts that run on two library-owned streams forked from / joined to `stream` by events
 * (DSV_SPLIT=0 in the environment at dsv_init keeps every launch on `stream` itself).  Batches of
 * <= 2^14 items run an eight-lanes-per-signature kernel that trades throughput for latency, with
 * their window tables built on a library-owned stream beside the hash (DSV_QUAD=0 keeps them on
 * the one-lane kernel, DSV_SMALL_OVERLAP=0 builds the tables inside the verify kernel).  Same verdicts on every path. */
size_t dsv_workspace_bytes(size_t n);
int dsv_verify_single_dev(const void *u, const void *R_uv, const void *PK_uv, const void *m,
                          size_t n, void *ok, void *workspace, void *stream);
int dsv_verify_double_dev(const void *u, const void *R_uv, const void *Rp_uv, const void *PK_uv,
                          const void *PKp_uv, const void *m, size_t n, void *ok, void *workspace,
                          void *stream);
int dsv_verify_vargen_dev(const void *u, const void *R_uv, const void *PK_uv, const void *Gen_uv,
                          const void *m, size_t n, void *ok, void *workspace, void *stream);

/* ---- optional fast accept: random-linear-combination batch verification ----
 * Same inputs and the same verdict vector as dsv_verify_single_dev — what differs is the time.  Per
 * group of up to 2^22 items the call first runs ONE aggregate test (schnorr_amd/csrc/k_rlc.hip):
 *   every PK_i and R_i lies in the prime-order subgroup, and
 *   (sum z_i u_i) G + sum (z_i c_i) PK_i - sum z_i R_i == O   for secret 128-bit z_i drawn per call
 * (getrandom).  If it holds, every well-formed item's verdict is `true`, exactly as
 * `PublicKey::verify` (/root/reference/src/keys/public.rs:121-130) would say one by one — up to an
 * error probability <= 2^-112 of accepting a batch that holds a wrong signature, where the
 * per-signature path has none.  If it does not hold (one wrong signature, one point with a
 * small-order component — the reference's types can hold those and its equation is cofactorless —
 * or a point off the curve), the group is verified by dsv_verify_single_dev's kernels and gets their
 * verdicts: nothing is ever decided by the aggregate except "all true".  Worth it where batches are
 * expected to be entirely valid (~2.7x less arithmetic then).
 * ENQUEUE-ONLY like every other *_dev call: the per-signature kernels are launched unconditionally
 * behind the aggregate and return in their first instructions where it accepted (they read its flag
 * words from device memory); no decision is taken on the host.
 * Failure localisation.  A rejected group costs both paths, so the library adapts per device: a counter
 * kept by the device itself (8 after a call that held a rejected aggregate, one less after a call whose
 * aggregates all accepted, 1 after dsv_init; the host reads it without waiting, possibly one call late)
 * decides how the NEXT calls run.  While it is > 0 a group is cut into up to 16 sub-groups of >= 2^16 items
 * that share the hash and one set of launches, each with its own aggregate — one wrong signature in 2^20
 * then sends 2^16 items to the per-signature kernels, not 2^20 —, and 1024 consecutive items at a position
 * drawn from the call's secret key are verified first: a wrong one among them skips the aggregates
 * altogether (a batch tampered with throughout pays hash + sample + per-signature path; a heuristic,
 * DSV_RLC_SAMPLE=0 switches it off, DSV_RLC_SUB_LOG2 / DSV_RLC_SUBGROUPS tune / force the sub-groups).
 * A caller whose batches are valid runs one aggregate per group and nothing else.
 * window_bits: 0 = chosen from n — and groups below 2^17 items (single signatures; double and
 * var-generator batches: below 2^14), where an aggregate does not pay, go straight to the per-signature
 * kernels —, else one of 4, 6, 8, 12, 14, 16 (bucket windows; tests).
 * accepted (may be NULL): receives 1 if every group was decided by its aggregates, else 0.  If it points
 * to memory the device can write — device memory, or host memory from hipHostMalloc / hipHostRegister —
 * it is written by a kernel when `stream` gets there and the call does not block; if it is ordinary
 * host memory the call waits for `stream` at its end and stores the value itself.
 * workspace: dsv_rlc_workspace_bytes(n, window_bits) device bytes, 256-byte aligned (never less for a
 * larger n: a workspace sized for n serves any smaller batch).
 * Do NOT capture these calls into a hipGraph: the 32-byte weight key is drawn on the host (getrandom) per
 * call and passed by value — a replayed graph would reuse it, and predictable weights void the soundness
 * of the aggregate (the per-signature entry points have no such state and may be captured). */
size_t dsv_rlc_workspace_bytes(size_t n, int window_bits);
/* geometry of one group's aggregate (tests, sizing; works without a GPU): scheme 0 single / 1 double /
 * 2 var-generator, `groups` sub-groups (0 or 1: one); out[24] = window bits c, c/2, key windows, nonce
 * windows, windows, row/column segments, bit-sum segments, low digit bits sorted through LDS, multiples
 * of r added to a key scalar, long / short points and fixed-base terms per item, (digit, point) pairs
 * per sub-group, buckets per sub-group, points of the two scratch areas, high digit bits (bins per
 * window = 2^that), digit rows, digits per row, bins, slots per bin, sub-groups, items per sub-group,
 * workspace bytes of this plan */
int dsv_rlc_plan_info(int scheme, size_t n, int window_bits, int groups, uint64_t *out /*[24]*/);
/* the device's history counter (see above; tests and tools): returns it, or a negative dsv_status;
 * set >= 0 overrides it */
int dsv_debug_rlc_history(int device, int set);
/* the device's LONG history counter (128 after a call that held a rejected aggregate, one less after a call
 * whose aggregates all accepted): while it is > 0 and the short one is 0, groups run "guarded" — one aggregate
 * as in the steady state plus, behind it, a second stage of sub-group aggregates that a kernel switches off
 * when the first accepted (+0.03 ms on an accepted 2^20-item call), so that the first rejected batch after a run of valid
 * ones is localised too instead of paying both paths in full.  DSV_RLC_GUARD=0 switches that off. */
int dsv_debug_rlc_history_long(int device, int set);
/* sub-groups per group, process-wide: groups >= 1 forces that many (whatever the history says),
 * 0 = automatic, < 0 = leave; returns the previous setting ($DSV_RLC_SUBGROUPS initialises it) */
int dsv_debug_rlc_subgroups(int groups);
int dsv_verify_single_rlc_dev(const void *u, const void *R_uv, const void *PK_uv, const void *m,
                              size_t n, void *ok, void *workspace, void *stream, int window_bits,
                              int *accepted);
/* the same in front of dsv_verify_double_dev (`PublicKeyDouble::verify`, src/keys/public.rs:222-244:
 * both equations of an item enter the sum, each with a weight of its own) and of
 * dsv_verify_vargen_dev (`PublicKeyVarGen::verify`, :401-415: the generator is a third variable
 * point, its scalar z_i u_i) */
int dsv_verify_double_rlc_dev(const void *u, const void *R_uv, const void *Rp_uv, const void *PK_uv,
                              const void *PKp_uv, const void *m, size_t n, void *ok, void *workspace,
                              void *stream, int window_bits, int *accepted);
int dsv_verify_vargen_rlc_dev(const void *u, const void *R_uv, const void *PK_uv, const void *Gen_uv,
                              const void *m, size_t n, void *ok, void *workspace, void *stream,
                              int window_bits, int *accepted);

/* ... and in front of dsv_verify_*_wire_dev (serialized records resident in HBM; layouts below): decode,
 * then the aggregate.  A record that does not decode has verdict 0 and stays out of the sum; decoded
 * points lie on the curve but not necessarily in the prime-order subgroup (`from_bytes`,
 * src/keys/public.rs:94-100), which is what the aggregate's subgroup test is for.
 * workspace: dsv_wire_rlc_workspace_bytes(n, window_bits). */
size_t dsv_wire_rlc_workspace_bytes(size_t n, int window_bits);
int dsv_verify_single_wire_rlc_dev(const void *sig64, const void *pk32, const void *m, size_t n, void *ok,
                                   void *workspace, void *stream, int window_bits, int *accepted);
int dsv_verify_double_wire_rlc_dev(const void *sig96, const void *pk64, const void *m, size_t n, void *ok,
                                   void *workspace, void *stream, int window_bits, int *accepted);
int dsv_verify_vargen_wire_rlc_dev(const void *sig64, const void *pk64, const void *m, size_t n, void *ok,
                                   void *workspace, void *stream, int window_bits, int *accepted);

/* ... and from serialized records in HOST memory (the layouts of dsv_verify_*_wire): the pipeline decodes
 * chunk by chunk into a per-device arena while the transfers run (128 B per single signature on the bus,
 * half of the typed-object form), one aggregate follows.  One group, 2^17 <= n <= 2^22, on the calling
 * thread's device; other sizes take dsv_verify_*_wire. */
int dsv_verify_single_wire_rlc(const uint8_t *sig64, const uint8_t *pk32, const uint8_t *m, size_t n, uint8_t *ok,
                               int *accepted);
int dsv_verify_double_wire_rlc(const uint8_t *sig96, const uint8_t *pk64, const uint8_t *m, size_t n, uint8_t *ok,
                               int *accepted);
int dsv_verify_vargen_wire_rlc(const uint8_t *sig64, const uint8_t *pk64, const uint8_t *m, size_t n, uint8_t *ok,
                               int *accepted);

/* second stage alone: ok[i] = (accumulate ? ok[i] : valid[i]) & [u*Gen + c*PK == R], Gen = G
 * (which = 0) or G' (which = 1); c and valid as produced by dsv_challenge_*_dev */
int dsv_verify_core_dev(const void *u, const void *c, const void *valid, const void *PK_uv,
                        const void *R_uv, int which, int accumulate, size_t n, void *ok,
                        void *workspace, void *stream);

/* both equations of PublicKeyDouble::verify from a precomputed challenge, ONE launch (what
 * dsv_verify_double_dev runs after k_challenge): ok[i] = valid[i] & [uG + cPK == R] & [uG' + cPK' == R'] */
int dsv_verify_core_double_dev(const void *u, const void *c, const void *valid, const void *PK_uv,
                               const void *R_uv, const void *PKp_uv, const void *Rp_uv, size_t n,
                               void *ok, void *workspace, void *stream);

/* ---- mixed batches (single and double signatures interleaved), device buffers ----------------
 * kinds[i] = 0: item i is a Signature / PublicKey pair, 1: a SignatureDouble / PublicKeyDouble
 * pair; the batch is a structure of arrays over ALL n items (Rp_uv / PKp_uv rows of single items
 * are ignored).  The library splits the batch by kind ON THE DEVICE (stable compaction of the
 * index vector + row gathers), runs each kind through its own kernels and scatters the verdicts
 * back into batch order.  n_double = number of kind-1 items; every other item must be kind 0.  If
 * the counts do not match the kind vector, or an item has another kind, the affected verdicts
 * (all of them for a count mismatch) are 0.  kinds must be 16-byte aligned.
 * workspace: dsv_mixed_workspace_bytes(n) device bytes, 256-byte aligned. */
size_t dsv_mixed_workspace_bytes(size_t n);
int dsv_verify_mixed_dev(const void *kinds, const void *u, const void *R_uv, const void *Rp_uv,
                         const void *PK_uv, const void *PKp_uv, const void *m, size_t n,
                         size_t n_double, void *ok, void *workspace, void *stream);
/* ... with each kind's items through the batch fast accept (dsv_verify_*_rlc_dev above: same verdicts;
 * `accepted` as there: 1 = every group of both kinds was decided by its aggregates).
 * workspace: dsv_mixed_rlc_workspace_bytes(n). */
size_t dsv_mixed_rlc_workspace_bytes(size_t n);
int dsv_verify_mixed_rlc_dev(const void *kinds, const void *u, const void *R_uv, const void *Rp_uv,
                             const void *PK_uv, const void *PKp_uv, const void *m, size_t n,
                             size_t n_double, void *ok, void *workspace, void *stream, int *accepted);
/* the pieces, for callers that shard each kind separately (schnorr_amd/distributed.py):
 *   split : idx_single[j] / idx_double[j] = batch position (uint32) of the j-th item of that kind,
 *           at most cap_* entries written; scratch: dsv_split_scratch_bytes(n) device bytes whose
 *           last two uint32 (at offset dsv_split_scratch_bytes(n) - 256) receive the two counts
 *   gather: dst row j = src row idx[j], rows of row_bytes (multiple of 16) bytes, src has src_rows rows
 *   scatter: dst[idx[j]] = src[j] (verdict bytes back into batch order), dst has dst_len bytes
 *   Both touch entries j < min(count, *count_limit) only (count_limit: device uint32, e.g. one of
 *   the split's two counts; NULL = no limit) and skip an index that is out of range, so an index
 *   vector is never dereferenced beyond what the split wrote into it. */
size_t dsv_split_scratch_bytes(size_t n);
int dsv_split_kinds_dev(const void *kinds, size_t n, void *idx_single, size_t cap_single,
                        void *idx_double, size_t cap_double, void *scratch, void *stream);
int dsv_gather_rows_dev(const void *src, size_t src_rows, size_t row_bytes, const void *idx,
                        size_t count, const void *count_limit, void *dst, void *stream);
int dsv_scatter_verdicts_dev(const void *src, const void *idx, size_t count, const void *count_limit,
                             void *dst, size_t dst_len, void *stream);

/* ---- challenge hash only (c = trunc250(Poseidon(R.., m))), 32 B LE per item ---- */
int dsv_challenge_single(const uint8_t *R_uv, const uint8_t *m, size_t n, uint8_t *c);
int dsv_challenge_double(const uint8_t *R_uv, const uint8_t *Rp_uv, const uint8_t *m, size_t n,
                         uint8_t *c);
int dsv_challenge_single_dev(const void *R_uv, const void *m, size_t n, void *c, void *valid,
                             void *stream);
int dsv_challenge_double_dev(const void *R_uv, const void *Rp_uv, const void *m, size_t n,
                             void *c, void *valid, void *stream);

/* ---- signing / key derivation: input generation for tests and benchmarks ("next" row f-1) ----
 * sk, m, r canonical 32 B.  r is the caller-drawn nonce (the reference draws it from its RNG
 * inside sign()).  Outputs: u (32 B), R_uv / Rp_uv (64 B).  gen_uv == NULL means the standard
 * generator G;  for the var-generator scheme pass the per-key generator (variable base).
 * NOT A PRODUCTION SIGNER: unlike dusk-jubjub's constant-time multiplication, these kernels index
 * global-memory tables with digits of sk and of the nonce (memory addresses depend on secrets), and
 * they normalise r*G / sk*G with a VARIABLE-TIME inversion of the projective z (extended Euclid,
 * inv29.h: its step count, and whether it falls back to Fermat, depend on z — a value derived from
 * the secret scalar; projective coordinates are known to leak scalar bits).
 * The host entry points zero their device staging of sk / nonce before returning (and the
 * library zeroes staging it releases); the *_dev forms work in the caller's buffers only.
 * A scalar that is not canonical (>= r): host entry points return DSV_ERR_INVALID_ARGUMENT; the
 * *_dev forms write 0xff..ff (never a valid encoding) to that item's outputs. */
int dsv_sign_single(const uint8_t *sk, const uint8_t *m, const uint8_t *r, size_t n, uint8_t *u,
                    uint8_t *R_uv);
int dsv_sign_double(const uint8_t *sk, const uint8_t *m, const uint8_t *r, size_t n, uint8_t *u,
                    uint8_t *R_uv, uint8_t *Rp_uv);
int dsv_sign_vargen(const uint8_t *sk, const uint8_t *Gen_uv, const uint8_t *m, const uint8_t *r,
                    size_t n, uint8_t *u, uint8_t *R_uv);
/* PK = sk * G (which = 0), sk * G' (which = 1); or sk * Gen when gen_uv != NULL */
int dsv_public_keys(const uint8_t *sk, int which, const uint8_t *gen_uv, size_t n,
                    uint8_t *PK_uv);
int dsv_sign_single_dev(const void *sk, const void *m, const void *r, size_t n, void *u,
                        void *R_uv, void *stream);
int dsv_sign_double_dev(const void *sk, const void *m, const void *r, size_t n, void *u,
                        void *R_uv, void *Rp_uv, void *stream);
int dsv_public_keys_dev(const void *sk, int which, size_t n, void *PK_uv, void *stream);
/* variable-base forms (var-generator scheme); workspace: dsv_workspace_bytes(n) device bytes */
int dsv_public_keys_vargen_dev(const void *sk, const void *Gen_uv, size_t n, void *PK_uv,
                               void *workspace, void *stream);
int dsv_sign_vargen_dev(const void *sk, const void *Gen_uv, const void *m, const void *r, size_t n,
                        void *u, void *R_uv, void *workspace, void *stream);

/* ---- wire formats (the reference's Serializable impls) ------------------------------------
 * compressed point = JubJubAffine::to_bytes(): canonical v with bit 255 = lowest bit of u.
 * dsv_decompress_points: JubJubAffine::from_bytes for n points; ok[i] = 0 where the reference
 * would return Err (v >= q, or no square root).  The _dev form reads one 32-byte record every
 * in_stride bytes (16-byte aligned) and, with accumulate != 0, ANDs into ok[] instead of
 * overwriting it. */
int dsv_compress_points(const uint8_t *in_uv, size_t n, uint8_t *out32); /* JubJubAffine::to_bytes */
int dsv_decompress_points(const uint8_t *in32, size_t n, uint8_t *out_uv, uint8_t *ok);
int dsv_decompress_points_dev(const void *in, size_t in_stride, size_t n, void *out_uv, void *ok,
                              int accumulate, void *stream);
/* verify straight from serialized values (host buffers, array-of-records as the Rust
 * `to_bytes()` produce them):
 *   single : Signature (64 B = u || compressed R)            PublicKey (32 B)
 *            /root/reference/src/signatures.rs:106-123, src/keys/public.rs:87-101
 *   double : SignatureDouble (96 B = u || R || R')           PublicKeyDouble (64 B = pk || pk')
 *            /root/reference/src/signatures.rs:245-270, src/keys/public.rs:282-299
 *   vargen : SignatureVarGen (64 B)                          PublicKeyVarGen (64 B = pk || generator)
 *            /root/reference/src/signatures.rs:387-404, src/keys/public.rs:347-372
 * ok[i] = 1 iff every from_bytes would succeed AND verify() would return true. */
int dsv_verify_single_wire(const uint8_t *sig64, const uint8_t *pk32, const uint8_t *m, size_t n,
                           uint8_t *ok);
int dsv_verify_double_wire(const uint8_t *sig96, const uint8_t *pk64, const uint8_t *m, size_t n,
                           uint8_t *ok);
int dsv_verify_vargen_wire(const uint8_t *sig64, const uint8_t *pk64, const uint8_t *m, size_t n,
                           uint8_t *ok);
/* the same with the serialized records already in device memory (16-byte aligned; enqueue only);
 * workspace: dsv_wire_workspace_bytes(n) device bytes, 256-byte aligned */
size_t dsv_wire_workspace_bytes(size_t n);
int dsv_verify_single_wire_dev(const void *sig64, const void *pk32, const void *m, size_t n, void *ok,
                               void *workspace, void *stream);
int dsv_verify_double_wire_dev(const void *sig96, const void *pk64, const void *m, size_t n, void *ok,
                               void *workspace, void *stream);
int dsv_verify_vargen_wire_dev(const void *sig64, const void *pk64, const void *m, size_t n, void *ok,
                               void *workspace, void *stream);

/* ---- the reference harness's input generator (rand 0.8 StdRng::seed_from_u64(seed) = ChaCha12,
 * draw order per item: sk = JubJubScalar::random, m = BlsScalar::random, nonce r =
 * JubJubScalar::random; /root/reference/tests/schnorr.rs:16-22, benches/signature.rs:48-60).
 * Items first_item .. first_item + n - 1 of that stream; 32 B canonical each. */
int dsv_stdrng_sign_inputs(uint64_t seed, size_t first_item, size_t n, uint8_t *sk, uint8_t *m,
                           uint8_t *r);
int dsv_stdrng_sign_inputs_dev(uint64_t seed, size_t first_item, size_t n, void *sk, void *m,
                               void *r, void *stream);
/* var-generator harness: SecretKeyVarGen::random draws sk, then the generator scalar g
 * (Gen = g*G), then the message, then the nonce — four draws per item
 * (/root/reference/src/keys/secret.rs:371-373, tests/schnorr_var_generator.rs:16-22) */
int dsv_stdrng_vargen_inputs_dev(uint64_t seed, size_t first_item, size_t n, void *sk, void *g,
                                 void *m, void *r, void *stream);

/* ---- introspection for tests: copy one fixed-base table entry (affine niels v+u, v-u, 2duv
 * as canonical LE, 96 B) for generator `which` (0 = G, 1 = G'), window w (signed windows of
 * dsv_fixed_window_bits() bits), digit magnitude d ---- */
int dsv_debug_table_entry(int which, int window, int digit, uint8_t out96[96]);
int dsv_fixed_window_bits(void); /* width of the (signed) fixed-base windows; digit <= 2^(bits-1) */
/* ---- introspection: the three short scalars of the var-generator kernel (lattice3.h) for (u, c):
 * per item 128 B = |x| || |y| || |z| (32 B LE each) || sign bytes of x, y, z || padding;
 * x = z*u, y = z*c (mod 8r), z odd.  u is taken mod 2^252, c mod 2^250. */
int dsv_debug_lattice3(const uint8_t *u, const uint8_t *c, size_t n, uint8_t *out128);
/* ---- introspection: the half-size scalars of the fixed-generator kernels (halfgcd.h) for c (taken
 * mod 2^250): per item 96 B = |a| || |b| (32 B LE each) || sign byte of b || padding;
 * a = b*c (mod 8r), b odd. */
int dsv_debug_half_scalars(const uint8_t *c, size_t n, uint8_t *out96);
/* ---- introspection: field-multiplier self test on the device: out = a*b mod q (canonical) */
int dsv_debug_fq_mul(const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif /* DSV_H */
