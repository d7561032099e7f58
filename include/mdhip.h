/*
 * mdhip.h — C ABI of libmdhip.so, the MI355X (gfx950) backend for the
 * RDF/CN and MSD/Green-Kubo hot path of molmd/mdproptools.
 *
 * The reference has no plugin/FFI seam of its own: its hot path is a set of
 * private array-in/array-out Python functions. Each entry point below replaces
 * exactly one of them and cites it (paths under /root/reference/mdproptools/).
 * INTEGRATION.md shows the ctypes stub a maintainer adds on the reference side.
 *
 * Conventions
 *  - plain C types, caller-owned buffers, C-contiguous, float64 unless noted;
 *  - every call returns 0 on success or a negative MDHIP_E* code; the text is
 *    available from mdhip_last_error(); nothing throws across the boundary;
 *  - "host|dev" inputs: the big per-frame planes may live in host memory
 *    (the library stages them) or already in device memory (flag `on_device`);
 *    small tables (relations, cutoffs, offsets, boxes) are always host
 *    pointers; results are written to host pointers, except by the entry points
 *    named *_dev, which write the same values to a DEVICE buffer of the caller
 *    (one process per GPU: the multi-GPU layer hands that buffer to RCCL);
 *  - calls are synchronous on the context's stream: results are complete when
 *    the call returns — except for the entry points named *_async (SURVEY.md
 *    8b: "an _async variant + mdhip_sync"), which return with their work queued
 *    on the stream; see "asynchronous calls" below. One context per thread;
 *    contexts are independent;
 *  - there is NO CPU fallback: without a usable HIP device mdhip_create fails.
 *
 * Integer results (histograms, counts) are exact and independent of the
 * launch geometry. Floating-point reductions use a fixed order per launch
 * geometry (no float atomics), so they are reproducible run to run.
 */
#ifndef MDHIP_H
#define MDHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDHIP_VERSION 600 /* 0.6.0: + mdhip_row_displacement, MDHIP_EUNKNOWN (mdhip_ticket_status of a forgotten ticket), mdhip_lag_msd* finishes on the device at every length; 0.5.0: + mdhip_ticket_status / mdhip_fallbacks / MDHIP_EPENDING (a call's own completion status), mdhip_lag_msd_status_dev; 0.4.0: the *_async entry points with mdhip_sync / mdhip_wait / mdhip_call_stats, mdhip_green_kubo, mdhip_cumtrapz_dev */

#define MDHIP_OK 0
#define MDHIP_EINVAL (-1)  /* bad argument (shape, NULL, unsupported size) */
#define MDHIP_EHIP (-2)    /* a HIP runtime call failed */
#define MDHIP_ENOMEM (-3)  /* device or host allocation failed */
#define MDHIP_ENODEV (-4)  /* no usable gfx950 device */
#define MDHIP_ELIMIT (-5)  /* problem exceeds a kernel limit (e.g. LDS for the histogram rows) */
#define MDHIP_EPENDING (-6) /* mdhip_ticket_status: the call is still in flight */
#define MDHIP_EUNKNOWN (-7) /* mdhip_ticket_status: no such call is remembered (more than 64 completed calls ago) */

typedef struct mdhip_ctx mdhip_ctx;

/* ---- lifecycle ---------------------------------------------------------- */
int mdhip_version(void);
/* Binds `device` (ordinal), creates a stream and a growable device workspace. */
int mdhip_create(mdhip_ctx **out, int device);
void mdhip_destroy(mdhip_ctx *ctx);
/* Last error text of this context (or of mdhip_create when ctx == NULL). */
const char *mdhip_last_error(mdhip_ctx *ctx);
/* Launch on a caller-owned hipStream_t (e.g. torch's current stream); NULL restores the own stream. */
int mdhip_set_stream(mdhip_ctx *ctx, void *hip_stream);
/*
 * ---- asynchronous calls ----------------------------------------------------
 * A *_async entry point takes the arguments of its synchronous twin (a trailing `*_on_device` flag where the twin
 * comes as a host / _dev pair), queues ALL of its device work — staging of the small tables, kernels, the copy of
 * the results into page-locked staging — on the context's stream and returns. The call COMPLETES later, inside
 * mdhip_sync or mdhip_wait, in issue order: only then are host results in the caller's arrays, device results final
 * (a rare internal re-run may rewrite them at completion), and the error code of the call known — mdhip_sync /
 * mdhip_wait return the first error of the calls they completed (text: mdhip_last_error). Until then every array
 * passed to the call — inputs and outputs, host and device, the small tables too — must stay valid and unchanged.
 * Several calls may be in flight; their kernels run back to back on the stream with no host round trip between
 * them, which is the point: a step of k calls costs its kernels plus ONE wait (reference callers are plain
 * synchronous Python, e.g. dynamical/viscosity.py:178-190, so the overlap has to come from here). Inputs the
 * asynchronous path does not take (frames in pageable host memory, class passes, the edge-table kernels) are
 * handled by completing that part of the work before the call returns: correct, just not overlapped.
 * A synchronous call completes everything issued before it first. mdhip_destroy completes what is still in flight.
 */
/* Completes every call in flight, then waits for the stream. Returns the first error among them, 0 if none. */
int mdhip_sync(mdhip_ctx *ctx);
/* Completes the calls in flight in issue order until at most `keep_in_flight` remain (double buffering: issue step
 * k + 1, then mdhip_wait(ctx, 1) for the results of step k while k + 1 runs). */
int mdhip_wait(mdhip_ctx *ctx, int keep_in_flight);
/* Number of asynchronous calls issued and not yet completed. */
int mdhip_pending(mdhip_ctx *ctx);
/* Kernel time / preparation time / launches / dominant kernel of the call completed `back` calls ago (0 = the last
 * one, as mdhip_last_kernel_ms reports; the last 64 are remembered). Any output pointer may be NULL. */
int mdhip_call_stats(mdhip_ctx *ctx, int back, double *kernel_ms, double *aux_ms, int *n_launches, const char **kernel);
/* The number of the entry-point call issued last on this context (1, 2, ...; calls made by the library itself while it
 * completes another one are not counted), and the same statistics looked up by that number once the call has
 * completed — how a caller with several calls in flight finds the times of each. */
long long mdhip_last_ticket(mdhip_ctx *ctx);
int mdhip_ticket_stats(mdhip_ctx *ctx, long long ticket, double *kernel_ms, double *aux_ms, int *n_launches,
                       const char **kernel);
/* What the COMPLETION of call `ticket` returned: 0, or the negative code of its failure (text: mdhip_last_error) —
 * whoever completed it (its own mdhip_wait, a later mdhip_sync, a synchronous call that drained it). MDHIP_EPENDING
 * while it is in flight, MDHIP_EUNKNOWN for a number more than 64 completed calls old (the context's error text is left as it is: MDHIP_EINVAL
 * can be the completion status of the call itself). An error handed out here is no
 * longer reported by a later mdhip_sync / mdhip_wait. *n_fallbacks (may be NULL): slow-path repeats the call took —
 * the staged full-lag MSD kernel's ring timed out (grid not resident as a whole: a co-tenant on the GPU) and the
 * call was repeated over a transposed copy — results are the same, the call took seconds instead of milliseconds.
 * The wrapper warns when it is non-zero. */
int mdhip_ticket_status(mdhip_ctx *ctx, long long ticket, int *n_fallbacks);
/* Slow-path repeats (see mdhip_ticket_status) since the context was created. */
long long mdhip_fallbacks(mdhip_ctx *ctx);
/* Device time (ms, hipEvent pair on the launch stream) of the dominant kernel of the last call,
 * and the number of times that kernel was launched by that call. */
double mdhip_last_kernel_ms(mdhip_ctx *ctx, int *n_launches);
/* Device time (ms) of the preparation kernels of the last call that are not part of the dominant kernel
 * (the spatial sort / tile lists of the culled pair path); 0 when there were none. */
double mdhip_last_aux_ms(mdhip_ctx *ctx);
/* Name of the dominant kernel the last call launched (as rocprofv3 lists it, without the namespace), "" if none. */
const char *mdhip_last_kernel_name(mdhip_ctx *ctx);
/* Estimated relative rounding-error bound of the last mdhip_lag_msd call when it was answered by the FFT
 * path (lag_variant 2 or 3); 0 when the exact-difference kernel answered. */
double mdhip_last_rel_bound(mdhip_ctx *ctx);
/* Identity of the BUILD: sha256 (first 16 hex digits) of the sources the library was compiled from, as the build
 * recipe computed it (mdproptools_amd/build.py); "unknown" for a library built by other means. bench.py prints it. */
const char *mdhip_build_id(void);
/* Writes the device name (e.g. "gfx950...") into buf. */
int mdhip_device_name(mdhip_ctx *ctx, char *buf, int buflen);
/* Kernel organisation knobs, for A/B measurements only; results never depend on them. Keys:
 *   "rdf_variant"  1 fast pair kernels (default), 0 edge-table lookup per pair
 *   "rdf_cull"     -1 auto, 0 dense sweep, 1 spatially culled sweep (when applicable)
 *   "rdf_sj"       culled sweep: 1 scalar-j kernel (default), 2 the same without the persistent grid,
 *                  0 LDS-tile kernel
 *   "rdf_rows"     scalar-j RDF: -1/1 ordered-pair rows without a class-row table when they fit LDS, 0 class rows
 *   "rdf_pk"       scalar-j RDF with ordered rows: -1/1 packed-f32 classification of the pairs (two per VALU
 *                  instruction) with every pair inside the error band of a bin edge or of the cutoff resolved by
 *                  the exact f64 chain (default whenever the band is narrow enough; ordered rows when they fit LDS,
 *                  else class rows with their row table), 0 all-f64 sweep
 *   "rdf_sort"     spatial sort: -1 auto, 1 one block per frame (LDS counters), 0 multi-block (global counters), 3 as
 *                  auto without the read-once form for frames of <= 12288 atoms
 *   "rdf_inflight" per-frame output: frames in flight per XCD
 *   "rdf_jsplit", "rdf_fpb", "rdf_batch", "rdf_slots"  launch geometry of the pair kernels
 *   "lag_variant"  full-lag MSD: 3 / -1 (default) = autocorrelation theorem (O(F log F)) when the relative error
 *                  bound it computes for the data (mdhip_last_rel_bound) is <= 1e-10, else the exact-difference
 *                  kernel; 1 = always the series-resident difference kernel, 0 = staged difference kernel,
 *                  2 = always the autocorrelation theorem (the one knob that changes results, within that bound)
 *   "lag_fft_kernel" fused full-lag MSD kernel: 3 (default) the 12288-point kernel (radix-12 first pass, twelve waves each a
 *                  512-point transform in registers) where 3072 <= F and F + max_lag <= 12288, else as 2; 2 first pass
 *                  from registers + wave-private sub-transforms of a power-of-two length where the series is long
 *                  enough, 1 block-wide passes, 0 the round-2 kernel (results agree within the bound)
 *   "xcorr_tile"   time slabs of the direct correlation kernel
 *   "fft_logr", "fft_logc"  FFT correlation: largest radix of a pass (log2, default 10) and columns per tile of the
 *                  radix-4 network; "fft_net8" 1 (default) radix-8 network for passes of radix >= 2^9, 0 off, 2 one
 *                  workgroup per CU; "fft_specfuse" 1 (default) spectrum step inside the inverse's first pass;
 *                  "fft_mid" 1 (default) an autocorrelation through a two-pass transform runs in three launches (second
 *                  pass + spectrum step + inverse's first pass in one), 0 four, 2 three with narrower tiles
 *   "seg_frame"    segment COM / flux: 1 (default) one (run, frame) per block, 0 the software-pipelined staged kernel;
 *                  "seg_cap", "seg_vec", "seg_gy" geometry of the staged kernel
 *   "h2d_overlap"  host-resident frames: 1 (default) staged batch by batch under the sweeps, 0 copied first;
 *                  "h2d_ring" 1 (default) pageable sources go through a page-locked ring filled by helper threads
 *   "small_copy"   1 (default) copies of <= 256 KB between page-locked host memory and the device are made by a kernel
 *                  on the launch stream, 0 by hipMemcpyAsync (a blit on another hardware queue)
 *   "rdf_guard", "cn_pk"  overflow guard / coordination counts through the packed sweep ("rdf_relblock": retired in
 *                  round 4, accepted and ignored — the f32 records are relative to their whole tile's centre)
 *   "lag_direct"   fused full-lag MSD path: -1 default (= 2), 0 transposed copy made by a pass of its own, 1 the kernel reads
 *                  the trajectory in place, 2 clusters of 16 workgroups transpose their tiles inside the kernel (results of
 *                  the three agree within the reported bound)
 *   "sync_spin"    waiting for device work: 1 (default) poll the completion event (no interrupt wake-up latency; turns
 *                  into a blocking wait after 100 ms), 0 block at once */
int mdhip_set_option(mdhip_ctx *ctx, const char *key, int value);

/* ---- R2/R3 binning table ------------------------------------------------ */
/*
 * Exact bin edges of the reference's binning rule
 *     bin = (int64) ( sqrt(rsq) / bin_size )           structural/rdf_cn.py:68,85
 * edges[k] = the smallest double rsq whose bin is >= k, k = 0..nbins (edges[0] = 0).
 * Found by bisection on the bit pattern with IEEE sqrt and divide, so that
 * binning on the device is a pure comparison of rsq against this table and
 * never depends on device sqrt/div rounding. `edges` has nbins+1 entries.
 */
int mdhip_bin_edges(double bin_size, int nbins, double *edges);
/*
 * Bound (in bins) on |g32 - sqrt(rsq_ref)/bin_size - addend| of the packed-f32 bin guess the pair kernel uses to
 * classify pairs: coordinates relative to a tile centre with |xr_i| + |xr_j| <= s_cap per axis, boxes up
 * to l_max, n_rows histogram rows of nbins+1 words riding in the guess. Pairs whose guess is within twice this bound
 * (plus slack) of an integer are resolved by the exact f64 chain instead. Exposed so that the bound can be checked
 * against an emulation of the f32 chain (tests/test_abi_cpu.py); a pure function, no device needed.
 */
double mdhip_pk_error_bound(double r_cut, double bin_size, int nbins, int n_rows, double s_cap, double l_max);

/*
 * Row layout of the table-free ("ordered rows") sweep when the plain n_ti x n_tj rows do not fit LDS:
 * integers a[n_ti], b[n_tj] such that row(ti, tj) = a[ti] + b[tj] never holds two type pairs of different classes
 * (cls[ti * n_tj + tj] >= 0: the class of the ordered pair), with *n_rows = max a + max b + 1 < n_ti * n_tj rows and
 * row_cls[row] the class of every row (-1: unused; at least n_ti * n_tj entries). *n_rows = 0: no layout with fewer
 * rows than the plain one was found (a, b, row_cls untouched). Exposed so that the layout's defining property can be
 * checked without a device (tests/test_abi_cpu.py); a pure function, deterministic.
 */
int mdhip_row_displacement(int n_ti, int n_tj, const int32_t *cls, int32_t *a, int32_t *b, int32_t *row_cls,
                           int *n_rows);

/* ---- R3: _rdf_loop (+ _calc_rsq, _remove_outliers) ------------------------ */
/*
 * structural/rdf_cn.py:72-97 for n_frames frames at once.
 *   xyz        host|dev [n_frames][3][n_atoms]  SoA planes, atoms sorted by id
 *   type       host int32 [n_atoms] labels as in the dump (or altered ids, rdf_cn.py:197-215);
 *              type_frame_stride = 0 when shared by all frames, n_atoms when per frame
 *   box        host [n_frames][3] edge lengths lx, ly, lz (rdf_cn.py:80)
 *   rel        host int32 [n_rel][2] = relation_matrix (rdf_cn.py:484)
 *   r_cut_sq   the value the reference compares against, r_cut**2 (rdf_cn.py:66)
 *   edges      host [nbins+1] from mdhip_bin_edges, or NULL to have it computed from bin_size
 *   hist_full  host uint64 [n_frames][nbins] (per_frame=1) or [nbins] summed over frames (per_frame=0);
 *              +2 per in-cutoff pair (rdf_cn.py:85-86)
 *   hist_part  host uint64 [n_frames][n_rel][nbins] or [n_rel][nbins]; +1 per (head a, other b) and per
 *              (head b, other a) (rdf_cn.py:87-96)
 *   overflow   host uint64 [1]: in-cutoff pairs whose bin index is >= nbins. The reference indexes out of
 *              bounds for them; here they are dropped and counted.
 * Semantics per pair (i < j): d = head - other; one shift by -sign(d)*L iff |d| > L/2;
 * rsq = (dx*dx + dy*dy) + dz*dz with no fused multiply-add; keep iff rsq < r_cut_sq.
 */
int mdhip_rdf_atomic(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz,
                     int on_device, const int32_t *type, int64_t type_frame_stride,
                     const double *box, int n_rel, const int32_t *rel, double r_cut_sq,
                     double bin_size, int nbins, const double *edges, int per_frame,
                     uint64_t *hist_full, uint64_t *hist_part, uint64_t *overflow);

/*
 * The frame-summed form of mdhip_rdf_atomic (per_frame = 0) with the sums left ON THE DEVICE, for the multi-GPU
 * path: frames shard over one process per GPU and the (1 + n_rel) * nbins + 1 words are all-reduced over RCCL
 * straight from this buffer (mdproptools_amd/dist.py), no host round trip.
 *   out_dev    DEVICE uint64 [(1 + n_rel) * nbins + 1] = hist_full | hist_part | overflow (overwritten)
 * Complete (stream synchronised) on return, like every other call.
 */
int mdhip_rdf_atomic_dev(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz, int on_device,
                         const int32_t *type, int64_t type_frame_stride, const double *box, int n_rel,
                         const int32_t *rel, double r_cut_sq, double bin_size, int nbins, const double *edges,
                         uint64_t *out_dev);

int mdhip_rdf_atomic_async(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz, int on_device,
                           const int32_t *type, int64_t type_frame_stride, const double *box, int n_rel,
                           const int32_t *rel, double r_cut_sq, double bin_size, int nbins, const double *edges,
                           int per_frame, uint64_t *hist_full, uint64_t *hist_part, uint64_t *overflow);
int mdhip_rdf_atomic_dev_async(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz, int on_device,
                               const int32_t *type, int64_t type_frame_stride, const double *box, int n_rel,
                               const int32_t *rel, double r_cut_sq, double bin_size, int nbins, const double *edges,
                               uint64_t *out_dev);

/* ---- R4: _cn_loop --------------------------------------------------------- */
/*
 * structural/rdf_cn.py:100-119: cn[kl] += #pairs(rsq < r_cut_sq[kl]) with the same a/b double test;
 * every relation has its own cutoff. cn: host uint64 [n_frames][n_rel] or [n_rel].
 */
int mdhip_cn_atomic(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz,
                    int on_device, const int32_t *type, int64_t type_frame_stride,
                    const double *box, int n_rel, const int32_t *rel, const double *r_cut_sq,
                    int per_frame, uint64_t *cn);

/* mdhip_cn_atomic with per_frame = 0 and the n_rel counts written to a DEVICE buffer (all-reduced over RCCL by
 * mdproptools_amd/dist.py: cn_sharded). cn_dev: device uint64 [n_rel]. */
int mdhip_cn_atomic_dev(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz, int on_device,
                        const int32_t *type, int64_t type_frame_stride, const double *box, int n_rel,
                        const int32_t *rel, const double *r_cut_sq, uint64_t *cn_dev);

/* cn: host [n_frames][n_rel] / [n_rel], or (cn_on_device, per_frame = 0) device [n_rel]. */
int mdhip_cn_atomic_async(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz, int on_device,
                          const int32_t *type, int64_t type_frame_stride, const double *box, int n_rel,
                          const int32_t *rel, const double *r_cut_sq, int per_frame, uint64_t *cn, int cn_on_device);

/* ---- R3 + R4 in one sweep ------------------------------------------------------------------ */
/*
 * calc_atomic_rdf and calc_atomic_cn walk the same pairs of the same frames (structural/rdf_cn.py:385-530 and
 * 533-651; BASELINE config 3 asks for both): this call returns the outputs of mdhip_rdf_atomic AND of mdhip_cn_atomic
 * from ONE sweep. The packed-f32 sweep sends every pair that can lie inside the largest coordination cutoff to the
 * exact f64 chain, which bins it and counts it below the cutoffs it is below — the integers are those of the two
 * separate calls. cn_r_cut_sq host [n_rel] (each <= r_cut_sq for the single sweep; otherwise, and for geometries the
 * packed sweep does not cover, the two sweeps run one after the other inside this call).
 */
int mdhip_rdf_cn_atomic(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz, int on_device,
                        const int32_t *type, int64_t type_frame_stride, const double *box, int n_rel,
                        const int32_t *rel, double r_cut_sq, double bin_size, int nbins, const double *edges,
                        const double *cn_r_cut_sq, int per_frame, uint64_t *hist_full, uint64_t *hist_part,
                        uint64_t *overflow, uint64_t *cn);

int mdhip_rdf_cn_atomic_async(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz, int on_device,
                              const int32_t *type, int64_t type_frame_stride, const double *box, int n_rel,
                              const int32_t *rel, double r_cut_sq, double bin_size, int nbins, const double *edges,
                              const double *cn_r_cut_sq, int per_frame, uint64_t *hist_full, uint64_t *hist_part,
                              uint64_t *overflow, uint64_t *cn);

/* ---- R5: _rdf_mol_loop / _cn_mol_loop (atoms x sites, rectangular) --------- */
/*
 * structural/rdf_cn.py:122-141 and 144-162. Sites are molecule centres of mass
 * (or any second coordinate set): sites host|dev [n_frames][3][n_sites],
 * site_type host int32 [n_sites] (molecule type labels, shared by all frames).
 * +1 per in-cutoff (atom of type rel[kl][0], site of type rel[kl][1]); an atom's
 * own molecule is not excluded. hist_part / cn as above (no hist_full).
 */
int mdhip_rdf_sites(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz,
                    int xyz_on_device, const int32_t *type, int64_t n_sites, const double *sites,
                    int sites_on_device, const int32_t *site_type, const double *box, int n_rel,
                    const int32_t *rel, double r_cut_sq, double bin_size, int nbins,
                    const double *edges, int per_frame, uint64_t *hist_part, uint64_t *overflow);

int mdhip_cn_sites(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz,
                   int xyz_on_device, const int32_t *type, int64_t n_sites, const double *sites,
                   int sites_on_device, const int32_t *site_type, const double *box, int n_rel,
                   const int32_t *rel, const double *r_cut_sq, int per_frame, uint64_t *cn);

/* ---- R6 / M3: per-molecule centre of mass ---------------------------------- */
/*
 * structural/rdf_cn.py:218-241 (_define_mol_cols), common/com_mols.py:58-60 (calc_com),
 * dynamical/diffusion.py:83-89: mass-weighted mean over contiguous atom runs.
 *   attr       host|dev [n_frames][n_attr][n_atoms]
 *   atom_mass  host [n_atoms]; atom_q host [n_atoms] or NULL
 *   seg_off    host int64 [n_seg+1] (atoms of segment s are seg_off[s]..seg_off[s+1]-1)
 *   out        host|dev (out_on_device) [n_frames][n_attr][n_seg]:  sum(m*a) / sum(m), each product
 *              and sum a separate rounding, atoms added in index order
 *   seg_mass   host [n_seg] or NULL; seg_q host [n_seg] or NULL (sum of atom_q)
 * Floating point: sums are in a different order from pandas' Kahan group sum / BLAS dot; agreement with
 * the reference is to ~1e-15 relative (tests use rtol 1e-13).
 */
int mdhip_segment_com(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, int n_attr,
                      const double *attr, int attr_on_device, const double *atom_mass,
                      const double *atom_q, int64_t n_seg, const int64_t *seg_off, double *out,
                      int out_on_device, double *seg_mass, double *seg_q);

int mdhip_segment_com_async(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, int n_attr, const double *attr,
                            int attr_on_device, const double *atom_mass, const double *atom_q, int64_t n_seg,
                            const int64_t *seg_off, double *out, int out_on_device, double *seg_mass, double *seg_q);

/* ---- M1 / M2: frame-pair displacement reductions --------------------------- */
/*
 * dynamical/diffusion.py:212-218 and 225-237 as one primitive over a list of frame pairs.
 *   r          host|dev [n_frames][3][n_ent] entity coordinates (atoms or molecule COMs)
 *   scale      unit factor applied to every coordinate first (r*scale, diffusion.py:201-203)
 *   pairs      host int32 [n_pairs][2] = (t0, t1); M1 is {(0,t)}, M2 is {((k-1)tao, k tao)}
 *   group_off  host int64 [n_groups+1] contiguous entity groups (molecule types; one group for atoms)
 *   sums       host [n_pairs][n_groups][4]: sum over the group's entities of dx2, dy2, dz2 and
 *              (dx2+dy2)+dz2 (the caller divides by the group size: diffusion.py:218)
 *   per_entity host|dev (pe_on_device) [n_pairs][n_ent][4] = msd_all rows, or NULL
 * Tolerance vs the reference: rtol 1e-10 on means (fixed-order tree sums vs pandas' compensated sums).
 */
int mdhip_msd_pairs(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r,
                    int on_device, double scale, int n_pairs, const int32_t *pairs, int n_groups,
                    const int64_t *group_off, double *sums, double *per_entity, int pe_on_device);

/*
 * M1 for a FRAME SHARD (dynamical/diffusion.py:212-218 when the frames are dealt to one process per GPU): every frame
 * t of r against ONE origin frame that need not be among them — the frame at time 0, broadcast by the rank that owns
 * it — i.e. the frame pairs {(origin, t)}, t = 0..n_frames-1.
 *   origin     host|dev (origin_on_device) [3][n_ent]
 *   sums       host|dev (sums_on_device) [n_frames][n_groups][4]
 *   cols       NULL, or host|dev (cols_on_device) the four per-entity columns as in mdhip_msd_pairs_cols
 */
int mdhip_msd_origin(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r, int on_device,
                     const double *origin, int origin_on_device, double scale, int n_groups,
                     const int64_t *group_off, double *sums, int sums_on_device, double *cols, int64_t col_stride,
                     int cols_on_device);

/* mdhip_msd_pairs without per-entity rows and with the sums left in a DEVICE buffer [n_pairs][n_groups][4]: a rank's
 * frame shard of the single-origin MSD, all-gathered over RCCL from there (dist.py: msd_single_origin_sharded;
 * reference: dynamical/diffusion.py:212-218). */
int mdhip_msd_pairs_dev(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r, int on_device, double scale,
                        int n_pairs, const int32_t *pairs, int n_groups, const int64_t *group_off, double *sums_dev);

/*
 * The same reduction with the per-entity values stored as COLUMNS: dx2, dy2, dz2, msd, each [n_pairs][n_ent],
 * column k at cols + k * col_stride (col_stride >= n_pairs * n_ent, in doubles). These are the column blocks the
 * `msd_all` DataFrame (diffusion.py:212-217, 222) is made of, so the caller wraps them without a transpose.
 *   cols       host|dev (cols_on_device)
 */
int mdhip_msd_pairs_cols(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r,
                         int on_device, double scale, int n_pairs, const int32_t *pairs, int n_groups,
                         const int64_t *group_off, double *sums, double *cols, int64_t col_stride,
                         int cols_on_device);

/*
 * dynamical/diffusion.py:225-237 per entity: frames kept = 0, tao, 2 tao, ...; for every entity the
 * sums over the n_kept-1 windows of (x_k - x_{k-1})^2 per axis and of their total.
 *   win_sums   host [n_ent][4] (the caller applies /(n_kept-1) and, for the total, /n_kept — the
 *              reference's NaN-row quirk)
 */
int mdhip_msd_windows(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r,
                      int on_device, double scale, int tao, double *win_sums);

/* The same with win_sums in a DEVICE buffer [n_ent][4] (a rank's windows of the fixed-lag MSD, all-reduced over RCCL). */
int mdhip_msd_windows_dev(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r, int on_device,
                          double scale, int tao, double *win_sums_dev);

/*
 * Superset (not in the reference): full lag average
 *   msd[lag][g][c] = mean over t0 in [0, n_frames-lag) and entities of group g of the squared
 *   displacement, c = x, y, z, total; lag = 0..max_lag.   out host [max_lag+1][n_groups][4]
 * Tolerance: rtol 1e-10 against the plain double sum. The default evaluates S1(k) - 2 S2(k) with S2 from the power
 * spectrum (O(F log F)) whenever the rounding bound it computes for the data is <= 1e-10 relative to the MSD
 * (mdhip_last_rel_bound reports it); otherwise — e.g. coordinates far from the origin against a small displacement —
 * the difference kernel sums (r(t0+k) - r(t0))^2 directly.
 */
int mdhip_lag_msd(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r, int on_device,
                  double scale, int max_lag, int n_groups, const int64_t *group_off, double *out);

/* The same with the means in a DEVICE buffer [max_lag+1][n_groups][4] (a rank's entity slice; dist.py: lag_msd_sharded
 * turns them into sums on the device and all-reduces them). */
int mdhip_lag_msd_dev(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r, int on_device,
                      double scale, int max_lag, int n_groups, const int64_t *group_off, double *out_dev);

/* Asynchronous twins of the MSD calls of a sharded step (dist.py issues them back to back and waits once):
 * mdhip_msd_origin, mdhip_msd_pairs_dev, mdhip_msd_windows[_dev] (out_on_device), mdhip_lag_msd[_dev] (out_on_device;
 * the spectral path's bound check and its fallback to the difference kernel happen at completion). */
int mdhip_msd_origin_async(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r, int on_device,
                           const double *origin, int origin_on_device, double scale, int n_groups,
                           const int64_t *group_off, double *sums, int sums_on_device, double *cols, int64_t col_stride,
                           int cols_on_device);
int mdhip_msd_pairs_dev_async(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r, int on_device,
                              double scale, int n_pairs, const int32_t *pairs, int n_groups, const int64_t *group_off,
                              double *sums_dev);
int mdhip_msd_windows_async(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r, int on_device,
                            double scale, int tao, double *win_sums, int out_on_device);
int mdhip_lag_msd_async(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r, int on_device, double scale,
                        int max_lag, int n_groups, const int64_t *group_off, double *out, int out_on_device);
/* The STATUS of the mdhip_lag_msd* call issued last on this context, as one number in DEVICE memory, copied to
 * *dst_dev on the context's stream (behind that call's kernels, no host wait): the relative error bound of the spectral
 * path (what mdhip_last_rel_bound returns once the call has completed), +infinity when the in-kernel transposition of
 * that path did not complete (its result is then rewritten when the call completes), 0 when the exact-difference
 * kernel answered. The multi-GPU step adds this word to the buffer it all-reduces: every rank then knows, from the
 * reduced value alone, whether the lag sums have to be redone — without a host wait between the kernels and the
 * collective (mdproptools_amd/dist.py: msd_step_sharded_async). */
int mdhip_lag_msd_status_dev(mdhip_ctx *ctx, double *dst_dev);

/* ---- G1: per-frame charge flux --------------------------------------------- */
/*
 * dynamical/_conductivity.py:11-35: J[k][type] = sum over molecules of that type of
 * q_mol * v_com,k with v_com = sum(m v)/sum(m), both converted to SI first.
 *   vel        host|dev [n_frames][3][n_atoms]; atom_mass, atom_q host [n_atoms]
 *   seg_off    host int64 [n_seg+1]; seg_type host int32 [n_seg] in 0..n_types-1
 *   flux       host [3][n_types][n_frames] (the layout Conductivity.get_charge_flux returns)
 */
int mdhip_charge_flux(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *vel,
                      int on_device, const double *atom_mass, const double *atom_q, int64_t n_seg,
                      const int64_t *seg_off, const int32_t *seg_type, int n_types,
                      double vel_conv, double charge_conv, double *flux);

/* The same with flux in a DEVICE buffer [3][n_types][n_frames] (a rank's frames; all-gathered over RCCL — the reference's
 * only frame-parallel gather, dynamical/conductivity.py:190-194). */
int mdhip_charge_flux_dev(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *vel,
                          int on_device, const double *atom_mass, const double *atom_q, int64_t n_seg,
                          const int64_t *seg_off, const int32_t *seg_type, int n_types,
                          double vel_conv, double charge_conv, double *flux_dev);

int mdhip_charge_flux_async(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *vel, int on_device,
                            const double *atom_mass, const double *atom_q, int64_t n_seg, const int64_t *seg_off,
                            const int32_t *seg_type, int n_types, double vel_conv, double charge_conv, double *flux,
                            int flux_on_device);

/* ---- G2 / G3: correlation functions ---------------------------------------- */
#define MDHIP_XCORR_FFT 0    /* zero-padded FFT (reference: length 2n; here the next power of two >= 2n, same linear correlation): conductivity.py:109-114, viscosity.py:111-115 */
#define MDHIP_XCORR_DIRECT 1 /* direct lag sums:           viscosity.py:103-108 ("brute_force")        */
/*
 * out[p][k] = sum_{t=0}^{n-1-k} a_p[t+k] * b_p[t] / (n-k),  k = 0..n_lags-1, for n_pairs series pairs.
 *   a, b       host|dev [n_pairs][n] (b == a for an autocorrelation)
 *   out        host [n_pairs][n_lags]
 * Tolerance vs the reference: |err| <= 1e-10 * out[p][0].
 */
int mdhip_xcorr(mdhip_ctx *ctx, int64_t n, int n_pairs, const double *a, const double *b,
                int on_device, int method, int64_t n_lags, double *out);

/*
 * The lags lag_begin .. lag_begin + n_lags - 1 only (out host [n_pairs][n_lags]): the unit of the multi-GPU split of
 * the direct estimator, whose cost per lag k is n - k products — every GPU owns a lag range and needs the whole
 * series (mdproptools_amd/dist.py: xcorr_direct_sharded). Direct method only unless lag_begin == 0.
 */
int mdhip_xcorr_lags(mdhip_ctx *ctx, int64_t n, int n_pairs, const double *a, const double *b, int on_device,
                     int method, int64_t lag_begin, int64_t n_lags, double *out);

/* The same with out in a DEVICE buffer [n_pairs][n_lags] (a rank's lag range, all-gathered over RCCL). */
int mdhip_xcorr_lags_dev(mdhip_ctx *ctx, int64_t n, int n_pairs, const double *a, const double *b, int on_device,
                         int method, int64_t lag_begin, int64_t n_lags, double *out_dev);

int mdhip_xcorr_async(mdhip_ctx *ctx, int64_t n, int n_pairs, const double *a, const double *b, int on_device,
                      int method, int64_t n_lags, double *out);
int mdhip_xcorr_lags_dev_async(mdhip_ctx *ctx, int64_t n, int n_pairs, const double *a, const double *b, int on_device,
                               int method, int64_t lag_begin, int64_t n_lags, double *out_dev);

/* ---- G4: cumulative trapezoid ---------------------------------------------- */
/*
 * dynamical/viscosity.py:151 (cumtrapz) and conductivity.py:231 (cumulative_trapezoid):
 * I[k] = sum_{m<k} dx*(y[m]+y[m+1])/2. y host|dev [n_series][n];
 * out host [n_series][n-1], or [n_series][n] with a leading 0 when leading_zero != 0.
 */
int mdhip_cumtrapz(mdhip_ctx *ctx, int64_t n, int n_series, const double *y, int on_device,
                   double dx, int leading_zero, double *out);

/* The same with out in a DEVICE buffer [n_series][n-1 (+1)] (the running integral stays on the GPU). */
int mdhip_cumtrapz_dev(mdhip_ctx *ctx, int64_t n, int n_series, const double *y, int on_device, double dx,
                       int leading_zero, double *out_dev);
int mdhip_cumtrapz_async(mdhip_ctx *ctx, int64_t n, int n_series, const double *y, int on_device, double dx,
                         int leading_zero, double *out);
int mdhip_cumtrapz_dev_async(mdhip_ctx *ctx, int64_t n, int n_series, const double *y, int on_device, double dx,
                             int leading_zero, double *out_dev);

/* ---- G3 -> G4 in one call: the Green-Kubo chain without leaving the device ------------------------------------ */
/*
 * dynamical/viscosity.py:178-190 (_calc_3d_visc: autocorrelate, times PRESSURE_CONVERSION**2, calc_visc = V/(kB T) x
 * cumtrapz, mean over the three tensor components) and dynamical/conductivity.py:109-114 + 229-231 as ONE call: the
 * series go to the device once, the correlation functions never come back to be sent again.
 *   acf[p][k]       = ( sum_t a_p[t+k] b_p[t] / (n-k) ) * acf_scale,  k = 0..n-1     host [n_series][n], or NULL
 *   integral[p][k]  = integral_scale * cumtrapz(acf[p], dx)                          host [n_series][n-1 (+1 with leading_zero)]
 *   integral_mean   = mean over p of integral[p] (numpy's order: ((i0 + i1) + i2 ...) / n_series)   host [n-1 (+1)], or NULL
 * Each factor is applied as one multiplication of the finished value — the roundings of `acf * c**2` and
 * `np.multiply(c, integral)` on the host; 1.0 leaves values untouched. a, b host|dev [n_series][n] (b == a:
 * autocorrelation). Destinations in page-locked memory (mdhip_host_alloc) are written by DMA.
 */
int mdhip_green_kubo(mdhip_ctx *ctx, int64_t n, int n_series, const double *a, const double *b, int on_device,
                     int method, double acf_scale, double dx, double integral_scale, int leading_zero, double *acf,
                     double *integral, double *integral_mean);
int mdhip_green_kubo_async(mdhip_ctx *ctx, int64_t n, int n_series, const double *a, const double *b, int on_device,
                           int method, double acf_scale, double dx, double integral_scale, int leading_zero, double *acf,
                           double *integral, double *integral_mean);

/* ---- pinned host staging memory (reader -> pinned buffers -> H2D) -------- */
/*
 * Page-locked host memory for the frame batches a caller parses into (mdproptools_amd/stream.py): host|dev inputs
 * that live in it are copied by DMA at the PCIe rate, asynchronously to the host, instead of through the driver's
 * bounce buffers. MDHIP_ENODEV without a usable HIP runtime.
 */
int mdhip_host_alloc(size_t bytes, void **out);
/* The same from a thread that has made no HIP call yet (a reader thread): `device` is made the thread's current device
 * first, so that a rank bound to GPU k does not open a context on GPU 0; device < 0 = whatever is current. */
int mdhip_host_alloc_on(int device, size_t bytes, void **out);
void mdhip_host_free(void *p);

/* ---- I/O: native LAMMPS text-dump reader (host only, no GPU needed) ---------------------------- */
/* ---- neighbour-shell residence autocorrelation ---- */
/*
 * Replaces the two loops of ResidenceTime.calc_auto_correlation   dynamical/residence_time.py:70-146
 * for one relation: central atoms xi [F][3][n_i], shell atoms xj [F][3][n_j], box [F][3];
 *   h_ij(t) = (rsq > r_lo_sq) && (rsq <= r_hi_sq)                 residence_time.py:101-102
 * with the reference's single-wrap rsq (rdf_cn.py:44-57); exclude_diagonal != 0 (a type with itself; the two
 * sets must then be the same atoms in the same order) clears h_ii (residence_time.py:103-104).
 *   counts[k] = sum_{i,j} sum_t h_ij(t) h_ij(t+k),  k = 0..F-1    exact integers
 * are the numerators of the unbiased autocovariances the reference sums (residence_time.py:124-131):
 *   corr[k] = counts[k] / (F - k) / (n_i n_j) / corr[0]. n_records (optional) = sum_t |{h_ij(t) = 1}|.
 */
int mdhip_shell_residence(mdhip_ctx *ctx, int64_t n_frames, int64_t n_i, const double *xi, int xi_on_device,
                          int64_t n_j, const double *xj, int xj_on_device, const double *box, double r_lo_sq,
                          double r_hi_sq, int exclude_diagonal, uint64_t *counts, uint64_t *n_records);

/*
 * Replaces, for the inputs of the path, the un-vendored pymatgen `parse_lammps_dumps` + pandas
 * `read_csv` the reference uses (call sites structural/rdf_cn.py:176, dynamical/diffusion.py:172,
 * dynamical/conductivity.py:87). A file may hold several frames. Numbers are the correctly rounded
 * doubles of their text (identical to pandas for the <= 15 significant digits LAMMPS writes).
 */
typedef struct mdhip_dump mdhip_dump;
int mdhip_dump_open(const char *path, mdhip_dump **out); /* mmap + index the frames */
void mdhip_dump_close(mdhip_dump *d);
const char *mdhip_dump_error(mdhip_dump *d); /* d == NULL: error of the last failed open */
int64_t mdhip_dump_n_frames(mdhip_dump *d);
/* Header of frame f: bounds6 = xlo xhi ylo yhi zlo zhi as written, tilt3 = xy xz yz when triclinic;
 * columns = the names after "ITEM: ATOMS", space separated. Any output pointer may be NULL. */
int mdhip_dump_frame_info(mdhip_dump *d, int64_t f, int64_t *timestep, int64_t *natoms, double *bounds6,
                          double *tilt3, int *triclinic, int *n_cols, char *columns, int columns_len);
/* Columns col_idx[0..n_sel) of frame f as SoA planes out[n_sel][natoms], rows ordered by ascending value
 * of column sort_col (stable; e.g. the id column, as `sort_values("id")` in rdf_cn.py:192) or in file
 * order when sort_col < 0. Parsing is split over n_threads host threads. */
int mdhip_dump_read(mdhip_dump *d, int64_t f, int n_sel, const int32_t *col_idx, int sort_col, double *out,
                    int n_threads);

/* As mdhip_dump_read, with one destination plane [natoms] per selected column: outs[s] may point anywhere — the
 * streaming layer parses x, y, z straight into a slot of a page-locked batch buffer (mdhip_host_alloc) and the ids
 * and types into arrays of their own, so that no copy stands between the text and the H2D transfer. */
int mdhip_dump_read_cols(mdhip_dump *d, int64_t f, int n_sel, const int32_t *col_idx, int sort_col,
                         double *const *outs, int n_threads);

/*
 * MANY single-frame dump files in one call (the per-timestep files LAMMPS writes; the reference's glob pattern,
 * rdf_cn.py:176): file k is opened, indexed and parsed by one of n_threads host threads, column s of its atoms going to
 * dst[s] + k * dst_stride[s] (doubles), rows ordered by ascending sort_name (NULL: file order) — e.g. x, y, z straight
 * into frame slot k of a page-locked batch buffer [n_files][3][n_atoms] and the ids into a table of their own. Columns
 * are found by NAME in every file. timesteps [n_files], bounds6 [n_files][6] (as written), tilt3 [n_files][3],
 * triclinic [n_files]: headers, any may be NULL. Returns MDHIP_OK; a negative MDHIP_E* code with the first failure's
 * text in err; or 1 when some file is not ONE frame of n_atoms atoms with all the requested columns (err says which):
 * the caller then reads that batch frame by frame (mdhip_dump_open / mdhip_dump_read_cols).
 * cmp_equal (may be NULL) [n_files]: 1 where column cmp_sel of the file equals, bit for bit, cmp_ref [n_atoms] — or
 * the same column of the FIRST file when cmp_ref is NULL. The reference re-derives the atom types of every frame
 * (rdf_cn.py:462-470); a caller that knows they did not change skips that per-frame work.
 */
int mdhip_dump_read_files(const char *const *paths, int n_files, int n_sel, const char *const *col_names,
                          const char *sort_name, int64_t n_atoms, double *const *dst, const int64_t *dst_stride,
                          int64_t *timesteps, double *bounds6, double *tilt3, int32_t *triclinic, int n_threads,
                          char *err, int err_len, int cmp_sel, const double *cmp_ref, int32_t *cmp_equal);

/* ---- native LAMMPS log reader (host only) ---------------------------------------------------- */
/*
 * Role of pymatgen's parse_lammps_log (un-vendored; call sites dynamical/viscosity.py:211,
 * utilities/log.py:21, dynamical/diffusion.py:241): one thermo table per `run`, between the line starting with
 * "Memory usage per processor =" / "Per MPI rank memory allocation" and the line starting with "Loop time of";
 * first line = column names; lines starting with "WARNING" and blank lines are skipped. Numbers are parsed as in
 * the dump reader (correctly rounded). `regular` = 0 when some row is not one number per column (the caller then
 * falls back to its text reader for exact pandas behaviour). mdhip_log_read fills column planes
 * out[n_cols][n_rows]; is_int[c] = 1 when every token of column c is a plain integer (pandas makes it int64).
 */
typedef struct mdhip_log mdhip_log;
int mdhip_log_open(const char *path, mdhip_log **out);
void mdhip_log_close(mdhip_log *l);
const char *mdhip_log_error(mdhip_log *l); /* l == NULL: error of the last failed open */
int64_t mdhip_log_n_runs(mdhip_log *l);
int mdhip_log_run_info(mdhip_log *l, int64_t run, int64_t *n_rows, int *n_cols, int *regular, char *names,
                       int names_len);
int mdhip_log_read(mdhip_log *l, int64_t run, double *out, int32_t *is_int, int n_threads);

#ifdef __cplusplus
}
#endif
#endif /* MDHIP_H */
