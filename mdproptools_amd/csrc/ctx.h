// ctx.h — context, workspace and error plumbing shared by the libmdhip.so translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mdhip.h"

// Workspace slots: one growable device buffer each, reused across calls.
enum WsSlot {
    WS_XYZ_I = 0,  // staged coordinates of set i
    WS_XYZ_J,      // staged coordinates of set j / sites
    WS_TYPE_I,
    WS_TYPE_J,
    WS_BOX,
    WS_TABLES,  // edges + class table
    WS_HIST,    // class histograms (slots or per-frame)
    WS_MISC,    // overflow counters, small outputs
    WS_OUT,     // generic device output
    WS_PART,    // per-block partial sums
    WS_AUX0,
    WS_AUX1,
    WS_AUX2,
    WS_AUX3,
    WS_SORT_XYZ,   // spatially sorted coordinates (culled pair path)
    WS_SORT_TYPE,
    WS_KEYS,
    WS_CELLS,
    WS_BBOX,
    WS_LIST,
    WS_LISTCNT,
    WS_GSPH,
    WS_WSPH,
    WS_SORT_AOS,   // sorted atoms as (x, y, z, row-table offset) records, padded to whole tiles
    WS_ORIGIN,     // per-frame grid origin of the spatial sort
    WS_WORK,       // per-frame work counters of the scalar-j kernel
    WS_SORT_AOS_J, // culled atoms x sites: sorted records and boxes of the site set
    WS_BBOX_J,
    WS_GSPH_J,
    WS_WSPH_J,
    WS_GSPH4,      // 4-atom group boxes (scalar-j kernel)
    WS_GSPH4_J,
    WS_SLICES,     // per-block histogram copies of the scalar-j kernel
    WS_ROWS,       // their sums per output frame
    WS_REL,        // packed-f32 tile-relative records (scalar-j MODE 3)
    WS_CEN,        // tile centres + half extents
    WS_FFT_TMP,    // second transform buffer of fft_pow2.hip for the large-lag MSD path
    WS_FFT_TW,     // two-level table of the roots of unity of order L (fft_pow2.hip), kept from call to call
    WS_OUT2,       // second / third device result of a chained call (mdhip_green_kubo: running integrals, their mean)
    WS_OUT3,
    WS_SCAN,       // one-pass scan (scan.hip): block totals | flags | ticket counter, kept from call to call
    WS_XYZ_I2,     // second staging buffer of host-resident coordinates (asynchronous pair calls alternate between the two)
    WS_COUNT
};

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

// Pinned host staging memory: blocks of a pool owned by the context. A call takes the blocks it needs (small tables on
// their way to the device, results on their way back) and gives them back when it has COMPLETED — which for an
// asynchronous call is later than its return, so the copies are truly asynchronous and nothing a queued copy still
// reads is ever reused or freed under it.
struct PinBlock {
    void *p = nullptr;
    size_t cap = 0;
};

// What a completed call reports (mdhip_last_kernel_ms and friends; mdhip_call_stats for the calls before the last).
struct CallStats {
    double ms = 0.0, aux_ms = 0.0, rel_bound = 0.0;
    int launches = 0;
    const char *kernel = "";
    long long ticket = 0;  // the call's number (mdhip_last_ticket)
    int rc = 0;            // what the call's completion returned (mdhip_ticket_status), with its text
    std::string err;
    int fallbacks = 0;     // slow-path repeats the call took (the staged full-lag kernel's stalled ring, a guard re-run)
};

// One invocation of an entry point. Everything it enqueues goes to the context's stream; what is left to do on the
// host once that work has run (timer read-out, folding row sums, copying out of pinned staging, a rare re-run) is a list
// of completion steps. A synchronous call completes before it returns; an asynchronous one (the *_async entry points)
// returns with its work queued and completes inside mdhip_sync / mdhip_wait — in issue order.
struct mdhip_call {
    bool async = false;
    bool ended = false;
    mdhip_call *parent = nullptr;  // a call made from inside another one (always synchronous)
    std::vector<PinBlock> pins;
    std::vector<hipEvent_t> events;            // timing events, back to the pool at completion
    std::vector<std::function<int()>> steps;   // run in order once `done` has fired; the first error ends the list
    hipEvent_t done = nullptr;
    CallStats stats;                           // kernel name / launches as set while the call was issued
    bool inner = false;                        // made from inside another call or from a completion step
    CallStats outer;                           // ... the registers of the call around it, as they were
};

struct mdhip_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // calls (see mdhip_call): the one being issued, those issued asynchronously and not yet completed, and the pools
    mdhip_call *cur = nullptr;
    std::deque<mdhip_call *> inflight;
    std::vector<PinBlock> pin_free;
    std::vector<hipEvent_t> ev_free, done_free;
    std::deque<CallStats> history;  // completed calls, newest first (bounded)
    long long tickets = 0;          // top-level calls issued so far; the last one's number is mdhip_last_ticket
    bool want_async = false;        // set by an *_async entry point for the call it is about to make
    int completing = 0;             // > 0 while completion steps run (they may issue calls of their own)
    int deferred_rc = 0;            // first error of an asynchronous call that was completed on behalf of a later one
    std::string deferred_err;
    long long deferred_ticket = 0;  // ... and the call it belongs to (mdhip_ticket_status hands it to its owner)
    int cur_fallbacks = 0;          // slow-path repeats of the call being issued / completed (CallStats::fallbacks)
    long long fallbacks_total = 0;  // ... since the context was created (mdhip_fallbacks)
    int opt_sync_spin = 1;          // waiting for the stream: 1 poll the completion event (no interrupt wake-up latency)
                                    // for up to 100 ms, then block; 0 block at once (A/B)
    // host-resident pair inputs: the frames of batch k+1 are copied on this stream while batch k is swept (created on
    // first use); one event per batch in flight
    // Round 6: two streams whose kernels run on DISJOINT sets of CUs (hipExtStreamCreateWithCUMask; mask bit i = CU i div 8 of
    // XCD i mod 8): [0] three quarters of every XCD's CUs, for a compute-bound kernel with one workgroup per CU, [1] the other
    // quarter, for a streaming kernel that would otherwise run in front of it with the compute units idle. Created on first
    // use (mdhip_part_streams); part_state -1: not available on this device / driver.
    hipStream_t part_stream[2] = {nullptr, nullptr};
    hipEvent_t part_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // fork | ready[2] | free[2] | join
    int part_state = 0;
    int part_cus[2] = {0, 0};
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_ev[2] = {nullptr, nullptr};
    // asynchronous pair calls on host-resident frames: the whole trajectory of call k + 1 is copied (copy stream) into
    // the staging buffer call k does NOT use, while call k's kernels run; stage_ev[b]: recorded on the launch stream
    // behind the last kernels that read buffer b
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    bool stage_used[2] = {false, false};
    int stage_flip = 0;
    int opt_rdf_relblock = 0;  // (retired: the f32 records are relative to their whole tile's centre; accepted, ignored)
    int opt_h2d_overlap = 1;  // 1 (default): overlapped staging of host-resident pair inputs, 0: one copy up front (A/B)
    int opt_small_copy = 1;   // 1 (default): copies of up to MD_SMALL_COPY_MAX bytes between page-locked host memory and the
                              // device are made by a KERNEL on the launch stream (mdhip_copy_small) instead of
                              // hipMemcpyAsync: the runtime's copies run as blit kernels on another hardware queue,
                              // and every hand-over between the queues cost a C2 step ~17 us of idle GPU (round 5)
    int opt_h2d_ring = 1;     // 1 (default): pageable sources go through the context's page-locked ring (mdhip_h2d_any),
                              // 0: handed to hipMemcpyAsync as they are (A/B)
    // the ring: two halves, each with the event recorded behind its last DMA (a half is reused once that has fired)
    // (round 6, ADVICE r05: each half is H2D_RING_CHUNKS chunks of 8 MB cycled through with an event per chunk — a half
    // sized to the whole copy pinned up to twice the trajectory, ~4.8 GB at C3, until mdhip_destroy)
    static constexpr int H2D_RING_CHUNKS = 4;
    void *h2d_ring[2] = {nullptr, nullptr};
    hipEvent_t h2d_chunk_ev[2][H2D_RING_CHUNKS] = {};
    bool h2d_chunk_used[2][H2D_RING_CHUNKS] = {};
    int h2d_chunk_next[2] = {0, 0};
    struct CopyPool *copy_pool = nullptr;  // helper threads of mdhip_h2d_any (created on first use)
    std::string err;
    DevBuf ws[WS_COUNT];
    double last_ms = 0.0;
    double last_aux_ms = 0.0;  // device time of the preparation kernels of the last call (e.g. spatial sort)
    int last_launches = 0;
    const char *last_kernel = "";  // dominant kernel of the last call (static string)
    int cu_count = 256;
    size_t lds_max = 65536;
    char name[256] = {0};
    // options (A/B knobs; results never depend on them)
    int opt_rdf_variant = 1;   // 1 = fast kernel (default), 0 = table-lookup loops (A/B baseline)
    int opt_rdf_jsplit = 0;   // 0 = auto
    int opt_rdf_fpb = 0;      // frames per block of the fast kernel, 0 = auto
    int opt_rdf_sj = 1;       // culled path: 1 = scalar-j kernel (waves independent; persistent grid when the
                              // output is frame-summed), 2 = scalar-j with one block per (frame, tile, slice),
                              // 0 = LDS-tile kernel
    int opt_rdf_batch = 0;    // frames per batch of the pair path, 0 = auto (workspace-bounded)
    int opt_rdf_cull = -1;    // spatial culling of tile pairs: -1 = auto, 0 = never, 1 = always (when applicable)
    int opt_rdf_slots = 16;   // replicas of the frame-summed histogram in HBM
    int opt_rdf_inflight = 1; // per-frame output: frames in flight per XCD (their records should stay in its L2)
    int opt_rdf_rows = -1;    // scalar-j RDF: -1/1 ordered-pair rows without a row table when they fit, 0 class rows + table
    int opt_rdf_disp = -1;    // ordered rows: -1/1 displaced rows (row = A[ti] + B[tj], pair_hist.hip) when the plain
                              // n_ti x n_tj layout does not fit LDS and the displaced one does, 0 never, 2 whenever
                              // it has fewer rows (A/B)
    int opt_rdf_big = -1;      // ordered rows that fit neither a third of LDS nor a displaced layout: -1/1 ONE 16-wave block per
                               // CU with the whole LDS (pair_hist_sj_kernel<., ., false, true>), 0 class rows in passes (A/B)
    int opt_rdf_pk_passes = 1; // packed class-row sweep in several passes when the classes do not fit LDS at once (round 6);
                               // 0: such calls take the all-f64 class-row kernel as before (A/B)
    int opt_residence_cap = 0; // mdhip_shell_residence: > 0 = slots of the first sweep's pair table (tests: forces the re-sweep)
    int opt_rdf_pk = -1;      // scalar-j RDF with ordered rows: -1/1 packed-f32 classification sweep with the exact
                              // deferred resolver (MODE 3) when its error bound allows, 0 the all-f64 sweep (MODE 2)
    int opt_rdf_sort = -1;    // spatial sort: -1 auto, 1 one block per frame (LDS counters), 0 multi-block (global counters),
                              // 3 as auto but without the read-once form for frames of <= 12288 atoms (A/B)
    int opt_cn_pk = 1;        // mdhip_cn_atomic: 1 (default since the scalar-stream cut of the packed sweep) =
                              // through the packed-f32 sweep (coarse 64-bin histogram up to the largest cutoff + split
                              // bins) when the geometry allows, 0 = the f64 edge-table kernel (also the fallback).
                              // Same integers (soaked against each other and the oracle); C3: 4.78 against 5.11 ms per
                              // 64 frames, C2 1.60 against 1.70 ms
    int opt_rdf_guard = 0;    // overflow guard of the 32-bit LDS histogram words: neighbour tiles a block may sweep
                              // per launch (0 = the real bound, 2^32 / (64 * 256) with margin; tests lower it)
    int opt_seg_cap = 0;      // segment kernels: atoms per block stage: 0 = 1024 or 512 by the shape of the segment table (by-frame
                              // kernel, pick_seg_cap), 1024 for the staged one; 1024, 512, or 256 = one wave per block (A/B)
    int opt_seg_vec = 1;      // segment kernels: 16-byte loads when alignment allows (default), 0 = 8-byte loads (A/B)
    int opt_fft_logr = 10;    // fft_pow2.hip: largest radix of a pass (log2, 4..10); passes of radix >= 2^9 run the radix-8
                              // network (fft_pass8_kernel), so that 2^20 points take two passes (round 2: 8 = three passes)
    int opt_fft_net8 = 1;     // 0: radix-4 network for every radix (A/B)
    int opt_fft_specfuse = 1; // mdhip_fft_xcorr: the spectrum step inside the inverse transform's first pass when that pass
                              // runs the radix-8 network (0: a kernel of its own, A/B)
    int opt_fft_mid = 1;      // mdhip_fft_xcorr, autocorrelation through a two-pass transform: 1 (default) three launches
                              // (fft_mid_acf_kernel: second pass + spectrum + inverse's first pass in one), 0 four
    int opt_fft_logc = 3;     // fft_pow2.hip: columns per tile (log2); 8 columns = 128-byte runs measured best (tools/ab_fft.py)
    int opt_seg_gy = 0;       // segment kernels: frame slices per block run (0 = auto)
    int opt_seg_frame = 1;    // mdhip_segment_com: one (run, frame) per block, nothing carried between frames (A/B: 0 =
                              // the software-pipelined staged kernel)
    int opt_xcorr_tile = 0;
    int opt_lag_residue = 1;      // full-lag MSD with 16 384 < F + max_lag <= 24 576: 1 (default) the residue-class kernel (msd_fft_w12r.h), 0 the batched transforms
    int opt_lag_mean_sample = -1; // long-series paths: the series are centred on the mean of about this many sampled frames (-1: 512; 0: every frame)
    int opt_lag_w1 = 1;           // full-lag MSD of F <= 1536 frames (F + max_lag <= 3072): 1 (default) one wave per series (msd_power_w1_kernel), 0 the block-wide kernels
    int opt_lag_overlap = 0;      // long-series full-lag MSD: 1 = the transposition of batch k + 1 runs on a quarter of the CUs while the
                                  // transform kernel of batch k runs on the others (CU-masked streams; 2: whatever the size, tests);
                                  // 0 (default) one after the other — measured: 16.7 against 13.2 ms at F = 10 000, a quarter of the CUs
                                  // moves 1.7 TB/s where the whole chip moves 4.8 (profiles/r06_ab_lag_overlap.txt)
    int opt_lag_ends = 1;         // lag_variant 3 with the bound missed at <= 24 lags per end of the lag range: 1 (default) those lags from the
                                  // difference form, the rest of the spectral result stands; 0 the whole call to the difference kernel
    int opt_lag_batch_mb = 4096;  // batched full-lag path: device memory of one batch of series (centred series + transform buffers), MB
    int opt_lag_w12_min_f = 1536;  // full-lag MSD with 2048 < F + max_lag <= 8192: from this many frames on the 12288-point
                                   // kernel (msd_fft_w12.h) instead of the 8192-point one; 0 = never, >= 1536
    int opt_lag_batched_fuse = 2;  // full-lag MSD of series beyond the fused kernels: 1 (round 6) the first pass reads
                                   // the centred series where they are (implicit padding) and |X|^2 is reduced straight
                                   // from the packed transform; 2 (default) as 1 in two passes, the second one fused with
                                   // that reduction (the packed transform is never written); 0 the round-2 sequence
                                   // (padded copy, half spectra) — A/B
    int opt_lag_variant = 3;  // full-lag MSD: 3 (default) = autocorrelation theorem (msd_fft.hip) when its error bound
                              // stays below 1e-10, else the exact-difference kernel; 1 = series-resident LDS
                              // difference kernel when it fits, 0 = staged difference kernel, 2 = always the
                              // autocorrelation theorem, 4 = 2 through batched global transforms (fft_pow2.hip)
    int scan_capacity = 0;     // scan.hip: tiles per launch of the one-pass scan; 0: its words have to be filled (first use, after a stall)
    int scan_parity = 0;       // ... which of the two sets of totals the next launch publishes into
    int fft_tw_logL = -1;         // length the table in WS_FFT_TW was built for (-1: none)
    const void *fft_tw_ptr = nullptr;
    int opt_lag_fft_kernel = 3;   // fused full-lag MSD path: 3 (default, round 5) = as 2, and padded length 12288 = 12 x 1024
                                  // (twelve waves, register-resident 512-point sub-transforms: msd_fft_w12.h) where the
                                  // next power of two would be 16384; 2 = first pass from registers + wave-private
                                  // sub-transforms where the series is long enough (msd_power_lds3_kernel), else as 1;
                                  // 1 = conflict-free LDS layout, bilinear spectrum sums (msd_power_lds2_kernel);
                                  // 0 = the round-2 kernel (A/B; also what short series fall back to)
    int opt_lag_direct = -1;      // fused full-lag MSD path, where the power kernel gets its time series from (-1: the default,
                                  // LAG_DIRECT_DEFAULT in msd_fft.hip = 2): 0 = a transposed copy [3 E][F] made by a pass
                                  // of its own (rounds 2-3: 2.2 of the call's 6.8 ms at C4, 6 GB of workspace); 1 = the
                                  // trajectory [F][3][E] itself, blocks in XCD clusters of 16 that share every line they fetch
                                  // (a third of the traffic, but 9.4 ms: a lane's 8 bytes cost the vector-memory path a whole
                                  // line); 2 = clusters of 16 blocks transpose their tiles inside the kernel through a ring in
                                  // device memory, handed from block to block (5.0 ms; shapes it does not take fall back to 0)
    double last_rel_bound = 0.0;  // error bound reported by the FFT MSD path of the last mdhip_lag_msd call (0: exact path)
    const double *lag_status_dev = nullptr;  // device word of the last mdhip_lag_msd call issued: its error bound, +inf when
                                             // the staged kernel's ring stalled; nullptr when the exact path answered
};

int mdhip_fail(mdhip_ctx *ctx, int code, const char *fmt, ...);
// Host-to-device copy of `bytes` on `stream` that overlaps with kernels WHATEVER kind of host memory `src` is (round 5).
// Page-locked sources: one hipMemcpyAsync, a DMA the host does not wait for. Pageable sources (what a caller holding
// plain numpy arrays passes): hipMemcpyAsync would stage them through the runtime's bounce buffers with the host
// waiting AND — measured at C2 — without overlapping the kernels of the call before (pipelined steps 3.64 ms against
// 2.86 for page-locked frames); here the bytes go through a page-locked ring of the context instead: a few helper
// threads copy them chunk by chunk (the host waits for that memcpy, ~40 GB/s), each chunk's DMA is queued as soon as
// it is there. `slot` 0 / 1: which half of the ring (two calls' worth may be in flight). Returns when every byte of
// `src` has been read.
bool mdhip_part_streams(mdhip_ctx *ctx);  // the CU-partitioned streams exist (created on first use)
int mdhip_h2d_any(mdhip_ctx *ctx, void *dst_dev, const void *src, size_t bytes, hipStream_t stream, int slot);
// msd_fft.hip: full-lag MSD through batched FFTs; d_r device [F][3][E], out host [max_lag+1][G][4]
// fft_pow2.hip: batched power-of-two FP64 real transforms (half spectra [batch][L/2+1]); d_tmp holds batch * L/2
// complex points; r2c overwrites its input, c2r is the unnormalised inverse
int mdhip_fft_r2c(mdhip_ctx *ctx, double *d_real, double2 *d_tmp, double2 *d_spec, long long L, int batch);
int mdhip_fft_c2r(mdhip_ctx *ctx, const double2 *d_spec, double2 *d_tmp, double *d_real, long long L, int batch);
int mdhip_fft_r2c_packed(mdhip_ctx *ctx, const double *d_series, long long n, double2 *d_buf0, double2 *d_buf1, long long L,
                         int batch, const double2 **Z_out);
int mdhip_fft_power_rows(mdhip_ctx *ctx, const double2 *Z, long long L, long long row0, long long row1, int splits,
                         double *d_partial);
// ... in two passes, the second one fused with the column sums (fft_power_pass_kernel): the packed transform is never written
bool mdhip_fft_power2_plan(const mdhip_ctx *ctx, long long L);
int mdhip_fft_first_perm(mdhip_ctx *ctx, const double *d_series, long long n, double2 *d_buf, long long L, int batch);
int mdhip_fft_power_pass(mdhip_ctx *ctx, const double2 *d_buf, long long L, long long row0, long long row1, int splits,
                         double *d_partial);
// the fused FFT estimator of xcorr.hip: series in, scaled lags out; buf0..3 hold batch * L/2 complex points each
int mdhip_fft_xcorr(mdhip_ctx *ctx, const double *d_a, const double *d_b, long long n, long long L, int batch,
                    double2 *buf0, double2 *buf1, double2 *buf2, double2 *buf3, long long n_lags, double *d_lags,
                    double out_scale);
// scan.hip: cumulative trapezoid of device series y [n_series][n] into d_out [n_series][n - 1 + lead], on the stream
// (every finished value times post_scale: one more rounding, as a multiplication of the finished array)
int mdhip_cumtrapz_enqueue(mdhip_ctx *ctx, int64_t n, int n_series, const double *d_y, double dx, int lead, double *d_out,
                           double post_scale);

void *mdhip_ws(mdhip_ctx *ctx, int slot, size_t bytes);  // nullptr on failure (error set)
// pinned host memory that belongs to the call being issued (ctx->cur) until it completes; nullptr on failure (error set)
void *mdhip_pin(mdhip_ctx *ctx, size_t bytes);
hipEvent_t mdhip_timer_event(mdhip_ctx *ctx);  // a timing event that belongs to the call being issued; nullptr on failure
// Waits until everything queued on the context's stream so far has run (polling or blocking: opt_sync_spin).
hipError_t mdhip_stream_wait(mdhip_ctx *ctx);
mdhip_call *mdhip_call_begin(mdhip_ctx *ctx);
int mdhip_call_end(mdhip_ctx *ctx, mdhip_call *call);      // see CallScope::end
void mdhip_call_abandon(mdhip_ctx *ctx, mdhip_call *call);  // an error return: drain the stream, drop the steps
int mdhip_complete_inflight(mdhip_ctx *ctx, size_t keep);  // completes all but the newest `keep` asynchronous calls

#define MD_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess)                                                                \
            return mdhip_fail(ctx, MDHIP_EHIP, "%s failed: %s (%s:%d)", #call,                \
                              hipGetErrorString(e__), __FILE__, __LINE__);                    \
    } while (0)

#define MD_REQUIRE(cond, ...)                                  \
    do {                                                       \
        if (!(cond)) return mdhip_fail(ctx, MDHIP_EINVAL, __VA_ARGS__); \
    } while (0)

#define MD_WS(var, type, slot, bytes)                          \
    type *var = (type *)mdhip_ws(ctx, slot, bytes);            \
    if (!var) return MDHIP_ENOMEM;

#define MD_PIN(var, type, bytes)                               \
    type *var = (type *)mdhip_pin(ctx, bytes);                 \
    if (!var) return MDHIP_ENOMEM;

// Scope of one entry-point invocation (see mdhip_call). Construct it first thing; `return cs.end()` on the way out
// with success; every other return (MD_HIP / MD_REQUIRE failures) abandons the call in the destructor.
struct CallScope {
    mdhip_ctx *ctx;
    mdhip_call *call;
    explicit CallScope(mdhip_ctx *c) : ctx(c), call(c ? mdhip_call_begin(c) : nullptr) {}
    CallScope(const CallScope &) = delete;
    CallScope &operator=(const CallScope &) = delete;
    ~CallScope()
    {
        if (call) mdhip_call_abandon(ctx, call);  // (never reached end(): an error return)
    }
    bool async() const { return call && call->async; }
    bool ended() const { return call == nullptr; }
    // host work to do once the queued device work has run: fn() -> MDHIP_* code
    template <class F>
    void defer(F &&fn)
    {
        call->steps.emplace_back(std::forward<F>(fn));
    }
    // Synchronous call: waits for the stream, runs the completion steps, returns the first error. Asynchronous call:
    // leaves the call in flight and returns MDHIP_OK.
    int end()
    {
        mdhip_call *c = call;
        if (!c) return MDHIP_EINVAL;
        call = nullptr;  // (the call object belongs to the context from here on: it may be gone when this returns)
        return mdhip_call_end(ctx, c);
    }
};

// What an *_async entry point does before it runs the body it shares with its synchronous twin: the next top-level
// call of this context is issued asynchronously. Cleared on the way out whatever happened.
struct AsyncCall {
    mdhip_ctx *ctx;
    explicit AsyncCall(mdhip_ctx *c) : ctx(c) { c->want_async = true; }
    ~AsyncCall() { ctx->want_async = false; }
};

// msd_fft.hip: full-lag MSD through the autocorrelation theorem. d_r device [F][3][E]. Enqueues the transforms and
// defers the long-hand finish: once the call has completed, res->out [max_lag+1][G][4] holds the means (as
// mdhip_lag_msd) and res->bound the largest estimated relative rounding error over all (lag >= 1, group, axis) entries.
struct LagFftResult {
    std::vector<double> out;       // the means, when the path finished on the host (!delivered)
    std::vector<int64_t> group_off;
    double bound = 0.0;
    bool delivered = false;        // the fused path: finished on the device and copied to the caller's buffer on the stream
    // (round 6) which lags miss the 1e-10 bound: none beyond [1, k_lo] and [k_hi, max_lag]; the bound of all the others
    bool ends_valid = false;
    long long k_lo = 0, k_hi = 0;
    double bound_ok = 0.0;
};
int mdhip_lag_msd_fft(CallScope &cs, int64_t n_frames, int64_t n_ent, const double *d_r, double scale, int max_lag,
                      int n_groups, const int64_t *group_off, const std::shared_ptr<LagFftResult> &res, double *out,
                      int out_on_device);

// Finished values (host memory) on their way to a DEVICE result buffer from inside a completion step. The copy runs on
// the context's copy stream, not on the launch stream: a later call's kernels may already be queued there (calls are
// issued ahead of the completion of the ones before them), and a copy behind them would make this completion — and
// with it the caller's next issue — wait for kernels it has nothing to do with. Complete when this returns.
int mdhip_deliver_to_device(mdhip_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);

// dst <- src, `bytes` bytes, either side page-locked host memory (device-accessible at the same address) or device
// memory, as ONE kernel on the launch stream when the copy is small (see opt_small_copy), else hipMemcpyAsync of `kind`.
constexpr size_t MD_SMALL_COPY_MAX = (size_t)256 << 10;
int mdhip_copy_small(mdhip_ctx *ctx, void *dst, const void *src, size_t bytes, hipMemcpyKind kind);

// A small host table on its way to device memory: through pinned staging of the call, so that the copy is asynchronous
// and the caller's (or this function's) memory may go away at once.
static inline int mdhip_h2d_small(mdhip_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes)
{
    if (bytes == 0) return MDHIP_OK;
    void *h = mdhip_pin(ctx, bytes);
    if (!h) return MDHIP_ENOMEM;
    memcpy(h, src_host, bytes);
    return mdhip_copy_small(ctx, dst_dev, h, bytes, hipMemcpyHostToDevice);
}

// A result on its way to the caller. Device destination: a device-to-device copy on the stream. Host destination: up to
// MD_STAGE_MAX bytes go through pinned staging of the call (asynchronous copy now, memcpy in a completion step);
// larger results are copied straight into the caller's memory (a DMA when that memory is page-locked, else the
// runtime's own staged copy, which blocks the host until it is done — correct either way).
constexpr size_t MD_STAGE_MAX = (size_t)4 << 20;
static inline int mdhip_result(CallScope &cs, void *dst, const void *d_src, size_t bytes, int dst_on_device)
{
    mdhip_ctx *ctx = cs.ctx;
    if (bytes == 0) return MDHIP_OK;
    if (dst_on_device) {
        MD_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        return MDHIP_OK;
    }
    if (bytes > MD_STAGE_MAX) {
        MD_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        return MDHIP_OK;
    }
    void *h = mdhip_pin(ctx, bytes);
    if (!h) return MDHIP_ENOMEM;
    {
        const int rcc = mdhip_copy_small(ctx, h, d_src, bytes, hipMemcpyDeviceToHost);
        if (rcc) return rcc;
    }
    cs.defer([dst, h, bytes]() {
        memcpy(dst, h, bytes);
        return MDHIP_OK;
    });
    return MDHIP_OK;
}

// Stage `bytes` from a host-or-device source into workspace `slot` unless it is already on the device.
static inline const void *mdhip_stage(mdhip_ctx *ctx, int slot, const void *src, size_t bytes,
                                      int on_device, int *rc)
{
    *rc = MDHIP_OK;
    if (on_device) return src;
    void *d = mdhip_ws(ctx, slot, bytes);
    if (!d) {
        *rc = MDHIP_ENOMEM;
        return nullptr;
    }
    hipError_t e = hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) {
        *rc = mdhip_fail(ctx, MDHIP_EHIP, "H2D staging failed: %s", hipGetErrorString(e));
        return nullptr;
    }
    return d;
}

// Results: device workspace -> the caller's buffer, which is host memory for the plain entry points and device
// memory for the *_dev ones (the multi-GPU layer hands those buffers to RCCL without a host round trip).
static inline hipError_t mdhip_deliver(mdhip_ctx *ctx, void *dst, const void *d_src, size_t bytes, int dst_on_device)
{
    if (bytes == 0) return hipSuccess;
    return hipMemcpyAsync(dst, d_src, bytes, dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                          ctx->stream);
}

// Zeroes a result buffer (host: at once; device: on the stream — complete when the call is).
static inline int mdhip_zero_result(mdhip_ctx *ctx, void *dst, size_t bytes, int dst_on_device)
{
    if (bytes == 0 || !dst) return MDHIP_OK;
    if (!dst_on_device) {
        memset(dst, 0, bytes);
        return MDHIP_OK;
    }
    const hipError_t e = hipMemsetAsync(dst, 0, bytes, ctx->stream);
    if (e != hipSuccess) return mdhip_fail(ctx, MDHIP_EHIP, "zeroing a device result failed: %s", hipGetErrorString(e));
    return MDHIP_OK;
}

// Device time of a run of launches: an event pair of the call being issued. collect() belongs in a completion step
// (the events have fired by then); the main timer's value becomes the call's `ms`, an `aux` timer's is returned only.
struct KernelTimer {
    mdhip_ctx *ctx;
    hipEvent_t e0, e1;
    bool aux;
    explicit KernelTimer(mdhip_ctx *c, int launches = 1, bool aux_ = false)
        : ctx(c), e0(mdhip_timer_event(c)), e1(mdhip_timer_event(c)), aux(aux_)
    {
        if (!aux) ctx->last_launches = launches;
        if (e0) (void)hipEventRecord(e0, ctx->stream);
    }
    void stop()
    {
        if (e1) (void)hipEventRecord(e1, ctx->stream);
    }
    double collect() const
    {
        float ms = 0.f;
        if (!e0 || !e1 || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return 0.0;
        if (!aux) ctx->last_ms = ms;
        return ms;
    }
};
