// mdhip_ctx.hip — lifecycle, workspace, options and the host-side bin-edge table.
#include <algorithm>
#include <vector>
#include <cmath>
#include <cstdlib>
#include <ctime>

#include <condition_variable>
#include <mutex>
#include <thread>

#include "ctx.h"

static thread_local std::string g_create_error;

int mdhip_fail(mdhip_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->err = buf;
    else
        g_create_error = buf;
    return code;
}

void *mdhip_ws(mdhip_ctx *ctx, int slot, size_t bytes)
{
    DevBuf &b = ctx->ws[slot];
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return b.p;
    if (b.p) {
        // the buffer may still be in use by queued work
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    size_t cap = bytes + (bytes >> 3);
    cap = (cap + 255) & ~size_t(255);
    hipError_t e = hipMalloc(&b.p, cap);
    if (e != hipSuccess) {
        b.p = nullptr;
        mdhip_fail(ctx, MDHIP_ENOMEM, "hipMalloc(%zu) failed for workspace %d: %s", cap, slot,
                   hipGetErrorString(e));
        return nullptr;
    }
    b.cap = cap;
    return b.p;
}

void *mdhip_pin(mdhip_ctx *ctx, size_t bytes)
{
    if (!ctx->cur) {
        mdhip_fail(ctx, MDHIP_EINVAL, "internal: pinned staging requested outside a call");
        return nullptr;
    }
    if (bytes == 0) bytes = 16;
    // the smallest free block that is large enough
    int best = -1;
    for (size_t k = 0; k < ctx->pin_free.size(); ++k)
        if (ctx->pin_free[k].cap >= bytes && (best < 0 || ctx->pin_free[k].cap < ctx->pin_free[(size_t)best].cap))
            best = (int)k;
    PinBlock b;
    if (best >= 0) {
        b = ctx->pin_free[(size_t)best];
        ctx->pin_free.erase(ctx->pin_free.begin() + best);
    } else {
        size_t cap = bytes + (bytes >> 2);
        if (cap < 65536) cap = 65536;
        cap = (cap + 4095) & ~size_t(4095);
        const hipError_t e = hipHostMalloc(&b.p, cap, hipHostMallocDefault);
        if (e != hipSuccess) {
            mdhip_fail(ctx, MDHIP_ENOMEM, "hipHostMalloc(%zu) failed for a staging block: %s", cap, hipGetErrorString(e));
            return nullptr;
        }
        b.cap = cap;
    }
    ctx->cur->pins.push_back(b);
    return b.p;
}

hipEvent_t mdhip_timer_event(mdhip_ctx *ctx)
{
    hipEvent_t e = nullptr;
    if (!ctx->ev_free.empty()) {
        e = ctx->ev_free.back();
        ctx->ev_free.pop_back();
    } else if (hipEventCreate(&e) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (ctx->cur)
        ctx->cur->events.push_back(e);
    else
        ctx->ev_free.push_back(e);  // (no call to own it: usable at once, never collected)
    return e;
}

static inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}

// Polls `ev` (already recorded) instead of sleeping on it: a blocked host thread is woken by an interrupt and pays the
// scheduler's latency on top — tens of microseconds on a quiet host, milliseconds on a busy one — and the calls of this
// library are a few milliseconds long. After 100 ms of polling the wait turns into a blocking one.
static hipError_t wait_event(mdhip_ctx *ctx, hipEvent_t ev)
{
    if (ctx->opt_sync_spin) {
        timespec t0;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (unsigned it = 0;; ++it) {
            const hipError_t q = hipEventQuery(ev);
            if (q != hipErrorNotReady) return q;
            for (int k = 0; k < 32; ++k) cpu_relax();
            if ((it & 63) == 63) {
                timespec t1;
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if ((t1.tv_sec - t0.tv_sec) * 1000000000LL + (t1.tv_nsec - t0.tv_nsec) > 100000000LL) break;
            }
        }
    }
    return hipEventSynchronize(ev);
}

static hipEvent_t take_done_event(mdhip_ctx *ctx)
{
    hipEvent_t e = nullptr;
    if (!ctx->done_free.empty()) {
        e = ctx->done_free.back();
        ctx->done_free.pop_back();
        return e;
    }
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return e;
}

hipError_t mdhip_stream_wait(mdhip_ctx *ctx)
{
    if (!ctx->opt_sync_spin) return hipStreamSynchronize(ctx->stream);
    hipEvent_t e = take_done_event(ctx);
    if (!e) return hipStreamSynchronize(ctx->stream);
    hipError_t rc = hipEventRecord(e, ctx->stream);
    if (rc == hipSuccess) rc = wait_event(ctx, e);
    ctx->done_free.push_back(e);
    return rc;
}

mdhip_call *mdhip_call_begin(mdhip_ctx *ctx)
{
    const bool top = ctx->cur == nullptr && ctx->completing == 0;
    const bool async = ctx->want_async && top;
    ctx->want_async = false;
    if (top && !async && !ctx->inflight.empty()) {
        // a synchronous call drains what is in flight BEFORE it issues anything: its own staging (copy stream, the
        // batch-by-batch path of host-resident frames) is not ordered behind the kernels of the calls before it
        const int prev = mdhip_complete_inflight(ctx, 0);
        if (prev != MDHIP_OK && ctx->deferred_rc == MDHIP_OK) ctx->deferred_rc = prev;
    }
    mdhip_call *c = new mdhip_call();
    c->parent = ctx->cur;
    c->async = async;
    if (!c->parent && ctx->completing == 0) {
        c->stats.ticket = ++ctx->tickets;
    } else {
        c->inner = true;
        c->outer.ms = ctx->last_ms;
        c->outer.aux_ms = ctx->last_aux_ms;
        c->outer.launches = ctx->last_launches;
        c->outer.kernel = ctx->last_kernel;
    }
    ctx->cur = c;
    // the registers a call reports through start from nothing: what is in them when the call has been issued is its own
    ctx->last_ms = 0.0;
    ctx->last_aux_ms = 0.0;
    ctx->last_launches = 0;
    ctx->last_kernel = "";
    return c;
}

static void release_call(mdhip_ctx *ctx, mdhip_call *c)
{
    for (const PinBlock &b : c->pins) ctx->pin_free.push_back(b);
    for (hipEvent_t e : c->events) ctx->ev_free.push_back(e);
    if (c->done) ctx->done_free.push_back(c->done);
    delete c;
}

// The call's queued work has to be over: waits for its `done` event, runs the completion steps, records the stats.
static int complete_call(mdhip_ctx *ctx, mdhip_call *c)
{
    int rc = MDHIP_OK;
    hipError_t e = c->done ? wait_event(ctx, c->done) : mdhip_stream_wait(ctx);
    if (e != hipSuccess) rc = mdhip_fail(ctx, MDHIP_EHIP, "waiting for a call's device work failed: %s", hipGetErrorString(e));
    // the registers the steps work on start from what the call set while it was issued (a part of the work that
    // completed inside the call has its times there already)
    ctx->last_ms = c->stats.ms;
    ctx->last_aux_ms = c->stats.aux_ms;
    ctx->last_kernel = c->stats.kernel;
    ctx->last_launches = c->stats.launches;
    ctx->last_rel_bound = c->stats.rel_bound;
    ++ctx->completing;
    for (size_t k = 0; k < c->steps.size() && rc == MDHIP_OK; ++k) rc = c->steps[k]();
    --ctx->completing;
    CallStats st;
    st.ticket = c->stats.ticket;
    st.ms = ctx->last_ms;
    st.aux_ms = ctx->last_aux_ms;
    st.launches = ctx->last_launches;
    st.kernel = ctx->last_kernel;
    st.rel_bound = ctx->last_rel_bound;
    st.rc = rc;
    if (rc != MDHIP_OK) st.err = ctx->err;
    if (!c->inner) {
        st.fallbacks = ctx->cur_fallbacks;
        ctx->cur_fallbacks = 0;
    }
    if (c->inner) {
        // a helper call (a copy of finished values, say) reports nothing of its own: the call around it keeps its numbers;
        // a re-run of the work replaces them
        if (st.ms == 0.0 && st.aux_ms == 0.0) {
            ctx->last_ms = c->outer.ms;
            ctx->last_aux_ms = c->outer.aux_ms;
            ctx->last_launches = c->outer.launches;
            ctx->last_kernel = c->outer.kernel;
        }
    } else {
        ctx->history.push_front(st);
        if (ctx->history.size() > 64) ctx->history.pop_back();
    }
    release_call(ctx, c);
    return rc;
}

int mdhip_complete_inflight(mdhip_ctx *ctx, size_t keep)
{
    int first = MDHIP_OK;
    while (ctx->inflight.size() > keep) {
        mdhip_call *c = ctx->inflight.front();
        ctx->inflight.pop_front();
        const int rc = complete_call(ctx, c);
        if (rc != MDHIP_OK && first == MDHIP_OK) {
            first = rc;
            if (ctx->deferred_rc == MDHIP_OK) {
                ctx->deferred_err = ctx->err;
                ctx->deferred_ticket = ctx->history.empty() ? 0 : ctx->history.front().ticket;
            }
        }
    }
    return first;
}

int mdhip_call_end(mdhip_ctx *ctx, mdhip_call *c)
{
    c->ended = true;
    ctx->cur = c->parent;
    c->stats.ms = ctx->last_ms;
    c->stats.aux_ms = ctx->last_aux_ms;
    c->stats.kernel = ctx->last_kernel;
    c->stats.launches = ctx->last_launches;
    c->stats.rel_bound = ctx->last_rel_bound;
    c->done = take_done_event(ctx);
    if (c->done && hipEventRecord(c->done, ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        ctx->done_free.push_back(c->done);
        c->done = nullptr;  // complete_call then drains the stream instead
    }
    if (c->async) {
        ctx->inflight.push_back(c);
        return MDHIP_OK;
    }
    // a synchronous call: everything issued before it completes first (stream order) — unless this call was made from
    // inside a completion step, which only answers for itself
    if (!c->parent && ctx->completing == 0) {
        const int prev = mdhip_complete_inflight(ctx, 0);
        if (prev != MDHIP_OK && ctx->deferred_rc == MDHIP_OK) ctx->deferred_rc = prev;
    }
    return complete_call(ctx, c);
}

void mdhip_call_abandon(mdhip_ctx *ctx, mdhip_call *c)
{
    // queued copies may still read the call's staging blocks (or, on the copy stream, the caller's arrays)
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    (void)hipGetLastError();
    ctx->cur = c->parent;
    release_call(ctx, c);
}

// 16 bytes per lane where both sides allow it, the tail (and unaligned copies) byte by byte
__global__ __launch_bounds__(256) void copy_small_kernel(unsigned char *__restrict__ dst, const unsigned char *__restrict__ src,
                                                         size_t n16, size_t bytes)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(src)[i];
    const size_t t = n16 * 16 + i;
    if (i < 16 && t < bytes) dst[t] = src[t];
}
__global__ __launch_bounds__(256) void copy_bytes_kernel(unsigned char *__restrict__ dst, const unsigned char *__restrict__ src,
                                                         size_t bytes)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < bytes) dst[i] = src[i];
}

int mdhip_copy_small(mdhip_ctx *ctx, void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    if (bytes == 0) return MDHIP_OK;
    if (!ctx->opt_small_copy || bytes > MD_SMALL_COPY_MAX) {
        MD_HIP(hipMemcpyAsync(dst, src, bytes, kind, ctx->stream));
        return MDHIP_OK;
    }
    const bool al = ((reinterpret_cast<unsigned long long>(dst) | reinterpret_cast<unsigned long long>(src)) & 15ull) == 0ull;
    if (al) {
        const size_t n16 = bytes / 16, n = n16 > 16 ? n16 : 16;
        hipLaunchKernelGGL(copy_small_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                           (unsigned char *)dst, (const unsigned char *)src, n16, bytes);
    } else {
        hipLaunchKernelGGL(copy_bytes_kernel, dim3((unsigned)((bytes + 255) / 256)), dim3(256), 0, ctx->stream,
                           (unsigned char *)dst, (const unsigned char *)src, bytes);
    }
    MD_HIP(hipGetLastError());
    return MDHIP_OK;
}

bool mdhip_part_streams(mdhip_ctx *ctx)
{
    if (ctx->part_state) return ctx->part_state > 0;
    ctx->part_state = -1;
    const int n = ctx->cu_count;
    if (n < 64 || n % 32 != 0) return false;  // (8 XCDs x a multiple of 4 CUs)
    std::vector<uint32_t> mask[2] = {std::vector<uint32_t>((size_t)(n + 31) / 32, 0u), std::vector<uint32_t>((size_t)(n + 31) / 32, 0u)};
    int cnt[2] = {0, 0};
    for (int i = 0; i < n; ++i) {
        const int part = ((i / 8) % 4 == 3) ? 1 : 0;  // every fourth CU of each XCD to the streaming side
        mask[part][(size_t)i / 32] |= 1u << (i % 32);
        ++cnt[part];
    }
    for (int k = 0; k < 2; ++k)
        if (hipExtStreamCreateWithCUMask(&ctx->part_stream[k], (uint32_t)mask[k].size(), mask[k].data()) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
    for (auto &e : ctx->part_ev)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
    ctx->part_cus[0] = cnt[0];
    ctx->part_cus[1] = cnt[1];
    ctx->part_state = 1;
    return true;
}

int mdhip_deliver_to_device(mdhip_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes)
{
    if (bytes == 0) return MDHIP_OK;
    if (!ctx->copy_stream) {
        MD_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        MD_HIP(hipEventCreateWithFlags(&ctx->copy_ev[0], hipEventDisableTiming));
        MD_HIP(hipEventCreateWithFlags(&ctx->copy_ev[1], hipEventDisableTiming));
        MD_HIP(hipEventCreateWithFlags(&ctx->stage_ev[0], hipEventDisableTiming));
        MD_HIP(hipEventCreateWithFlags(&ctx->stage_ev[1], hipEventDisableTiming));
    }
    // a staging block of its own for the length of the copy (this runs outside any call's issue phase)
    int best = -1;
    for (size_t k = 0; k < ctx->pin_free.size(); ++k)
        if (ctx->pin_free[k].cap >= bytes && (best < 0 || ctx->pin_free[k].cap < ctx->pin_free[(size_t)best].cap)) best = (int)k;
    PinBlock b;
    if (best >= 0) {
        b = ctx->pin_free[(size_t)best];
        ctx->pin_free.erase(ctx->pin_free.begin() + best);
    } else {
        size_t cap = std::max<size_t>(65536, (bytes + 4095) & ~size_t(4095));
        if (hipHostMalloc(&b.p, cap, hipHostMallocDefault) != hipSuccess)
            return mdhip_fail(ctx, MDHIP_ENOMEM, "hipHostMalloc(%zu) failed for a staging block", cap);
        b.cap = cap;
    }
    memcpy(b.p, src_host, bytes);
    hipError_t e = hipMemcpyAsync(dst_dev, b.p, bytes, hipMemcpyHostToDevice, ctx->copy_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->copy_stream);
    ctx->pin_free.push_back(b);
    if (e != hipSuccess) return mdhip_fail(ctx, MDHIP_EHIP, "copy of finished values to the device failed: %s", hipGetErrorString(e));
    return MDHIP_OK;
}

// ---- mdhip_h2d_any: pageable sources through a page-locked ring, copied there by a few helper threads ----
struct CopyPool {
    static constexpr int N = 4;
    struct Job {
        char *dst = nullptr;
        const char *src = nullptr;
        size_t n = 0;
    };
    std::thread th[N];
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    Job job[N];
    unsigned long long gen = 0;
    int left = 0;
    bool quit = false;
    CopyPool()
    {
        for (int k = 0; k < N; ++k) th[k] = std::thread([this, k]() { run(k); });
    }
    ~CopyPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
            ++gen;
        }
        cv_go.notify_all();
        for (auto &t : th) t.join();
    }
    void run(int k)
    {
        unsigned long long seen = 0;
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_go.wait(lk, [&]() { return gen != seen; });
                seen = gen;
                if (quit) return;
                j = job[k];
            }
            if (j.n) memcpy(j.dst, j.src, j.n);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--left == 0) cv_done.notify_one();
            }
        }
    }
    // dst[0, n) = src[0, n), split over the helpers; returns when done
    void copy(void *dst, const void *src, size_t n)
    {
        const size_t per = ((n + N - 1) / N + 63) & ~size_t(63);
        {
            std::lock_guard<std::mutex> lk(mu);
            for (int k = 0; k < N; ++k) {
                const size_t o = std::min(n, per * k), e = std::min(n, per * (k + 1));
                job[k] = {static_cast<char *>(dst) + o, static_cast<const char *>(src) + o, e - o};
            }
            left = N;
            ++gen;
        }
        cv_go.notify_all();
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&]() { return left == 0; });
    }
};

static void copy_pool_destroy(mdhip_ctx *ctx)
{
    delete ctx->copy_pool;
    ctx->copy_pool = nullptr;
    for (int k = 0; k < 2; ++k) {
        if (ctx->h2d_ring[k]) (void)hipHostFree(ctx->h2d_ring[k]);
        ctx->h2d_ring[k] = nullptr;
        for (int c = 0; c < mdhip_ctx::H2D_RING_CHUNKS; ++c) {
            if (ctx->h2d_chunk_ev[k][c]) (void)hipEventDestroy(ctx->h2d_chunk_ev[k][c]);
            ctx->h2d_chunk_ev[k][c] = nullptr;
            ctx->h2d_chunk_used[k][c] = false;
        }
    }
}

int mdhip_h2d_any(mdhip_ctx *ctx, void *dst_dev, const void *src, size_t bytes, hipStream_t stream, int slot)
{
    if (bytes == 0) return MDHIP_OK;
    bool pageable = false;
    if (ctx->opt_h2d_ring) {
        hipPointerAttribute_t at;
        const hipError_t e = hipPointerGetAttributes(&at, src);
        if (e != hipSuccess) {
            (void)hipGetLastError();  // (memory the runtime has never seen: ordinary pageable memory)
            pageable = true;
        } else {
            pageable = at.type == hipMemoryTypeUnregistered;
        }
    }
    if (!pageable) {
        MD_HIP(hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, stream));
        return MDHIP_OK;
    }
    slot &= 1;
    if (!ctx->copy_pool) ctx->copy_pool = new CopyPool();
    // A ring of a few 8 MB chunks per half, cycled through with an event per chunk: the DMA of chunk c runs while the helper
    // threads fill chunk c + 1, and a chunk is written again only when the DMA that last read it is over. The pinned memory
    // is bounded (2 x 4 x 8 MB) whatever the copy's size; two copies into the same half (xi, then xj) wait for single
    // chunks, not for each other's whole transfer.
    constexpr int NC = mdhip_ctx::H2D_RING_CHUNKS;
    const size_t chunk = (size_t)8 << 20;
    if (!ctx->h2d_ring[slot]) {
        if (hipHostMalloc(&ctx->h2d_ring[slot], chunk * NC, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            ctx->h2d_ring[slot] = nullptr;
            MD_HIP(hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, stream));  // (no page-locked memory to be had)
            return MDHIP_OK;
        }
    }
    char *ring = static_cast<char *>(ctx->h2d_ring[slot]);
    for (size_t o = 0; o < bytes; o += chunk) {
        const size_t n = std::min(chunk, bytes - o);
        const int c = ctx->h2d_chunk_next[slot];
        ctx->h2d_chunk_next[slot] = (c + 1) % NC;
        if (!ctx->h2d_chunk_ev[slot][c]) MD_HIP(hipEventCreateWithFlags(&ctx->h2d_chunk_ev[slot][c], hipEventDisableTiming));
        if (ctx->h2d_chunk_used[slot][c]) MD_HIP(hipEventSynchronize(ctx->h2d_chunk_ev[slot][c]));  // (NC - 1 chunks ago)
        ctx->copy_pool->copy(ring + (size_t)c * chunk, static_cast<const char *>(src) + o, n);
        MD_HIP(hipMemcpyAsync(static_cast<char *>(dst_dev) + o, ring + (size_t)c * chunk, n, hipMemcpyHostToDevice, stream));
        MD_HIP(hipEventRecord(ctx->h2d_chunk_ev[slot][c], stream));
        ctx->h2d_chunk_used[slot][c] = true;
    }
    return MDHIP_OK;
}

extern "C" {

int mdhip_version(void) { return MDHIP_VERSION; }

#ifndef MDHIP_BUILD_ID
#define MDHIP_BUILD_ID "unknown"
#endif
const char *mdhip_build_id(void) { return MDHIP_BUILD_ID; }

int mdhip_create(mdhip_ctx **out, int device)
{
    if (!out) return mdhip_fail(nullptr, MDHIP_EINVAL, "mdhip_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return mdhip_fail(nullptr, MDHIP_ENODEV,
                          "mdhip_create: no HIP device (%s); libmdhip.so has no CPU fallback",
                          e == hipSuccess ? "count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= n)
        return mdhip_fail(nullptr, MDHIP_EINVAL, "mdhip_create: device %d out of range [0,%d)",
                          device, n);
    mdhip_ctx *ctx = new mdhip_ctx();
    ctx->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess) {
        delete ctx;
        return mdhip_fail(nullptr, MDHIP_EHIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        ctx->cu_count = prop.multiProcessorCount;
        snprintf(ctx->name, sizeof ctx->name, "%s (%s)", prop.name, prop.gcnArchName);
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            std::string nm = ctx->name;
            delete ctx;
            return mdhip_fail(nullptr, MDHIP_ENODEV,
                              "mdhip_create: device %d is %s; this library is built for gfx950 only",
                              device, nm.c_str());
        }
        ctx->lds_max = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor
                                                              : 65536;
    }
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return mdhip_fail(nullptr, MDHIP_EHIP, "mdhip_create: stream creation failed");
    }
    ctx->stream = ctx->own_stream;
    if (const char *v = getenv("MDHIP_RDF_VARIANT")) ctx->opt_rdf_variant = atoi(v);  // A/B knobs
    if (const char *v = getenv("MDHIP_RDF_JSPLIT")) ctx->opt_rdf_jsplit = atoi(v);
    if (const char *v = getenv("MDHIP_RDF_FPB")) ctx->opt_rdf_fpb = atoi(v);
    if (const char *v = getenv("MDHIP_RDF_CULL")) ctx->opt_rdf_cull = atoi(v);
    if (const char *v = getenv("MDHIP_RDF_SJ")) ctx->opt_rdf_sj = atoi(v);
    *out = ctx;
    return MDHIP_OK;
}

void mdhip_destroy(mdhip_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)mdhip_complete_inflight(ctx, 0);  // (results of calls nobody waited for still reach their destinations)
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &b : ctx->ws)
        if (b.p) (void)hipFree(b.p);
    for (auto &b : ctx->pin_free)
        if (b.p) (void)hipHostFree(b.p);
    for (hipEvent_t e : ctx->ev_free) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->done_free) (void)hipEventDestroy(e);
    if (ctx->copy_stream) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamDestroy(ctx->copy_stream);
    }
    for (auto &st : ctx->part_stream)
        if (st) {
            (void)hipStreamSynchronize(st);
            (void)hipStreamDestroy(st);
        }
    for (auto &e : ctx->part_ev)
        if (e) (void)hipEventDestroy(e);
    copy_pool_destroy(ctx);
    for (auto &e : ctx->copy_ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : ctx->stage_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char *mdhip_last_error(mdhip_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

// first error of the asynchronous calls completed so far that nobody has been told about, then `rc`
static int take_deferred(mdhip_ctx *ctx, int rc)
{
    if (ctx->deferred_rc != MDHIP_OK) {
        const int d = ctx->deferred_rc;
        ctx->deferred_rc = MDHIP_OK;
        ctx->deferred_ticket = 0;
        if (!ctx->deferred_err.empty()) ctx->err = ctx->deferred_err;
        ctx->deferred_err.clear();
        return d;
    }
    if (rc != MDHIP_OK && !ctx->deferred_err.empty()) {
        ctx->err = ctx->deferred_err;
        ctx->deferred_err.clear();
        ctx->deferred_ticket = 0;
    }
    return rc;
}

int mdhip_set_stream(mdhip_ctx *ctx, void *hip_stream)
{
    if (!ctx) return MDHIP_EINVAL;
    MD_HIP(hipSetDevice(ctx->device));
    const int rc = mdhip_complete_inflight(ctx, 0);
    MD_HIP(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return take_deferred(ctx, rc);
}

int mdhip_sync(mdhip_ctx *ctx)
{
    if (!ctx) return MDHIP_EINVAL;
    MD_HIP(hipSetDevice(ctx->device));
    const int rc = mdhip_complete_inflight(ctx, 0);
    const hipError_t e = mdhip_stream_wait(ctx);
    if (e != hipSuccess && rc == MDHIP_OK && ctx->deferred_rc == MDHIP_OK)
        return mdhip_fail(ctx, MDHIP_EHIP, "mdhip_sync: %s", hipGetErrorString(e));
    return take_deferred(ctx, rc);
}

int mdhip_wait(mdhip_ctx *ctx, int keep_in_flight)
{
    if (!ctx) return MDHIP_EINVAL;
    MD_HIP(hipSetDevice(ctx->device));
    return take_deferred(ctx, mdhip_complete_inflight(ctx, keep_in_flight > 0 ? (size_t)keep_in_flight : 0));
}

int mdhip_pending(mdhip_ctx *ctx) { return ctx ? (int)ctx->inflight.size() : 0; }

double mdhip_last_kernel_ms(mdhip_ctx *ctx, int *n_launches)
{
    if (!ctx) return 0.0;
    if (n_launches) *n_launches = ctx->last_launches;
    return ctx->last_ms;
}

long long mdhip_last_ticket(mdhip_ctx *ctx) { return ctx ? ctx->tickets : 0; }

int mdhip_ticket_status(mdhip_ctx *ctx, long long ticket, int *n_fallbacks)
{
    if (!ctx) return MDHIP_EINVAL;
    if (n_fallbacks) *n_fallbacks = 0;
    for (size_t k = 0; k < ctx->history.size(); ++k) {
        const CallStats &st = ctx->history[k];
        if (st.ticket != ticket || ticket <= 0) continue;
        if (n_fallbacks) *n_fallbacks = st.fallbacks;
        if (st.rc != MDHIP_OK) {
            ctx->err = st.err;
            if (ctx->deferred_ticket == ticket) {  // its owner has been told: not parked for a later mdhip_sync any more
                ctx->deferred_rc = MDHIP_OK;
                ctx->deferred_err.clear();
                ctx->deferred_ticket = 0;
            }
        }
        return st.rc;
    }
    for (const mdhip_call *c : ctx->inflight)
        if (c->stats.ticket == ticket) return MDHIP_EPENDING;
    // (a code of its own, and ctx->err untouched: MDHIP_EINVAL is a legitimate completion status of a call — a
    // requirement failing inside a deferred re-run — and the error text belongs to whoever checks the wait's code next)
    return MDHIP_EUNKNOWN;
}

long long mdhip_fallbacks(mdhip_ctx *ctx) { return ctx ? ctx->fallbacks_total : 0; }

int mdhip_ticket_stats(mdhip_ctx *ctx, long long ticket, double *kernel_ms, double *aux_ms, int *n_launches,
                       const char **kernel)
{
    if (!ctx) return MDHIP_EINVAL;
    for (size_t k = 0; k < ctx->history.size(); ++k)
        if (ctx->history[k].ticket == ticket && ticket > 0) return mdhip_call_stats(ctx, (int)k, kernel_ms, aux_ms, n_launches, kernel);
    return mdhip_fail(ctx, MDHIP_EINVAL, "mdhip_ticket_stats: call %lld has not completed, or more than 64 calls ago", ticket);
}

int mdhip_call_stats(mdhip_ctx *ctx, int back, double *kernel_ms, double *aux_ms, int *n_launches, const char **kernel)
{
    if (!ctx) return MDHIP_EINVAL;
    if (back < 0 || (size_t)back >= ctx->history.size())
        return mdhip_fail(ctx, MDHIP_EINVAL, "mdhip_call_stats: only %zu completed calls are remembered", ctx->history.size());
    const CallStats &st = ctx->history[(size_t)back];
    if (kernel_ms) *kernel_ms = st.ms;
    if (aux_ms) *aux_ms = st.aux_ms;
    if (n_launches) *n_launches = st.launches;
    if (kernel) *kernel = st.kernel;
    return MDHIP_OK;
}

double mdhip_last_aux_ms(mdhip_ctx *ctx) { return ctx ? ctx->last_aux_ms : 0.0; }
const char *mdhip_last_kernel_name(mdhip_ctx *ctx) { return ctx ? ctx->last_kernel : ""; }
double mdhip_last_rel_bound(mdhip_ctx *ctx) { return ctx ? ctx->last_rel_bound : 0.0; }

int mdhip_host_alloc_on(int device, size_t bytes, void **out)
{
    if (!out) return MDHIP_EINVAL;
    *out = nullptr;
    if (bytes == 0) return MDHIP_OK;
    // the calling thread may be a reader thread that never touched HIP: its current device would be 0 whatever GPU
    // the process (one rank per GPU) works on, and the allocation would open a context there
    if (device >= 0 && hipSetDevice(device) != hipSuccess) {
        (void)hipGetLastError();
        return MDHIP_ENODEV;
    }
    void *p = nullptr;
    // non-coherent = ordinary cached host memory that is page-locked: the reader threads scatter 8-byte values into
    // it (fine-grained coherent host memory is uncached for the CPU on this platform: that scatter ran 4x slower), and
    // the only consumer is an explicit hipMemcpyAsync, which needs no CPU/GPU coherence. Portable: page-locked for
    // every device of the process, not only for the one that was current at the time.
    const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocNonCoherent | hipHostMallocPortable);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return e == hipErrorOutOfMemory ? MDHIP_ENOMEM : MDHIP_ENODEV;
    }
    *out = p;
    return MDHIP_OK;
}

int mdhip_host_alloc(size_t bytes, void **out) { return mdhip_host_alloc_on(-1, bytes, out); }

void mdhip_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

int mdhip_device_name(mdhip_ctx *ctx, char *buf, int buflen)
{
    if (!ctx || !buf || buflen <= 0) return MDHIP_EINVAL;
    snprintf(buf, (size_t)buflen, "%s", ctx->name);
    return MDHIP_OK;
}

int mdhip_set_option(mdhip_ctx *ctx, const char *key, int value)
{
    if (!ctx || !key) return MDHIP_EINVAL;
    if (!strcmp(key, "rdf_variant"))
        ctx->opt_rdf_variant = value;
    else if (!strcmp(key, "rdf_sj"))
        ctx->opt_rdf_sj = value;
    else if (!strcmp(key, "rdf_batch"))
        ctx->opt_rdf_batch = value;
    else if (!strcmp(key, "rdf_cull"))
        ctx->opt_rdf_cull = value;
    else if (!strcmp(key, "rdf_fpb"))
        ctx->opt_rdf_fpb = value;
    else if (!strcmp(key, "rdf_jsplit"))
        ctx->opt_rdf_jsplit = value;
    else if (!strcmp(key, "rdf_inflight"))
        ctx->opt_rdf_inflight = value < 1 ? 1 : value;
    else if (!strcmp(key, "rdf_rows"))
        ctx->opt_rdf_rows = value;
    else if (!strcmp(key, "rdf_big"))
        ctx->opt_rdf_big = value;
    else if (!strcmp(key, "rdf_pk_passes"))
        ctx->opt_rdf_pk_passes = value < 0 ? 1 : value;
    else if (!strcmp(key, "residence_cap"))
        ctx->opt_residence_cap = value;
    else if (!strcmp(key, "rdf_disp"))
        ctx->opt_rdf_disp = value;
    else if (!strcmp(key, "rdf_sort"))
        ctx->opt_rdf_sort = value;
    else if (!strcmp(key, "rdf_pk"))
        ctx->opt_rdf_pk = value;
    else if (!strcmp(key, "lag_batched_fuse"))
        ctx->opt_lag_batched_fuse = value < 0 ? 2 : value;
    else if (!strcmp(key, "lag_residue"))
        ctx->opt_lag_residue = value < 0 ? 1 : value;
    else if (!strcmp(key, "lag_mean_sample"))
        ctx->opt_lag_mean_sample = value;
    else if (!strcmp(key, "lag_w1"))
        ctx->opt_lag_w1 = value < 0 ? 1 : value;
    else if (!strcmp(key, "lag_overlap"))
        ctx->opt_lag_overlap = value < 0 ? 0 : value;
    else if (!strcmp(key, "lag_ends"))
        ctx->opt_lag_ends = value < 0 ? 1 : value;
    else if (!strcmp(key, "lag_batch_mb"))
        ctx->opt_lag_batch_mb = value <= 0 ? 4096 : std::min(value, 65536);
    else if (!strcmp(key, "lag_w12_min_f"))
        ctx->opt_lag_w12_min_f = value < 0 ? 1536 : value;  // -1 restores the default
    else if (!strcmp(key, "lag_variant"))
        ctx->opt_lag_variant = value < 0 ? 3 : value;  // -1 restores the default
    else if (!strcmp(key, "seg_cap"))
        ctx->opt_seg_cap = value;
    else if (!strcmp(key, "seg_vec"))
        ctx->opt_seg_vec = value;
    else if (!strcmp(key, "fft_logr"))
        ctx->opt_fft_logr = value;
    else if (!strcmp(key, "fft_logc"))
        ctx->opt_fft_logc = value;
    else if (!strcmp(key, "seg_gy"))
        ctx->opt_seg_gy = value;
    else if (!strcmp(key, "fft_specfuse"))
        ctx->opt_fft_specfuse = value;
    else if (!strcmp(key, "fft_net8"))
        ctx->opt_fft_net8 = value;
    else if (!strcmp(key, "seg_frame"))
        ctx->opt_seg_frame = value;
    else if (!strcmp(key, "cn_pk"))
        ctx->opt_cn_pk = value;
    else if (!strcmp(key, "rdf_guard"))
        ctx->opt_rdf_guard = value < 0 ? 0 : value;
    else if (!strcmp(key, "rdf_slots"))
        ctx->opt_rdf_slots = value < 1 ? 1 : value;
    else if (!strcmp(key, "xcorr_tile"))
        ctx->opt_xcorr_tile = value;
    else if (!strcmp(key, "rdf_relblock"))
        ctx->opt_rdf_relblock = value;
    else if (!strcmp(key, "lag_fft_kernel"))
        ctx->opt_lag_fft_kernel = value;
    else if (!strcmp(key, "h2d_overlap"))
        ctx->opt_h2d_overlap = value;
    else if (!strcmp(key, "fft_mid"))
        ctx->opt_fft_mid = value;
    else if (!strcmp(key, "small_copy"))
        ctx->opt_small_copy = value;
    else if (!strcmp(key, "h2d_ring"))
        ctx->opt_h2d_ring = value;
    else if (!strcmp(key, "lag_direct"))
        ctx->opt_lag_direct = value;
    else if (!strcmp(key, "sync_spin"))
        ctx->opt_sync_spin = value;
    else
        return mdhip_fail(ctx, MDHIP_EINVAL, "mdhip_set_option: unknown key '%s'", key);
    return MDHIP_OK;
}

// The reference's rule, evaluated on the host with IEEE sqrt and divide (structural/rdf_cn.py:68,85).
static inline int64_t ref_bin(double rsq, double ddr) { return (int64_t)(std::sqrt(rsq) / ddr); }

int mdhip_bin_edges(double bin_size, int nbins, double *edges)
{
    if (!(bin_size > 0.0) || nbins < 1 || !edges) return MDHIP_EINVAL;
    edges[0] = 0.0;
    for (int k = 1; k <= nbins; ++k) {
        // Smallest double with ref_bin >= k: bisection on the (monotone) bit pattern of positive doubles.
        double guess = ((double)k * bin_size) * ((double)k * bin_size);
        uint64_t lo, hi;
        double g_lo = guess * 0.999999, g_hi = guess * 1.000001;
        memcpy(&lo, &g_lo, 8);
        memcpy(&hi, &g_hi, 8);
        double d;
        // widen until lo is below the edge and hi is at/above it
        for (;;) {
            memcpy(&d, &lo, 8);
            if (ref_bin(d, bin_size) < k) break;
            lo -= (lo >> 20) + 1;
        }
        for (;;) {
            memcpy(&d, &hi, 8);
            if (ref_bin(d, bin_size) >= k) break;
            hi += (hi >> 20) + 1;
        }
        while (hi - lo > 1) {
            uint64_t mid = lo + ((hi - lo) >> 1);
            memcpy(&d, &mid, 8);
            if (ref_bin(d, bin_size) >= k)
                hi = mid;
            else
                lo = mid;
        }
        memcpy(&edges[k], &hi, 8);
    }
    return MDHIP_OK;
}

}  // extern "C"
