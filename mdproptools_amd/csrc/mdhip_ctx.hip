// mdhip_ctx.hip — lifecycle, workspace, options and the host-side bin-edge table.
#include <cmath>
#include <cstdlib>

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

void *mdhip_pin(mdhip_ctx *ctx, int slot, size_t bytes)
{
    DevBuf &b = ctx->pin[slot];
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return b.p;
    if (b.p) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipHostFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    size_t cap = bytes + (bytes >> 2);
    cap = (cap + 4095) & ~size_t(4095);
    hipError_t e = hipHostMalloc(&b.p, cap, hipHostMallocDefault);
    if (e != hipSuccess) {
        b.p = nullptr;
        mdhip_fail(ctx, MDHIP_ENOMEM, "hipHostMalloc(%zu) failed for staging buffer %d: %s", cap, slot,
                   hipGetErrorString(e));
        return nullptr;
    }
    b.cap = cap;
    return b.p;
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
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
        hipEventCreate(&ctx->ev2) != hipSuccess || hipEventCreate(&ctx->ev3) != hipSuccess) {
        delete ctx;
        return mdhip_fail(nullptr, MDHIP_EHIP, "mdhip_create: stream/event creation failed");
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
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &b : ctx->ws)
        if (b.p) (void)hipFree(b.p);
    for (auto &b : ctx->pin)
        if (b.p) (void)hipHostFree(b.p);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->copy_stream) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamDestroy(ctx->copy_stream);
    }
    for (auto &e : ctx->copy_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->ev2) (void)hipEventDestroy(ctx->ev2);
    if (ctx->ev3) (void)hipEventDestroy(ctx->ev3);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char *mdhip_last_error(mdhip_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int mdhip_set_stream(mdhip_ctx *ctx, void *hip_stream)
{
    if (!ctx) return MDHIP_EINVAL;
    MD_HIP(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return MDHIP_OK;
}

int mdhip_sync(mdhip_ctx *ctx)
{
    if (!ctx) return MDHIP_EINVAL;
    MD_HIP(hipStreamSynchronize(ctx->stream));
    return MDHIP_OK;
}

double mdhip_last_kernel_ms(mdhip_ctx *ctx, int *n_launches)
{
    if (!ctx) return 0.0;
    if (n_launches) *n_launches = ctx->last_launches;
    return ctx->last_ms;
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
    else if (!strcmp(key, "rdf_sort"))
        ctx->opt_rdf_sort = value;
    else if (!strcmp(key, "rdf_pk"))
        ctx->opt_rdf_pk = value;
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
