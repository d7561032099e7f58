// tools/ubench_hbm.hip — what a read stream of the shapes this library's HBM-bound kernels have reaches on the box
// (segment_frame_kernel 0.64-0.68, msd_windows_kernel 0.69, msd_pairs_kernel 0.73-0.74 of the 8 TB/s
// the roofline is priced against — is the distance to 1.0 the kernels' or the memory system's?):
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/ubench_hbm tools/ubench_hbm.hip && tools/_bin/ubench_hbm [json-path]
//
// Every kernel streams a buffer of `total` bytes ONCE with 16-byte non-temporal loads, 256 lanes per block, each block
// one contiguous piece (or three pieces a plane apart, the [frame][xyz][atom] layout), every lane's loads issued before
// the first use, and folds what it read into one double per block (so nothing is removed). Variants add what the real
// kernels add: a second, L2-resident read of a third of the bytes (the masses of segment_frame_kernel: 8 B per atom
// beside 24 B of coordinates) or of as many bytes again (the origin frame of msd_pairs_kernel), a write stream of a tenth of the bytes (its per-molecule results), a block barrier
// between the loads and the use (its LDS stage). Rates are bytes of the streamed buffer (+ bytes written) per second of
// HIP-event time, best of 5 launches; `frac` is against 8 TB/s. The buffer holds pseudo-random doubles (zeros read 1-3 %
// faster).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                      \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

// PER 16-byte loads per lane and plane; PLANES pieces of PER * 256 * 16 bytes, `plane_stride` doubles apart (0: the
// block's bytes are one run); MASS: one more 16-byte load per PER from an 8 KB-per-block table that stays in L2;
// WRITE: 1/10 of the bytes read are written back; BARRIER: __syncthreads between loads and use (through LDS).
template <int PER, int PLANES, int MASS, bool WRITE, bool BARRIER, bool NTW = false>
__global__ __launch_bounds__(256) void stream_kernel(const double *__restrict__ x, long long plane_stride,
                                                     long long block_stride, const double *__restrict__ mass,
                                                     int mass_blocks, double *__restrict__ sink, double *__restrict__ wout)
{
    __shared__ double s[BARRIER ? PLANES * PER * 512 : 1];
    const int tid = threadIdx.x;
    const double *p = x + (size_t)blockIdx.x * block_stride;
    d2 v[PLANES][PER], m[MASS ? MASS : 1][PER];
#pragma unroll
    for (int r = 0; r < PER; ++r) {
#pragma unroll
        for (int k = 0; k < PLANES; ++k)
            v[k][r] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(p + (size_t)k * plane_stride + 2 * tid + r * 512));
#pragma unroll
        for (int k = 0; k < MASS; ++k)
            m[k][r] = *reinterpret_cast<const d2 *>(mass + ((size_t)(blockIdx.x % mass_blocks) * MASS + k) * (PER * 512) + 2 * tid + r * 512);
    }
    double acc = 0.0;
    if (BARRIER) {
#pragma unroll
        for (int r = 0; r < PER; ++r)
#pragma unroll
            for (int k = 0; k < PLANES; ++k) {
                const d2 t = MASS ? v[k][r] * m[MASS > 1 ? k % (MASS ? MASS : 1) : 0][r] : v[k][r];
                s[(k * PER + r) * 512 + 2 * tid] = t[0];
                s[(k * PER + r) * 512 + 2 * tid + 1] = t[1];
            }
        __syncthreads();
        // (a lane sums a short run, as one lane per molecule does)
        for (int i = 0; i < PLANES * PER * 2; ++i) acc += s[(tid * PLANES * PER * 2 + i) % (PLANES * PER * 512)];
    } else {
#pragma unroll
        for (int r = 0; r < PER; ++r)
#pragma unroll
            for (int k = 0; k < PLANES; ++k) {
                const d2 t = MASS ? v[k][r] * m[MASS > 1 ? k % (MASS ? MASS : 1) : 0][r] : v[k][r];
                acc += t[0] + t[1];
            }
    }
    if (WRITE) {
        // a tenth of the bytes read: PLANES * PER * 256 * 16 / 10 bytes per block
        constexpr int NW = PLANES * PER * 512 / 10;
        for (int i = tid; i < NW; i += 256) {
            if (NTW) __builtin_nontemporal_store(acc, wout + (size_t)blockIdx.x * NW + i);
            else wout[(size_t)blockIdx.x * NW + i] = acc;
        }
    }
    if (acc == 1.2345e300) sink[blockIdx.x] = acc;  // (never true: the loads stay)
}

// pseudo-random doubles in [1, 2) (a buffer of zeros reads faster than data does: less switching on the bus)
__global__ void fill_kernel(double *x, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        unsigned long long h = (unsigned long long)i * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        h *= 0xBF58476D1CE4E5B9ull;
        h ^= h >> 32;
        x[i] = __longlong_as_double((long long)(0x3FF0000000000000ull | (h >> 12)));
    }
}

struct Result {
    std::string name;
    double gbs, frac, ms;
};

template <int PER, int PLANES, int MASS, bool WRITE, bool BARRIER, bool NTW = false>
Result run(const char *name, const double *d_x, long long n_doubles, bool planar, const double *d_mass, double *d_sink,
           double *d_w, hipStream_t st)
{
    const long long piece = (long long)PER * 512;         // doubles per block and plane
    const long long per_block = piece * PLANES;
    const long long blocks = n_doubles / per_block;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipEventRecord(e0, st));
        if (planar) {
            // the buffer read as [PLANES][all atoms]: a block's pieces are a third of the buffer apart — three streams a
            // large stride apart, the access shape of one frame's x, y, z planes
            hipLaunchKernelGGL((stream_kernel<PER, PLANES, MASS, WRITE, BARRIER, NTW>), dim3((unsigned)blocks), dim3(256), 0, st, d_x,
                               n_doubles / PLANES, piece, d_mass, 50, d_sink, d_w);
        } else {
            hipLaunchKernelGGL((stream_kernel<PER, PLANES, MASS, WRITE, BARRIER, NTW>), dim3((unsigned)blocks), dim3(256), 0, st, d_x,
                               piece, per_block, d_mass, 50, d_sink, d_w);
        }
        CHECK(hipEventRecord(e1, st));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)blocks * per_block * 8 * (WRITE ? 1.1 : 1.0);
    Result r{name, bytes / (best * 1e-3) * 1e-9, bytes / (best * 1e-3) / 8e12, best};
    printf("%-92s %8.1f GB/s  %.3f of 8 TB/s  %.3f ms\n", name, r.gbs, r.frac, r.ms);
    fflush(stdout);
    return r;
}

int main(int argc, char **argv)
{
    const long long n = 750000000LL / 1536 * 1536;  // 6 GB of doubles (the C4 trajectory: 5000 x 3 x 50000)
    double *d_x, *d_mass, *d_sink, *d_w;
    CHECK(hipMalloc(&d_x, (size_t)n * 8));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, d_x, n);
    CHECK(hipMalloc(&d_mass, 50 * 3 * 4 * 512 * 8));
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, d_mass, 50LL * 3 * 4 * 512);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMalloc(&d_sink, 8 << 20));
    CHECK(hipMalloc(&d_w, (size_t)n * 8 / 10 + 4096));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    std::vector<Result> rs;
    // one run per block: 4 KB, 8 KB, 16 KB, 32 KB, 48 KB in flight per block
    rs.push_back(run<1, 1, 0, false, false>("flat  4 KB per block", d_x, n, false, d_mass, d_sink, d_w, st));
    rs.push_back(run<2, 1, 0, false, false>("flat  8 KB per block", d_x, n, false, d_mass, d_sink, d_w, st));
    rs.push_back(run<4, 1, 0, false, false>("flat 16 KB per block", d_x, n, false, d_mass, d_sink, d_w, st));
    rs.push_back(run<2, 3, 0, false, false>("flat 24 KB per block (3 x 8 KB adjacent)", d_x, n, false, d_mass, d_sink, d_w, st));
    rs.push_back(run<4, 3, 0, false, false>("flat 48 KB per block (3 x 16 KB adjacent)", d_x, n, false, d_mass, d_sink, d_w, st));
    // three planes far apart (the [xyz][atom] layout): 3 x 8 KB = segment_frame_kernel's 1024 atoms
    rs.push_back(run<2, 3, 0, false, false>("planes 3 x 8 KB a third of the buffer apart", d_x, n, true, d_mass, d_sink, d_w, st));
    rs.push_back(run<2, 3, 1, false, false>("  + 8 KB per block from an L2-resident table (masses)", d_x, n, true, d_mass, d_sink, d_w, st));
    rs.push_back(run<2, 3, 1, true, false>("  + masses + a tenth of the bytes written", d_x, n, true, d_mass, d_sink, d_w, st));
    rs.push_back(run<2, 3, 1, true, true>("  + masses + writes + LDS stage and barrier", d_x, n, true, d_mass, d_sink, d_w, st));
    rs.push_back(run<2, 3, 0, true, false>("planes 3 x 8 KB + a tenth written (no masses)", d_x, n, true, d_mass, d_sink, d_w, st));
    rs.push_back(run<4, 3, 1, true, true>("planes 3 x 16 KB + masses + writes + LDS stage", d_x, n, true, d_mass, d_sink, d_w, st));
    rs.push_back(run<2, 3, 3, false, false>("planes 3 x 8 KB + 24 KB per block from an L2-resident 1.2 MB table (msd_pairs' origin)", d_x, n, true, d_mass, d_sink, d_w, st));
    rs.push_back(run<2, 3, 1, true, false, true>("planes 3 x 8 KB + masses + a tenth written non-temporal", d_x, n, true, d_mass, d_sink, d_w, st));
    rs.push_back(run<2, 1, 0, true, false>("flat 8 KB per block + a tenth written", d_x, n, false, d_mass, d_sink, d_w, st));
    if (argc > 1) {
        FILE *f = fopen(argv[1], "w");
        if (f) {
            fprintf(f, "{\"peak_gbs\": 8000.0, \"buffer_bytes\": %lld, \"cases\": [", n * 8);
            for (size_t i = 0; i < rs.size(); ++i)
                fprintf(f, "%s{\"name\": \"%s\", \"gbs\": %.1f, \"frac\": %.4f, \"ms\": %.4f}", i ? ", " : "", rs[i].name.c_str(),
                        rs[i].gbs, rs[i].frac, rs[i].ms);
            fprintf(f, "]}\n");
            fclose(f);
        }
    }
    return 0;
}
