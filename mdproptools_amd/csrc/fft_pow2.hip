// fft_pow2.hip — batched power-of-two FP64 transforms for the correlation paths (G2/G3 of SURVEY §8a:
// dynamical/conductivity.py:97-114 `correlate`, dynamical/viscosity.py:103-115 `autocorrelate`), written for gfx950.
//
// Why not the library FFT: rocFFT compiles its kernels at run time for every new length — 1.4-1.8 s on the first
// transform of a process, three orders of magnitude more than the transform itself — and a correlation call is
// usually made ONCE per process. These kernels are compiled with the library; the first call costs what every call
// costs.
//
// Layout of the work (L real samples, H = L/2 complex points, both powers of two):
//   real -> spectrum   the L reals ARE H complex points z[j] = x[2j] + i x[2j+1] (no repacking); Z = FFT_H(z);
//                      X(k) = E(k) + e^{-2 pi i k/L} O(k), E/O from Z(k) and conj Z(H-k)       (r2c_post_kernel)
//   spectrum -> real   Y(k) = [S(k) + conj S(H-k)] + i e^{+2 pi i k/L} [S(k) - conj S(H-k)];     (c2r_pre_kernel)
//                      y = IFFT_H(Y) = conj FFT_H(conj Y); c[2j] = Re y[j], c[2j+1] = Im y[j]  (unnormalised)
//   FFT_H              Stockham autosort, decimation in frequency, passes of radix R = 2^1..2^8 (option fft_logr):
//                      pass (n, s) with n*s = H reads column c of the [R][H/R] view of its input, does the R-point
//                      DFT, multiplies output j by e^{-2 pi i j p/n} (p = c div s) and stores it at
//                      (c mod s) + s*(R*p + j); then n /= R, s *= R. Natural order in, natural order out.
//
// fft_pass_kernel: a workgroup of 256 lanes owns a tile of C = 16 neighbouring columns (256-byte runs in HBM for the
// loads of every pass and for the stores of every pass but the first, whose stores are runs of R points): the
// R x C tile sits in LDS (64 KB at R = 256), the DFT is an in-place Gentleman-Sande network over the rows, radix-4
// butterflies in registers = two radix-2 stages per LDS round trip
// (lanes walk the columns: consecutive 16-byte LDS words, no bank conflicts) with the R/2 roots of unity in an LDS
// table made once per workgroup by sincospi of an EXACT argument (k/R, dyadic), and the bit-reversed row order is
// undone by the store loop. The per-pass twiddle e^{-2 pi i jp/n} is again sincospi of an exact dyadic argument
// (jp < n <= 2^29 fits an integer, n is a power of two), so twiddle error is the ~1 ulp of ocml's sincospi and does
// not grow with the length. HBM-bound: 32 B per point per pass.
#include <algorithm>

#include "ctx.h"

namespace {

#ifndef FFT_NT_STORE
#define FFT_NT_STORE 0  // (A/B build) 1: the passes' complex outputs leave with non-temporal stores
#endif
__device__ __forceinline__ void fft_store(double2 *p, double2 v)
{
#if FFT_NT_STORE
    typedef double d2_t __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(d2_t{v.x, v.y}, reinterpret_cast<d2_t *>(p));
#else
    *p = v;
#endif
}
constexpr int FFT_THREADS = 256;
constexpr int FFT_MAX_LOGR = 10;   // largest radix of a pass the `fft_logr` knob may ask for (default 8: R <= 256, 64 KB tiles)
constexpr int FFT_MAX_PASSES = 8;  // PassPlan::logR slots: H <= 2^32 at the smallest allowed radix, 2^4
constexpr int FFT_LOGC = 4;      // C = 16 columns per tile
constexpr int FFT_MAX_BATCH = 32768;  // series per launch (grid.y)

__device__ __forceinline__ double2 cmul(double2 a, double2 b)
{
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// e^{-2 pi i num/den}, den a power of two, 0 <= num: the argument 2*num/den is exact in double
__device__ __forceinline__ double2 root_of_unity(long long num, long long den)
{
    double s, c;
    sincospi(-2.0 * (double)num / (double)den, &s, &c);
    return make_double2(c, s);
}

// Roots of unity from a two-level table instead of sincospi per point (round 3: the per-output twiddle of a pass was
// ~50 of the ~150 FP64 instructions a point costs, and the passes were as VALU-busy (58 %) as HBM-busy (60 %)):
//   e^{-2 pi i m / 2^logn} = A[idx >> 10] * B[idx & 1023],  idx = m << (logL - logn)  (a root of order n is a root of order L)
//   A[k] = e^{-2 pi i k 2^10 / L}, k < L / 2^10 (one entry for L <= 2^10);  B[k] = e^{-2 pi i k / L}, k < min(L, 2^10)
// Every entry is sincospi of an exact dyadic argument (~1 ulp); the product adds ~2 ulp: the twiddle error still does
// not grow with the length. The table (<= 16 MB at L = 2^30, 48 KB at L = 2^21: L2-resident) is built once per length
// and context (fft_twiddle_table).
struct TwTab {
    const double2 *A, *B;
    int logL;
};

__device__ __forceinline__ double2 tw_lookup(const TwTab &t, unsigned long long m, int logn)
{
    const unsigned long long idx = m << (t.logL - logn);
    return cmul(t.A[idx >> 10], t.B[idx & 1023ull]);
}

__global__ void fft_twiddle_table_kernel(double2 *A, long long nA, double2 *B, long long nB, long long L)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nA) A[i] = root_of_unity(i << 10, L);
    if (i < nB) B[i] = root_of_unity(i, L);
}

__device__ __forceinline__ unsigned bit_reverse(unsigned r, int bits) { return bits ? __brev(r) >> (32 - bits) : 0u; }

enum { IN_PLAIN = 0, IN_PAD = 1, IN_SPEC = 2 };  // IN_SPEC (fft_pass8_kernel only): see spec_input
enum { OUT_PLAIN = 0, OUT_CONJ = 1, OUT_LAGS = 2, OUT_PERM = 3 };  // OUT_PERM (fft_pass8_kernel, first of two passes): pair_phys

// The order in which the first pass of a two-pass autocorrelation leaves the R outputs of a column (fft_mid_acf_kernel):
// j and R - j next to each other — 0, R/2, 1, R-1, 2, R-2, ... — so that the frequencies k and H - k, which the
// spectrum step combines, sit in NEIGHBOURING columns of the second pass and every tile of >= 2 columns holds whole pairs.
__host__ __device__ inline int pair_phys(int j, int R) { return j == 0 ? 0 : j == R / 2 ? 1 : j < R / 2 ? 2 * j : 2 * (R - j) + 1; }
__host__ __device__ inline int pair_logical(int x, int R) { return (x & 1) ? (x == 1 ? R / 2 : R - (x >> 1)) : (x >> 1); }

// what the first / last pass of the fused correlation pipeline reads / writes instead of a complex buffer
struct PassIo {
    const double *series;  // IN_PAD: [batch][n] real samples; the transform input is the zero-padded series read
    long long n;           //         as complex pairs (x[2e], x[2e+1])
    double *lags;          // OUT_LAGS: [batch][n_lags]; the complex result y = conj(v) holds c[2o] = Re, c[2o+1] = Im,
    long long n_lags;      //           lags[t] = (c[t] / L) / (n - t) for t < n_lags
    double L;
    double scale;          //           ... times `scale` (a unit factor of the caller: one more rounding, as a multiplication
                           //           of the finished array would be; 1.0 leaves every value as it is)
    const double2 *zb;     // IN_SPEC: the second series' transform [batch][H] (nullptr: autocorrelation, Zb = Za = `in`)
};

// grid (H/R/C tiles, batch). in/out [batch][H].
template <int IN, int OUT>
__global__ __launch_bounds__(FFT_THREADS) void fft_pass_kernel(const double2 *__restrict__ in,
                                                               double2 *__restrict__ out, long long H, int logR,
                                                               int logC, int logS, long long n, PassIo io, TwTab tt)
{
    const long long s = 1LL << logS;
    extern __shared__ double2 lds[];
    const int R = 1 << logR, C = 1 << logC;
    double2 *buf = lds;              // [R][C]
    double2 *tw = lds + (R << logC);  // [R/2]
    const long long cols = H >> logR;
    const long long c0 = (long long)blockIdx.x << logC;
    if (IN == IN_PAD) {
        const double *x = io.series + (size_t)blockIdx.y * io.n;
        for (int idx = threadIdx.x; idx < (R << logC); idx += FFT_THREADS) {
            const int k = idx >> logC, cc = idx & (C - 1);
            const long long e = 2 * (c0 + cc + (long long)k * cols);
            double2 v = make_double2(0.0, 0.0);
            if (e < io.n) {  // (rows k >= R/2 lie in the zero padding: L >= 2n)
                v.x = x[e];
                if (e + 1 < io.n) v.y = x[e + 1];
            }
            buf[idx] = v;
        }
    } else {
        in += (size_t)blockIdx.y * H;
        for (int idx = threadIdx.x; idx < (R << logC); idx += FFT_THREADS) {
            const int k = idx >> logC, cc = idx & (C - 1);
            buf[idx] = in[c0 + cc + (long long)k * cols];
        }
    }
    for (int t = threadIdx.x; t < (R >> 1); t += FFT_THREADS) tw[t] = tw_lookup(tt, (unsigned long long)t, logR);
    __syncthreads();
    // Gentleman-Sande network over the rows, two radix-2 stages per LDS round trip (a radix-4 butterfly in registers:
    // the same operations in the same order as the two stages one after the other, half the LDS traffic and barriers);
    // an odd log2 R starts with one radix-2 stage.
    int st = 0;
    if (logR & 1) {
        const int lh = logR - 1, half = 1 << lh;
        for (int b = threadIdx.x; b < (R >> 1 << logC); b += FFT_THREADS) {
            const int cc = b & (C - 1), i = b >> logC;
            double2 *pa = buf + (i << logC) + cc, *pb = pa + (half << logC);
            const double2 a = *pa, bb = *pb;
            *pa = make_double2(a.x + bb.x, a.y + bb.y);
            *pb = cmul(make_double2(a.x - bb.x, a.y - bb.y), tw[i]);
        }
        __syncthreads();
        st = 1;
    }
    for (; st < logR; st += 2) {
        const int lq = logR - 2 - st;  // log2 of the quarter-span q
        const int q = 1 << lq;
        for (int b = threadIdx.x; b < (R >> 2 << logC); b += FFT_THREADS) {
            const int cc = b & (C - 1), pr = b >> logC;
            const int i = pr & (q - 1);
            const int i0 = ((pr >> lq) << (lq + 2)) + i;
            double2 *p0 = buf + (i0 << logC) + cc, *p1 = p0 + (q << logC), *p2 = p1 + (q << logC), *p3 = p2 + (q << logC);
            const double2 x0 = *p0, x1 = *p1, x2 = *p2, x3 = *p3;
            // stage st (half-span 2q): (x0, x2) with W_4q^i, (x1, x3) with W_4q^(i+q)
            const double2 a0 = make_double2(x0.x + x2.x, x0.y + x2.y);
            const double2 a2 = cmul(make_double2(x0.x - x2.x, x0.y - x2.y), tw[i << st]);
            const double2 a1 = make_double2(x1.x + x3.x, x1.y + x3.y);
            const double2 a3 = cmul(make_double2(x1.x - x3.x, x1.y - x3.y), tw[(i + q) << st]);
            // stage st + 1 (half-span q): (a0, a1) and (a2, a3), both with W_2q^i
            const double2 w2 = tw[i << (st + 1)];
            *p0 = make_double2(a0.x + a1.x, a0.y + a1.y);
            *p1 = cmul(make_double2(a0.x - a1.x, a0.y - a1.y), w2);
            *p2 = make_double2(a2.x + a3.x, a2.y + a3.y);
            *p3 = cmul(make_double2(a2.x - a3.x, a2.y - a3.y), w2);
        }
        __syncthreads();
    }
    const bool last = n == R;  // p = 0 for every column: no twiddle
    const int logn = 63 - __builtin_clzll((unsigned long long)n);
    // store order: columns fastest when a tile's columns share their p (s >= C: runs of C points), output index
    // fastest otherwise (first pass, s = 1: the tile's outputs are one contiguous block of R*C points)
    const bool col_fast = s >= C;
    out += (size_t)blockIdx.y * H;
    double *lags = OUT == OUT_LAGS ? io.lags + (size_t)blockIdx.y * io.n_lags : nullptr;
    for (int idx = threadIdx.x; idx < (R << logC); idx += FFT_THREADS) {
        int cc, j;
        if (col_fast) {
            cc = idx & (C - 1);
            j = idx >> logC;
        } else {
            j = idx & (R - 1);
            cc = idx >> logR;
        }
        const int r = (int)bit_reverse((unsigned)j, logR);
        double2 v = buf[(r << logC) + cc];
        const long long c = c0 + cc;
        const long long p = c >> logS, q = c & (s - 1);
        if (!last && j != 0 && p != 0) v = cmul(v, tw_lookup(tt, (unsigned long long)j * (unsigned long long)p, logn));
        const long long o = q + s * (((long long)p << logR) + j);
        if (OUT == OUT_LAGS) {
            const long long t = 2 * o;
            if (t < io.n_lags) lags[t] = ((v.x / io.L) / (double)(io.n - t)) * io.scale;
            if (t + 1 < io.n_lags) lags[t + 1] = ((-v.y / io.L) / (double)(io.n - t - 1)) * io.scale;
        } else {
            if (OUT == OUT_CONJ) v.y = -v.y;
            fft_store(out + o, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 3: the same pass with a radix-8 network, for the large radices (R = 2^9 .. 2^11) that let a transform of 2^20
// points run in TWO passes instead of three (C5: 144 MB per series through HBM instead of 208). The radix-4 network above
// needs log2(R)/2 LDS round trips with a block barrier each and 256 lanes — at R = 1024 five of them over an 8192-point
// tile, which is why two passes measured slower than three. Here a lane does a whole radix-8 butterfly in
// registers per round (the butterflies of msd_fft.hip's in-LDS transform), R·C/8 lanes per workgroup, the last round is a
// 16-, 8- or 4-point transform without twiddles: 1024 = 8 x 8 x 16 in three rounds. Rows sit C points apart with C
// points of padding per final sub-transform, so that the last round — a lane walks 16 consecutive rows — spreads over
// the banks. Decimation in frequency, output row of frequency j by digit reversal (net8_row).
#ifndef NET8_MIN_LOGR
#define NET8_MIN_LOGR 9  // passes of radix >= 2^9 take the radix-8 network (fft_net8 = 0: the radix-4 one, A/B)
#endif
#ifndef NET8_MAX_LOGC
#define NET8_MAX_LOGC 5
#endif
struct Net8Plan {
    int n8;  // rounds of radix 8 (with twiddles)
    int lf;  // log2 of the last round's transform (1 .. 4), without twiddles
};

__host__ __device__ constexpr inline Net8Plan net8_plan(int logR)
{
    Net8Plan p{0, logR};
    while (p.lf > 4) {
        ++p.n8;
        p.lf -= 3;
    }
    return p;
}

__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 cmul_mi(double2 a) { return make_double2(a.y, -a.x); }  // a * (-i)

__device__ __forceinline__ void net_dft4(double2 c0, double2 c1, double2 c2, double2 c3, double2 &z0, double2 &z1,
                                         double2 &z2, double2 &z3)
{
    const double2 d0 = cadd(c0, c2), d1 = csub(c0, c2), d2 = cadd(c1, c3), d3 = cmul_mi(csub(c1, c3));
    z0 = cadd(d0, d2);
    z2 = csub(d0, d2);
    z1 = cadd(d1, d3);
    z3 = csub(d1, d3);
}

// 8-point DFT of a[0..7] (elements i + e·stride of a sub-transform), output digit d left in a[d]
__device__ __forceinline__ void net_dft8(double2 *a)
{
    constexpr double H = 0.70710678118654752440;
    const double2 b0 = cadd(a[0], a[4]), b4 = csub(a[0], a[4]);
    const double2 b1 = cadd(a[1], a[5]), t5 = csub(a[1], a[5]);
    const double2 b2 = cadd(a[2], a[6]), t6 = csub(a[2], a[6]);
    const double2 b3 = cadd(a[3], a[7]), t7 = csub(a[3], a[7]);
    const double2 b5 = make_double2((t5.x + t5.y) * H, (t5.y - t5.x) * H);
    const double2 b6 = cmul_mi(t6);
    const double2 b7 = make_double2((t7.y - t7.x) * H, -(t7.x + t7.y) * H);
    net_dft4(b0, b1, b2, b3, a[0], a[2], a[4], a[6]);
    net_dft4(b4, b5, b6, b7, a[1], a[3], a[5], a[7]);
}

// 16-point DFT, two radix-4 stages; frequency f0 + 4 f1 ends up in x[4 f0 + f1]
__device__ __forceinline__ void net_dft16(double2 *x)
{
    constexpr double C1 = 0.92387953251128675613, S1 = 0.38268343236508977173, H = 0.70710678118654752440;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        double2 y0, y1, y2, y3;
        net_dft4(x[e], x[e + 4], x[e + 8], x[e + 12], y0, y1, y2, y3);
        if (e == 1) {
            y1 = cmul(y1, make_double2(C1, -S1));
            y2 = make_double2((y2.x + y2.y) * H, (y2.y - y2.x) * H);
            y3 = cmul(y3, make_double2(S1, -C1));
        } else if (e == 2) {
            y1 = make_double2((y1.x + y1.y) * H, (y1.y - y1.x) * H);
            y2 = cmul_mi(y2);
            y3 = make_double2((y3.y - y3.x) * H, -(y3.x + y3.y) * H);
        } else if (e == 3) {
            y1 = cmul(y1, make_double2(S1, -C1));
            y2 = make_double2((y2.y - y2.x) * H, -(y2.x + y2.y) * H);
            y3 = cmul(y3, make_double2(-C1, S1));
        }
        x[e] = y0;
        x[e + 4] = y1;
        x[e + 8] = y2;
        x[e + 12] = y3;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        double2 y0, y1, y2, y3;
        net_dft4(x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3], y0, y1, y2, y3);
        x[4 * g] = y0;
        x[4 * g + 1] = y1;
        x[4 * g + 2] = y2;
        x[4 * g + 3] = y3;
    }
}

// row that holds frequency j after the rounds of `pl`
__device__ __forceinline__ int net8_row(int j, int logR, const Net8Plan &pl)
{
    int row = 0, left = logR;
    for (int q = 0; q < pl.n8; ++q) {
        left -= 3;
        row += (j & 7) << left;
        j >>= 3;
    }
    // the 16-point transform leaves frequency f0 + 4 f1 at local index 4 f0 + f1, the smaller ones digit d at d
    return pl.lf == 4 ? row + 4 * (j & 3) + (j >> 2) : row + j;
}

// W_R^m for 0 <= m < R from the table of the first R/2 powers
__device__ __forceinline__ double2 net8_root(const double2 *tw, int m, int half)
{
    const double2 w = tw[m & (half - 1)];
    return m & half ? make_double2(-w.x, -w.y) : w;
}

// IN_SPEC: the inverse transform's first pass computes its own input — what xcorr_spectrum_kernel writes in place, W(o) =
// conj Y(o) from the half spectra of Za, Zb (the zero-padded series' transforms read as complex pairs) — from Z(o) and
// Z(H - o) of both series, so that the spectrum never makes a round trip through HBM: 32 MB read instead of 16 read +
// 16 written + 16 read per 10^6-sample series. Same operations in the same order as that kernel (thread k of it owns
// the points k and H - k; here every point is computed by the lane that loads it).
__device__ __forceinline__ double2 spec_input(const double2 *__restrict__ Za, const double2 *__restrict__ Zb, long long o,
                                              long long H, const TwTab &tt)
{
    const long long om = (H - o) & (H - 1);
    const bool upper = o > om;              // o = H - k with k < H/2: the mirror point of the pair
    const long long k = upper ? om : o, kk = upper ? o : om;
    const double2 w = tw_lookup(tt, (unsigned long long)k, tt.logL);  // e^{-2 pi i k/L}
    auto half_spectrum = [&](const double2 zk, const double2 zh, double2 &xk, double2 &xh) {
        const double2 E = make_double2(0.5 * (zk.x + zh.x), 0.5 * (zk.y - zh.y));
        const double2 O = make_double2(0.5 * (zk.y + zh.y), -0.5 * (zk.x - zh.x));
        const double2 wo = cmul(w, O);
        xk = make_double2(E.x + wo.x, E.y + wo.y);
        xh = make_double2(E.x - wo.x, -(E.y - wo.y));
    };
    double2 ak, ah, bk, bh;
    half_spectrum(Za[k], Za[kk], ak, ah);
    if (Zb == Za) {
        bk = ak;
        bh = ah;
    } else {
        half_spectrum(Zb[k], Zb[kk], bk, bh);
    }
    const double2 sk = make_double2(ak.x * bk.x + ak.y * bk.y, ak.y * bk.x - ak.x * bk.y);
    const double2 sh = make_double2(ah.x * bh.x + ah.y * bh.y, ah.y * bh.x - ah.x * bh.y);
    const double2 se = make_double2(sk.x + sh.x, sk.y - sh.y);
    const double2 sd = make_double2(sk.x - sh.x, sk.y + sh.y);
    if (!upper) {
        const double2 t = cmul(make_double2(w.x, -w.y), sd);
        return make_double2(se.x - t.y, -(se.y + t.x));
    }
    const double2 u = cmul(w, make_double2(sd.x, -sd.y));
    return make_double2(se.x - u.y, -(-se.y + u.x));
}

// grid (H/R/C tiles, batch), R·C/8 lanes (at most 1024). in/out [batch][H]. LDS: (R·C + (R >> lf)·C) points + R/2 roots.
template <int IN, int OUT>
__global__ __launch_bounds__(1024) void fft_pass8_kernel(const double2 *__restrict__ in, double2 *__restrict__ out,
                                                         long long H, int logR, int logC, int logS, long long n,
                                                         PassIo io, TwTab tt)
{
    const long long s = 1LL << logS;
    extern __shared__ double2 lds[];
    const int R = 1 << logR, C = 1 << logC, NT = blockDim.x;
    const Net8Plan pl = net8_plan(logR);
    const int lf = pl.lf;
    double2 *buf = lds;                                        // point (row k, column cc) at P(k, cc)
    double2 *tw = lds + (R << logC) + ((R >> lf) << logC);     // [R/2]
#define NET8_P(k, cc) ((((k) + ((k) >> lf)) << logC) + (cc))
    const long long cols = H >> logR;
    // Tiles narrower than a 128-byte line (C = 4 points) share every line with a neighbour: workgroups are dealt to the
    // XCDs round robin, so the neighbours are given to the SAME XCD, one dispatch round apart (w = 8 m + x -> tile
    // (m & 1) + 2 x + 16 (m >> 1)); dealt in launch order the two halves of a line were fetched by two L2s (FETCH_SIZE
    // twice the algorithmic bytes).
    long long tile = blockIdx.x;
    if ((gridDim.x & 15u) == 0u) {
        const unsigned m = blockIdx.x >> 3, xc = blockIdx.x & 7u;
        tile = (long long)((m & 1u) + 2u * xc) + 16LL * (m >> 1);
    }
    const long long c0 = tile << logC;
    if (IN == IN_PAD) {
        const double *x = io.series + (size_t)blockIdx.y * io.n;
        for (int idx = threadIdx.x; idx < (R << logC); idx += NT) {
            const int k = idx >> logC, cc = idx & (C - 1);
            const long long e = 2 * (c0 + cc + (long long)k * cols);
            double2 v = make_double2(0.0, 0.0);
            if (e < io.n) {
                v.x = x[e];
                if (e + 1 < io.n) v.y = x[e + 1];
            }
            buf[NET8_P(k, cc)] = v;
        }
    } else if (IN == IN_SPEC) {
        const double2 *za = in + (size_t)blockIdx.y * H;
        const double2 *zb = io.zb ? io.zb + (size_t)blockIdx.y * H : za;
        for (int idx = threadIdx.x; idx < (R << logC); idx += NT) {
            const int k = idx >> logC, cc = idx & (C - 1);
            buf[NET8_P(k, cc)] = spec_input(za, zb, c0 + cc + (long long)k * cols, H, tt);
        }
    } else {
        in += (size_t)blockIdx.y * H;
        for (int idx = threadIdx.x; idx < (R << logC); idx += NT) {
            const int k = idx >> logC, cc = idx & (C - 1);
            buf[NET8_P(k, cc)] = in[c0 + cc + (long long)k * cols];
        }
    }
    for (int t = threadIdx.x; t < (R >> 1); t += NT) tw[t] = tw_lookup(tt, (unsigned long long)t, logR);
    __syncthreads();
    // ---- rounds with twiddles: radix 8 ----
    int logn = logR;  // log2 of the current sub-transform length
    for (int q = 0; q < pl.n8; ++q) {
        const int lst = logn - 3;  // log2 of the butterfly stride (rows)
        for (int b = threadIdx.x; b < (R >> 3 << logC); b += NT) {
            const int cc = b & (C - 1), bf = b >> logC;
            const int i = bf & ((1 << lst) - 1), blk = bf >> lst;
            const int row0 = (blk << logn) + i;
            double2 a[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = buf[NET8_P(row0 + (e << lst), cc)];
            net_dft8(a);
            // output digit d times W_len^(i d) = W_R^(i d R/len): the powers 1, 2, 4 from the table, the others products
            const int sh = logR - logn;
            const double2 w1 = net8_root(tw, i << sh, R >> 1), w2 = net8_root(tw, (2 * i) << sh, R >> 1),
                          w4 = net8_root(tw, (4 * i) << sh, R >> 1);
            const double2 w3 = cmul(w1, w2), w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
            a[1] = cmul(a[1], w1);
            a[2] = cmul(a[2], w2);
            a[3] = cmul(a[3], w3);
            a[4] = cmul(a[4], w4);
            a[5] = cmul(a[5], w5);
            a[6] = cmul(a[6], w6);
            a[7] = cmul(a[7], w7);
#pragma unroll
            for (int e = 0; e < 8; ++e) buf[NET8_P(row0 + (e << lst), cc)] = a[e];
        }
        __syncthreads();
        logn -= 3;
    }
    // ---- last round: 2^lf consecutive rows per lane, no twiddles ----
    for (int b = threadIdx.x; b < (R >> lf << logC); b += NT) {
        const int cc = b & (C - 1), row0 = (b >> logC) << lf;
        if (lf == 4) {
            double2 x[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) x[e] = buf[NET8_P(row0 + e, cc)];
            net_dft16(x);
#pragma unroll
            for (int e = 0; e < 16; ++e) buf[NET8_P(row0 + e, cc)] = x[e];
        } else if (lf == 3) {
            double2 x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = buf[NET8_P(row0 + e, cc)];
            net_dft8(x);
#pragma unroll
            for (int e = 0; e < 8; ++e) buf[NET8_P(row0 + e, cc)] = x[e];
        } else if (lf == 2) {
            double2 y0, y1, y2, y3;
            net_dft4(buf[NET8_P(row0, cc)], buf[NET8_P(row0 + 1, cc)], buf[NET8_P(row0 + 2, cc)], buf[NET8_P(row0 + 3, cc)],
                     y0, y1, y2, y3);
            buf[NET8_P(row0, cc)] = y0;
            buf[NET8_P(row0 + 1, cc)] = y1;
            buf[NET8_P(row0 + 2, cc)] = y2;
            buf[NET8_P(row0 + 3, cc)] = y3;
        } else {
            const double2 u = buf[NET8_P(row0, cc)], v = buf[NET8_P(row0 + 1, cc)];
            buf[NET8_P(row0, cc)] = cadd(u, v);
            buf[NET8_P(row0 + 1, cc)] = csub(u, v);
        }
    }
    __syncthreads();
    const bool last = n == R;  // p = 0 for every column: no twiddle
    const int logn_p = 63 - __builtin_clzll((unsigned long long)n);
    const bool col_fast = s >= C;
    out += (size_t)blockIdx.y * H;
    double *lags = OUT == OUT_LAGS ? io.lags + (size_t)blockIdx.y * io.n_lags : nullptr;
    if (OUT == OUT_PERM) {
        // (s = 1, the host's condition: a column's R outputs are one run of R points; lanes walk its POSITIONS)
        for (int idx = threadIdx.x; idx < (R << logC); idx += NT) {
            const int x = idx & (R - 1), cc = idx >> logR;
            const int j = pair_logical(x, R);
            double2 v = buf[NET8_P(net8_row(j, logR, pl), cc)];
            const long long c = c0 + cc;
            if (!last && j != 0 && c != 0) v = cmul(v, tw_lookup(tt, (unsigned long long)j * (unsigned long long)c, logn_p));
            fft_store(out + (c << logR) + x, v);
        }
        return;
    }
    for (int idx = threadIdx.x; idx < (R << logC); idx += NT) {
        int cc, j;
        if (col_fast) {
            cc = idx & (C - 1);
            j = idx >> logC;
        } else {
            j = idx & (R - 1);
            cc = idx >> logR;
        }
        const int r = net8_row(j, logR, pl);
        double2 v = buf[NET8_P(r, cc)];
        const long long c = c0 + cc;
        const long long p = c >> logS, q = c & (s - 1);
        if (!last && j != 0 && p != 0) v = cmul(v, tw_lookup(tt, (unsigned long long)j * (unsigned long long)p, logn_p));
        const long long o = q + s * (((long long)p << logR) + j);
        if (OUT == OUT_LAGS) {
            const long long t = 2 * o;
            if (t < io.n_lags) lags[t] = ((v.x / io.L) / (double)(io.n - t)) * io.scale;
            if (t + 1 < io.n_lags) lags[t + 1] = ((-v.y / io.L) / (double)(io.n - t - 1)) * io.scale;
        } else {
            if (OUT == OUT_CONJ) v.y = -v.y;
            fft_store(out + o, v);
        }
    }
#undef NET8_P
}

// Round 5: the middle of a TWO-pass autocorrelation in one launch — the forward transform's second pass, the spectrum
// step and the inverse transform's first pass — so that the pipeline is three launches and the spectrum never leaves
// the CU: 8 + 16 | 16 + 16 | 16 + 8 MB per 10^6-sample series through HBM instead of 8 + 16 | 16 + 16 | 32 + 16 | 16 + 8.
// H = Ra Rb. The first pass (radix Ra, OUT_PERM) left in[Ra r + pair_phys(j)] = output j of column r; this kernel's
// tile is C neighbouring positions x of every row r < Rb: the columns c = pair_logical(x) of the second pass, whose
// outputs are the frequencies k = c + Ra j. H - k = (Ra - c) + Ra (Rb - 1 - j) (c > 0) lies in the column next door,
// Ra (Rb - j) (c = 0) in the same one: the spectrum step (xcorr_spectrum_kernel's operations, SAME) works in place on
// the tile, each pair by the lane that owns its lower frequency. The forward network (decimation in frequency, as
// fft_pass8_kernel) leaves frequency j in row net8_row(j); the inverse runs the TRANSPOSED network — the rounds in
// reverse order, twiddles before the butterflies — which takes exactly that order and ends in natural order (F = P A
// is symmetric, so F w' = A^T P^T w', and P^T undoes the digit reversal the data already carries): no reordering
// between the two. Output as the inverse's first pass (n = H, s = 1): out[Rb c + j], times e^{-2 pi i j c / H}.
// grid (Ra / C tiles, batch), Rb C / 8 lanes; LDS as fft_pass8_kernel at radix Rb.
__global__ __launch_bounds__(1024) void fft_mid_acf_kernel(const double2 *__restrict__ in, double2 *__restrict__ out,
                                                           long long H, int logRa, int logR, int logC, TwTab tt)
{
    extern __shared__ double2 lds[];
    const int R = 1 << logR, C = 1 << logC, NT = blockDim.x, Ra = 1 << logRa;
    const Net8Plan pl = net8_plan(logR);
    const int lf = pl.lf;
    double2 *buf = lds;
    double2 *tw = lds + (R << logC) + ((R >> lf) << logC);
#define NET8_P(k, cc) ((((k) + ((k) >> lf)) << logC) + (cc))
    long long tile = blockIdx.x;
    if ((gridDim.x & 15u) == 0u) {  // (half-line tiles: the two halves of a line to the same XCD, as fft_pass8_kernel)
        const unsigned m = blockIdx.x >> 3, xc = blockIdx.x & 7u;
        tile = (long long)((m & 1u) + 2u * xc) + 16LL * (m >> 1);
    }
    const int x0 = (int)(tile << logC);
    in += (size_t)blockIdx.y * H;
    out += (size_t)blockIdx.y * H;
    for (int idx = threadIdx.x; idx < (R << logC); idx += NT) {
        const int k = idx >> logC, cc = idx & (C - 1);
        buf[NET8_P(k, cc)] = in[x0 + cc + (long long)k * Ra];
    }
    for (int t = threadIdx.x; t < (R >> 1); t += NT) tw[t] = tw_lookup(tt, (unsigned long long)t, logR);
    __syncthreads();
    // ---- forward: radix-8 rounds with twiddles, then the last round (fft_pass8_kernel) ----
    int logn = logR;
    for (int q = 0; q < pl.n8; ++q) {
        const int lst = logn - 3;
        for (int b = threadIdx.x; b < (R >> 3 << logC); b += NT) {
            const int cc = b & (C - 1), bf = b >> logC;
            const int i = bf & ((1 << lst) - 1), blk = bf >> lst;
            const int row0 = (blk << logn) + i;
            double2 a[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = buf[NET8_P(row0 + (e << lst), cc)];
            net_dft8(a);
            const int sh = logR - logn;
            const double2 w1 = net8_root(tw, i << sh, R >> 1), w2 = net8_root(tw, (2 * i) << sh, R >> 1),
                          w4 = net8_root(tw, (4 * i) << sh, R >> 1);
            const double2 w3 = cmul(w1, w2), w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
            a[1] = cmul(a[1], w1);
            a[2] = cmul(a[2], w2);
            a[3] = cmul(a[3], w3);
            a[4] = cmul(a[4], w4);
            a[5] = cmul(a[5], w5);
            a[6] = cmul(a[6], w6);
            a[7] = cmul(a[7], w7);
#pragma unroll
            for (int e = 0; e < 8; ++e) buf[NET8_P(row0 + (e << lst), cc)] = a[e];
        }
        __syncthreads();
        logn -= 3;
    }
    // the last round and its transpose: net_dft16 leaves frequency f0 + 4 f1 at 4 f0 + f1 (sg16: that digit swap), the
    // smaller ones are in natural order. PRE: the rows are permuted on the way in, POST: on the way out.
    auto last_round = [&](bool transposed) {
        for (int b = threadIdx.x; b < (R >> lf << logC); b += NT) {
            const int cc = b & (C - 1), row0 = (b >> logC) << lf;
            if (lf == 4) {
                double2 x[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) x[e] = buf[NET8_P(row0 + (transposed ? 4 * (e & 3) + (e >> 2) : e), cc)];
                net_dft16(x);
#pragma unroll
                for (int e = 0; e < 16; ++e) buf[NET8_P(row0 + e, cc)] = x[transposed ? 4 * (e & 3) + (e >> 2) : e];
            } else if (lf == 3) {
                double2 x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = buf[NET8_P(row0 + e, cc)];
                net_dft8(x);
#pragma unroll
                for (int e = 0; e < 8; ++e) buf[NET8_P(row0 + e, cc)] = x[e];
            } else if (lf == 2) {
                double2 y0, y1, y2, y3;
                net_dft4(buf[NET8_P(row0, cc)], buf[NET8_P(row0 + 1, cc)], buf[NET8_P(row0 + 2, cc)], buf[NET8_P(row0 + 3, cc)],
                         y0, y1, y2, y3);
                buf[NET8_P(row0, cc)] = y0;
                buf[NET8_P(row0 + 1, cc)] = y1;
                buf[NET8_P(row0 + 2, cc)] = y2;
                buf[NET8_P(row0 + 3, cc)] = y3;
            } else {
                const double2 u = buf[NET8_P(row0, cc)], v = buf[NET8_P(row0 + 1, cc)];
                buf[NET8_P(row0, cc)] = cadd(u, v);
                buf[NET8_P(row0 + 1, cc)] = csub(u, v);
            }
        }
    };
    last_round(false);
    __syncthreads();
    // ---- spectrum step, in place (xcorr_spectrum_kernel<true>, pair by pair) ----
    // (k = c + Ra j <= H / 2 exactly for j < Rb / 2, whatever the column — and for the one point H / 2 itself, c = 0,
    // j = Rb / 2, which rides as one more item of the tile that holds column 0: every item is a pair to do)
    const int n_items = (R >> 1 << logC) + (x0 == 0 ? 1 : 0);
    for (int idx = threadIdx.x; idx < n_items; idx += NT) {
        const bool mid_point = idx == (R >> 1 << logC);
        const int cc = mid_point ? 0 : idx & (C - 1), j = mid_point ? R >> 1 : idx >> logC;
        const int x = x0 + cc, c = pair_logical(x, Ra);
        const long long k = (long long)c + (long long)Ra * j;
        const int ccp = (x < 2 ? x : x ^ 1) - x0;
        const int jp = c == 0 ? (R - j) & (R - 1) : R - 1 - j;
        const int pos = NET8_P(net8_row(j, logR, pl), cc), posp = NET8_P(net8_row(jp, logR, pl), ccp);
        const double2 zk = buf[pos], zh = buf[posp];
        const double2 w = tw_lookup(tt, (unsigned long long)k, tt.logL);  // e^{-2 pi i k/L}
        const double2 E = make_double2(0.5 * (zk.x + zh.x), 0.5 * (zk.y - zh.y));
        const double2 O = make_double2(0.5 * (zk.y + zh.y), -0.5 * (zk.x - zh.x));
        const double2 wo = cmul(w, O);
        const double2 ak = make_double2(E.x + wo.x, E.y + wo.y);     // X(k)
        const double2 ah = make_double2(E.x - wo.x, -(E.y - wo.y));  // X(H-k)
        const double2 sk = make_double2(ak.x * ak.x + ak.y * ak.y, ak.y * ak.x - ak.x * ak.y);  // A conj(A), as mul_conj
        const double2 sh = make_double2(ah.x * ah.x + ah.y * ah.y, ah.y * ah.x - ah.x * ah.y);
        const double2 se = make_double2(sk.x + sh.x, sk.y - sh.y);
        const double2 sd = make_double2(sk.x - sh.x, sk.y + sh.y);
        const double2 t = cmul(make_double2(w.x, -w.y), sd);
        buf[pos] = make_double2(se.x - t.y, -(se.y + t.x));
        if (k != 0 && 2 * k != H) {
            const double2 u = cmul(w, make_double2(sd.x, -sd.y));
            buf[posp] = make_double2(se.x - u.y, -(-se.y + u.x));
        }
    }
    __syncthreads();
    // ---- inverse: the transposed network — last round first, then the radix-8 rounds, twiddles BEFORE the butterflies ----
    last_round(true);
    __syncthreads();
    for (int q = pl.n8 - 1; q >= 0; --q) {
        logn = logR - 3 * q;
        const int lst = logn - 3;
        for (int b = threadIdx.x; b < (R >> 3 << logC); b += NT) {
            const int cc = b & (C - 1), bf = b >> logC;
            const int i = bf & ((1 << lst) - 1), blk = bf >> lst;
            const int row0 = (blk << logn) + i;
            double2 a[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = buf[NET8_P(row0 + (e << lst), cc)];
            const int sh = logR - logn;
            const double2 w1 = net8_root(tw, i << sh, R >> 1), w2 = net8_root(tw, (2 * i) << sh, R >> 1),
                          w4 = net8_root(tw, (4 * i) << sh, R >> 1);
            const double2 w3 = cmul(w1, w2), w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
            a[1] = cmul(a[1], w1);
            a[2] = cmul(a[2], w2);
            a[3] = cmul(a[3], w3);
            a[4] = cmul(a[4], w4);
            a[5] = cmul(a[5], w5);
            a[6] = cmul(a[6], w6);
            a[7] = cmul(a[7], w7);
            net_dft8(a);
#pragma unroll
            for (int e = 0; e < 8; ++e) buf[NET8_P(row0 + (e << lst), cc)] = a[e];
        }
        __syncthreads();
    }
    // ---- the inverse's first pass ends: twiddle e^{-2 pi i j c / H}, a column's outputs one run of Rb points ----
    int logH = 0;
    while ((1LL << logH) < H) ++logH;
    for (int idx = threadIdx.x; idx < (R << logC); idx += NT) {
        const int j = idx & (R - 1), cc = idx >> logR;
        const int c = pair_logical(x0 + cc, Ra);
        double2 v = buf[NET8_P(j, cc)];
        if (j != 0 && c != 0) v = cmul(v, tw_lookup(tt, (unsigned long long)j * (unsigned long long)c, logH));
        fft_store(out + ((long long)c << logR) + j, v);
    }
#undef NET8_P
}

// spec[b][k], k = 0..H, from Z = FFT_H of the reals read as complex pairs. grid (ceil((H/2+1)/256), batch)
__global__ void r2c_post_kernel(const double2 *__restrict__ Z, double2 *__restrict__ spec, long long H, TwTab tt)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > H / 2) return;
    Z += (size_t)blockIdx.y * H;
    spec += (size_t)blockIdx.y * (H + 1);
    const long long kk = (H - k) & (H - 1);
    const double2 zk = Z[k], zh = Z[kk];
    const double2 E = make_double2(0.5 * (zk.x + zh.x), 0.5 * (zk.y - zh.y));
    const double2 O = make_double2(0.5 * (zk.y + zh.y), -0.5 * (zk.x - zh.x));  // (zk - conj zh) / (2i)
    const double2 w = tw_lookup(tt, (unsigned long long)k, tt.logL);
    const double2 wo = cmul(w, O);
    spec[k] = make_double2(E.x + wo.x, E.y + wo.y);
    spec[H - k] = make_double2(E.x - wo.x, -(E.y - wo.y));  // X(H-k) = conj(E - w O); k = 0 -> X(H)
}

// r2c_post_kernel and the column sums of |X_k|^2 in one pass over Z (round 6: the full-lag MSD of series too long for the
// fused kernels wrote the half spectra only to read them once): partial[split][k], k = 0..H, = the sum over the split's
// rows of |X_row(k)|^2, X from Z exactly as r2c_post_kernel makes it (the same operations in the same order, the rows
// added alternately into two sums as power_rows_kernel of msd_fft.hip did: bit-identical results).
// grid (ceil((H/2+1)/256), splits); rows [row0, row1) of Z [.][H].
__global__ __launch_bounds__(256) void r2c_power_rows_kernel(const double2 *__restrict__ Z, long long H, long long row0,
                                                             long long row1, double *__restrict__ partial, TwTab tt)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > H / 2) return;
    const long long n = row1 - row0;
    const long long ra = row0 + n * blockIdx.y / gridDim.y, rb = row0 + n * (blockIdx.y + 1) / gridDim.y;
    const long long kk = (H - k) & (H - 1);
    const double2 w = tw_lookup(tt, (unsigned long long)k, tt.logL);
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;  // |X_k|^2 and |X_(H-k)|^2, even / odd rows
    for (long long q = ra; q < rb; ++q) {
        const double2 zk = Z[(size_t)q * H + k], zh = Z[(size_t)q * H + kk];
        const double2 E = make_double2(0.5 * (zk.x + zh.x), 0.5 * (zk.y - zh.y));
        const double2 O = make_double2(0.5 * (zk.y + zh.y), -0.5 * (zk.x - zh.x));
        const double2 wo = cmul(w, O);
        const double2 xk = make_double2(E.x + wo.x, E.y + wo.y);
        const double2 xh = make_double2(E.x - wo.x, -(E.y - wo.y));
        const double pk = xk.x * xk.x + xk.y * xk.y, ph = xh.x * xh.x + xh.y * xh.y;
        if ((q - ra) & 1) {
            a1 += pk;
            b1 += ph;
        } else {
            a0 += pk;
            b0 += ph;
        }
    }
    double *p = partial + (size_t)blockIdx.y * (H + 1);
    p[k] = a0 + a1;
    if (H - k != k) p[H - k] = b0 + b1;  // (k = 0 -> X(H))
}

// Round 6, second step: the SECOND pass of a two-pass transform and the column sums of |X_k|^2 in one kernel — the packed
// transform Z never exists in memory. H = Ra Rb; the first pass (radix Ra, IN_PAD, OUT_PERM) left in[Ra r + pair_phys(j)] =
// output j of column r, as for fft_mid_acf_kernel, whose tile this kernel shares: C neighbouring positions x of every row
// r < Rb = the columns c = pair_logical(x) of the second pass, so that the frequencies k = c + Ra j and H - k, which the
// real spectrum combines, sit in the same tile. A workgroup owns ONE tile and walks the rows (series) of its split: load,
// transform (the forward network of fft_pass8_kernel), the sums of its 4 pairs (k, H - k) per lane into registers; partial[split][k], k = 0..H, once at the end. Per series and pass over HBM: 16 H bytes read,
// nothing written (the plain sequence: 16 H read + 16 H written by the pass, 16 H read by r2c_power_rows_kernel).
// grid (Ra / C tiles, splits), EXACTLY Rb C / 8 lanes (64 .. 512); rows [row0, row1) of in [.][H]; LDS as fft_pass8_kernel
// at radix Rb.
template <int logR, int logC>
__global__ __launch_bounds__(((1 << logR) << logC) >> 3, 3) void fft_power_pass_kernel(const double2 *__restrict__ in, long long H,
                                                                                      int logRa, long long row0, long long row1,
                                                                                      double *__restrict__ partial, TwTab tt)
{
    extern __shared__ double2 lds[];
    constexpr int R = 1 << logR, C = 1 << logC, NT = (R * C) >> 3;
    const int Ra = 1 << logRa;
    constexpr Net8Plan pl = net8_plan(logR);
    constexpr int lf = pl.lf;
    double2 *buf = lds;
    double2 *tw = lds + (R << logC) + ((R >> lf) << logC);
#define NET8_P(k, cc) ((((k) + ((k) >> lf)) << logC) + (cc))
    long long tile = blockIdx.x;
    if ((gridDim.x & 15u) == 0u) {  // (half-line tiles: the two halves of a line to the same XCD, as fft_pass8_kernel)
        const unsigned m = blockIdx.x >> 3, xc = blockIdx.x & 7u;
        tile = (long long)((m & 1u) + 2u * xc) + 16LL * (m >> 1);
    }
    const int x0 = (int)(tile << logC);
    const long long n_rows = row1 - row0;
    const long long ra = row0 + n_rows * blockIdx.y / gridDim.y, rb = row0 + n_rows * (blockIdx.y + 1) / gridDim.y;
    for (int t = threadIdx.x; t < (R >> 1); t += NT) tw[t] = tw_lookup(tt, (unsigned long long)t, logR);
    // This lane's four pairs (item idx = lane + i NT < R/2 * C: column cc = idx mod C, output j = idx div C < R/2) and, for
    // lane 0 of the tile that holds column 0, the point H/2 (c = 0, j = R/2: its own partner). The sums are the BILINEAR ones
    // of the fused kernels (msd_fft.hip, msd_power_lds2_kernel): S = |Z_k|^2, S' = |Z_(H-k)|^2, T = Im(Z_k Z_(H-k)) per series —
    // six fused multiply-adds per pair — and the frequencies are sorted out once, after the last series:
    //   |X_k|^2 = (S + S')/2 + Im(w) (S - S')/2 + Re(w) T,  |X_(H-k)|^2 = (S + S')/2 - Im(w) (S - S')/2 - Re(w) T,  w = e^{-2 pi i k/L}
    int pos[4], posp[4];
    double sk[4], sh[4], tk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = threadIdx.x + i * NT;
        const int cc = idx & (C - 1), j = idx >> logC;
        const int x = x0 + cc, c = pair_logical(x, Ra);
        const int ccp = (x < 2 ? x : x ^ 1) - x0;
        const int jp = c == 0 ? (R - j) & (R - 1) : R - 1 - j;
        pos[i] = NET8_P(net8_row(j, logR, pl), cc);
        posp[i] = NET8_P(net8_row(jp, logR, pl), ccp);
        sk[i] = sh[i] = tk[i] = 0.0;
    }
    const bool has_mid = x0 == 0 && threadIdx.x == 0;
    const int pos_mid = NET8_P(net8_row(R >> 1, logR, pl), 0);
    double s_mid = 0.0;
    double2 v[8];
    // (`lane` = threadIdx.x through an opaque copy made inside the series loop: every address, LDS position and twiddle of
    // the transform is the same for all series, and hoisted out of the loop they would fill the register budget — 100+
    // spilled registers; recomputed per series they are a few integer operations)
    auto fetch = [&](long long q, int lane) {
        const double2 *row = in + (size_t)q * H + x0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {  // (R C = 8 NT points: all eight loads in flight)
            const int idx = lane + i * NT;
            typedef double d2_t __attribute__((ext_vector_type(2)));
            const d2_t t = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(row + (idx & (C - 1)) + (long long)(idx >> logC) * Ra));
            v[i] = make_double2(t[0], t[1]);
        }
    };
    if (ra < rb) fetch(ra, threadIdx.x);
    for (long long q = ra; q < rb; ++q) {
        int lane = threadIdx.x;
        asm volatile("" : "+v"(lane));
        __syncthreads();  // (the previous series' pairs are read; the first trip: the twiddle table is written)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = lane + i * NT;
            buf[NET8_P(idx >> logC, idx & (C - 1))] = v[i];
        }
        __syncthreads();
        int logn = logR;
        for (int r8 = 0; r8 < pl.n8; ++r8) {
            const int lst = logn - 3;
            {  // (R/8 * C butterflies = NT: one per lane)
                const int b = lane;
                const int cc = b & (C - 1), bf = b >> logC;
                const int i = bf & ((1 << lst) - 1), blk = bf >> lst;
                const int rw0 = (blk << logn) + i;
                double2 a[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = buf[NET8_P(rw0 + (e << lst), cc)];
                net_dft8(a);
                const int sft = logR - logn;
                const double2 w1 = net8_root(tw, i << sft, R >> 1), w2 = net8_root(tw, (2 * i) << sft, R >> 1),
                              w4 = net8_root(tw, (4 * i) << sft, R >> 1);
                const double2 w3 = cmul(w1, w2), w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
                a[1] = cmul(a[1], w1);
                a[2] = cmul(a[2], w2);
                a[3] = cmul(a[3], w3);
                a[4] = cmul(a[4], w4);
                a[5] = cmul(a[5], w5);
                a[6] = cmul(a[6], w6);
                a[7] = cmul(a[7], w7);
#pragma unroll
                for (int e = 0; e < 8; ++e) buf[NET8_P(rw0 + (e << lst), cc)] = a[e];
            }
            __syncthreads();
            logn -= 3;
        }
        for (int b = lane; b < (R >> lf << logC); b += NT) {
            const int cc = b & (C - 1), rw0 = (b >> logC) << lf;
            if (lf == 4) {
                double2 x[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) x[e] = buf[NET8_P(rw0 + e, cc)];
                net_dft16(x);
#pragma unroll
                for (int e = 0; e < 16; ++e) buf[NET8_P(rw0 + e, cc)] = x[e];
            } else if (lf == 3) {
                double2 x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = buf[NET8_P(rw0 + e, cc)];
                net_dft8(x);
#pragma unroll
                for (int e = 0; e < 8; ++e) buf[NET8_P(rw0 + e, cc)] = x[e];
            } else if (lf == 2) {
                double2 y0, y1, y2, y3;
                net_dft4(buf[NET8_P(rw0, cc)], buf[NET8_P(rw0 + 1, cc)], buf[NET8_P(rw0 + 2, cc)], buf[NET8_P(rw0 + 3, cc)],
                         y0, y1, y2, y3);
                buf[NET8_P(rw0, cc)] = y0;
                buf[NET8_P(rw0 + 1, cc)] = y1;
                buf[NET8_P(rw0 + 2, cc)] = y2;
                buf[NET8_P(rw0 + 3, cc)] = y3;
            } else {
                const double2 u = buf[NET8_P(rw0, cc)], w = buf[NET8_P(rw0 + 1, cc)];
                buf[NET8_P(rw0, cc)] = cadd(u, w);
                buf[NET8_P(rw0 + 1, cc)] = csub(u, w);
            }
        }
        __syncthreads();
        if (q + 1 < rb) fetch(q + 1, lane);  // (the next series' tile lands under the sums and the barrier)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double2 zk = buf[pos[i]], zh = buf[posp[i]];
            sk[i] = __builtin_fma(zk.x, zk.x, sk[i]);
            sk[i] = __builtin_fma(zk.y, zk.y, sk[i]);
            sh[i] = __builtin_fma(zh.x, zh.x, sh[i]);
            sh[i] = __builtin_fma(zh.y, zh.y, sh[i]);
            tk[i] = __builtin_fma(zk.x, zh.y, tk[i]);
            tk[i] = __builtin_fma(zk.y, zh.x, tk[i]);
        }
        if (has_mid) {
            const double2 z = buf[pos_mid];
            s_mid = __builtin_fma(z.x, z.x, s_mid);
            s_mid = __builtin_fma(z.y, z.y, s_mid);
        }
    }
    double *p = partial + (size_t)blockIdx.y * (H + 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = threadIdx.x + i * NT;
        const int cc = idx & (C - 1), j = idx >> logC;
        const long long k = (long long)pair_logical(x0 + cc, Ra) + (long long)Ra * j;
        const double2 w = tw_lookup(tt, (unsigned long long)k, tt.logL);  // e^{-2 pi i k/L}
        if (k == 0) {  // Z_0 is its own partner: X_0 = Re + Im, X_H = Re - Im
            p[0] = sk[i] + tk[i];
            p[H] = sk[i] - tk[i];
        } else {
            const double m = 0.5 * (sk[i] + sh[i]), d = 0.5 * (sk[i] - sh[i]);
            p[k] = m + w.y * d + w.x * tk[i];
            p[H - k] = m - w.y * d - w.x * tk[i];
        }
    }
    if (has_mid) p[H >> 1] = s_mid;  // X_(H/2) = conj Z_(H/2)
#undef NET8_P
}

// W = conj Y (the input of the forward transform that stands for the inverse one) from the Hermitian half spectrum
// S[b][0..H]. grid (ceil((H/2+1)/256), batch)
__global__ void c2r_pre_kernel(const double2 *__restrict__ S, double2 *__restrict__ W, long long H, TwTab tt)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > H / 2) return;
    S += (size_t)blockIdx.y * (H + 1);
    W += (size_t)blockIdx.y * H;
    const double2 sk = S[k], sh = S[H - k];
    const double2 w = tw_lookup(tt, (unsigned long long)k, tt.logL);  // e^{-2 pi i k/L}
    const double2 se = make_double2(sk.x + sh.x, sk.y - sh.y);       // S(k) + conj S(H-k)
    const double2 sd = make_double2(sk.x - sh.x, sk.y + sh.y);       // S(k) - conj S(H-k)
    const double2 t = cmul(make_double2(w.x, -w.y), sd);             // e^{+2 pi i k/L} sd
    const double2 yk = make_double2(se.x - t.y, se.y + t.x);         // se + i t
    W[k] = make_double2(yk.x, -yk.y);
    if (k != 0 && 2 * k != H) {
        // Y(H-k) = conj(se) - i w (S(H-k) - conj S(k)) = conj(se) + i w conj(sd)
        const double2 u = cmul(w, make_double2(sd.x, -sd.y));
        const double2 yh = make_double2(se.x - u.y, -se.y + u.x);
        W[H - k] = make_double2(yh.x, -yh.y);
    }
}

// The correlation pipeline's one pointwise step, in place: Za, Zb = FFT_H of the two zero-padded series read as
// complex pairs  ->  half spectra A, B (as r2c_post_kernel)  ->  S = A conj(B)  ->  W = conj Y (as c2r_pre_kernel),
// written over Za (thread k owns the points k and H-k of both inputs and of the output). SAME: Zb is Za.
template <bool SAME>
__global__ void xcorr_spectrum_kernel(double2 *__restrict__ Za, const double2 *__restrict__ Zb, long long H, TwTab tt)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > H / 2) return;
    Za += (size_t)blockIdx.y * H;
    Zb += (size_t)blockIdx.y * H;
    const long long kk = (H - k) & (H - 1);
    const double2 w = tw_lookup(tt, (unsigned long long)k, tt.logL);  // e^{-2 pi i k/L}
    auto half_spectrum = [&](const double2 zk, const double2 zh, double2 &xk, double2 &xh) {
        const double2 E = make_double2(0.5 * (zk.x + zh.x), 0.5 * (zk.y - zh.y));
        const double2 O = make_double2(0.5 * (zk.y + zh.y), -0.5 * (zk.x - zh.x));
        const double2 wo = cmul(w, O);
        xk = make_double2(E.x + wo.x, E.y + wo.y);     // X(k)
        xh = make_double2(E.x - wo.x, -(E.y - wo.y));  // X(H-k)
    };
    double2 ak, ah, bk, bh;
    half_spectrum(Za[k], Za[kk], ak, ah);
    if (SAME) {
        bk = ak;
        bh = ah;
    } else {
        half_spectrum(Zb[k], Zb[kk], bk, bh);
    }
    const double2 sk = make_double2(ak.x * bk.x + ak.y * bk.y, ak.y * bk.x - ak.x * bk.y);  // A conj(B), as mul_conj
    const double2 sh = make_double2(ah.x * bh.x + ah.y * bh.y, ah.y * bh.x - ah.x * bh.y);
    const double2 se = make_double2(sk.x + sh.x, sk.y - sh.y);
    const double2 sd = make_double2(sk.x - sh.x, sk.y + sh.y);
    const double2 t = cmul(make_double2(w.x, -w.y), sd);
    Za[k] = make_double2(se.x - t.y, -(se.y + t.x));
    if (k != 0 && 2 * k != H) {
        const double2 u = cmul(w, make_double2(sd.x, -sd.y));
        Za[kk] = make_double2(se.x - u.y, -(-se.y + u.x));
    }
}

__global__ void conj_copy_kernel(const double2 *__restrict__ in, double2 *__restrict__ out, long long count)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = make_double2(in[i].x, -in[i].y);
}

// The table of roots of order L = 2 H for this context (built on the launch stream when the length changes).
bool fft_twiddle_table(mdhip_ctx *ctx, long long H, TwTab &tt)
{
    const long long L = 2 * H;
    int logL = 0;
    while ((1LL << logL) < L) ++logL;
    const long long nA = L > 1024 ? L >> 10 : 1, nB = L > 1024 ? 1024 : L;
    double2 *tab = (double2 *)mdhip_ws(ctx, WS_FFT_TW, (size_t)(nA + nB) * sizeof(double2));
    if (!tab) return false;
    tt.A = tab;
    tt.B = tab + nA;
    tt.logL = logL;
    if (ctx->fft_tw_logL != logL || ctx->fft_tw_ptr != tab) {
        const long long m = std::max(nA, nB);
        hipLaunchKernelGGL(fft_twiddle_table_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, tab,
                           nA, tab + nA, nB, L);
        ctx->fft_tw_logL = logL;
        ctx->fft_tw_ptr = tab;
    }
    return true;
}

struct PassPlan {
    int n_pass = 0;
    int logR[FFT_MAX_PASSES] = {0, 0, 0, 0, 0, 0, 0, 0};
};

PassPlan plan_passes(const mdhip_ctx *ctx, long long H, int batch)
{
    int logH = 0;
    while ((1LL << logH) < H) ++logH;
    int max_logr = std::min(std::max(ctx->opt_fft_logr, 4), FFT_MAX_LOGR);
    // a one-column tile of the radix (R + R/2 complex points) must fit the CU's LDS
    while (max_logr > 4 && ((size_t)24 << max_logr) > ctx->lds_max) --max_logr;
    if (max_logr > 8 && logH > 0) {
        // the large radices save a pass over HBM only while their tiles (>= 64 KB, 4 columns) still fill the chip:
        // 3 x 2^17 points as 9 + 8 put 96 workgroups on 256 CUs and measured 0.050 ms against 0.044 ms for three passes
        const int np = (logH + max_logr - 1) / max_logr;
        const int lr0 = (logH + np - 1) / np;
        if (lr0 > 8 && ((H >> lr0) >> 2) * (long long)std::max(batch, 1) < (long long)ctx->cu_count) max_logr = 8;
    }
    PassPlan p;
    p.n_pass = (logH + max_logr - 1) / max_logr;
    if (p.n_pass > FFT_MAX_PASSES) {  // H > 2^32 at radix 16: no caller gets here (xcorr caps n at 2^29)
        p.n_pass = -1;
        return p;
    }
    for (int i = 0, left = logH; i < p.n_pass; ++i) {
        p.logR[i] = (left + (p.n_pass - i) - 1) / (p.n_pass - i);  // as even as possible, larger radices first
        left -= p.logR[i];
    }
    return p;
}

// Tile width (log2 columns) and LDS bytes of a radix-2^logR pass through the radix-8 network; false when that pass runs the
// radix-4 network instead (small radix, switched off, or no tile fits)
bool net8_tile(const mdhip_ctx *ctx, long long H, int logR, int &logC, size_t &lds)
{
    if (logR < NET8_MIN_LOGR || ctx->opt_fft_net8 == 0) return false;
    const long long cols = H >> logR;
    const int lf = net8_plan(logR).lf;
    auto bytes = [&](int lc) {
        return ((((size_t)1 << logR) + ((size_t)1 << logR >> lf)) << lc) * sizeof(double2) + ((size_t)1 << logR >> 1) * sizeof(double2);
    };
    // two workgroups per CU (their loads, transforms and stores overlap), i.e. <= 78 KB of LDS each
    const size_t cap = ctx->opt_fft_net8 == 2 ? ctx->lds_max : (size_t)78 * 1024;  // (2: one workgroup per CU, A/B)
    logC = NET8_MAX_LOGC;
    while (logC > 0 && (bytes(logC) > cap || (1LL << logC) > cols)) --logC;
    lds = bytes(logC);
    return lds <= ctx->lds_max;
}

template <int IN, int OUT>
bool launch_pass(mdhip_ctx *ctx, const double2 *in, double2 *out, long long H, int batch, int logR, int logS,
                 const PassIo &io, const TwTab &tt)
{
    const long long cols = H >> logR;
    {
        int logC;
        size_t lds;
        if (net8_tile(ctx, H, logR, logC, lds)) {
            const int threads = (int)std::min<long long>(1024, std::max<long long>(64, ((1LL << logR) << logC) >> 3));
            const dim3 grid((unsigned)(cols >> logC), (unsigned)batch);
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fft_pass8_kernel<IN, OUT>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) {
                mdhip_fail(ctx, MDHIP_EHIP, "fft pass of radix 2^%d needs %zu bytes of LDS: %s", logR, lds, hipGetErrorString(e));
                return false;
            }
            hipLaunchKernelGGL((fft_pass8_kernel<IN, OUT>), grid, dim3((unsigned)threads), lds, ctx->stream, in, out, H, logR,
                               logC, logS, H >> logS, io, tt);
            return true;
        }
    }
    if (IN == IN_SPEC) {  // (callers ask for the fused spectrum input only when net8_tile says the first pass takes it)
        mdhip_fail(ctx, MDHIP_EHIP, "internal: spectrum-input pass of radix 2^%d has no radix-8 tile", logR);
        return false;
    }
    // 16 columns per tile (256-byte runs) while the tile fits 64 KB of LDS, fewer for the larger radices
    int logC = std::min(std::max(ctx->opt_fft_logc, 0), 6);
    while (logC > 0 && ((size_t)16 << logR << logC) > (size_t)128 * 1024) --logC;
    while ((1LL << logC) > cols) --logC;
    const size_t lds = ((size_t)(1 << logR << logC) + (size_t)(1 << logR >> 1)) * sizeof(double2);
    const dim3 grid((unsigned)(cols >> logC), (unsigned)batch);
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fft_pass_kernel<IN, OUT>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        mdhip_fail(ctx, MDHIP_EHIP, "fft pass of radix 2^%d needs %zu bytes of LDS: %s", logR, lds, hipGetErrorString(e));
        return false;
    }
    hipLaunchKernelGGL((fft_pass_kernel<IN, OUT>), grid, dim3(FFT_THREADS), lds, ctx->stream, in, out, H, logR, logC,
                       logS, H >> logS, io, tt);
    return true;
}

// FFT_H of `batch` series. The passes alternate between the two buffers; returns the buffer that holds the result
// (x after an even number of passes, y after an odd one), nullptr when a pass could not be launched (error set).
// x is overwritten. in_mode IN_PAD: the first pass reads io.series instead of x; out_mode OUT_LAGS: the last pass
// writes io.lags instead of a buffer.
double2 *fft_forward(mdhip_ctx *ctx, double2 *x, double2 *y, long long H, int batch, int in_mode, int out_mode,
                     const PassIo &io, const TwTab &tt)
{
    const PassPlan p = plan_passes(ctx, H, batch);
    if (p.n_pass < 0) {
        mdhip_fail(ctx, MDHIP_ELIMIT, "transform of %lld complex points needs more than %d passes", H, FFT_MAX_PASSES);
        return nullptr;
    }
    if (p.n_pass == 0) {  // H = 1: the transform is the identity (plain buffers only)
        if (out_mode == OUT_CONJ)
            hipLaunchKernelGGL(conj_copy_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, ctx->stream, x, y,
                               (long long)batch);
        return out_mode == OUT_CONJ ? y : x;
    }
    double2 *src = x, *dst = y;
    int s = 0;  // log2 of the product of the radices done
    for (int i = 0; i < p.n_pass; ++i) {
        const int im = i == 0 ? in_mode : IN_PLAIN, om = i == p.n_pass - 1 ? out_mode : OUT_PLAIN;
        bool ok;
        if (im == IN_SPEC && om == OUT_PLAIN) ok = launch_pass<IN_SPEC, OUT_PLAIN>(ctx, src, dst, H, batch, p.logR[i], s, io, tt);
        else if (im == IN_SPEC && om == OUT_LAGS) ok = launch_pass<IN_SPEC, OUT_LAGS>(ctx, src, dst, H, batch, p.logR[i], s, io, tt);
        else if (im == IN_PAD && om == OUT_PLAIN) ok = launch_pass<IN_PAD, OUT_PLAIN>(ctx, src, dst, H, batch, p.logR[i], s, io, tt);
        else if (im == IN_PLAIN && om == OUT_LAGS) ok = launch_pass<IN_PLAIN, OUT_LAGS>(ctx, src, dst, H, batch, p.logR[i], s, io, tt);
        else if (im == IN_PLAIN && om == OUT_CONJ) ok = launch_pass<IN_PLAIN, OUT_CONJ>(ctx, src, dst, H, batch, p.logR[i], s, io, tt);
        else ok = launch_pass<IN_PLAIN, OUT_PLAIN>(ctx, src, dst, H, batch, p.logR[i], s, io, tt);
        if (!ok) return nullptr;
        s += p.logR[i];
        std::swap(src, dst);
    }
    return src;
}

}  // namespace

// ---- interface used by xcorr.hip / msd_fft.hip (declared in ctx.h) ------------------------------------------------

// Half spectra X[b][0..L/2] of `batch` real series x[b][0..L-1], L a power of two >= 2.
// d_real is overwritten (it is one of the two transform buffers); d_tmp holds batch * L/2 complex points.
int mdhip_fft_r2c(mdhip_ctx *ctx, double *d_real, double2 *d_tmp, double2 *d_spec, long long L, int batch)
{
    MD_REQUIRE(L >= 2 && (L & (L - 1)) == 0, "transform length %lld is not a power of two", L);
    const long long H = L / 2;
    if (batch > FFT_MAX_BATCH) {  // grid.y
        for (int b0 = 0; b0 < batch; b0 += FFT_MAX_BATCH) {
            const int rc = mdhip_fft_r2c(ctx, d_real + (size_t)b0 * L, d_tmp + (size_t)b0 * H,
                                         d_spec + (size_t)b0 * (H + 1), L, std::min(FFT_MAX_BATCH, batch - b0));
            if (rc) return rc;
        }
        return MDHIP_OK;
    }
    TwTab tt;
    if (!fft_twiddle_table(ctx, H, tt)) return MDHIP_ENOMEM;
    double2 *Z = fft_forward(ctx, reinterpret_cast<double2 *>(d_real), d_tmp, H, batch, IN_PLAIN, OUT_PLAIN, PassIo{}, tt);
    if (!Z) return MDHIP_EHIP;
    hipLaunchKernelGGL(r2c_post_kernel, dim3((unsigned)((H / 2 + 1 + 255) / 256), (unsigned)batch), dim3(256), 0,
                       ctx->stream, Z, d_spec, H, tt);
    MD_HIP(hipGetLastError());
    return MDHIP_OK;
}

// Z[b][0..L/2) = FFT_(L/2) of the zero-padded real series d_series[b][0..n) read as complex pairs — the transform WITHOUT
// the half-spectrum post pass, for a caller that reduces |X_k|^2 over many series itself (mdhip_fft_power_rows): the first
// pass reads the n samples where they are (no padded copy: the padding is implicit), *Z_out is whichever of the two
// buffers (batch * L/2 complex points each) holds the result.
int mdhip_fft_r2c_packed(mdhip_ctx *ctx, const double *d_series, long long n, double2 *d_buf0, double2 *d_buf1, long long L,
                         int batch, const double2 **Z_out)
{
    MD_REQUIRE(L >= 4 && (L & (L - 1)) == 0 && n <= L, "transform length %lld is not a power of two >= the series", L);
    const long long H = L / 2;
    TwTab tt;
    if (!fft_twiddle_table(ctx, H, tt)) return MDHIP_ENOMEM;
    const double2 *Z = nullptr;
    for (int b0 = 0; b0 < batch; b0 += FFT_MAX_BATCH) {  // (grid.y)
        PassIo io{};
        io.series = d_series + (size_t)b0 * n;
        io.n = n;
        double2 *z = fft_forward(ctx, d_buf0 + (size_t)b0 * H, d_buf1 + (size_t)b0 * H, H, std::min(FFT_MAX_BATCH, batch - b0),
                                 IN_PAD, OUT_PLAIN, io, tt);
        if (!z) return MDHIP_EHIP;
        if (b0 == 0) Z = z - (size_t)b0 * H;  // (the same number of passes for every part: the same buffer)
    }
    *Z_out = Z;
    MD_HIP(hipGetLastError());
    return MDHIP_OK;
}

// partial[split][0..L/2] = sum over the rows [row0, row1) of Z (mdhip_fft_r2c_packed) of |X_row(k)|^2, in `splits` parts
int mdhip_fft_power_rows(mdhip_ctx *ctx, const double2 *Z, long long L, long long row0, long long row1, int splits,
                         double *d_partial)
{
    const long long H = L / 2;
    TwTab tt;
    if (!fft_twiddle_table(ctx, H, tt)) return MDHIP_ENOMEM;
    hipLaunchKernelGGL(r2c_power_rows_kernel, dim3((unsigned)((H / 2 + 1 + 255) / 256), (unsigned)splits), dim3(256), 0,
                       ctx->stream, Z, H, row0, row1, d_partial, tt);
    MD_HIP(hipGetLastError());
    return MDHIP_OK;
}

// The two-pass form of mdhip_fft_r2c_packed + mdhip_fft_power_rows (fft_power_pass_kernel): H = L/2 = Ra Rb.
// mdhip_fft_power2_plan: false when this length does not take it (H < 2^10: the callers' lengths are far beyond).
// mdhip_fft_first_perm: the first pass over `batch` zero-padded series d_series[b][0..n) into d_buf [batch][H].
// mdhip_fft_power_pass: partial[split][0..L/2] = sum over the rows [row0, row1) of d_buf of |X_row(k)|^2 (`splits` parts).
// (radix, tile width) pairs fft_power_pass_kernel is compiled for: what power2_plan's rule gives for H = 2^10 .. 2^22
#define POWER2_INSTANCES(X) X(5, 4) X(6, 4) X(7, 4) X(8, 3) X(9, 2) X(10, 1) X(11, 1)
namespace {
struct Power2Plan {
    int logRa, logRb, lc1, lc2;
    size_t lds1, lds2;
};

bool power2_plan(const mdhip_ctx *ctx, long long H, Power2Plan &pp)
{
    int logH = 0;
    while ((1LL << logH) < H) ++logH;
    if (logH < 10 || logH > 22) return false;
    pp.logRa = (logH + 1) / 2;
    pp.logRb = logH - pp.logRa;
    auto bytes = [](int logR, int lc) {
        const int lf = net8_plan(logR).lf;
        return ((((size_t)1 << logR) + ((size_t)1 << logR >> lf)) << lc) * sizeof(double2) + ((size_t)1 << logR >> 1) * sizeof(double2);
    };
    // tiles of <= 40 KB: four workgroups per CU, whose loads, transforms and sums overlap; 16 columns (256-byte runs) at most,
    // R C / 8 lanes between 64 and 1024
    auto pick = [&](int logR, int &lc, size_t &lds) {
        lc = 4;
        while (lc > 1 && (bytes(logR, lc) > (size_t)40 * 1024 || ((1 << logR << lc) >> 3) > 512)) --lc;
        lds = bytes(logR, lc);
        return lds <= ctx->lds_max && ((1 << logR << lc) >> 3) >= 64 && ((1 << logR << lc) >> 3) <= 512;
    };
    if (!pick(pp.logRa, pp.lc1, pp.lds1) || !pick(pp.logRb, pp.lc2, pp.lds2)) return false;
    bool have = false;
#define MD_HAVE(LR, LC) have = have || (pp.logRb == LR && pp.lc2 == LC);
    POWER2_INSTANCES(MD_HAVE)
#undef MD_HAVE
    return have;
}
}  // namespace

bool mdhip_fft_power2_plan(const mdhip_ctx *ctx, long long L)
{
    Power2Plan pp;
    return L >= 512 && (L & (L - 1)) == 0 && power2_plan(ctx, L / 2, pp);
}

int mdhip_fft_first_perm(mdhip_ctx *ctx, const double *d_series, long long n, double2 *d_buf, long long L, int batch)
{
    const long long H = L / 2;
    Power2Plan pp;
    MD_REQUIRE((L & (L - 1)) == 0 && n <= L && power2_plan(ctx, H, pp), "no two-pass plan for the transform length %lld", L);
    TwTab tt;
    if (!fft_twiddle_table(ctx, H, tt)) return MDHIP_ENOMEM;
    MD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fft_pass8_kernel<IN_PAD, OUT_PERM>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)pp.lds1));
    const long long cols = H >> pp.logRa;
    const int threads = (1 << pp.logRa << pp.lc1) >> 3;
    for (int b0 = 0; b0 < batch; b0 += FFT_MAX_BATCH) {  // (grid.y)
        PassIo io{};
        io.series = d_series + (size_t)b0 * n;
        io.n = n;
        hipLaunchKernelGGL((fft_pass8_kernel<IN_PAD, OUT_PERM>), dim3((unsigned)(cols >> pp.lc1), (unsigned)std::min(FFT_MAX_BATCH, batch - b0)),
                           dim3((unsigned)threads), pp.lds1, ctx->stream, (const double2 *)nullptr, d_buf + (size_t)b0 * H, H,
                           pp.logRa, pp.lc1, 0, H, io, tt);
    }
    MD_HIP(hipGetLastError());
    return MDHIP_OK;
}

int mdhip_fft_power_pass(mdhip_ctx *ctx, const double2 *d_buf, long long L, long long row0, long long row1, int splits,
                         double *d_partial)
{
    const long long H = L / 2;
    Power2Plan pp;
    MD_REQUIRE((L & (L - 1)) == 0 && power2_plan(ctx, H, pp), "no two-pass plan for the transform length %lld", L);
    TwTab tt;
    if (!fft_twiddle_table(ctx, H, tt)) return MDHIP_ENOMEM;
    const int threads = (1 << pp.logRb << pp.lc2) >> 3;
#define MD_POWER_PASS(LR, LC)                                                                                                 \
    if (pp.logRb == LR && pp.lc2 == LC) {                                                                                     \
        MD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fft_power_pass_kernel<LR, LC>),                             \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)pp.lds2));                                \
        hipLaunchKernelGGL((fft_power_pass_kernel<LR, LC>), dim3((unsigned)((1LL << pp.logRa) >> pp.lc2), (unsigned)splits),  \
                           dim3((unsigned)threads), pp.lds2, ctx->stream, d_buf, H, pp.logRa, row0, row1, d_partial, tt);     \
        MD_HIP(hipGetLastError());                                                                                            \
        return MDHIP_OK;                                                                                                      \
    }
    POWER2_INSTANCES(MD_POWER_PASS)
#undef MD_POWER_PASS
    mdhip_fail(ctx, MDHIP_EHIP, "internal: no instance of the fused pass for radix 2^%d, 2^%d columns", pp.logRb, pp.lc2);
    return MDHIP_EHIP;
}

// Unnormalised inverse: c[b][t] = sum_{k=0}^{L-1} S[b][k] e^{+2 pi i k t/L} (S Hermitian, given for k = 0..L/2) into
// d_real [batch][L]. d_spec is left as it is; d_tmp holds batch * L/2 complex points.
int mdhip_fft_c2r(mdhip_ctx *ctx, const double2 *d_spec, double2 *d_tmp, double *d_real, long long L, int batch)
{
    MD_REQUIRE(L >= 2 && (L & (L - 1)) == 0, "transform length %lld is not a power of two", L);
    const long long H = L / 2;
    if (batch > FFT_MAX_BATCH) {  // grid.y
        for (int b0 = 0; b0 < batch; b0 += FFT_MAX_BATCH) {
            const int rc = mdhip_fft_c2r(ctx, d_spec + (size_t)b0 * (H + 1), d_tmp + (size_t)b0 * H,
                                         d_real + (size_t)b0 * L, L, std::min(FFT_MAX_BATCH, batch - b0));
            if (rc) return rc;
        }
        return MDHIP_OK;
    }
    const PassPlan p = plan_passes(ctx, H, batch);
    MD_REQUIRE(p.n_pass >= 0, "transform length %lld needs more than %d passes", L, FFT_MAX_PASSES);
    // the result must land in d_real: start in d_real for an even number of buffer hops, in d_tmp for an odd one
    // (H = 1: one hop, the conjugating copy)
    const int hops = p.n_pass == 0 ? 1 : p.n_pass;
    double2 *real_c = reinterpret_cast<double2 *>(d_real);
    double2 *first = hops % 2 == 0 ? real_c : d_tmp, *second = hops % 2 == 0 ? d_tmp : real_c;
    TwTab tt;
    if (!fft_twiddle_table(ctx, H, tt)) return MDHIP_ENOMEM;
    hipLaunchKernelGGL(c2r_pre_kernel, dim3((unsigned)((H / 2 + 1 + 255) / 256), (unsigned)batch), dim3(256), 0,
                       ctx->stream, d_spec, first, H, tt);
    // W = conj Y went in; y = conj FFT(W): conjugate on the way out. c[2j] = Re y[j], c[2j+1] = Im y[j]: the complex
    // result read as reals IS the series.
    double2 *res = fft_forward(ctx, first, second, H, batch, IN_PLAIN, OUT_CONJ, PassIo{}, tt);
    if (!res) return MDHIP_EHIP;
    MD_HIP(hipGetLastError());
    if (res != real_c) return mdhip_fail(ctx, MDHIP_EHIP, "internal: inverse transform landed in the wrong buffer");
    return MDHIP_OK;
}

// The whole FFT estimator of `batch` series pairs (xcorr.hip): lags[b][t] = sum_u a[b][u+t] b[b][u] / (n - t),
// t < n_lags, through L-point transforms (L a power of two >= 2n, n >= 2). Five kinds of launches: the forward passes
// of a (the first reads the series and pads on the fly), the same for b unless d_b == d_a, one pointwise kernel
// (half spectra, product, inverse-transform input), the inverse passes (the last writes the scaled lags).
// buf0..buf3 hold batch * L/2 complex points each (buf2, buf3 unused for an autocorrelation).
int mdhip_fft_xcorr(mdhip_ctx *ctx, const double *d_a, const double *d_b, long long n, long long L, int batch,
                    double2 *buf0, double2 *buf1, double2 *buf2, double2 *buf3, long long n_lags, double *d_lags,
                    double out_scale)
{
    MD_REQUIRE(L >= 4 && (L & (L - 1)) == 0 && L >= 2 * n, "bad transform length %lld for %lld samples", L, n);
    MD_REQUIRE(batch <= 65535, "more than 65535 series per launch");
    const long long H = L / 2;
    PassIo io{};
    io.n = n;
    io.series = d_a;
    TwTab tt;
    if (!fft_twiddle_table(ctx, H, tt)) return MDHIP_ENOMEM;
    const bool same = d_a == d_b;
    {
        // Autocorrelation in THREE launches where the transform runs in two passes of the radix-8 network (round 5,
        // fft_mid_acf_kernel): first pass (pairs next to each other) | second pass + spectrum + inverse's first | last
        const PassPlan fp = plan_passes(ctx, H, batch);
        int lc1 = 0, lcm = 0;
        size_t lds1 = 0, ldsm = 0;
        if (same && ctx->opt_fft_mid != 0 && fp.n_pass == 2 && net8_tile(ctx, H, fp.logR[0], lc1, lds1) &&
            net8_tile(ctx, H, fp.logR[1], lcm, ldsm) && lcm >= 1) {
            const int logRa = fp.logR[0], logRb = fp.logR[1];
            if (ctx->opt_fft_mid == 2 && lcm > 1) {  // (A/B: tiles half as wide, more workgroups per CU)
                --lcm;
                ldsm = ((((size_t)1 << logRb) + ((size_t)1 << logRb >> net8_plan(logRb).lf)) << lcm) * sizeof(double2) +
                       ((size_t)1 << logRb >> 1) * sizeof(double2);
            }
            if (!launch_pass<IN_PAD, OUT_PERM>(ctx, buf0, buf1, H, batch, logRa, 0, io, tt)) return MDHIP_EHIP;
            const int threads = (int)std::min<long long>(1024, std::max<long long>(64, ((1LL << logRb) << lcm) >> 3));
            MD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fft_mid_acf_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsm));
            hipLaunchKernelGGL(fft_mid_acf_kernel, dim3((unsigned)((1LL << logRa) >> lcm), (unsigned)batch),
                               dim3((unsigned)threads), ldsm, ctx->stream, buf1, buf0, H, logRa, logRb, lcm, tt);
            io.lags = d_lags;
            io.n_lags = n_lags;
            io.L = (double)L;
            io.scale = out_scale;
            io.zb = nullptr;
            if (!launch_pass<IN_PLAIN, OUT_LAGS>(ctx, buf0, buf1, H, batch, logRa, logRb, io, tt)) return MDHIP_EHIP;
            MD_HIP(hipGetLastError());
            return MDHIP_OK;
        }
    }
    double2 *Za = fft_forward(ctx, buf0, buf1, H, batch, IN_PAD, OUT_PLAIN, io, tt);
    if (!Za) return MDHIP_EHIP;
    double2 *Zb = Za;
    if (!same) {
        io.series = d_b;
        Zb = fft_forward(ctx, buf2, buf3, H, batch, IN_PAD, OUT_PLAIN, io, tt);
        if (!Zb) return MDHIP_EHIP;
    }
    // the pointwise step (half spectra, product, inverse-transform input) rides on the inverse transform's first pass when
    // that pass runs the radix-8 network (IN_SPEC); else it is a kernel of its own, in place
    const PassPlan ip = plan_passes(ctx, H, batch);
    int lc_;
    size_t lds_;
    const bool fuse = ctx->opt_fft_specfuse != 0 && ip.n_pass >= 1 && net8_tile(ctx, H, ip.logR[0], lc_, lds_);
    if (!fuse) {
        const dim3 grid((unsigned)((H / 2 + 1 + 255) / 256), (unsigned)batch);
        if (same)
            hipLaunchKernelGGL(xcorr_spectrum_kernel<true>, grid, dim3(256), 0, ctx->stream, Za, Zb, H, tt);
        else
            hipLaunchKernelGGL(xcorr_spectrum_kernel<false>, grid, dim3(256), 0, ctx->stream, Za, Zb, H, tt);
    }
    io.lags = d_lags;
    io.n_lags = n_lags;
    io.L = (double)L;
    io.scale = out_scale;
    io.zb = same ? nullptr : Zb;
    if (!fft_forward(ctx, Za, Za == buf0 ? buf1 : buf0, H, batch, fuse ? IN_SPEC : IN_PLAIN, OUT_LAGS, io, tt)) return MDHIP_EHIP;
    MD_HIP(hipGetLastError());
    return MDHIP_OK;
}
