// msd_fft_w12r.h — included by msd_fft.hip inside its anonymous namespace, behind msd_fft_w12.h.
//
// Round 6: the power spectra of series LONGER than the fused kernels hold (F + max_lag > 16 384), without transform passes
// through HBM. The padded length is L' = D x 6144 (D = 4: 24 576 >= F + max_lag, F <= 12 288 — BASELINE's C4 shape at 10 000
// frames), and the spectrum is computed one RESIDUE CLASS of frequencies at a time:
//     X[D j + r] = sum_{n < 6144} y_r[n] w_6144^(n j),     y_r[n] = w_L'^(r n) sum_q x[n + 6144 q] w_D^(r q)
// — a 6144-point complex transform per class, exactly the size msd_power_w12_kernel's machinery (twelve register-resident
// 512-point sub-transforms, one per wave) was built for. x is real, so |X[L' - k]|^2 = |X[k]|^2: the classes r = 0 .. D/2
// cover every frequency 0 .. L'/2 (r = 0 and r = D/2 twice: their upper halves are dropped). Nothing is untangled and no
// wave reads another's results: the sums are |Y_r[j]|^2 in the lane that holds Y_r[j].
// Per series and class, three block barriers (msd_power_w12_kernel's structure):
//   inputs     all 12 waves, 8 points per lane: y_r[n] from the lane's samples x[n + 6144 q], n = tid + 768 i (loaded once
//              per series, coalesced from the time-major centred copy, kept in registers for every class), times the part of
//              the class twiddle that depends on e = n div 512 (w_L'^(512 r e), a table of 12 D-th roots) -> region e, position j;
//   -- barrier --
//   head       waves 0 .. 7, lane = position j < 512: the radix-12 butterfly over the regions, in place;
//   -- barrier --
//   wave d     msd_power_w12_kernel's three register passes over its region; the twiddles of the head (w_6144^(j d)) AND the rest of
//              the class twiddle (w_L'^(r j)) ride in its two factors: the per-register one from a table [class][d][n2], the
//              per-lane one folded into the first pass's twiddle chain; |Y|^2 into the class's 8 accumulators;
//   -- barrier --
// The frequencies are sorted out once per block (Ppart [rows][L'/2 + 1]); msd_residue_inverse_kernel turns the folded spectra
// into correlations by direct summation in double-double (a handful of segments: 12 288 terms per lag).
//
// What this file holds, in the order of the text (DESIGN.md 4.4b has the measurements):
//   msd_power_w12r_kernel<4>     the first form described above: classes 0, 1, 2 (option lag_residue 2, kept for A/B);
//   msd_power_w12p_kernel<SHARE> the default for F <= 12 288: TWO transforms per series — the even frequencies as
//                                msd_power_w12_kernel's packed transform (bilinear sums, partner phase), the odd ones as class 1;
//   msd_power_w12o_kernel<R>     12 288 < F <= 24 576 (L' = 8 x 6144, the series folded once by the transposition): the odd
//                                frequencies, classes 1 and 3 (mod 8); the even ones are msd_power_w12p_kernel<false> over the folded rows;
//   msd_power_w1_kernel<D2>      F <= 1536: the 512-point sub-transform as the whole transform, one wave per series;
//   msd_residue_inverse_kernel   the correlations of a non-power-of-two length, directly.

#ifndef W12R_EXP
#define W12R_EXP 0  // timing experiments only (WRONG results): 1 no input stage, 2 no head, 4 no register passes
#endif

template <int K, int DD>
__device__ __forceinline__ Cx w12r_root_mul(double a)  // a * w_DD^K (a real), DD = 4 or 8
{
    constexpr int k = ((K % DD) + DD) % DD;
    constexpr double H = 0.70710678118654752440;
    if constexpr (DD == 4) {
        if constexpr (k == 0) return {a, 0.0};
        else if constexpr (k == 1) return {0.0, -a};
        else if constexpr (k == 2) return {-a, 0.0};
        else return {0.0, a};
    } else {
        if constexpr (k == 0) return {a, 0.0};
        else if constexpr (k == 1) return {a * H, -a * H};
        else if constexpr (k == 2) return {0.0, -a};
        else if constexpr (k == 3) return {-a * H, -a * H};
        else if constexpr (k == 4) return {-a, 0.0};
        else if constexpr (k == 5) return {-a * H, a * H};
        else if constexpr (k == 6) return {0.0, a};
        else return {a * H, a * H};
    }
}

template <int K, int DD>
__device__ __forceinline__ Cx w12r_root_mul_d(double a)  // a * w_DD^K (a real), DD = 2, 4 or 6
{
    if constexpr (DD == 4) return w12r_root_mul<K, 4>(a);
    else if constexpr (DD == 2) return (K & 1) ? Cx{-a, 0.0} : Cx{a, 0.0};
    else {
        constexpr int k = ((K % 6) + 6) % 6;
        constexpr double S60 = 0.86602540378443864676;
        if constexpr (k == 0) return {a, 0.0};
        else if constexpr (k == 1) return {0.5 * a, -S60 * a};
        else if constexpr (k == 2) return {-0.5 * a, -S60 * a};
        else if constexpr (k == 3) return {-a, 0.0};
        else if constexpr (k == 4) return {-0.5 * a, S60 * a};
        else return {0.5 * a, S60 * a};
    }
}

// The 3-point DFT (w_3 = e^{-2 pi i / 3})
__device__ __forceinline__ void w12r_dft3(Cx a, Cx b, Cx c, Cx &y0, Cx &y1, Cx &y2)
{
    constexpr double S60 = 0.86602540378443864676;
    const Cx t1 = cx_add(b, c);
    const Cx t2 = {a.x - 0.5 * t1.x, a.y - 0.5 * t1.y};
    const Cx t3 = {S60 * (b.x - c.x), S60 * (b.y - c.y)};
    y0 = cx_add(a, t1);
    y1 = {t2.x + t3.y, t2.y - t3.x};
    y2 = {t2.x - t3.y, t2.y + t3.x};
}

// The full 12-point DFT of position j over the regions, in place, by the prime-factor map 12 = 3 x 4 (no twiddles between the
// two stages): input n = (4 n1 + 3 n2) mod 12, output k = (4 k1 + 9 k2) mod 12 — four 3-point and three 4-point transforms,
// ~100 additions and 8 multiplications (the three pruned radix-12 butterflies of msd_power_w12_kernel's head, unpruned: ~300).
__device__ __forceinline__ void w12r_dft12(double2 *col)  // col = R + j: element e at col[e * W12_RS]
{
    Cx t[3][4];
    {
        const Cx a = w12_ld(col), b = w12_ld(col + 4 * W12_RS), c = w12_ld(col + 8 * W12_RS);
        w12r_dft3(a, b, c, t[0][0], t[1][0], t[2][0]);
    }
    {
        const Cx a = w12_ld(col + 3 * W12_RS), b = w12_ld(col + 7 * W12_RS), c = w12_ld(col + 11 * W12_RS);
        w12r_dft3(a, b, c, t[0][1], t[1][1], t[2][1]);
    }
    {
        const Cx a = w12_ld(col + 6 * W12_RS), b = w12_ld(col + 10 * W12_RS), c = w12_ld(col + 2 * W12_RS);
        w12r_dft3(a, b, c, t[0][2], t[1][2], t[2][2]);
    }
    {
        const Cx a = w12_ld(col + 9 * W12_RS), b = w12_ld(col + 1 * W12_RS), c = w12_ld(col + 5 * W12_RS);
        w12r_dft3(a, b, c, t[0][3], t[1][3], t[2][3]);
    }
    Cx o[4];
    dft4(t[0][0], t[0][1], t[0][2], t[0][3], o[0], o[1], o[2], o[3]);  // k = 0, 9, 6, 3
    w12_st(col, o[0]);
    w12_st(col + 9 * W12_RS, o[1]);
    w12_st(col + 6 * W12_RS, o[2]);
    w12_st(col + 3 * W12_RS, o[3]);
    dft4(t[1][0], t[1][1], t[1][2], t[1][3], o[0], o[1], o[2], o[3]);  // k = 4, 1, 10, 7
    w12_st(col + 4 * W12_RS, o[0]);
    w12_st(col + 1 * W12_RS, o[1]);
    w12_st(col + 10 * W12_RS, o[2]);
    w12_st(col + 7 * W12_RS, o[3]);
    dft4(t[2][0], t[2][1], t[2][2], t[2][3], o[0], o[1], o[2], o[3]);  // k = 8, 5, 2, 11
    w12_st(col + 8 * W12_RS, o[0]);
    w12_st(col + 5 * W12_RS, o[1]);
    w12_st(col + 2 * W12_RS, o[2]);
    w12_st(col + 11 * W12_RS, o[3]);
}

// LDS: regions | two-level table of w_L' ([256] w^i, [L'/256] w^(256 i)) | class tables [NCLS][12][8] | w_512^lane [64] |
// w_64^(n0 k1) [8][9] | head roots w_(12 D)^m [12 D] | the sums of class 0 [8][768] (the registers hold the other classes')
inline size_t w12r_lds_bytes(int D)
{
    const int ncls = D / 2 + 1;
    return (size_t)W12_NW * W12_RS * 16 + (size_t)(256 + D * W12_N / 256) * 16 + (size_t)ncls * W12_NW * 8 * 16 + (64 + 72) * 16 +
           (size_t)12 * D * 16 + (size_t)8 * W12_THREADS * 8;
}

// x: the time-major centred copy [rows][F] of a batch of series (scaled); items: row ranges [c_lo, c_hi) of it, one block
// each; tab2: [256] w_L'^i | [L'/256] w_L'^(256 i); Ppart [items][L'/2 + 1].
template <int D>
__global__ __launch_bounds__(W12_THREADS) void msd_power_w12r_kernel(const double *__restrict__ x, int F,
                                                                     const FftItem *__restrict__ items,
                                                                     const double2 *__restrict__ tab2,
                                                                     double *__restrict__ Ppart)
{
    constexpr int N = W12_N, RS = W12_RS, NCLS = D / 2 + 1, NQ = D / 2, LP = D * N, NA = LP / 256;
    extern __shared__ double ft_lds[];
    double2 *R = reinterpret_cast<double2 *>(ft_lds);
    double2 *tB = R + W12_NW * RS, *tA = tB + 256;
    double2 *btab = tA + NA;                   // [cls][d][n2] = w_N^(64 d n2) w_L'^(64 cls n2)
    double2 *t1tab = btab + NCLS * W12_NW * 8;  // [lane] = w_512^lane
    double2 *t2tab = t1tab + 64;               // [9 n0 + k1] = w_64^(n0 k1)
    double2 *ctab = t2tab + 72;                // [m] = w_(12 D)^m = w_L'^(512 m)
    double *sacc0 = reinterpret_cast<double *>(ctab + 12 * D);  // [k0][tid]: class 0's sums (the kernel is short of registers)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 256 + NA; i += W12_THREADS) tB[i] = tab2[i];
    const FftItem it = items[blockIdx.x];
    __syncthreads();
    auto tw2 = [&](int k) {  // w_L'^k, 0 <= k < L'
        const double2 a = tA[k >> 8], b = tB[k & 255];
        return Cx{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
    };
    auto twn = [&](long long k) { return tw2((int)((k % (2 * N)) * (D / 2))); };  // w_(2 N)^k
    for (int i = tid; i < NCLS * W12_NW * 8; i += W12_THREADS) {
        const int cls = i / (W12_NW * 8), d = (i >> 3) % W12_NW, n2 = i & 7;
        const Cx w = cx_mul(twn(128LL * d * n2), tw2(64 * cls * n2));
        btab[i] = make_double2(w.x, w.y);
    }
    for (int i = tid; i < 64 + 72 + 12 * D; i += W12_THREADS) {
        Cx w;
        if (i < 64) w = twn(24LL * i);
        else if (i < 64 + 72) w = twn(192LL * ((i - 64) / 9) * ((i - 64) % 9));
        else w = tw2(512 * (i - 64 - 72));
        t1tab[i] = make_double2(w.x, w.y);
    }
    // per-lane factor of the twiddles that sit between the head and the first register pass: w_N^(lane d); the class's
    // w_L'^(cls lane) comes from the table where it is used (w_L'^i, i < 256)
    const Cx tw_a = twn(2LL * lane * wv);
    double sacc[NCLS - 1][8];  // classes 1 .. NCLS - 1
#pragma unroll
    for (int cls = 0; cls < NCLS - 1; ++cls)
#pragma unroll
        for (int i = 0; i < 8; ++i) sacc[cls][i] = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) sacc0[i * W12_THREADS + tid] = 0.0;
    // this lane's samples of the series: x[n + 6144 q], n = tid + 768 i (loaded once per series, used by every class)
    double xs[8][NQ];
    auto fetch = [&](long long c) {
        const double *row = x + (size_t)c * F;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int t = tid + W12_THREADS * i + N * q;
                xs[i][q] = t < F ? __builtin_nontemporal_load(row + t) : 0.0;
            }
    };
    double2 *myR = R + wv * RS;
    const int hj = (wv & 7) * 64 + lane;  // the head's position j (waves 0 .. 7)
    __syncthreads();
    if (it.c_lo < it.c_hi) fetch(it.c_lo);
    for (long long c = it.c_lo; c < it.c_hi; ++c) {
        const bool more = c + 1 < it.c_hi;
        // (the tables' addresses opaque per series: their entries are the same for every series, and hoisted out of the loop
        // they would take a hundred registers)
        int opq = 0;  // (an opaque ZERO added to the tables' addresses — not the pointers themselves through the asm: that
                      // loses their address space, and every table read becomes a flat load that waits for ALL memory counters,
                      // the prefetched samples included)
        asm volatile("" : "+v"(opq));
        const double2 *ctab_s = ctab + opq, *btab_s = btab + opq, *tB_s = tB + opq;
        auto one_class = [&](auto ck) {
            constexpr int cls = decltype(ck)::value;
            // ---- the class's inputs y_cls[n] (but for the factor w_L'^(cls j), which rides in the register passes) ----
#pragma unroll
            for (int i = 0; i < ((W12R_EXP & 1) ? 1 : 8); ++i) {
                const int n = tid + W12_THREADS * i, e = n >> 9, j = n & 511;
                // sum_q x[n + 6144 q] w_D^(cls q)
                Cx s = {xs[i][0], 0.0};
                if constexpr (NQ > 1) s = cx_add(s, w12r_root_mul<cls, D>(xs[i][1]));
                if constexpr (NQ > 2) s = cx_add(s, w12r_root_mul<2 * cls, D>(xs[i][2]));
                if constexpr (NQ > 3) s = cx_add(s, w12r_root_mul<3 * cls, D>(xs[i][3]));
                if constexpr (cls != 0) s = cx_mul(s, w12_ld(ctab_s + cls * e));
                w12_st(R + e * RS + j, s);
            }
            if constexpr (cls == NCLS - 1) {
                if (more) fetch(c + 1);  // (the samples are consumed: the next series' land under the passes)
            }
            __syncthreads();
            // ---- head: waves 0 .. 7, the radix-12 butterfly of position j in place ----
            if (wv < 8 && !(W12R_EXP & 2)) w12r_dft12(R + hj);
            __syncthreads();
            // ---- this wave's 512-point sub-transform, in registers (msd_power_w12_kernel) ----
            Cx a[8];
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) a[n2] = w12_ld(myR + lane + 64 * n2);
            if (!(W12R_EXP & 4)) {
            if (wv != 0 || cls != 0) {
#pragma unroll
                for (int n2 = 1; n2 < 8; ++n2) a[n2] = cx_mul(a[n2], w12_ld(btab_s + (cls * W12_NW + wv) * 8 + n2));
            }
            f2_bfly8(a, Cx{1.0, 0.0}, false);
            {
                const Cx tw_1 = w12_ld(t1tab + lane);
                Cx t = tw_a;
                if constexpr (cls != 0) t = cx_mul(t, w12_ld(tB_s + cls * lane));
                a[0] = cx_mul(a[0], t);
#pragma unroll
                for (int k2 = 1; k2 < 8; ++k2) {
                    t = cx_mul(t, tw_1);
                    a[k2] = cx_mul(a[k2], t);
                }
            }
            // exchange 1: (n0, n1 | k2) -> (n0, k2 | n1): point n0 + 8 k2 + 64 n1
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) w12_st(myR + (lane & 7) + 8 * k2 + 64 * (lane >> 3), a[k2]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) a[n1] = w12_ld(myR + lane + 64 * n1);
            f2_bfly8(a, Cx{1.0, 0.0}, false);
            {
                const double2 *t2 = t2tab + 9 * (lane & 7);
#pragma unroll
                for (int k1 = 1; k1 < 8; ++k1) a[k1] = cx_mul(a[k1], w12_ld(t2 + k1));
            }
            // exchange 2: (n0, k2 | k1) -> (k1, k2 | n0): point k1 + 8 k2 + 65 n0
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k1 = 0; k1 < 8; ++k1) w12_st(myR + k1 + 8 * (lane >> 3) + 65 * (lane & 7), a[k1]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int n0 = 0; n0 < 8; ++n0) a[n0] = w12_ld(myR + lane + 65 * n0);
            f2_bfly8(a, Cx{1.0, 0.0}, false);
            }
            // a[k0] = Y_cls at j = d + 12 (k2 + 8 k1 + 64 k0), lane = k1 + 8 k2
            if constexpr (cls == 0) {
#pragma unroll
                for (int k0 = 0; k0 < 8; ++k0) {
                    double v = sacc0[k0 * W12_THREADS + tid];
                    v = __builtin_fma(a[k0].x, a[k0].x, v);
                    v = __builtin_fma(a[k0].y, a[k0].y, v);
                    sacc0[k0 * W12_THREADS + tid] = v;
                }
            } else {
#pragma unroll
                for (int k0 = 0; k0 < 8; ++k0) {
                    sacc[cls - 1][k0] = __builtin_fma(a[k0].x, a[k0].x, sacc[cls - 1][k0]);
                    sacc[cls - 1][k0] = __builtin_fma(a[k0].y, a[k0].y, sacc[cls - 1][k0]);
                }
            }
            __syncthreads();
        };
        one_class(std::integral_constant<int, 0>());
        one_class(std::integral_constant<int, 1>());
        one_class(std::integral_constant<int, 2>());
        if constexpr (NCLS > 3) {
            one_class(std::integral_constant<int, 3>());
            one_class(std::integral_constant<int, (NCLS > 3 ? 4 : 0)>());
        }
    }
    // frequencies: class r, j = d + 12 k' -> k = D j + r, or its mirror L' - k (r = 0 and r = D/2 hold both: the upper one is dropped)
    double *pp = Ppart + (size_t)it.row * (LP / 2 + 1);
#pragma unroll
    for (int cls = 0; cls < NCLS; ++cls) {
#pragma unroll
        for (int k0 = 0; k0 < 8; ++k0) {
            const int kp = (lane >> 3) + 8 * (lane & 7) + 64 * k0;  // k' = k2 + 8 k1 + 64 k0
            const int k = D * (wv + W12_NW * kp) + cls;
            const double v = cls == 0 ? sacc0[k0 * W12_THREADS + tid] : sacc[cls == 0 ? 0 : cls - 1][k0];
            if (k <= LP / 2) pp[k] = v;
            else if (cls != 0 && cls != D / 2) pp[LP - k] = v;
        }
    }
}

// The three register passes of one wave over its region (msd_power_w12_kernel's, verbatim): a[k0] = the transform at
// j = d + 12 (k2 + 8 k1 + 64 k0), lane = k1 + 8 k2. brow: the per-register factors [8] applied in front (nullptr: none);
// t: the per-lane factor folded into the first pass's twiddle chain.
__device__ __forceinline__ void w12r_passes(Cx *a, double2 *myR, int lane, const double2 *brow, Cx t, const double2 *t1tab,
                                            const double2 *t2tab, bool from_region = true)
{
    if (from_region) {  // (false: the caller has put point lane + 64 n2 into a[n2] itself)
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) a[n2] = w12_ld(myR + lane + 64 * n2);
    }
    if (brow) {
#pragma unroll
        for (int n2 = 1; n2 < 8; ++n2) a[n2] = cx_mul(a[n2], w12_ld(brow + n2));
    }
    f2_bfly8(a, Cx{1.0, 0.0}, false);
    {
        const Cx tw_1 = w12_ld(t1tab + lane);
        a[0] = cx_mul(a[0], t);
#pragma unroll
        for (int k2 = 1; k2 < 8; ++k2) {
            t = cx_mul(t, tw_1);
            a[k2] = cx_mul(a[k2], t);
        }
    }
    // exchange 1: (n0, n1 | k2) -> (n0, k2 | n1): point n0 + 8 k2 + 64 n1
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) w12_st(myR + (lane & 7) + 8 * k2 + 64 * (lane >> 3), a[k2]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) a[n1] = w12_ld(myR + lane + 64 * n1);
    f2_bfly8(a, Cx{1.0, 0.0}, false);
    {
        const double2 *t2 = t2tab + 9 * (lane & 7);
#pragma unroll
        for (int k1 = 1; k1 < 8; ++k1) a[k1] = cx_mul(a[k1], w12_ld(t2 + k1));
    }
    // exchange 2: (n0, k2 | k1) -> (k1, k2 | n0): point k1 + 8 k2 + 65 n0
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) w12_st(myR + k1 + 8 * (lane >> 3) + 65 * (lane & 7), a[k1]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n0 = 0; n0 < 8; ++n0) a[n0] = w12_ld(myR + lane + 65 * n0);
    f2_bfly8(a, Cx{1.0, 0.0}, false);
}

// Second form (the default): TWO transforms per series instead of three. The even frequencies k = 2 k'' are the real transform
// of length 12 288 of the series itself (F <= 12 288: nothing folds), i.e. msd_power_w12_kernel's packed transform z[m] = x[2 m] +
// i x[2 m + 1] with its bilinear sums S, S', T and partner phase; the odd ones are class r = 1 above (r = 3 is its mirror).
// LDS: regions | w_L' tables | btab [2][12][8] | w_512^lane | w_64^(n0 k1) | w_L'^(512 e) [12] | the odd class's sums [8][768].
inline size_t w12p_lds_bytes()
{
    return (size_t)W12_NW * W12_RS * 16 + (size_t)(256 + 4 * W12_N / 256) * 16 + (size_t)2 * W12_NW * 8 * 16 + (64 + 72) * 16 +
           (size_t)12 * 16 + (size_t)8 * W12_THREADS * 8;
}

// x: rows of `stride` doubles. Padded length 24 576 (F <= 12 288): the centred series itself, stride = Fp = Fo = F, off_odd = 0,
// kmul = 1, prow = 12 289. Padded length 49 152 (F <= 24 576; the EVEN frequencies of that length, msd_power_w12o_kernel
// makes the odd ones): rows [g | h] of 12 288 samples each, g[n] = x[n] + x[n + 12288], h[n] = x[n] - x[n + 12288]
// (transpose_fold64_sq_kernel) — the 24 576-point transform of the series folded once is the packed transform of g and class
// 1 of h —, stride = 24 576, Fp = Fo = off_odd = 12 288, kmul = 2, prow = 24 577.
// SHARE (the D = 4 use): both transforms of a series from ONE set of samples. The packed transform's lane holds x[2 m],
// x[2 m + 1] for m = tid + 768 i, i < 8 — and 6144 samples further is 4 i further in the same lane: the lane also holds the
// odd class's pairs (x[n], x[n + 6144]) for its own 8 points n = 2 (tid + 768 i) + p, i < 4, p < 2. The series is read once.
template <bool SHARE>
__global__ __launch_bounds__(W12_THREADS) void msd_power_w12p_kernel(const double *__restrict__ x, long long stride, int Fp,
                                                                     int off_odd, int Fo, int kmul, int prow,
                                                                     const FftItem *__restrict__ items,
                                                                     const double2 *__restrict__ tab2,
                                                                     double *__restrict__ Ppart)
{
    constexpr int D = 4, N = W12_N, RS = W12_RS, LP = D * N, NA = LP / 256;
    extern __shared__ double ft_lds[];
    double2 *R = reinterpret_cast<double2 *>(ft_lds);
    double2 *tB = R + W12_NW * RS, *tA = tB + 256;
    double2 *btab = tA + NA;                   // [cls][d][n2] = w_N^(64 d n2) w_L'^(64 cls n2), cls = 0 (packed), 1 (odd)
    double2 *t1tab = btab + 2 * W12_NW * 8;    // [lane] = w_512^lane
    double2 *t2tab = t1tab + 64;               // [9 n0 + k1] = w_64^(n0 k1)
    double2 *ctab = t2tab + 72;                // [e] = w_L'^(512 e)
    double *saccg = reinterpret_cast<double *>(ctab + 12);  // [k0][tid]: the odd class's sums
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 256 + NA; i += W12_THREADS) tB[i] = tab2[i];
    const FftItem it = items[blockIdx.x];
    __syncthreads();
    auto tw2 = [&](int k) {  // w_L'^k, 0 <= k < L'
        const double2 a = tA[k >> 8], b = tB[k & 255];
        return Cx{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
    };
    auto twn = [&](long long k) { return tw2((int)((k % (2 * N)) * (D / 2))); };  // w_(2 N)^k
    for (int i = tid; i < 2 * W12_NW * 8; i += W12_THREADS) {
        const int cls = i / (W12_NW * 8), d = (i >> 3) % W12_NW, n2 = i & 7;
        const Cx w = cx_mul(twn(128LL * d * n2), tw2(64 * cls * n2));
        btab[i] = make_double2(w.x, w.y);
    }
    for (int i = tid; i < 64 + 72 + 12; i += W12_THREADS) {
        Cx w;
        if (i < 64) w = twn(24LL * i);
        else if (i < 64 + 72) w = twn(192LL * ((i - 64) / 9) * ((i - 64) % 9));
        else w = tw2(512 * (i - 64 - 72));
        t1tab[i] = make_double2(w.x, w.y);
    }
    const Cx tw_a = twn(2LL * lane * wv);  // w_N^(lane d)
    // the partner of frequency j = d + 12 k' of the packed transform (msd_power_w12_kernel): wave 12 - d, lane 63 - lane,
    // register 7 - k0 for d > 0; wave 0 pairs inside itself, lane 0 of it inside its own registers
    const int pw = wv == 0 ? 0 : W12_NW - wv;
    int plane = 63 - lane;
    if (wv == 0) {
        const int mneg = (64 - ((lane >> 3) + 8 * (lane & 7))) & 63;
        plane = (mneg >> 3) + 8 * (mneg & 7);
    }
    const bool self0 = wv == 0 && lane == 0;
    double sacc[8], tacc[5];
#pragma unroll
    for (int i = 0; i < 8; ++i) sacc[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 5; ++i) tacc[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) saccg[i * W12_THREADS + tid] = 0.0;
    // this lane's samples, one set at a time: the packed transform's x[2 m], x[2 m + 1] (m = tid + 768 i), then the odd class's
    // x[n], x[n + 6144] (n = tid + 768 i); each set is requested when the other has been consumed and lands under its passes
    double xs[8][2];
    auto fetch_packed = [&](long long c) {
        const double *row = x + (size_t)c * (size_t)stride;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int t = 2 * (tid + W12_THREADS * i);
            xs[i][0] = t < Fp ? __builtin_nontemporal_load(row + t) : 0.0;
            xs[i][1] = t + 1 < Fp ? __builtin_nontemporal_load(row + t + 1) : 0.0;
        }
    };
    auto fetch_odd = [&](long long c) {
        const double *row = x + (size_t)c * (size_t)stride + off_odd;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int t = tid + W12_THREADS * i;
            xs[i][0] = t < Fo ? __builtin_nontemporal_load(row + t) : 0.0;
            xs[i][1] = t + N < Fo ? __builtin_nontemporal_load(row + t + N) : 0.0;
        }
    };
    double2 *myR = R + wv * RS;
    const double2 *pR = R + pw * RS;
    const int hj = (wv & 7) * 64 + lane;  // the head's position j (waves 0 .. 7)
    __syncthreads();
    if (it.c_lo < it.c_hi) fetch_packed(it.c_lo);
    for (long long c = it.c_lo; c < it.c_hi; ++c) {
        const bool more = c + 1 < it.c_hi;
        // (the tables' addresses through an opaque copy per series: see msd_power_w12r_kernel)
        int opq = 0;  // (an opaque ZERO added to the tables' addresses — not the pointers themselves through the asm: that
                      // loses their address space, and every table read becomes a flat load that waits for ALL memory counters,
                      // the prefetched samples included)
        asm volatile("" : "+v"(opq));
        const double2 *ctab_s = ctab + opq, *btab_s = btab + opq, *tB_s = tB + opq;
        // ======== the even frequencies: the packed transform ========
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = tid + W12_THREADS * i;
            w12_st(R + (m >> 9) * RS + (m & 511), Cx{xs[i][0], xs[i][1]});
        }
        if constexpr (!SHARE) fetch_odd(c);
        __syncthreads();
        if (wv < 8) w12r_dft12(R + hj);
        __syncthreads();
        Cx a[8];
        w12r_passes(a, myR, lane, wv != 0 ? btab_s + wv * 8 : nullptr, tw_a, t1tab, t2tab);
#pragma unroll
        for (int k0 = 0; k0 < 8; ++k0) {
            sacc[k0] = __builtin_fma(a[k0].x, a[k0].x, sacc[k0]);
            sacc[k0] = __builtin_fma(a[k0].y, a[k0].y, sacc[k0]);
        }
        // registers 4 .. 7 are what the partner reads (its registers 3 .. 0)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k0 = 4; k0 < 8; ++k0) w12_st(myR + lane + 64 * (k0 - 4), a[k0]);
        if (self0) {
            // lane 0 of wave 0: frequencies 12 * 64 k0, pairs (k0, 8 - k0): 0 and 4 with themselves
            tacc[0] = __builtin_fma(2.0 * a[0].x, a[0].y, tacc[0]);
            tacc[4] = __builtin_fma(2.0 * a[4].x, a[4].y, tacc[4]);
#pragma unroll
            for (int u = 1; u < 4; ++u) {
                tacc[u] = __builtin_fma(a[u].x, a[8 - u].y, tacc[u]);
                tacc[u] = __builtin_fma(a[u].y, a[8 - u].x, tacc[u]);
            }
        }
        __syncthreads();
        if (!self0) {
            Cx pz[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) pz[u] = w12_ld(pR + plane + 64 * (3 - u));  // the partner's register 7 - u
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                tacc[u] = __builtin_fma(a[u].x, pz[u].y, tacc[u]);
                tacc[u] = __builtin_fma(a[u].y, pz[u].x, tacc[u]);
            }
        }
        __syncthreads();
        // ======== the odd frequencies: class r = 1 ========
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // x[n] + w_4 x[n + 6144] = x[n] - i x[n + 6144], times w_L'^(512 e) (w_L'^j rides in the register passes)
            const int n = SHARE ? 2 * (tid + W12_THREADS * (i >> 1)) + (i & 1) : tid + W12_THREADS * i, e = n >> 9;
            const Cx xv = SHARE ? Cx{xs[i >> 1][i & 1], -xs[(i >> 1) + 4][i & 1]} : Cx{xs[i][0], -xs[i][1]};
            const Cx sv = cx_mul(xv, w12_ld(ctab_s + e));
            w12_st(R + e * RS + (n & 511), sv);
        }
        if (more) fetch_packed(c + 1);
        __syncthreads();
        if (wv < 8) w12r_dft12(R + hj);
        __syncthreads();
        w12r_passes(a, myR, lane, btab_s + (W12_NW + wv) * 8, cx_mul(tw_a, w12_ld(tB_s + lane)), t1tab, t2tab);
#pragma unroll
        for (int k0 = 0; k0 < 8; ++k0) {
            double v = saccg[k0 * W12_THREADS + tid];
            v = __builtin_fma(a[k0].x, a[k0].x, v);
            v = __builtin_fma(a[k0].y, a[k0].y, v);
            saccg[k0 * W12_THREADS + tid] = v;
        }
        __syncthreads();
    }
    double *pp = Ppart + (size_t)it.row * (size_t)prow;
    // odd frequencies: j = d + 12 k' -> k = 4 j + 1, or its mirror L' - k
#pragma unroll
    for (int k0 = 0; k0 < 8; ++k0) {
        const int kp = (lane >> 3) + 8 * (lane & 7) + 64 * k0;  // k' = k2 + 8 k1 + 64 k0
        const int k = D * (wv + W12_NW * kp) + 1;
        pp[kmul * (k <= LP / 2 ? k : LP - k)] = saccg[k0 * W12_THREADS + tid];
    }
    // even frequencies, as msd_power_w12_kernel sorts them out (the loop ended on a barrier): point lane + 64 k0 of the wave's
    // region = {S, T}; |X_(2 k'')|^2 = (S + S')/2 + Im(w) (S - S')/2 + Re(w) T, w = e^{-2 pi i k''/12288}
#pragma unroll
    for (int k0 = 0; k0 < 8; ++k0) myR[lane + 64 * k0] = make_double2(sacc[k0], k0 < 4 ? tacc[k0] : 0.0);
    __syncthreads();
#pragma unroll
    for (int k0 = 0; k0 < 8; ++k0) {
        const int kp = (lane >> 3) + 8 * (lane & 7) + 64 * k0;
        const int k = wv + W12_NW * kp;
        double sn, tk;
        if (self0) {
            const int kq = (8 - k0) & 7;
            sn = myR[64 * kq].x;  // (lane 0: its own registers, through the region)
            tk = k0 == 0 ? tacc[0] : k0 == 4 ? tacc[4] : k0 < 4 ? tacc[k0] : tacc[8 - k0];
        } else {
            const double2 pv = pR[plane + 64 * (7 - k0)];
            sn = pv.x;
            tk = k0 < 4 ? tacc[k0] : pv.y;
        }
        const double sk = sacc[k0];
        const Cx w = twn(k);  // (cos, -sin) of 2 pi k / 12288
        pp[kmul * 2 * k] = 0.5 * (sk + sn) + w.y * (0.5 * (sk - sn)) + w.x * tk;
        if (k == 0) pp[kmul * 2 * N] = sk - tk;
    }
}

// The ODD frequencies of the padded length L'' = 8 x 6144 = 49 152 (12 288 < F <= 24 576): class R = 1 or 3 (mod 8; 7 and 5 are
// their mirrors) of the series x[n] = (g[n] + h[n]) / 2, x[n + 12288] = (g[n] - h[n]) / 2 that transpose_fold64_sq_kernel
// left folded (rows [g | h], see msd_power_w12p_kernel, which makes the even frequencies from the same rows):
//     y_R[n] = w_L''^(R n) sum_{q < 4} x[n + 6144 q] w_8^(R q),  n < 6144;   X[8 j + R] = FFT_6144(y_R)[j]
// One class per launch: a lane's 8 points take 32 samples, which fill the registers the other kernels keep sums in — the
// sums live in LDS here. LDS: regions | w_L'' tables | btab [12][8] | w_512^lane | w_64^(n0 k1) | w_L''^(512 R e) [12] | sums.
inline size_t w12o_lds_bytes()
{
    return (size_t)W12_NW * W12_RS * 16 + (size_t)(256 + 8 * W12_N / 256) * 16 + (size_t)W12_NW * 8 * 16 + (64 + 72) * 16 +
           (size_t)12 * 16 + (size_t)8 * W12_THREADS * 8;
}

template <int RC>
__global__ __launch_bounds__(W12_THREADS) void msd_power_w12o_kernel(const double *__restrict__ x,
                                                                     const FftItem *__restrict__ items,
                                                                     const double2 *__restrict__ tab2,
                                                                     double *__restrict__ Ppart)
{
    constexpr int D = 8, N = W12_N, RS = W12_RS, LP = D * N, NA = LP / 256, G2 = 2 * N;  // G2: samples of g (and of h) per row
    extern __shared__ double ft_lds[];
    double2 *R = reinterpret_cast<double2 *>(ft_lds);
    double2 *tB = R + W12_NW * RS, *tA = tB + 256;
    double2 *btab = tA + NA;              // [d][n2] = w_N^(64 d n2) w_L''^(64 RC n2)
    double2 *t1tab = btab + W12_NW * 8;   // [lane] = w_512^lane
    double2 *t2tab = t1tab + 64;          // [9 n0 + k1] = w_64^(n0 k1)
    double2 *ctab = t2tab + 72;           // [e] = w_L''^(512 RC e)
    double *saccg = reinterpret_cast<double *>(ctab + 12);  // [k0][tid]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 256 + NA; i += W12_THREADS) tB[i] = tab2[i];
    const FftItem it = items[blockIdx.x];
    __syncthreads();
    auto tw2 = [&](int k) {  // w_L''^k, 0 <= k < L''
        const double2 a = tA[k >> 8], b = tB[k & 255];
        return Cx{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
    };
    auto twn = [&](long long k) { return tw2((int)((k % (2 * N)) * (D / 2))); };  // w_(2 N)^k
    for (int i = tid; i < W12_NW * 8; i += W12_THREADS) {
        const int d = i >> 3, n2 = i & 7;
        const Cx w = cx_mul(twn(128LL * d * n2), tw2(64 * RC * n2));
        btab[i] = make_double2(w.x, w.y);
    }
    for (int i = tid; i < 64 + 72 + 12; i += W12_THREADS) {
        Cx w;
        if (i < 64) w = twn(24LL * i);
        else if (i < 64 + 72) w = twn(192LL * ((i - 64) / 9) * ((i - 64) % 9));
        else w = tw2(512 * RC * (i - 64 - 72));
        t1tab[i] = make_double2(w.x, w.y);
    }
    __syncthreads();
    const Cx tw_l = cx_mul(twn(2LL * lane * wv), w12_ld(tB + RC * lane));  // w_N^(lane d) w_L''^(RC lane)
#pragma unroll
    for (int i = 0; i < 8; ++i) saccg[i * W12_THREADS + tid] = 0.0;
    // the lane's samples: g[n], h[n], g[n + 6144], h[n + 6144], n = tid + 768 i
    double xs[8][4];
    auto fetch = [&](long long c) {
        const double *row = x + (size_t)c * (size_t)(2 * G2);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = tid + W12_THREADS * i;
            xs[i][0] = __builtin_nontemporal_load(row + n);
            xs[i][1] = __builtin_nontemporal_load(row + G2 + n);
            xs[i][2] = __builtin_nontemporal_load(row + n + N);
            xs[i][3] = __builtin_nontemporal_load(row + G2 + n + N);
        }
    };
    double2 *myR = R + wv * RS;
    const int hj = (wv & 7) * 64 + lane;
    if (it.c_lo < it.c_hi) fetch(it.c_lo);
    for (long long c = it.c_lo; c < it.c_hi; ++c) {
        const bool more = c + 1 < it.c_hi;
        int opq = 0;  // (see msd_power_w12r_kernel)
        asm volatile("" : "+v"(opq));
        const double2 *ctab_s = ctab + opq, *btab_s = btab + opq;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = tid + W12_THREADS * i, e = n >> 9;
            const double x0 = 0.5 * (xs[i][0] + xs[i][1]), x2 = 0.5 * (xs[i][0] - xs[i][1]);
            const double x1 = 0.5 * (xs[i][2] + xs[i][3]), x3 = 0.5 * (xs[i][2] - xs[i][3]);
            Cx sv = {x0, 0.0};
            sv = cx_add(sv, w12r_root_mul<RC, 8>(x1));
            sv = cx_add(sv, w12r_root_mul<2 * RC, 8>(x2));
            sv = cx_add(sv, w12r_root_mul<3 * RC, 8>(x3));
            sv = cx_mul(sv, w12_ld(ctab_s + e));
            w12_st(R + e * RS + (n & 511), sv);
        }
        if (more) fetch(c + 1);
        __syncthreads();
        if (wv < 8) w12r_dft12(R + hj);
        __syncthreads();
        Cx a[8];
        w12r_passes(a, myR, lane, btab_s + wv * 8, tw_l, t1tab, t2tab);
#pragma unroll
        for (int k0 = 0; k0 < 8; ++k0) {
            double v = saccg[k0 * W12_THREADS + tid];
            v = __builtin_fma(a[k0].x, a[k0].x, v);
            v = __builtin_fma(a[k0].y, a[k0].y, v);
            saccg[k0 * W12_THREADS + tid] = v;
        }
        __syncthreads();
    }
    double *pp = Ppart + (size_t)it.row * (LP / 2 + 1);
#pragma unroll
    for (int k0 = 0; k0 < 8; ++k0) {
        const int kp = (lane >> 3) + 8 * (lane & 7) + 64 * k0;
        const int k = D * (wv + W12_NW * kp) + RC;
        pp[k <= LP / 2 ? k : LP - k] = saccg[k0 * W12_THREADS + tid];
    }
}

// SHORT trajectories (F + max_lag <= 3072, F <= 1536 — below msd_power_w12_kernel's range, where rounds 2-5 ran block-wide
// transforms through LDS at one 8-wave block per CU: 2.1-3.9 ms per call at E = 50 000 whatever F): the same residue classes with
// the 512-point sub-transform as the WHOLE transform — padded length L' = D2 x 1024, D2 = 1 .. 3; the frequencies k = D2 k'' are
// the real transform of length 1024 of the series folded at 1024 (packed, 512 complex points, bilinear sums, the partner of
// k'' in the same wave: msd_power_w12_kernel's wave 0); the classes r = 1 .. D2 - 1 (mod D = 2 D2) are 512-point transforms of
// y_r[n] = w_L'^(r n) sum_q x[n + 512 q] w_D^(r q). One WAVE per series, twelve independent waves per block, no block
// barrier in the series loop; a wave's region (8.3 KB) is the medium of its exchanges only. Samples straight from the
// time-major centred copy (a series is <= 12 KB: the classes' re-reads hit L1 / L2).
// LDS: regions [12][520] | w_L' tables ([256] w^i, [L'/256] w^(256 i)) | class tables [D2][8] | w_512^lane | w_64^(n0 k1) | the
// sums of the last class [12][8][64] (registers hold the others').
inline size_t w1_lds_bytes(int D2)
{
    return (size_t)W12_NW * W12_RS * 16 + (size_t)(256 + D2 * 1024 / 256) * 16 + (size_t)D2 * 8 * 16 + (64 + 72) * 16 +
           (D2 >= 3 ? (size_t)W12_NW * 8 * 64 * 8 : 0);
}

// x: rows of F doubles; items: row ranges, one block each (wave w takes rows c_lo + w, + 12, ...); Ppart row it.row, L'/2 + 1
// doubles.
template <int D2>
__global__ __launch_bounds__(W12_THREADS) void msd_power_w1_kernel(const double *__restrict__ x, int F,
                                                                   const FftItem *__restrict__ items,
                                                                   const double2 *__restrict__ tab2,
                                                                   double *__restrict__ Ppart)
{
    constexpr int D = 2 * D2, N = 512, RS = W12_RS, LP = D2 * 1024, NA = LP / 256;
    extern __shared__ double ft_lds[];
    double2 *R = reinterpret_cast<double2 *>(ft_lds);
    double2 *tB = R + W12_NW * RS, *tA = tB + 256;
    double2 *btab = tA + NA;            // [r][n2] = w_L'^(64 r n2)
    double2 *t1tab = btab + D2 * 8;     // [lane] = w_512^lane
    double2 *t2tab = t1tab + 64;        // [9 n0 + k1] = w_64^(n0 k1)
    double *sacc_l = reinterpret_cast<double *>(t2tab + 72);  // D2 == 3: class 2's sums [wave][k0][lane]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 256 + NA; i += W12_THREADS) tB[i] = tab2[i];
    const FftItem it = items[blockIdx.x];
    __syncthreads();
    auto tw2 = [&](int k) {  // w_L'^k, 0 <= k < L'
        const double2 a = tA[k >> 8], b = tB[k & 255];
        return Cx{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
    };
    for (int i = tid; i < D2 * 8 + 64 + 72; i += W12_THREADS) {
        Cx w;
        if (i < D2 * 8) w = tw2((64 * (i >> 3) * (i & 7)) % LP);
        else if (i < D2 * 8 + 64) w = tw2(((i - D2 * 8) * D) % LP);  // w_512^i = w_L'^(i D)
        else w = tw2((((i - D2 * 8 - 64) / 9) * ((i - D2 * 8 - 64) % 9) * 8 * D) % LP);  // w_64^m = w_L'^(8 D m)
        btab[i] = make_double2(w.x, w.y);
    }
    __syncthreads();
    // the partner of frequency k'' = k2 + 8 k1 + 64 k0 of the packed transform: 512 - k'', register 7 - k0 of the lane that
    // holds -(k2 + 8 k1) mod 64; lane 0 pairs inside its own registers (msd_power_w12_kernel's wave 0)
    const int mneg = (64 - ((lane >> 3) + 8 * (lane & 7))) & 63;
    const int plane = (mneg >> 3) + 8 * (mneg & 7);
    const bool self0 = lane == 0;
    double2 *myR = R + wv * RS;
    double sacc[8], tacc[5], sacc1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sacc[i] = sacc1[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 5; ++i) tacc[i] = 0.0;
    double *my_l = sacc_l + (size_t)wv * 8 * 64;
    if constexpr (D2 >= 3) {
#pragma unroll
        for (int i = 0; i < 8; ++i) my_l[i * 64 + lane] = 0.0;
    }
    Cx tw_c[D2 >= 2 ? D2 - 1 : 1];  // w_L'^(r lane): the per-lane factor of class r's twiddle
#pragma unroll
    for (int r = 1; r < D2; ++r) tw_c[r - 1] = tw2(r * lane);
    for (long long c = it.c_lo + wv; c < it.c_hi; c += W12_NW) {
        const double *row = x + (size_t)c * F;
        // (the tables' addresses through an opaque copy per series: see msd_power_w12r_kernel)
        int opq = 0;  // (see msd_power_w12r_kernel)
        asm volatile("" : "+v"(opq));
        const double2 *btab_s = btab + opq;
        Cx a[8];
        // ======== the frequencies k = D2 k'': the packed transform of the series folded at 1024 ========
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) {
            const int t = 2 * (lane + 64 * n2);
            double re = t < F ? row[t] : 0.0, im = t + 1 < F ? row[t + 1] : 0.0;
            if constexpr (D2 >= 2) {  // (F > 1024 only then)
                if (t + 1024 < F) re += row[t + 1024];
                if (t + 1025 < F) im += row[t + 1025];
            }
            a[n2] = {re, im};
        }
        w12r_passes(a, myR, lane, nullptr, Cx{1.0, 0.0}, t1tab, t2tab, false);
#pragma unroll
        for (int k0 = 0; k0 < 8; ++k0) {
            sacc[k0] = __builtin_fma(a[k0].x, a[k0].x, sacc[k0]);
            sacc[k0] = __builtin_fma(a[k0].y, a[k0].y, sacc[k0]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k0 = 4; k0 < 8; ++k0) w12_st(myR + lane + 64 * (k0 - 4), a[k0]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (self0) {
            tacc[0] = __builtin_fma(2.0 * a[0].x, a[0].y, tacc[0]);
            tacc[4] = __builtin_fma(2.0 * a[4].x, a[4].y, tacc[4]);
#pragma unroll
            for (int u = 1; u < 4; ++u) {
                tacc[u] = __builtin_fma(a[u].x, a[8 - u].y, tacc[u]);
                tacc[u] = __builtin_fma(a[u].y, a[8 - u].x, tacc[u]);
            }
        } else {
            Cx pz[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) pz[u] = w12_ld(myR + plane + 64 * (3 - u));  // the partner's register 7 - u
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                tacc[u] = __builtin_fma(a[u].x, pz[u].y, tacc[u]);
                tacc[u] = __builtin_fma(a[u].y, pz[u].x, tacc[u]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ======== the classes r = 1 .. D2 - 1 ========
        auto one_class = [&](auto rk) {
            constexpr int r = decltype(rk)::value;
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) {
                const int n = lane + 64 * n2;
                Cx sv = {n < F ? row[n] : 0.0, 0.0};
                if (n + 512 < F) sv = cx_add(sv, w12r_root_mul_d<r, D>(row[n + 512]));
                if (n + 1024 < F) sv = cx_add(sv, w12r_root_mul_d<2 * r, D>(row[n + 1024]));
                a[n2] = sv;
            }
            w12r_passes(a, myR, lane, btab_s + r * 8, tw_c[r - 1], t1tab, t2tab, false);
            if constexpr (r == 1) {
#pragma unroll
                for (int k0 = 0; k0 < 8; ++k0) {
                    sacc1[k0] = __builtin_fma(a[k0].x, a[k0].x, sacc1[k0]);
                    sacc1[k0] = __builtin_fma(a[k0].y, a[k0].y, sacc1[k0]);
                }
            } else {
#pragma unroll
                for (int k0 = 0; k0 < 8; ++k0) {
                    double v = my_l[k0 * 64 + lane];
                    v = __builtin_fma(a[k0].x, a[k0].x, v);
                    v = __builtin_fma(a[k0].y, a[k0].y, v);
                    my_l[k0 * 64 + lane] = v;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        if constexpr (D2 >= 2) one_class(std::integral_constant<int, 1>());
        if constexpr (D2 >= 3) one_class(std::integral_constant<int, 2>());
    }
    // The packed transform's frequencies first, as msd_power_w12_kernel sorts them out (the wave's own region: {S, T} at point
    // lane + 64 k0): |X_(D2 k'')|^2 = (S + S')/2 + Im(w) (S - S')/2 + Re(w) T, w = e^{-2 pi i k''/1024} — into registers.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k0 = 0; k0 < 8; ++k0) myR[lane + 64 * k0] = make_double2(sacc[k0], k0 < 4 ? tacc[k0] : 0.0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double pk[8], p_top = 0.0;
#pragma unroll
    for (int k0 = 0; k0 < 8; ++k0) {
        const int k = (lane >> 3) + 8 * (lane & 7) + 64 * k0;
        double sn, tk;
        if (self0) {
            const int kq = (8 - k0) & 7;
            sn = myR[64 * kq].x;
            tk = k0 == 0 ? tacc[0] : k0 == 4 ? tacc[4] : k0 < 4 ? tacc[k0] : tacc[8 - k0];
        } else {
            const double2 pv = myR[plane + 64 * (7 - k0)];
            sn = pv.x;
            tk = k0 < 4 ? tacc[k0] : pv.y;
        }
        const double sk = sacc[k0];
        const Cx w = tw2(k * D2);  // (cos, -sin) of 2 pi k / 1024
        pk[k0] = 0.5 * (sk + sn) + w.y * (0.5 * (sk - sn)) + w.x * tk;
        if (k == 0) p_top = sk - tk;
    }
    // The twelve waves' spectra are added in LDS, wave after wave in a fixed order (the regions are free behind the barrier): ONE
    // row of L'/2 + 1 sums per block goes to memory (twelve rows per block made the fold of the partial spectra the most
    // expensive kernel of the call).
    __syncthreads();
    double *rowsum = reinterpret_cast<double *>(R);
    for (int i = tid; i <= LP / 2; i += W12_THREADS) rowsum[i] = 0.0;
    __syncthreads();
    for (int w = 0; w < W12_NW; ++w) {
        if (wv == w) {
#pragma unroll
            for (int k0 = 0; k0 < 8; ++k0) {
                const int j = (lane >> 3) + 8 * (lane & 7) + 64 * k0;
                rowsum[D2 * j] += pk[k0];
                if (j == 0) rowsum[D2 * N] += p_top;
                if constexpr (D2 >= 2) {  // classes r >= 1: k = D j + r, or its mirror L' - k
                    const int k = D * j + 1;
                    rowsum[k <= LP / 2 ? k : LP - k] += sacc1[k0];
                }
                if constexpr (D2 >= 3) {
                    const int k = D * j + 2;
                    rowsum[k <= LP / 2 ? k : LP - k] += my_l[k0 * 64 + lane];
                }
            }
        }
        __syncthreads();
    }
    double *pp = Ppart + (size_t)it.row * (LP / 2 + 1);
    for (int i = tid; i <= LP / 2; i += W12_THREADS) pp[i] = rowsum[i];
}

// corr[s][k] = (1 / L') sum_{f < L'} P_s[f] e^{2 pi i f k / L'}, P_s[L' - f] = P_s[f] given for f = 0 .. L'/2, k < n_lags — by direct
// summation: lane = one lag, the RI_WAVES waves of a block take a share of the frequencies each (four independent sums per
// lane); cos from a quarter-wave table in LDS (L'/4 + 1 entries, cospi of 2 m / L'); products and sums in double-double
// (two_prod by fma, two_sum), so that the 12 288 terms cost no accuracy against a transform's log2 L' stages.
// grid (ceil(n_lags / 64), S), 64 RI_WAVES lanes.
constexpr int RI_WAVES = 16;
__global__ __launch_bounds__(64 * RI_WAVES) void msd_residue_inverse_kernel(const double *__restrict__ P, int LP, int n_lags,
                                                                            double *__restrict__ corr)
{
#pragma clang fp contract(off)  // (the two-sums below take a product: a fused multiply-add would break their error terms)
    extern __shared__ double ft_lds[];
    double *qt = ft_lds;                 // [LP / 4 + 1]
    double *red = qt + LP / 4 + 1;       // [RI_WAVES][64][2]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, H = LP / 2, Q4 = LP / 4;
    for (int m = tid; m <= Q4; m += 64 * RI_WAVES) qt[m] = cospi(2.0 * (double)m / (double)LP);
    __syncthreads();
    const double *p = P + (size_t)blockIdx.y * (H + 1);
    const int k = blockIdx.x * 64 + lane;
    // frequencies f = 1 .. H - 1 in RI_WAVES parts, each in four runs with a sum of its own (four independent chains per lane)
    const int f0 = 1 + (int)((long long)(H - 1) * wv / RI_WAVES), f1 = 1 + (int)((long long)(H - 1) * (wv + 1) / RI_WAVES);
    auto cosv = [&](long long idx) {
        int m = (int)idx;
        if (m > H) m = LP - m;
        return m > Q4 ? -qt[H - m] : qt[m];
    };
    auto dd_add = [&](double &hi, double &lo, double pv, double cv) {
        const double ph = pv * cv, pl = __builtin_fma(pv, cv, -ph);
        const double s = hi + ph, bb = s - hi;
        lo += ((hi - (s - bb)) + (ph - bb)) + pl;
        hi = s;
    };
    const int nq = (f1 - f0) / 4;  // terms per run; the remainder (< 4 terms) goes to the first sum
    double sh[4] = {0.0, 0.0, 0.0, 0.0}, sl[4] = {0.0, 0.0, 0.0, 0.0};
    long long ix[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) ix[u] = ((long long)(f0 + u * nq) * (long long)k) % LP;
    for (int i = 0; i < nq; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            dd_add(sh[u], sl[u], p[f0 + u * nq + i], cosv(ix[u]));
            ix[u] += k;
            if (ix[u] >= LP) ix[u] -= LP;
        }
    }
    {
        long long ir = ((long long)(f0 + 4 * nq) * (long long)k) % LP;
        for (int f = f0 + 4 * nq; f < f1; ++f) {
            dd_add(sh[0], sl[0], p[f], cosv(ir));
            ir += k;
            if (ir >= LP) ir -= LP;
        }
    }
    double ah = sh[0], al = sl[0];
#pragma unroll
    for (int u = 1; u < 4; ++u) {
        const double s = ah + sh[u], bb = s - ah;
        al += ((ah - (s - bb)) + (sh[u] - bb)) + sl[u];
        ah = s;
    }
    red[(wv * 64 + lane) * 2] = ah;
    red[(wv * 64 + lane) * 2 + 1] = al;
    __syncthreads();
    if (wv == 0 && k < n_lags) {
        double sh = 0.0, sl = 0.0;
        for (int w = 0; w < RI_WAVES; ++w) {
            const double xh = red[(w * 64 + lane) * 2], xl = red[(w * 64 + lane) * 2 + 1];
            const double s = sh + xh, bb = s - sh;
            sl += ((sh - (s - bb)) + (xh - bb)) + xl;
            sh = s;
        }
        const double edge = p[0] + ((k & 1) ? -p[H] : p[H]);
        corr[(size_t)blockIdx.y * n_lags + k] = (edge + 2.0 * (sh + sl)) / (double)LP;
    }
}
