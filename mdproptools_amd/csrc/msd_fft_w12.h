// msd_fft_w12.h — included by msd_fft.hip inside its anonymous namespace (behind the helpers of the fused kernels).
//
// Round 5: the fused power-spectrum kernel for padded lengths L = 12288 = 3 * 2^12 (N = 6144 packed complex points):
// every series with 8192 < F + max_lag <= 12288 — BASELINE config 4, F = 5000 — took L = 16384 before. What the counters
// of msd_power_lds3_kernel said when read against the issue rates this library measured (profiles/r02_ubench_valu.txt):
// 9100 vector instructions per series on 4 SIMDs, 88 % of them f64, issue at 6.85 cycles each with TWO waves per SIMD
// (the 256-register budget) = 15 600 of the 16 200 cycles a series took. The kernel was bound by f64 ISSUE AT ITS
// OCCUPANCY, not by LDS or HBM. Two levers, both taken here:
//   * fewer points: 2 F - 1 = 9999 needs no power of two. N = 12 x 512: a radix-12 first pass (pruned: 5 of 12 inputs hold
//     data at F = 5000) and 512-point sub-transforms — 0.75 of the points, LDS traffic and butterflies of N = 8192;
//   * more waves: TWELVE waves per block, 3 per SIMD (f64 issue ~5.6 instead of 6.85 cycles). A wave owns one 512-point
//     sub-transform = 8 points per lane, which it keeps in REGISTERS through three radix-8 passes; LDS is only the
//     medium of the two 8 x 8 lane-register transpositions between them (16-byte accesses, conflict-free by the choice
//     of the exchange layout, which no in-place constraint ties down any more). ~150 registers instead of 244.
// Structure per series (three block barriers, as before):
//   first pass   item (j, c), 2 per lane: the inputs z[j + 512 e] (centred, from the raw plane in LDS) -> the four outputs
//                d = c + 3 q of the radix-12 butterfly (a 12-th-root pre-twist, then a DFT-4) -> region d, position j;
//   -- barrier -- (the raw plane is free: the next series' samples are requested into registers here)
//   wave d       reads its region (lane l: points l + 64 n2), applies the first pass's twiddle w_N^(j d) in two factors —
//                a wave-uniform one per register (w_96^(d n2), from a small table) and a per-lane one folded into the
//                pass's own twiddle chain —, radix 8 over n2, exchange, radix 8 over n1, exchange, radix 8 over n0;
//                |Z|^2 into 8 accumulators; registers 4..7 to LDS for the partner;
//   -- barrier -- (rides: the block sum of the next series' samples)
//   partner      frequency N - k sits in wave 12 - d, lane 63 - l, register 7 - k0 (wave 0: its own region): Im(Z Z')
//                for the lane's registers 0..3; the next series: centre, add the squares (S1), store to the raw plane;
//   -- barrier --
// Frequencies are sorted out once per block, as in msd_power_lds3_kernel (bilinear spectrum sums, see there).
// The staging of the trajectory (SRC == 2: clusters of 16 blocks transpose their tiles through a ring in device memory)
// is that kernel's, re-dealt to 768 lanes: 96 rows per round, four 16-byte units per lane and tile.

constexpr int W12_NW = 12;
constexpr int W12_THREADS = 64 * W12_NW;
constexpr int W12_SUB = 512;                // points per sub-transform
constexpr int W12_N = W12_NW * W12_SUB;     // 6144
constexpr int W12_RS = 520;                 // points between two regions (the second exchange uses 8 x 65)
constexpr int W12_UN = 4;                   // staging units per lane and tile (96 rows a round: rows per member <= 384)
#ifndef W12_EXP
#define W12_EXP 0  // timing experiments only (WRONG results): 1 no first barrier, 2 no second, 4 no third, 8 no partner phase,
                   // 16 no exchanges (LDS writes + reads of the register passes), 32 no butterfly arithmetic in the passes,
                   // 64 no first-pass arithmetic, 128 no staging (SRC == 2: loads, stores), 256 no staging loads, 512 no staging stores
#endif
// A 16-byte buffer store hands its data registers to the memory pipeline over more than one cycle: a vector instruction
// that WRITES one of them in the cycle behind the store changes what some lanes store. The compiler knows the hazard
// (one or two idle cycles behind stores of more than 8 bytes) but not for buffer stores whose scalar offset is a
// register — the form every staging store here has — and the QE = 6 instance of this kernel came out with
//     buffer_store_dwordx4 v[86:89], v114, s[56:59], s74 offen sc1
//     v_cndmask_b32_e32 v86, v148, v126, vcc        (the next unit's load offset, into the store's first data register)
// which stored that offset as the low word of one sample in every second 8-lane group (integer-ramp input: MSD(1) =
// 1.013 instead of 1; tools/w12.py ramp). The guard keeps the data registers live across two idle cycles behind the
// store; tests/test_codegen_cpu.py scans the compiled kernels for the pattern.
#define W12_STORE_GUARD(v) asm volatile("s_nop 1" ::"v"(v))
#define W12_BARRIER(bit) do { if (!(W12_EXP & (bit))) __syncthreads(); } while (0)

// LDS: regions | raw plane (QE x 512 points) | two-level twiddle table | b table [12][8] | w_512^lane [64] |
// w_64^(n0 k1) [8][9] | red
inline size_t w12_lds_bytes(int qe)
{
    return (size_t)W12_NW * W12_RS * 16 + (size_t)qe * W12_SUB * 16 + 256 * 16 + (size_t)W12_NW * 8 * 16 + (64 + 72) * 16 + 32 * 8;
}

template <int K12>
__device__ __forceinline__ Cx w12_rot(Cx a)  // a * w_12^K12
{
    constexpr int k = ((K12 % 12) + 12) % 12;
    constexpr double C30 = 0.86602540378443864676, H = 0.5;
    if constexpr (k == 0) return a;
    else if constexpr (k == 3) return {a.y, -a.x};
    else if constexpr (k == 6) return {-a.x, -a.y};
    else if constexpr (k == 9) return {-a.y, a.x};
    else {
        // w^k = (cos(30 k), -sin(30 k))
        constexpr double cr = (k == 1 || k == 11) ? C30 : (k == 2 || k == 10) ? H : (k == 4 || k == 8) ? -H : -C30;
        constexpr double ci = (k == 1 || k == 5) ? -H : (k == 2 || k == 4) ? -C30 : (k == 7 || k == 11) ? H : C30;
        return {a.x * cr - a.y * ci, a.x * ci + a.y * cr};
    }
}

// The radix-12 butterfly of the first pass, outputs d = C + 3 q (q = 0..3), inputs z[0 .. QE) (zeros beyond), 4 <= QE <= 6:
//   sum_e z_e w12^((C + 3 q) e) = sum_e0 w4^(q e0) u_e0,   u_e0 = sum_e1 z_(e0 + 4 e1) w12^(C (e0 + 4 e1))
template <int QE, int C>
__device__ __forceinline__ void w12_head(const Cx *z, Cx *o)
{
    Cx u0 = z[0], u1 = w12_rot<C>(z[1]);
    const Cx u2 = w12_rot<2 * C>(z[2]), u3 = w12_rot<3 * C>(z[3]);
    if constexpr (QE > 4) u0 = cx_add(u0, w12_rot<4 * C>(z[4]));
    if constexpr (QE > 5) u1 = cx_add(u1, w12_rot<5 * C>(z[5]));
    dft4(u0, u1, u2, u3, o[0], o[1], o[2], o[3]);
}

__device__ __forceinline__ Cx w12_ld(const double2 *p)
{
    const double2 v = *p;
    return {v.x, v.y};
}
__device__ __forceinline__ void w12_st(double2 *p, Cx v) { *p = make_double2(v.x, v.y); }

// x: SRC == 0 the time-major copy [cols][F] (scaled), SRC == 2 the trajectory [F][cols] (see msd_power_lds3_kernel).
// Qpart [rows][F], Ppart [rows][N + 1] as the other fused kernels (N = 6144).
// SHORT (round 6): trajectories of 1536 <= F < 3072 frames (QE = 4): only the first of a lane's four sample units is
// known to hold data in every lane (the instances for F >= 3072 skip the test for the first two).
template <int QE, int SRC, bool SHORT = false>
__global__ __launch_bounds__(W12_THREADS) void msd_power_w12_kernel(
    const double *__restrict__ x, int F, const FftItem *__restrict__ items, const double2 *__restrict__ tab,
    double *__restrict__ Qpart, double *__restrict__ Ppart, long long cols, double scale, const FftStage *__restrict__ stg,
    double *__restrict__ scratch, unsigned *__restrict__ ready, int Fc_arg)
{
    constexpr int N = W12_N, RS = W12_RS;
    constexpr int RFULL = SHORT ? 1 : 2;  // units r < RFULL hold data in every lane
    const int Fc = Fc_arg < 0 ? -Fc_arg : Fc_arg;
    const bool withhold = Fc_arg < 0 && blockIdx.x == 0;
    extern __shared__ double ft_lds[];
    double2 *R = reinterpret_cast<double2 *>(ft_lds);       // regions
    double2 *raw = R + W12_NW * RS;                          // centred samples of the series about to be transformed
    double2 *tabA = raw + QE * W12_SUB, *tabB = tabA + 128;  // w_L^(128 i), w_L^i
    double2 *btab = tabB + 128;                              // [d][n2] = w_N^(64 d n2)
    double2 *t1tab = btab + W12_NW * 8;                      // [lane] = w_512^lane (the twiddle step of the first register pass)
    double2 *t2tab = t1tab + 64;                             // [9 n0 + k1] = w_64^(n0 k1): the twiddles of the second, ready made
                                                             // (rows of 9: the 8 rows a read touches start 36 banks apart)
    double *red = reinterpret_cast<double *>(t2tab + 72);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < 256) tabA[tid] = tab[tid];
    const FftItem it = items[blockIdx.x];
    __syncthreads();
    if (tid < W12_NW * 8) {
        const int d = tid >> 3, n2 = tid & 7;
        const Cx w = ft_tw(tabA, tabB, (128 * d * n2) % (2 * N));  // w_N^(64 d n2) = w_L^(128 d n2)
        btab[tid] = make_double2(w.x, w.y);
    }
    if (tid >= 128 && tid < 128 + 64 + 72) {
        const int i = tid - 128;
        const int q = i - 64;  // (second table: row n0 = q / 9, column k1 = q % 9; column 8 is padding)
        const Cx w = ft_tw(tabA, tabB, i < 64 ? 24 * i : (192 * (q / 9) * (q % 9)) % (2 * N));
        t1tab[i] = make_double2(w.x, w.y);
    }
    // per-lane constants of the three register passes: lane = n0 + 8 n1 (pass 1) = n0 + 8 k2 (pass 2) = k1 + 8 k2 (pass 3).
    // Only w_N^(lane d) stays in registers; the passes' twiddle steps are read from LDS where they are used (a register
    // held across the whole series costs more than a 16-byte read: the kernel sits at its 168-register limit)
    const Cx tw_a = ft_tw(tabA, tabB, (2 * lane * wv) % (2 * N));  // w_N^(lane d): the per-lane factor of the first pass's twiddle
    // the partner of frequency k = d + 12 k' (k' = k2 + 8 k1 + 64 k0 in lane k1 + 8 k2, register k0): N - k, which is
    // wave 12 - d, lane 63 - lane, register 7 - k0 for d > 0 (511 - k': every digit complemented). Wave 0 pairs inside
    // itself, k' with 512 - k': register 7 - k0 of the lane that holds -(k2 + 8 k1) mod 64 — except lane 0 (k2 = k1 = 0),
    // whose pairs (k0, 8 - k0) are its own registers.
    const int pw = wv == 0 ? 0 : W12_NW - wv;
    int plane = 63 - lane;
    if (wv == 0) {
        const int mneg = (64 - ((lane >> 3) + 8 * (lane & 7))) & 63;  // (k2 + 8 k1 negated mod 64)
        plane = (mneg >> 3) + 8 * (mneg & 7);
    }
    const bool self0 = wv == 0 && lane == 0;
    double sacc[8], tacc[5], qa[4], qb[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) sacc[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 5; ++i) tacc[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) qa[i] = qb[i] = 0.0;
    typedef double st2_t __attribute__((ext_vector_type(2)));
    st2_t v[4];  // the next series' samples: packed points tid + 768 r

    // ---- SRC == 2: the cluster's staging ring (msd_power_lds3_kernel's protocol and ring layout [slot][column][time]) ----
    // What differs is how a unit is moved. There a lane loads 16 bytes of ONE row (two columns) and swaps a value with its
    // neighbour lane so that it can store 16 bytes of one column: 2 multiplications + 2 DPP moves + 6 selects per unit,
    // and a second form of the load for odd column counts. Here a lane loads the SAME column of TWO consecutive rows with
    // two 8-byte loads — lane (pair sp = 8 (wave / 2) + lane / 8, column sc = 8 (wave % 2) + lane % 8): a load instruction
    // touches 8 half lines, as there — and stores the 16 bytes as they are: no swap, no odd-column form (8-byte loads do
    // not care about the rows' alignment), whole 128-byte lines per column in the ring (the 8 lanes of a column hold 16
    // consecutive rows). 8 vector instructions per unit instead of ~26: the staging was 0.93 of the call's 4.5 ms.
    FftStage sg{};
    if constexpr (SRC == 2) sg = stg[blockIdx.x];
    const long long Fs = 16LL * Fc, nt = it.c_hi - it.c_lo;
    st2_t sx = {0.0, 0.0}, sy = {0.0, 0.0};
    constexpr int RPR = 16 * (W12_NW / 2);  // rows per round of the block
    const int st_sp = (wv >> 1) * 8 + (lane >> 3), st_sc = (wv & 1) * 8 + (lane & 7);
    constexpr unsigned ST_OOB = 0xFFFFF000u;
    // rows of this member's share [k Fc, k Fc + st_rows) that lie at or behind the lane's first row 2 sp: unit r moves the
    // pair (96 r + 2 sp, + 1) iff 96 r < st_have (the second row of a pair may be the one behind the series' end, F odd:
    // it reads as zero — beyond the buffer — and the ring's row F holds that zero)
    const long long st_rows = (long long)F - (long long)sg.k * Fc < Fc ? (long long)F - (long long)sg.k * Fc : (long long)Fc;
    const int st_have = (int)(st_rows > 0 ? st_rows : 0) - 2 * st_sp;
    unsigned st_vi = 0u, st_vo = 0u;
    if constexpr (SRC == 2) {
        st_vi = (unsigned)(((size_t)(2 * st_sp) * (size_t)cols + (size_t)st_sc) * 8);
        st_vo = (unsigned)(((size_t)st_sc * (size_t)Fs + (size_t)sg.k * Fc + 2 * st_sp) * 8);
    }
    const __amdgpu_buffer_rsrc_t traj = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double *>(x) + (SRC == 2 ? (size_t)sg.k * (size_t)Fc * (size_t)cols : 0), 0,
        SRC == 2 ? (int)(unsigned)((size_t)(st_rows > 0 ? st_rows : 0) * (size_t)cols * 8) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ring = __builtin_amdgcn_make_buffer_rsrc(
        scratch + (SRC == 2 ? (size_t)sg.cluster * ST_BUF * 16 * (size_t)Fs : 0), 0,
        SRC == 2 ? (int)((size_t)ST_BUF * 16 * (size_t)Fs * 8) : 0, 0x00020000);
    constexpr int SC1 = 16, NT_HINT = 2;
    typedef unsigned st2u_t __attribute__((ext_vector_type(2)));
    typedef unsigned st4_t __attribute__((ext_vector_type(4)));
    // (per tile, wave-uniform but for the column test of the matrix's last tile) rows this lane may move of tile i: 0
    // for a tile behind the cluster's last one or a column beyond the matrix
    auto st_lim_of = [&](long long i) {
        const long long T = it.c_lo + i;
        return (i < nt && 16 * T + st_sc < cols) ? st_have : 0;
    };
    // Nothing here branches (see msd_power_lds3_kernel): a lane without a part addresses the buffer beyond its end in its
    // PER-LANE offset, which is what the hardware's range check looks at; loads return zeros there, stores are dropped.
    auto stage_load = [&](long long i, int r, int lim, st2_t &sv) {
        const long long T = it.c_lo + i;
        const unsigned soff = (unsigned)(((size_t)(RPR * r) * (size_t)cols + (size_t)(16 * T)) * 8);
        const st2u_t a = __builtin_amdgcn_raw_buffer_load_b64(traj, RPR * r < lim ? st_vi : ST_OOB, soff, NT_HINT);
        const st2u_t b = __builtin_amdgcn_raw_buffer_load_b64(traj, RPR * r + 1 < lim ? st_vi : ST_OOB,
                                                             soff + (unsigned)((size_t)cols * 8), NT_HINT);
        sv = st2_t{__builtin_bit_cast(double, a), __builtin_bit_cast(double, b)};
    };
    auto stage_store = [&](long long i, int r, int lim, const st2_t &sv) {
        const st2_t out = st2_t{sv[0] * scale, sv[1] * scale};
        const unsigned soff = (unsigned)(((size_t)(i & (ST_BUF - 1)) * 16 * (size_t)Fs + (size_t)(RPR * r)) * 8);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(st4_t, out), ring, RPR * r < lim ? st_vo : ST_OOB, soff,
                                               (W12_EXP & 1024) ? 0 : SC1);  // (timing: a store that ends in this XCD's L2)
        W12_STORE_GUARD(out);
    };
    auto st_flag = [&](long long i) { return ready + ((size_t)sg.cluster * ST_BUF + (size_t)(i & (ST_BUF - 1))) * ST_FLAG_STRIDE; };
    auto st_signal = [&](long long i) {
        if (tid == 0 && i >= 0 && i < nt && !(withhold && i >= ST_AHEAD))
            __hip_atomic_fetch_add(st_flag(i), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    bool stalled = false;
    auto st_peek = [&](long long i) { return __hip_atomic_load(st_flag(i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto st_wait = [&](long long i, unsigned seen) {
        const unsigned need = 16u * (unsigned)((i >> 3) + 1);
        if (stalled || seen >= need) return;
        const unsigned *w = st_flag(i);
        unsigned *stall = ready + (size_t)(gridDim.x / 16) * ST_BUF * ST_FLAG_STRIDE;
        for (int spin = 1; __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need; ++spin) {
            __builtin_amdgcn_s_sleep(8);
            if ((spin & 255) == 0 && __hip_atomic_load(stall, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                stalled = true;
                break;
            }
            if (spin > (1 << 19)) {  // ~1 s: a member is not running
                __hip_atomic_store(stall, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                stalled = true;
                break;
            }
        }
    };

    // the samples of series / tile c into v[]: packed point n = tid + 768 r holds samples 2 n, 2 n + 1 (zeros beyond F)
    auto fetch = [&](long long c, unsigned seen = 0u) {
        if constexpr (SRC == 2) {
            const long long i = c - it.c_lo, col = 16 * c + sg.k;
            const bool valid = col >= sg.lo && col < sg.hi;
            st_wait(i, seen);
            const unsigned row = (unsigned)((((size_t)(i & (ST_BUF - 1))) * 16 + sg.k) * (size_t)Fs * 8);  // (scalar)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = tid + W12_THREADS * r;
                // (beyond the series, or a column outside the segment: zeros from beyond the buffer's end; row F of the
                // ring holds a zero where F is odd: the pair store wrote it)
                const bool in = valid && (r < RFULL || 2 * n < F);  // (F >= 1536 RFULL: the host's condition for this kernel)
                v[r] = __builtin_bit_cast(
                    st2_t, __builtin_amdgcn_raw_buffer_load_b128(ring, in ? (unsigned)n * 16u : ST_OOB, row, SC1));
            }
            return;
        }
        const double *row = x + (size_t)c * F;
        const bool al16 = (reinterpret_cast<unsigned long long>(row) & 15ull) == 0ull;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = tid + W12_THREADS * r;
            v[r] = st2_t{0.0, 0.0};
            if (2 * n + 1 < F) {
                if (al16) {
                    v[r] = __builtin_nontemporal_load(reinterpret_cast<const st2_t *>(row + 2 * n));
                } else {
                    v[r][0] = row[2 * n];
                    v[r][1] = row[2 * n + 1];
                }
            } else if (2 * n < F) {
                v[r][0] = row[2 * n];
            }
        }
    };
    // this wave's share of the block sum of v[] into red[slot + wv]
    auto wave_sum = [&](int slot) {
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) s += v[r][0] + v[r][1];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (lane == 0) red[slot + wv] = s;
    };
    // (behind the barrier that follows wave_sum) centre v[], add the squares, store to the raw plane
    auto centre_store = [&](int slot) {
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < W12_NW; ++w) sum += red[slot + w];
        const double mean = sum / (double)F;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = tid + W12_THREADS * r;
            // (QE >= 5: the first three units always lie inside the raw plane; QE = 4 — never instantiated before round 6,
            // whose shorter trajectories found it — holds 2048 points: the third unit reaches 2303 and would run over the
            // twiddle table behind the plane)
            if (r < (QE >= 5 ? 3 : 2) || n < QE * W12_SUB) {
                // (F >= 3072, the host's condition for this kernel: the first two units hold data in every lane)
                const double da = (r < RFULL || 2 * n < F) ? v[r][0] - mean : 0.0;
                const double db = (r < RFULL || 2 * n + 1 < F) ? v[r][1] - mean : 0.0;
                qa[r] = __builtin_fma(da, da, qa[r]);
                qb[r] = __builtin_fma(db, db, qb[r]);
                raw[n] = make_double2(da, db);
            }
        }
    };

    if constexpr (SRC == 2) {
        // the first ST_AHEAD tiles, before anything is transformed
        // (a tile at a time: its eight loads, one wait for all of them, then its four stores. The software-pipelined form
        // the compiler makes of the plain loop stored wrong values in every second 8-lane group; the cause, found later
        // in the series loop of the QE = 6 instance, is the one W12_STORE_GUARD describes — a store's data register
        // rewritten in the cycle behind it — and the guard covers these stores too. The prologue runs once per block:
        // the simple form stays.)
        int n_ahead = ST_AHEAD;
        asm volatile("" : "+s"(n_ahead));  // (opaque trip count: no unrolling across tiles)
        for (int i = 0; i < n_ahead; ++i) {
            const int lim = st_lim_of(i);
            st2_t u[W12_UN];
#pragma unroll
            for (int r = 0; r < W12_UN; ++r) stage_load(i, r, lim, u[r]);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3])::"memory");
#pragma unroll
            for (int r = 0; r < W12_UN; ++r) stage_store(i, r, lim, u[r]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int i = 0; i < ST_AHEAD; ++i) st_signal(i);
    }
    if (it.c_lo < it.c_hi) {  // (a cluster member whose column lies outside the segment transforms zeros)
        fetch(it.c_lo);
#ifdef W12_VERIFY
        if constexpr (SRC == 2) {
            const long long coln = 16 * it.c_lo + sg.k;
            if (coln >= sg.lo && coln < sg.hi) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = tid + W12_THREADS * r;
                    for (int h = 0; h < 2; ++h) {
                        const int t = 2 * n + h;
                        if (t < F && v[r][h] != 16384.0 * (double)coln + (double)t)
                            printf("BAD0 block %d k %d tile %lld t %d got %.1f want %.1f\n", (int)blockIdx.x, sg.k, it.c_lo, t,
                                   v[r][h], 16384.0 * (double)coln + (double)t);
                    }
                }
            }
        }
#endif
        wave_sum(0);
        __syncthreads();
        centre_store(0);
    }
    __syncthreads();
    const int cstep = SRC == 2 ? 1 : it.step;
#ifdef W12_VERIFY
    double vfirst = 0.0;
    int vfirst_set = 0;
#endif
    const int c3 = wv % 3, g4 = wv / 3;  // first pass: this wave's residue c = d mod 3 and its quarter of the positions j
    double2 *myR = R + wv * RS;
    const double2 *pR = R + pw * RS;
    for (long long c = it.c_lo; c < it.c_hi; c += cstep) {
        const long long st_i = c - it.c_lo + ST_AHEAD;
        // Staging points 0 .. 5 of the iteration: point P stores the unit requested two points ago (P - 2) and requests
        // unit P (< W12_UN), units alternating between the two register pairs (msd_power_lds3_kernel's scheme).
        // Two other schedules were measured against this one in one process (tools/w12.py exp, C4 call, 4.23-4.28 ms
        // here; 3.52 ms with no staging at all): every unit stored FOUR points after its request (four units, 16
        // registers, live throughout; the signal two series later) 4.33-4.37 ms; all four units requested at the end of
        // the series before and stored together behind the first register pass 4.72 ms. The staging's cost is not the
        // distance between request and use: with it the call moves 18 GB through the fabric port (trajectory in, ring
        // out, ring in: profiles/pmc_secondary.json) in ~4.2 ms.
        // What it is, from timing builds in one process (W12_EXP 256 / 512 / 128 / 1024; the full kernel 4.11-4.16 ms):
        // no trajectory loads (zeros stored, ring read back) 3.53 ms; loads but no ring stores 3.75 ms; neither 3.56 ms;
        // ring stores without sc1 4.07-4.17 ms (no change). The stores alone are free, the loads alone cost 0.2 ms, both
        // 0.57 ms: the price follows the BYTES through the fabric port — 6 GB (ring in) 3.56 ms, 12 GB 3.5-3.75 ms,
        // 18 GB 4.1 ms = 4.4 TB/s of mixed reads and writes, the rate this kernel's traffic is served at — and below
        // ~12 GB the arithmetic and LDS time of the series (3.5 ms) hides it. Under 4.0 ms needs fewer bytes, not a
        // better schedule — and not simply a ring next door: with every cluster's 16 members dealt to ONE XCD (block b
        // runs on XCD b % 8) and the ring going through that XCD's L2 (plain stores, plain or sc0 loads; a timing build,
        // right only by luck) the call took 4.16-4.23 ms against 4.14-4.20: two clusters' live tiles (8 x 655 KB each)
        // do not fit 4 MB of L2 beside the trajectory streaming through it. (With an L1 invalidate per series: 15 ms.)
        // (The non-temporal hint on the ring's stores, its loads or both: 4.26 / 4.11 / 4.21 ms against 4.14-4.19 — nothing.)
        // Nor is it the instruction count at the margin (same process, 4.20 ms): the mean by a reciprocal made once
        // instead of a division per series (-12 vector instructions of ~640) 4.20 ms; the two multiplications per unit
        // skipped behind a wave-uniform test of scale == 1.0 4.51-4.54 ms (the branches cut the schedule the compiler
        // makes of the points). Neither is kept.
        int st_lim = 0;
        if constexpr (SRC == 2) st_lim = st_lim_of(st_i);
        auto point = [&](auto pk) {
            constexpr int P = decltype(pk)::value;
            if constexpr (SRC == 2 && !(W12_EXP & 128)) {
                st2_t &reg = (P & 1) ? sy : sx;
                if constexpr (P >= 2 && P - 2 < W12_UN) {
                    if constexpr (W12_EXP & 512) asm volatile("" ::"v"(reg));  // (timing: the unit is loaded, not stored)
                    else stage_store(st_i, P - 2, st_lim, reg);
                }
                if constexpr (P < W12_UN) {
                    if constexpr (W12_EXP & 256) reg = st2_t{0.0, 0.0};  // (timing: nothing is loaded, zeros are stored)
                    else stage_load(st_i, P, st_lim, reg);
                }
            }
        };
#define W12_POINT(P) point(std::integral_constant<int, (P)>())
        const bool more = c + cstep < it.c_hi;
        unsigned seen = 0u;
        if constexpr (SRC == 2) {
            if (more) seen = st_peek(c + cstep - it.c_lo);  // (an early look at the next tile's counter: no round trip later)
        }
        W12_POINT(0);
        // ---- first pass: two items (j, c3) per lane ----
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = g4 * 128 + jj * 64 + lane;
            Cx z[QE], o[4];
#pragma unroll
            for (int e = 0; e < QE; ++e) z[e] = w12_ld(raw + j + W12_SUB * e);
            if (W12_EXP & 64) {
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = z[q];
            } else if (c3 == 0) w12_head<QE, 0>(z, o);
            else if (c3 == 1) w12_head<QE, 1>(z, o);
            else w12_head<QE, 2>(z, o);
#pragma unroll
            for (int q = 0; q < 4; ++q) w12_st(R + (c3 + 3 * q) * RS + j, o[q]);
        }
        W12_POINT(1);
        W12_BARRIER(1);
        // the raw plane is free: the next series' samples are requested into registers and land under the passes
        if (more) fetch(c + cstep, seen);
        // ---- this wave's 512-point sub-transform, in registers ----
        Cx a[8];
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) a[n2] = w12_ld(myR + lane + 64 * n2);
        if (wv != 0 && !(W12_EXP & 32)) {
#pragma unroll
            for (int n2 = 1; n2 < 8; ++n2) a[n2] = cx_mul(a[n2], w12_ld(btab + wv * 8 + n2));
        }
        if (!(W12_EXP & 32)) f2_bfly8(a, Cx{1.0, 0.0}, false);
        if (!(W12_EXP & 32)) {
            const Cx tw_1 = w12_ld(t1tab + lane);
            Cx t = tw_a;  // w_N^(lane d) w_512^(lane k2), k2 = 0..7
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
        if (!(W12_EXP & 16)) {
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) w12_st(myR + (lane & 7) + 8 * k2 + 64 * (lane >> 3), a[k2]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) a[n1] = w12_ld(myR + lane + 64 * n1);
        }
        if constexpr (SRC == 2) {
            // what this wave stored for the tile staged under the PREVIOUS series has long been issued, and the youngest
            // request in flight (the next series' samples) is a pass old: waiting for everything here lets the second
            // barrier below carry the signal for that tile. (A counted wait that leaves the eight loads issued behind
            // those stores in flight — memory operations complete in issue order — measured the same, 4.18-4.29 against
            // 4.22-4.24 ms in one process: not kept.)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        W12_POINT(2);
        if (!(W12_EXP & 32)) f2_bfly8(a, Cx{1.0, 0.0}, false);
        if (!(W12_EXP & 32)) {
            // w_64^(n0 k1) from the table (7 reads of 16 bytes, the 8 lanes of an n0 reading one address) instead of six
            // chained complex products: the kernel is short of vector issue, not of LDS reads
            const double2 *t2 = t2tab + 9 * (lane & 7);
#pragma unroll
            for (int k1 = 1; k1 < 8; ++k1) a[k1] = cx_mul(a[k1], w12_ld(t2 + k1));
        }
        // exchange 2: (n0, k2 | k1) -> (k1, k2 | n0): point k1 + 8 k2 + 65 n0
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (!(W12_EXP & 16)) {
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) w12_st(myR + k1 + 8 * (lane >> 3) + 65 * (lane & 7), a[k1]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int n0 = 0; n0 < 8; ++n0) a[n0] = w12_ld(myR + lane + 65 * n0);
        }
        W12_POINT(3);
        if (!(W12_EXP & 32)) f2_bfly8(a, Cx{1.0, 0.0}, false);
        // a[k0] = Z at frequency d + 12 (k2 + 8 k1 + 64 k0), lane = k1 + 8 k2
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
        W12_POINT(4);
        if (more) wave_sum(0);
        W12_BARRIER(2);
        if constexpr (SRC == 2) {
            if (c > it.c_lo) st_signal(st_i - 1);  // (the prologue signalled its own tiles)
        }
        if (!self0 && !(W12_EXP & 8)) {
            Cx pz[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) pz[u] = w12_ld(pR + plane + 64 * (3 - u));  // the partner's register 7 - u
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                tacc[u] = __builtin_fma(a[u].x, pz[u].y, tacc[u]);
                tacc[u] = __builtin_fma(a[u].y, pz[u].x, tacc[u]);
            }
        }
        W12_POINT(5);
        if (more) centre_store(0);
        W12_BARRIER(4);
#undef W12_POINT
    }
    if constexpr (SRC == 2) {
        // the units of the last tile still in the registers were stored at point 5; nothing is in flight but stores
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    double *q = Qpart + (size_t)it.row * F, *pp = Ppart + (size_t)it.row * (N + 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = tid + W12_THREADS * r;
        if (2 * n < F) q[2 * n] = qa[r];
        if (2 * n + 1 < F) q[2 * n + 1] = qb[r];
    }
    // frequencies, once per block (the loop ended on a barrier): point lane + 64 k0 of the wave's region = {S, T}
#pragma unroll
    for (int k0 = 0; k0 < 8; ++k0) myR[lane + 64 * k0] = make_double2(sacc[k0], k0 < 4 ? tacc[k0] : 0.0);
    __syncthreads();
#pragma unroll
    for (int k0 = 0; k0 < 8; ++k0) {
        const int kp = (lane >> 3) + 8 * (lane & 7) + 64 * k0;  // k' = k2 + 8 k1 + 64 k0
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
        const Cx w = ft_tw(tabA, tabB, k);  // (cos, -sin) of 2 pi k / L
        pp[k] = 0.5 * (sk + sn) + w.y * (0.5 * (sk - sn)) + w.x * tk;
        if (k == 0) pp[N] = sk - tk;
    }
}

// corr[s][k] = inverse real transform of P[s][0..N] (N = 6144), k < n_lags: msd_inverse_lds_kernel for this length —
// the same packing, a radix-12 pass and three radix-8 passes in place, block-wide (a handful of blocks: speed is no concern)
__global__ __launch_bounds__(512) void msd_inverse_w12_kernel(const double *__restrict__ P, const double2 *__restrict__ tab,
                                                              int n_lags, double *__restrict__ corr)
{
    constexpr int N = W12_N;
    extern __shared__ double ft_lds[];
    double2 *Z = reinterpret_cast<double2 *>(ft_lds);
    double2 *tabA = Z + N, *tabB = tabA + 128;
    const int tid = threadIdx.x;
    if (tid < 256) tabA[tid] = tab[tid];
    __syncthreads();
    const double *p = P + (size_t)blockIdx.x * (N + 1);
    for (int k = tid; k < N; k += 512) {
        const double pk = p[k], pn = p[N - k];
        const double e = 0.5 * (pk + pn), d = 0.5 * (pk - pn);
        const Cx w = ft_tw(tabA, tabB, k);  // (cos, -sin)
        Z[k] = make_double2(e + d * w.y, -d * w.x);
    }
    __syncthreads();
    // radix 12 at stride 512: y_d[j] = w_N^(j d) sum_e z[j + 512 e] w12^(d e), in place (thread j owns the points j + 512 x)
    {
        const int j = tid;
        Cx z[12], y[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) z[e] = w12_ld(Z + j + 512 * e);
#pragma unroll
        for (int d = 0; d < 12; ++d) {
            Cx acc = z[0];
#pragma unroll
            for (int e = 1; e < 12; ++e) {
                const Cx w = ft_tw(tabA, tabB, 1024 * ((d * e) % 12));  // w12^(d e) = w_L^(1024 d e)
                acc = cx_add(acc, cx_mul(z[e], w));
            }
            y[d] = cx_mul(acc, ft_tw(tabA, tabB, (2 * j * d) % (2 * N)));
        }
#pragma unroll
        for (int d = 0; d < 12; ++d) w12_st(Z + 512 * d + j, y[d]);
    }
    __syncthreads();
    // three radix-8 DIF passes on each of the 12 regions of 512 points, in place
    for (int ls = 6; ls >= 0; ls -= 3) {
        const int s = 1 << ls;
        for (int b = tid; b < 12 * 64; b += 512) {
            const int reg = b >> 6, bb = b & 63;
            const int jj = bb & (s - 1), blk = bb >> ls;
            double2 *base = Z + 512 * reg + (blk << (ls + 3)) + jj;
            Cx a[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = w12_ld(base + (e << ls));
            // w_len^jj, len = 8 s: w_L^(jj L / len)
            f2_bfly8(a, ft_tw(tabA, tabB, jj * (2 * N / (8 * s))), ls > 0);
#pragma unroll
            for (int e = 0; e < 8; ++e) w12_st(base + (e << ls), a[e]);
        }
        __syncthreads();
    }
    const double inv = 1.0 / (double)N;
    double *out = corr + (size_t)blockIdx.x * n_lags;
    for (int k = tid; k < n_lags; k += 512) {
        const int n = k >> 1, d = n % 12, kp = n / 12;  // frequency n = d + 12 kp, kp = k2 + 8 k1 + 64 k0 at 64 k2 + 8 k1 + k0
        const double2 zz = Z[512 * d + 64 * (kp & 7) + 8 * ((kp >> 3) & 7) + (kp >> 6)];
        out[k] = (k & 1) ? -zz.y * inv : zz.x * inv;
    }
}
