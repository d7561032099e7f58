// tools/ubench_valu.hip — issue cost (shader cycles per wave64 instruction per SIMD) of the VALU ops the pair
// kernels are made of, at 1, 2, 3, 4, 6 and 8 resident waves per SIMD, measured on the box:
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ub tools/ubench_valu.hip && /tmp/ub [json-path]
//
// Method. Every kernel runs N_IT iterations of 16 independent dependency chains of ONE instruction (inline asm, so
// the compiler can neither fuse nor remove them). Occupancy is pinned with dynamic LDS: a 256-thread block is one
// wave per SIMD, and a block that asks for floor(160 KiB / k) bytes lets exactly k blocks share a CU; the grid is
// 256 CUs x k blocks, one resident round. Each wave stamps s_memtime (shader clock) and s_memrealtime (100 MHz)
// around its loop, so cycles per instruction do not depend on an assumed clock:
//
//   cycles/inst/SIMD = median over waves of dt_memtime / (k * N_IT * 16)        (the k waves of a SIMD run together)
//   clock            = dt_memtime / dt_memrealtime * 100 MHz
//
// and the HIP-event time of the launch is printed beside it as a cross-check. The "mix" kernels replay the VALU
// sequence of the packed-f32 pair loop (pair_sj.hip, sweep_group_pk + bin_pair) for two j atoms, without its memory
// operations and branches: their rate is the issue roof bench.py prices the pair kernel against.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

constexpr int CHAINS = 16;

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Stamp {
    unsigned long long cyc, real;
};

__device__ __forceinline__ void stamp(unsigned long long &c, unsigned long long &r)
{
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c), "=s"(r)::"memory");
}

#define STORE_STAMP()                                                                           \
    if ((threadIdx.x & 63) == 0) {                                                              \
        Stamp s{c1 - c0, r1 - r0};                                                              \
        out[(size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = s;                   \
    }

// One loop iteration = ONE asm statement of 16 independent instructions (hipcc pads separate inline-asm statements
// with s_nop, which would be measured too). OP(i) expands to the instruction text for chain i; operand 16 is `vb`.
#define REP16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)
#define CHAIN_OPERANDS(r)                                                                                         \
    "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]),     \
        "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])

// TYPE of the chain registers, BTYPE/BINIT/BCON of the second operand (`vb`, constraint "v" or "s")
#define DEF_K(NAME, TYPE, INIT, BTYPE, BINIT, BCON, OP, SUM)                                    \
    __global__ __launch_bounds__(256) void NAME(Stamp *out, double a, double b, unsigned long long sb, double *sink, int n_it) \
    {                                                                                           \
        extern __shared__ unsigned char lds[];                                                  \
        TYPE r[CHAINS];                                                                         \
        BTYPE vb = BINIT;                                                                       \
        for (int i = 0; i < CHAINS; ++i) r[i] = INIT(a + i + threadIdx.x);                      \
        unsigned long long c0, r0, c1, r1;                                                      \
        stamp(c0, r0);                                                                          \
        for (int it = 0; it < n_it; ++it) asm volatile(REP16(OP) : CHAIN_OPERANDS(r) : BCON(vb) : "vcc"); \
        stamp(c1, r1);                                                                          \
        STORE_STAMP()                                                                           \
        double s = 0;                                                                           \
        for (int i = 0; i < CHAINS; ++i) s += (double)(SUM(r[i]));                              \
        if (s == 12345.678) sink[0] = s + lds[0];                                               \
    }
#define INIT_D(x) (x)
#define INIT_F(x) ((float)(x))
#define INIT_P(x) f32x2{(float)(x), (float)(x) + 0.25f}
#define SUM_S(x) (x)
#define SUM_P(x) (x[0] + x[1])
#define K64(NAME, OP) DEF_K(NAME, double, INIT_D, double, b, "v", OP, SUM_S)
#define KPK(NAME, OP) DEF_K(NAME, f32x2, INIT_P, f32x2, INIT_P(b), "v", OP, SUM_P)
#define KPKS(NAME, OP) DEF_K(NAME, f32x2, INIT_P, unsigned long long, sb, "s", OP, SUM_P)
#define K32(NAME, OP) DEF_K(NAME, float, INIT_F, float, (float)b, "v", OP, SUM_S)
#define K32S(NAME, OP) DEF_K(NAME, float, INIT_F, unsigned, (unsigned)sb, "s", OP, SUM_S)

#define OP_ADD_F64(i) "v_add_f64 %" #i ", %" #i ", %16\n\t"
#define OP_MUL_F64(i) "v_mul_f64 %" #i ", %" #i ", %16\n\t"
#define OP_FMA_F64(i) "v_fma_f64 %" #i ", %" #i ", %16, %16\n\t"
#define OP_MIN_F64(i) "v_min_f64 %" #i ", |%" #i "|, |%16|\n\t"
#define OP_PK_ADD(i) "v_pk_add_f32 %" #i ", %" #i ", %16\n\t"
#define OP_PK_ADD_S(i) "v_pk_add_f32 %" #i ", %" #i ", %16 neg_lo:[0,1] neg_hi:[0,1]\n\t"
#define OP_PK_MUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %16\n\t"
#define OP_PK_FMA(i) "v_pk_fma_f32 %" #i ", %16, %16, %" #i "\n\t"
#define OP_ADD_F32(i) "v_add_f32 %" #i ", %" #i ", %16\n\t"
#define OP_SUB_F32(i) "v_sub_f32 %" #i ", %" #i ", %16\n\t"
#define OP_MUL_F32(i) "v_mul_f32 %" #i ", %" #i ", %16\n\t"
#define OP_FMA_F32(i) "v_fma_f32 %" #i ", %" #i ", %16, %16\n\t"
#define OP_FMAC_F32(i) "v_fmac_f32 %" #i ", %16, %16\n\t"
#define OP_SQRT_F32(i) "v_sqrt_f32 %" #i ", %" #i "\n\t"
#define OP_FRACT_F32(i) "v_fract_f32 %" #i ", %" #i "\n\t"
#define OP_CVT_I32(i) "v_cvt_i32_f32 %" #i ", %" #i "\n\t"
#define OP_LSHL_ADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %16\n\t"
#define OP_CMP_F32(i) "v_cmp_gt_f32 vcc, %" #i ", %16\n\t"
#define OP_CMP_F32_S(i) "v_cmp_gt_f32 vcc, %16, %" #i "\n\t"

K64(k_add_f64, OP_ADD_F64)
K64(k_mul_f64, OP_MUL_F64)
K64(k_fma_f64, OP_FMA_F64)
K64(k_min_f64_abs, OP_MIN_F64)
KPK(k_pk_add_f32, OP_PK_ADD)
KPKS(k_pk_add_f32_s, OP_PK_ADD_S)
KPK(k_pk_mul_f32, OP_PK_MUL)
KPK(k_pk_fma_f32, OP_PK_FMA)
K32(k_add_f32, OP_ADD_F32)
K32(k_sub_f32, OP_SUB_F32)
K32S(k_sub_f32_s, OP_SUB_F32)
K32(k_mul_f32, OP_MUL_F32)
K32(k_fma_f32, OP_FMA_F32)
K32(k_fmac_f32, OP_FMAC_F32)
K32(k_sqrt_f32, OP_SQRT_F32)
K32(k_fract_f32, OP_FRACT_F32)
K32(k_cvt_i32_f32, OP_CVT_I32)
K32(k_lshl_add, OP_LSHL_ADD)
K32(k_cmp_f32, OP_CMP_F32)
K32S(k_cmp_f32_s, OP_CMP_F32_S)

// The VALU stream of the packed pair loop for one half group (two j atoms against 64 i atoms), as pair_sj.hip issues
// it: 3 v_pk_add (i - j, the j atoms in SGPR pairs), v_pk_mul, 2 v_pk_fma, then per pair the cutoff compare and — for
// BIN of every 16 slots, the share of (wave, pair) slots with some lane inside the cutoff — the six instructions of
// the bin guess (bin_pair). No exec masking, LDS add or branch here: this is the issue roof of the arithmetic alone.
// One asm statement per half group.
// (hard registers for the temporaries: a packed result is a 64-bit pair whose halves the compares address singly)
#define MIX_HEAD                                                        \
    "v_pk_add_f32 v[40:41], %[x2], %[sx] neg_lo:[0,1] neg_hi:[0,1]\n\t" \
    "v_pk_add_f32 v[42:43], %[y2], %[sy] neg_lo:[0,1] neg_hi:[0,1]\n\t" \
    "v_pk_add_f32 v[44:45], %[z2], %[sz] neg_lo:[0,1] neg_hi:[0,1]\n\t" \
    "v_pk_mul_f32 v[40:41], v[40:41], v[40:41]\n\t"                     \
    "v_pk_fma_f32 v[40:41], v[42:43], v[42:43], v[40:41]\n\t"           \
    "v_pk_fma_f32 v[40:41], v[44:45], v[44:45], v[40:41]\n\t"
// (the pair block masks with v_cmpx, VOP3 with an SGPR-pair destination; here the compares are always true — rc = +inf,
// n2 = -1 — so that exec stays whole without a restoring scalar instruction)
#define MIX_CMP(R) "v_cmpx_gt_f32_e64 s[60:61], %[rc], " R "\n\t"
#define MIX_BIN(R)                              \
    "v_sqrt_f32 v46, " R "\n\t"                 \
    "s_nop 0\n\t"                               \
    "v_fma_f32 v46, v46, %[gs], %[no]\n\t"      \
    "v_fract_f32 v47, v46\n\t"                  \
    "v_cvt_i32_f32 v46, v46\n\t"                \
    "v_lshl_add_u32 %[acc], v46, 2, %[acc]\n\t" \
    "v_cmpx_ge_f32_e64 s[62:63], v47, %[n2]\n\t"
#define MIX_OPERANDS                                                                                                  \
    [acc] "+v"(acc)                                                                                                   \
        : [x2] "v"(x2), [y2] "v"(y2), [z2] "v"(z2), [sx] "s"(sx), [sy] "s"(sy), [sz] "s"(sz), [rc] "s"(rc), [gs] "v"(gs), \
          [no] "s"(no), [n2] "v"(n2)                                                                                  \
        : "vcc", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "s60", "s61", "s62", "s63"

// NB = how many of the 16 pair slots of an iteration (8 half groups) run the bin guess: 0, 11 (= 0.69, the C2 share) or 16
template <int NB>
__global__ __launch_bounds__(256) void k_mix(Stamp *out, double a, double b, unsigned long long sb, double *sink, int n_it)
{
    extern __shared__ unsigned char lds[];
    f32x2 x2 = INIT_P(a + threadIdx.x), y2 = INIT_P(a + 1.0 + threadIdx.x), z2 = INIT_P(a + 2.0 + threadIdx.x);
    const unsigned long long sx = sb, sy = sb + 0x0000100000001000ull, sz = sb + 0x0000200000002000ull;
    const unsigned rc = 0x7f800000u, no = 0x3f000000u;  // +inf, 0.5f as SGPR operands
    float gs = (float)b, n2 = -1.0f;
    unsigned acc = threadIdx.x;
    unsigned long long c0, r0, c1, r1;
    stamp(c0, r0);
    for (int it = 0; it < n_it; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int nb = (2 * i < NB ? 1 : 0) + (2 * i + 1 < NB ? 1 : 0);
            if (nb == 2)
                asm volatile(MIX_HEAD MIX_CMP("v40") MIX_BIN("v40") MIX_CMP("v41") MIX_BIN("v41") : MIX_OPERANDS);
            else if (nb == 1)
                asm volatile(MIX_HEAD MIX_CMP("v40") MIX_BIN("v40") MIX_CMP("v41") : MIX_OPERANDS);
            else
                asm volatile(MIX_HEAD MIX_CMP("v40") MIX_CMP("v41") : MIX_OPERANDS);
        }
    }
    stamp(c1, r1);
    STORE_STAMP()
    if (acc == 12345u) sink[0] = acc + lds[0];
}

struct Result {
    std::string name;
    int waves;
    double cyc, clock_ghz, ginst_per_s;
};

// insts_per_it: VALU wave-instructions per loop iteration of one wave. The iteration count is chosen so that a launch
// lasts ~1 ms at every occupancy (launch ramp and tail well under 1 %); the launch is timed with HIP events and the
// shader clock it ran at comes from the waves' own s_memtime / s_memrealtime stamps:
//   cycles/inst/SIMD = t_launch * clock / (k * n_it * insts_per_it)
template <typename K>
Result run(K kern, const char *name, int k, double insts_per_it, Stamp *d_out, double *d_sink)
{
    const int blocks = 256 * k;
    const size_t lds = (size_t)(163840 / k) & ~size_t(1023);  // exactly k blocks fit a CU
    const int n_it = (int)(2.4e6 / 4.3 / (k * insts_per_it));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const unsigned long long sb = 0x3f8000013f800000ull;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d_out, 1.5, 0.999999, sb, d_sink, n_it);
    (void)hipDeviceSynchronize();
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d_out, 1.5, 0.999999, sb, d_sink, n_it);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, (double)ms);
    }
    std::vector<Stamp> h((size_t)blocks * 4);
    (void)hipMemcpy(h.data(), d_out, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> clk;
    for (auto &s : h) clk.push_back((double)s.cyc / (double)s.real * 0.1);
    std::sort(clk.begin(), clk.end());
    Result r;
    r.name = name;
    r.waves = k;
    r.clock_ghz = clk[clk.size() / 2];
    const double insts_per_simd = (double)k * n_it * insts_per_it;
    r.ginst_per_s = insts_per_simd / (best * 1e-3) / 1e9;
    r.cyc = r.clock_ghz / r.ginst_per_s;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return r;
}

int main(int argc, char **argv)
{
    Stamp *d_out;
    double *d_sink;
    (void)hipMalloc(&d_out, sizeof(Stamp) * 256 * 8 * 4);
    (void)hipMalloc(&d_sink, 64);
    const int occ[6] = {1, 2, 3, 4, 6, 8};  // (3: the round-5 full-lag MSD kernel, twelve waves per CU)
    std::vector<Result> all;
#define RUN(K, NAME)                                                  \
    for (int k : occ) all.push_back(run(K, NAME, k, 16.0, d_out, d_sink));
    RUN(k_pk_add_f32, "v_pk_add_f32")
    RUN(k_pk_add_f32_s, "v_pk_add_f32 v,-s")
    RUN(k_pk_mul_f32, "v_pk_mul_f32")
    RUN(k_pk_fma_f32, "v_pk_fma_f32")
    RUN(k_add_f32, "v_add_f32")
    RUN(k_sub_f32, "v_sub_f32")
    RUN(k_sub_f32_s, "v_sub_f32 v,s")
    RUN(k_mul_f32, "v_mul_f32")
    RUN(k_fma_f32, "v_fma_f32")
    RUN(k_fmac_f32, "v_fmac_f32")
    RUN(k_cmp_f32, "v_cmp_gt_f32")
    RUN(k_cmp_f32_s, "v_cmp_gt_f32 s,v")
    RUN(k_sqrt_f32, "v_sqrt_f32")
    RUN(k_fract_f32, "v_fract_f32")
    RUN(k_cvt_i32_f32, "v_cvt_i32_f32")
    RUN(k_lshl_add, "v_lshl_add_u32")
    RUN(k_add_f64, "v_add_f64")
    RUN(k_mul_f64, "v_mul_f64")
    RUN(k_fma_f64, "v_fma_f64")
    RUN(k_min_f64_abs, "v_min_f64 |a|,|b|")
    // mix kernels: VALU instructions per wave per iteration = 8 half groups x 6 packed + 16 compares + NB x 6
    for (int k : occ) all.push_back(run(k_mix<0>, "mix bin 0/16", k, 48 + 16, d_out, d_sink));
    for (int k : occ) all.push_back(run(k_mix<11>, "mix bin 11/16", k, 48 + 16 + 66, d_out, d_sink));
    for (int k : occ) all.push_back(run(k_mix<16>, "mix bin 16/16", k, 48 + 16 + 96, d_out, d_sink));

    printf("%-20s %5s %14s %10s %22s\n", "instruction", "waves", "cyc/inst/SIMD", "clock GHz", "G wave-inst/s per SIMD");
    for (auto &r : all)
        printf("%-20s %5d %14.2f %10.3f %22.4f\n", r.name.c_str(), r.waves, r.cyc, r.clock_ghz, r.ginst_per_s);
    if (argc > 1) {
        FILE *f = fopen(argv[1], "w");
        if (f) {
            fprintf(f, "{\n \"method\": \"tools/ubench_valu.hip: wave64 VALU instructions per second per SIMD (HIP events over a ~1 ms launch, k waves per SIMD pinned with LDS), cycles = in-kernel clock (s_memtime / s_memrealtime) / rate\",\n \"results\": [\n");
            for (size_t i = 0; i < all.size(); ++i)
                fprintf(f, "  {\"inst\": \"%s\", \"waves_per_simd\": %d, \"cycles\": %.3f, \"clock_ghz\": %.3f, \"ginst_per_s_per_simd\": %.5f}%s\n",
                        all[i].name.c_str(), all[i].waves, all[i].cyc, all[i].clock_ghz, all[i].ginst_per_s, i + 1 < all.size() ? "," : "");
            fprintf(f, " ]\n}\n");
            fclose(f);
        }
    }
    return 0;
}
