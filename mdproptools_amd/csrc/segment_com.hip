// segment_com.hip — per-molecule centre of mass and per-frame charge flux (R6, M3, G1).
//
// Replaces structural/rdf_cn.py:218-241 (_define_mol_cols), common/com_mols.py:58-60 (calc_com),
// dynamical/diffusion.py:83-89 and dynamical/_conductivity.py:11-35 of the reference.
// HBM-bound: 8*(1+n_attr) bytes read per atom, 8*n_attr written per molecule.
//
// Rounding: every product m*a and every addition is its own IEEE operation (contraction off), atoms
// are added in index order, then one correctly rounded division by the mass sum. The reference's
// pandas/BLAS sums use another order, so agreement is ~1e-15 relative, not bitwise.
#include <algorithm>
#include <vector>

#include <cmath>

#include "ctx.h"

#pragma clang fp contract(off)

namespace {

// Staged kernel (default). A block owns a run of whole segments with at most SC_CAP atoms in total and
// loops over frames and attribute planes: the run's atoms are read with coalesced 8-byte loads,
// multiplied by their mass and parked in LDS; then one lane per segment adds its atoms up in index
// order from LDS and divides by the mass sum. Same operations in the same order as one lane walking
// its segment in global memory (the fallback below, kept for segments longer than SC_CAP), but every
// HBM byte is fetched once by a full-width load.
struct SegBlock {
    long long s0, s1;  // segments [s0, s1)
};
constexpr int SC_CAP_MAX = 1024;   // atoms per block stage (CAP of the kernel: 1024 or 512)
constexpr int SC_PLANES = 3;   // attribute planes staged together (x, y, z): one barrier pair per frame
// LDS index with one pad double per 32: lanes whose segments start 4, 8, 16, ... atoms apart would otherwise
// all hit the same banks
__device__ __forceinline__ int sc_pad(int i) { return i + (i >> 5); }
constexpr int sc_stride(int cap) { return cap + cap / 32 + 2; }

// FLUX = false: out[f][k][s] = sum(m a) / sum(m)                                   (com_mols.py:58-60)
// FLUX = true : out[f][k][s] = (sum(m v) / sum(m) * vel_conv) * (sum(q) * charge_conv)   (_conductivity.py:21-25)
// NT = threads per block: 256 (four waves share a stage of SC_CAP atoms, two block barriers per step) or 64 — ONE wave
// per block with a stage of 256 atoms: the barriers then cost nothing and every wave runs its own pipeline, decoupled
// from its neighbours (the wave-private form; default whenever no segment is longer than 256 atoms).
template <bool FLUX, int SC_CAP, int NT>
__global__ __launch_bounds__(NT) void segment_staged_kernel(
    const double *__restrict__ attr, const double *__restrict__ mass, const double *__restrict__ q,
    const long long *__restrict__ seg_off, const SegBlock *__restrict__ blocks, double *__restrict__ out,
    int n_attr, long long n_atoms, long long n_seg, long long n_frames, double vel_conv, double charge_conv, int use_vec)
{
    constexpr int SC_STRIDE = sc_stride(SC_CAP);
    __shared__ double s_v[SC_PLANES * SC_STRIDE];
    __shared__ double s_m[SC_CAP];
    const int tid = threadIdx.x;
    const SegBlock b = blocks[blockIdx.x];
    const long long a0 = seg_off[b.s0];
    const int na = (int)(seg_off[b.s1] - a0);
    for (int i = tid; i < na; i += NT) s_m[i] = mass[a0 + i];
    // Sum tasks: (segment, plane) pairs of the block, dealt plane-major over the 256 lanes — consecutive lanes take
    // consecutive segments of one plane (coalesced stores, spread LDS banks), and a block of 64 sixteen-atom molecules
    // keeps 192 lanes busy in the sum phase instead of 64 (SC_TPL tasks per lane cover 256 segments x 3 planes).
    constexpr int SC_TPL = SC_PLANES;
    const int nseg = (int)(b.s1 - b.s0);
    int t_lo[SC_TPL], t_hi[SC_TPL], t_kk[SC_TPL], t_seg[SC_TPL];
    double t_msum[SC_TPL], t_qsi[SC_TPL];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SC_TPL; ++j) {
        const int t = tid + j * NT;
        const int kk = t / nseg, sg = t - kk * nseg;
        t_kk[j] = kk < SC_PLANES ? kk : -1;
        t_seg[j] = sg;
        t_lo[j] = t_hi[j] = 0;
        t_msum[j] = 1.0;
        t_qsi[j] = 0.0;
        if (t_kk[j] >= 0) {
            t_lo[j] = (int)(seg_off[b.s0 + sg] - a0);
            t_hi[j] = (int)(seg_off[b.s0 + sg + 1] - a0);
            double msum = 0.0;
            for (int a = t_lo[j]; a < t_hi[j]; ++a) msum += s_m[a];
            t_msum[j] = msum;
            if (FLUX) {
                double qsum = 0.0;
                for (int a = t_lo[j]; a < t_hi[j]; ++a) qsum += q[a0 + a];
                t_qsi[j] = qsum * charge_conv;  // _conductivity.py:25
            }
        }
    }
    // Software pipeline over (frame, plane group) steps: the values of step n+1 are loaded into registers while
    // the segment sums of step n run out of LDS, so that HBM requests are in flight all the time.
    constexpr int PER = SC_CAP / NT;  // atoms per lane per plane
    const int groups = (n_attr + SC_PLANES - 1) / SC_PLANES;
    const long long n_my = blockIdx.y < n_frames ? (n_frames - blockIdx.y + gridDim.y - 1) / gridDim.y : 0;
    const long long steps = n_my * groups;
    // The loads of step n+1 are issued right after the barrier that ends the LDS writes of step n and land while
    // its sums run. (A second register set, fetching at the very top of the step, measured slower: 102 VGPRs.)
    // A lane owns the atoms (2 tid, 2 tid + 1) + 512 r: 16-byte loads when the planes allow it (even atom count and
    // block start, 16-byte aligned base) — half the load instructions of 8-byte loads.
    double v[SC_PLANES][PER];
    const bool vec2 = use_vec && ((n_atoms | a0) & 1LL) == 0 && ((reinterpret_cast<unsigned long long>(attr) & 15ULL) == 0);
    auto fetch = [&](long long step) {
        const long long f = blockIdx.y + (step / groups) * gridDim.y;
        const int k0 = (int)(step % groups) * SC_PLANES;
        const double *p = attr + ((size_t)f * n_attr + k0) * n_atoms + a0;
#pragma unroll
        for (int kk = 0; kk < SC_PLANES; ++kk)
#pragma unroll
            for (int r = 0; r < PER / 2; ++r) {
                const int i = 2 * tid + r * (2 * NT);
                const bool ok = k0 + kk < n_attr;
                if (!use_vec) {  // A/B: the 8-byte layout (lane owns tid + 256 r)
                    const int i0 = tid + (2 * r) * NT, i1 = tid + (2 * r + 1) * NT;
                    v[kk][2 * r] = (ok && i0 < na) ? p[(size_t)kk * n_atoms + i0] : 0.0;
                    v[kk][2 * r + 1] = (ok && i1 < na) ? p[(size_t)kk * n_atoms + i1] : 0.0;
                    continue;
                }
                if (vec2 && ok && i + 1 < na) {
                    typedef double d2_t __attribute__((ext_vector_type(2)));
                    const d2_t t2 = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(p + (size_t)kk * n_atoms + i));
                    v[kk][2 * r] = t2[0];
                    v[kk][2 * r + 1] = t2[1];
                } else {
                    v[kk][2 * r] = (ok && i < na) ? p[(size_t)kk * n_atoms + i] : 0.0;
                    v[kk][2 * r + 1] = (ok && i + 1 < na) ? p[(size_t)kk * n_atoms + i + 1] : 0.0;
                }
            }
    };
    auto process = [&](long long step) {
        const long long f = blockIdx.y + (step / groups) * gridDim.y;
        const int k0 = (int)(step % groups) * SC_PLANES;
        const int nk = n_attr - k0 < SC_PLANES ? n_attr - k0 : SC_PLANES;
        __syncthreads();  // the previous step's sums have been taken
#pragma unroll
        for (int kk = 0; kk < SC_PLANES; ++kk)
#pragma unroll
            for (int r = 0; r < PER; ++r) {
                const int i = use_vec ? 2 * tid + (r >> 1) * (2 * NT) + (r & 1) : tid + r * NT;
                if (i < na) s_v[kk * SC_STRIDE + sc_pad(i)] = v[kk][r] * s_m[i];
            }
        __syncthreads();
        if (step + 1 < steps) fetch(step + 1);
#pragma unroll
        for (int j = 0; j < SC_TPL; ++j) {
            const int kk = t_kk[j];
            if (kk >= 0 && kk < nk) {
                // atoms added in index order (one rounding per addition, as a lane walking the segment would);
                // four LDS reads are issued before their additions so that the reads overlap
                const double *row = s_v + kk * SC_STRIDE;
                double acc = 0.0;
                int a = t_lo[j];
                for (; a + 4 <= t_hi[j]; a += 4) {
                    const double x0 = row[sc_pad(a)], x1 = row[sc_pad(a + 1)], x2 = row[sc_pad(a + 2)],
                                 x3 = row[sc_pad(a + 3)];
                    acc += x0;
                    acc += x1;
                    acc += x2;
                    acc += x3;
                }
                for (; a < t_hi[j]; ++a) acc += row[sc_pad(a)];
                double r = acc / t_msum[j];
                if (FLUX) r = (r * vel_conv) * t_qsi[j];  // com_mols.py:60 then _conductivity.py:21-23
                out[((size_t)f * n_attr + k0 + kk) * n_seg + b.s0 + t_seg[j]] = r;
            }
        }
    };
    if (steps > 0) fetch(0);
    for (long long step = 0; step < steps; ++step) process(step);
}

// One (run of segments, frame, plane group) per block and nothing carried between frames (round 3; option seg_frame):
// the run's table entry holds its atom range, the per-segment mass sums come from the host's index-order sums
// (mdhip_segment_com computes them anyway), the masses are re-read per block (L2 hits: 8 B per atom against 24 B of
// coordinates from HBM). No software pipeline, no register double buffer: the loads of many small blocks overlap
// instead, as in msd_pairs_kernel. Same products, same additions in the same order, one division.
struct SegRun {
    long long s0, s1, a0;
    int na, pad_;
};

// (88 / 100 VGPRs: 5 / 4 blocks per CU where LDS would hold 6; forcing 80 registers spills 3 / 11 of them and measured
// 5 % / 20 % slower.)
template <bool FLUX, int SC_CAP>
__global__ __launch_bounds__(256) void segment_frame_kernel(
    const double *__restrict__ attr, const double *__restrict__ mass, const double *__restrict__ seg_msum,
    const double *__restrict__ seg_qsi, const long long *__restrict__ seg_off, const SegRun *__restrict__ runs,
    double *__restrict__ out, int n_attr, long long n_atoms, long long n_seg, long long n_steps, double vel_conv)
{
    constexpr int NT = 256, SC_STRIDE = sc_stride(SC_CAP), PER = SC_CAP / NT;
    __shared__ double s_v[SC_PLANES * SC_STRIDE];
    const int tid = threadIdx.x;
    const SegRun b = runs[blockIdx.x];
    const int na = b.na, nseg = (int)(b.s1 - b.s0);
    const int groups = (n_attr + SC_PLANES - 1) / SC_PLANES;
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const bool vec2 = ((n_atoms | b.a0) & 1LL) == 0 &&
                      ((reinterpret_cast<unsigned long long>(attr) | reinterpret_cast<unsigned long long>(mass)) & 15ULL) == 0;
    for (long long step = blockIdx.y; step < n_steps; step += gridDim.y) {
        const long long f = step / groups;
        const int k0 = (int)(step % groups) * SC_PLANES;
        const int nk = n_attr - k0 < SC_PLANES ? n_attr - k0 : SC_PLANES;
        const double *p = attr + ((size_t)f * n_attr + k0) * n_atoms + b.a0;
        double v[SC_PLANES][PER], mm[PER];
#pragma unroll
        for (int r = 0; r < PER / 2; ++r) {
            const int i = 2 * tid + r * (2 * NT);
#pragma unroll
            for (int kk = 0; kk < SC_PLANES; ++kk) {
                v[kk][2 * r] = v[kk][2 * r + 1] = 0.0;
                if (kk < nk) {
                    if (vec2 && i + 1 < na) {
                        const d2_t t2 = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(p + (size_t)kk * n_atoms + i));
                        v[kk][2 * r] = t2[0];
                        v[kk][2 * r + 1] = t2[1];
                    } else {
                        if (i < na) v[kk][2 * r] = p[(size_t)kk * n_atoms + i];
                        if (i + 1 < na) v[kk][2 * r + 1] = p[(size_t)kk * n_atoms + i + 1];
                    }
                }
            }
            mm[2 * r] = mm[2 * r + 1] = 0.0;
            if (vec2 && i + 1 < na) {
                const d2_t t2 = *reinterpret_cast<const d2_t *>(mass + b.a0 + i);
                mm[2 * r] = t2[0];
                mm[2 * r + 1] = t2[1];
            } else {
                if (i < na) mm[2 * r] = mass[b.a0 + i];
                if (i + 1 < na) mm[2 * r + 1] = mass[b.a0 + i + 1];
            }
        }
        // the (segment, plane) sum tasks of this lane, plane-major over the lanes as in segment_staged_kernel
        int t_lo[SC_PLANES], t_hi[SC_PLANES], t_kk[SC_PLANES], t_seg[SC_PLANES];
        double t_msum[SC_PLANES], t_qsi[SC_PLANES];
#pragma unroll
        for (int j = 0; j < SC_PLANES; ++j) {
            const int t = tid + j * NT;
            const int kk = t / nseg, sg = t - kk * nseg;
            t_kk[j] = kk < nk ? kk : -1;
            t_seg[j] = sg;
            t_lo[j] = t_hi[j] = 0;
            t_msum[j] = 1.0;
            t_qsi[j] = 0.0;
            if (t_kk[j] >= 0) {
                t_lo[j] = (int)(seg_off[b.s0 + sg] - b.a0);
                t_hi[j] = (int)(seg_off[b.s0 + sg + 1] - b.a0);
                t_msum[j] = seg_msum[b.s0 + sg];
                if (FLUX) t_qsi[j] = seg_qsi[b.s0 + sg];
            }
        }
#pragma unroll
        for (int kk = 0; kk < SC_PLANES; ++kk)
#pragma unroll
            for (int r = 0; r < PER; ++r) {
                const int i = 2 * tid + (r >> 1) * (2 * NT) + (r & 1);
                if (i < na) s_v[kk * SC_STRIDE + sc_pad(i)] = v[kk][r] * mm[r];
            }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SC_PLANES; ++j) {
            const int kk = t_kk[j];
            if (kk >= 0) {
                const double *row = s_v + kk * SC_STRIDE;
                double acc = 0.0;
                int a = t_lo[j];
                for (; a + 4 <= t_hi[j]; a += 4) {
                    const double x0 = row[sc_pad(a)], x1 = row[sc_pad(a + 1)], x2 = row[sc_pad(a + 2)],
                                 x3 = row[sc_pad(a + 3)];
                    acc += x0;
                    acc += x1;
                    acc += x2;
                    acc += x3;
                }
                for (; a < t_hi[j]; ++a) acc += row[sc_pad(a)];
                double r = acc / t_msum[j];
                if (FLUX) r = (r * vel_conv) * t_qsi[j];
                // (non-temporal: a write stream of a tenth of the bytes read costs a plain read stream a quarter of its
                // rate with ordinary stores and a sixth with these — tools/ubench_hbm.hip, 0.65 against 0.75 of 8 TB/s)
                __builtin_nontemporal_store(r, out + ((size_t)f * n_attr + k0 + kk) * n_seg + b.s0 + t_seg[j]);
            }
        }
        if (step + gridDim.y < n_steps) __syncthreads();  // (only when the grid could not hold every step)
    }
}

// Runs of whole segments with <= SC_CAP atoms and <= 256 segments each; false when a segment is longer.
bool build_seg_blocks(int64_t n_seg, const int64_t *seg_off, std::vector<SegBlock> &blocks, int SC_CAP = SC_CAP_MAX,
                      int max_segs = 256)
{
    blocks.clear();
    int64_t s0 = 0;
    while (s0 < n_seg) {
        int64_t s1 = s0;
        while (s1 < n_seg && s1 - s0 < max_segs && seg_off[s1 + 1] - seg_off[s0] <= SC_CAP) ++s1;
        if (s1 == s0) return false;  // segment s0 alone exceeds the LDS stage
        blocks.push_back({(long long)s0, (long long)s1});
        s0 = s1;
    }
    return true;
}

// Stage size of the by-frame kernel for a segment table (round 5; measured at C4 size, tools/ab_seg_cap.py): 1024 atoms per
// block unless that leaves the block's second phase — one lane per (segment, plane) sum, 256 lanes a round — mostly idle
// rounds (10-atom molecules: 102 segments x 3 planes = 306 sums = two rounds for 1.2 rounds' worth: 0.515 -> 0.478 ms with
// 512-atom stages) or the stage mostly empty (3-atom molecules: the 256-segment limit fills 768 of 1024: 0.722 -> 0.69).
// Molecules of 4, 16, 40 atoms and mixes of them keep 1024 (512: 2-7 % slower).
int pick_seg_cap(int64_t n_seg, const int64_t *seg_off, int n_attr, std::vector<SegBlock> &blocks)
{
    if (!build_seg_blocks(n_seg, seg_off, blocks, SC_CAP_MAX, 256)) {
        blocks.clear();  // (a segment longer than the stage: the caller takes the one-lane-per-segment kernel)
        return SC_CAP_MAX;
    }
    const int planes = n_attr < SC_PLANES ? n_attr : SC_PLANES;
    double rounds = 0.0, tasks = 0.0, atoms = 0.0;
    for (const SegBlock &b : blocks) {
        const double t = (double)planes * (double)(b.s1 - b.s0);
        rounds += std::ceil(t / 256.0);
        tasks += t;
        atoms += (double)(seg_off[b.s1] - seg_off[b.s0]);
    }
    const bool idle_rounds = rounds > (double)blocks.size() && rounds * 256.0 > 1.5 * tasks;
    const bool empty_stage = atoms < 0.85 * (double)SC_CAP_MAX * (double)blocks.size() && blocks.size() > 1;
    if (!(idle_rounds || empty_stage)) return SC_CAP_MAX;
    std::vector<SegBlock> half;
    if (!build_seg_blocks(n_seg, seg_off, half, 512, 256)) return SC_CAP_MAX;
    blocks.swap(half);
    return 512;
}

// Fallback: one lane per (segment, frame) walking global memory
__global__ __launch_bounds__(256) void segment_com_kernel(
    const double *__restrict__ attr, const double *__restrict__ mass,
    const long long *__restrict__ seg_off, double *__restrict__ out, int n_attr, long long n_atoms,
    long long n_seg, long long n_frames)
{
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    const long long lo = seg_off[s], hi = seg_off[s + 1];
    double msum = 0.0;
    for (long long a = lo; a < hi; ++a) msum += mass[a];
    for (long long f = blockIdx.y; f < n_frames; f += gridDim.y) {
        for (int k = 0; k < n_attr; ++k) {
            const double *p = attr + ((size_t)f * n_attr + k) * n_atoms;
            double acc = 0.0;
            for (long long a = lo; a < hi; ++a) acc += p[a] * mass[a];
            out[((size_t)f * n_attr + k) * n_seg + s] = acc / msum;
        }
    }
}

// q_mol * v_com,k in SI for every molecule: tmp [F][3][M] (fallback, as above)
__global__ __launch_bounds__(256) void mol_flux_kernel(
    const double *__restrict__ vel, const double *__restrict__ mass, const double *__restrict__ q,
    const long long *__restrict__ seg_off, double *__restrict__ tmp, long long n_atoms,
    long long n_seg, long long n_frames, double vel_conv, double charge_conv)
{
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    const long long lo = seg_off[s], hi = seg_off[s + 1];
    double msum = 0.0, qsum = 0.0;
    for (long long a = lo; a < hi; ++a) {
        msum += mass[a];
        qsum += q[a];
    }
    const double q_si = qsum * charge_conv;  // _conductivity.py:25
    for (long long f = blockIdx.y; f < n_frames; f += gridDim.y) {
        for (int k = 0; k < 3; ++k) {
            const double *p = vel + ((size_t)f * 3 + k) * n_atoms;
            double acc = 0.0;
            for (long long a = lo; a < hi; ++a) acc += p[a] * mass[a];
            const double v_si = (acc / msum) * vel_conv;  // com_mols.py:60 then _conductivity.py:21-23
            tmp[((size_t)f * 3 + k) * n_seg + s] = v_si * q_si;
        }
    }
}

// J[k][type][f] = sum over the molecules of that type (a contiguous run) — fixed-order tree sum.
__global__ __launch_bounds__(256) void type_sum_kernel(const double *__restrict__ tmp,
                                                       const long long *__restrict__ type_off,
                                                       double *__restrict__ flux, long long n_seg,
                                                       long long n_frames, int n_types)
{
    __shared__ double red[256];
    const int t = blockIdx.x, k = blockIdx.y;
    const long long f = blockIdx.z;
    const long long lo = type_off[t], hi = type_off[t + 1];
    const double *p = tmp + ((size_t)f * 3 + k) * n_seg;
    double acc = 0.0;
    for (long long s = lo + threadIdx.x; s < hi; s += 256) acc += p[s];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) flux[((size_t)k * n_types + t) * n_frames + f] = red[0];
}

int check_segments(mdhip_ctx *ctx, int64_t n_atoms, int64_t n_seg, const int64_t *seg_off)
{
    MD_REQUIRE(n_seg >= 0 && (n_seg == 0 || seg_off), "bad segment table");
    for (int64_t s = 0; s < n_seg; ++s)
        MD_REQUIRE(seg_off[s] >= 0 && seg_off[s] < seg_off[s + 1] && seg_off[s + 1] <= n_atoms,
                   "seg_off must be increasing and within [0, n_atoms] (segment %lld)", (long long)s);
    return MDHIP_OK;
}

}  // namespace

extern "C" {

int mdhip_segment_com(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, int n_attr,
                      const double *attr, int attr_on_device, const double *atom_mass,
                      const double *atom_q, int64_t n_seg, const int64_t *seg_off, double *out,
                      int out_on_device, double *seg_mass, double *seg_q)
{
    if (!ctx) return MDHIP_EINVAL;
    CallScope cs(ctx);
    MD_REQUIRE(n_frames >= 0 && n_atoms >= 0 && n_attr >= 0, "negative sizes");
    MD_REQUIRE(atom_mass || n_atoms == 0, "atom_mass is NULL");
    int rc = check_segments(ctx, n_atoms, n_seg, seg_off);
    if (rc) return rc;
    // sums that do not depend on the frame are done on the host, in index order
    std::vector<double> msum_h((size_t)n_seg);
    for (int64_t s = 0; s < n_seg; ++s) {
        double m = 0.0, q = 0.0;
        for (int64_t a = seg_off[s]; a < seg_off[s + 1]; ++a) {
            m += atom_mass[a];
            if (atom_q) q += atom_q[a];
        }
        msum_h[s] = m;
        if (seg_mass) seg_mass[s] = m;
        if (seg_q && atom_q) seg_q[s] = q;
    }
    if (n_frames == 0 || n_seg == 0 || n_attr == 0) return cs.end();
    MD_REQUIRE(attr && out, "NULL attr/out");
    MD_HIP(hipSetDevice(ctx->device));
    const double *d_attr = (const double *)mdhip_stage(ctx, WS_AUX0, attr,
                                                       (size_t)n_frames * n_attr * n_atoms * 8,
                                                       attr_on_device, &rc);
    if (rc) return rc;
    MD_WS(d_mass, double, WS_AUX1, (size_t)n_atoms * 8);
    rc = mdhip_h2d_small(ctx, d_mass, atom_mass, (size_t)n_atoms * 8);
    if (rc) return rc;
    MD_WS(d_off, long long, WS_AUX2, (size_t)(n_seg + 1) * 8);
    rc = mdhip_h2d_small(ctx, d_off, seg_off, (size_t)(n_seg + 1) * 8);
    if (rc) return rc;
    const size_t out_b = (size_t)n_frames * n_attr * n_seg * 8;
    double *d_out = out;
    if (!out_on_device) {
        d_out = (double *)mdhip_ws(ctx, WS_OUT, out_b);
        if (!d_out) return MDHIP_ENOMEM;
    }
    std::vector<SegBlock> blocks;
    // default: four waves share a 1024-atom stage; seg_cap = 256 selects the wave-private form (one wave per block,
    // 256-atom stage, no block barrier that costs anything) — measured 0.61 against 0.635 of HBM spec at C4 shape, so
    // the two barriers per step are not what limits this kernel
    int cap = ctx->opt_seg_cap == 512 ? 512 : ctx->opt_seg_cap == 256 ? 256 : SC_CAP_MAX;
    bool staged;
    if (ctx->opt_seg_cap == 0 && ctx->opt_seg_frame != 0) {  // (default: by the shape of the segment table, pick_seg_cap)
        cap = pick_seg_cap(n_seg, seg_off, n_attr, blocks);
        staged = !blocks.empty();
    } else {
        staged = build_seg_blocks(n_seg, seg_off, blocks, cap, cap == 256 ? 64 : 256);
        if (!staged && cap != SC_CAP_MAX) {
            cap = SC_CAP_MAX;
            staged = build_seg_blocks(n_seg, seg_off, blocks, cap, 256);
        }
    }
    SegBlock *d_blocks = nullptr;
    const bool by_frame = staged && ctx->opt_seg_frame != 0 && (cap == SC_CAP_MAX || cap == 512);
    std::vector<SegRun> runs;
    SegRun *d_runs = nullptr;
    double *d_msum = nullptr;
    if (by_frame) {
        runs.resize(blocks.size());
        for (size_t i = 0; i < blocks.size(); ++i)
            runs[i] = {blocks[i].s0, blocks[i].s1, (long long)seg_off[blocks[i].s0],
                       (int)(seg_off[blocks[i].s1] - seg_off[blocks[i].s0]), 0};
        const size_t rb = runs.size() * sizeof(SegRun);
        unsigned char *d_tab = (unsigned char *)mdhip_ws(ctx, WS_AUX3, rb + (size_t)n_seg * 8);
        if (!d_tab) return MDHIP_ENOMEM;
        d_runs = reinterpret_cast<SegRun *>(d_tab);
        d_msum = reinterpret_cast<double *>(d_tab + rb);
        // (the tables are locals: through pinned staging of the call)
        rc = mdhip_h2d_small(ctx, d_runs, runs.data(), rb);
        if (rc) return rc;
        rc = mdhip_h2d_small(ctx, d_msum, msum_h.data(), (size_t)n_seg * 8);
        if (rc) return rc;
    } else if (staged) {
        d_blocks = (SegBlock *)mdhip_ws(ctx, WS_AUX3, blocks.size() * sizeof(SegBlock));
        if (!d_blocks) return MDHIP_ENOMEM;
        rc = mdhip_h2d_small(ctx, d_blocks, blocks.data(), blocks.size() * sizeof(SegBlock));
        if (rc) return rc;
    }
    KernelTimer timer(ctx);
    if (by_frame) {
        const long long n_steps = (long long)n_frames * ((n_attr + SC_PLANES - 1) / SC_PLANES);
        const unsigned gy = (unsigned)std::min<long long>(n_steps, 65535);
        if (cap == 512) {
            ctx->last_kernel = "segment_frame_kernel<false, 512>";
            hipLaunchKernelGGL((segment_frame_kernel<false, 512>), dim3((unsigned)runs.size(), gy), dim3(256), 0,
                               ctx->stream, d_attr, d_mass, d_msum, (const double *)nullptr, d_off, d_runs, d_out, n_attr,
                               (long long)n_atoms, (long long)n_seg, n_steps, 1.0);
        } else {
        ctx->last_kernel = "segment_frame_kernel<false, 1024>";
        hipLaunchKernelGGL((segment_frame_kernel<false, 1024>), dim3((unsigned)runs.size(), gy), dim3(256), 0,
                           ctx->stream, d_attr, d_mass, d_msum, (const double *)nullptr, d_off, d_runs, d_out, n_attr,
                           (long long)n_atoms, (long long)n_seg, n_steps, 1.0);
        }
    } else if (staged) {
        // enough (block, frame slice) pairs to fill the chip several times over; a block loops over its frames
        // frame slices: at least enough (block, slice) pairs to fill the chip several times over, and short runs of
        // ~10 frames per block (measured at C4 shape: 82 slices 0.57 of HBM spec, 512 slices 0.63, 1250 slices 0.60)
        int64_t want = ((int64_t)ctx->cu_count * 16 + (int64_t)blocks.size() - 1) / (int64_t)blocks.size();
        want = std::max<int64_t>(want, n_frames / 10);
        unsigned gy = (unsigned)std::max<int64_t>(1, std::min<int64_t>(n_frames, std::min<int64_t>(want, 65535)));
        if (ctx->opt_seg_gy > 0) gy = (unsigned)std::min<int64_t>(n_frames, ctx->opt_seg_gy);
        ctx->last_kernel = cap == 512 ? "segment_staged_kernel<false, 512, 256>" : "segment_staged_kernel<false, 1024, 256>";
        if (cap == 256) {
            ctx->last_kernel = "segment_staged_kernel<false, 256, 64>";
            hipLaunchKernelGGL((segment_staged_kernel<false, 256, 64>), dim3((unsigned)blocks.size(), gy), dim3(64), 0,
                               ctx->stream, d_attr, d_mass, (const double *)nullptr, d_off, d_blocks, d_out, n_attr,
                               (long long)n_atoms, (long long)n_seg, (long long)n_frames, 1.0, 1.0, ctx->opt_seg_vec);
        } else if (cap == 512)
            hipLaunchKernelGGL((segment_staged_kernel<false, 512, 256>), dim3((unsigned)blocks.size(), gy), dim3(256), 0,
                               ctx->stream, d_attr, d_mass, (const double *)nullptr, d_off, d_blocks, d_out, n_attr,
                               (long long)n_atoms, (long long)n_seg, (long long)n_frames, 1.0, 1.0, ctx->opt_seg_vec);
        else
            hipLaunchKernelGGL((segment_staged_kernel<false, 1024, 256>), dim3((unsigned)blocks.size(), gy), dim3(256), 0,
                               ctx->stream, d_attr, d_mass, (const double *)nullptr, d_off, d_blocks, d_out, n_attr,
                               (long long)n_atoms, (long long)n_seg, (long long)n_frames, 1.0, 1.0, ctx->opt_seg_vec);
    } else {
        const unsigned gy = (unsigned)std::min<int64_t>(n_frames, 4096);
        ctx->last_kernel = "segment_com_kernel";
        hipLaunchKernelGGL(segment_com_kernel, dim3((unsigned)((n_seg + 255) / 256), gy), dim3(256), 0,
                           ctx->stream, d_attr, d_mass, d_off, d_out, n_attr, (long long)n_atoms,
                           (long long)n_seg, (long long)n_frames);
    }
    timer.stop();
    MD_HIP(hipGetLastError());
    if (!out_on_device) {
        rc = mdhip_result(cs, out, d_out, out_b, 0);
        if (rc) return rc;
    }
    cs.defer([timer]() {
        timer.collect();
        return MDHIP_OK;
    });
    return cs.end();
}

static int charge_flux_impl(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *vel,
                            int on_device, const double *atom_mass, const double *atom_q, int64_t n_seg,
                            const int64_t *seg_off, const int32_t *seg_type, int n_types,
                            double vel_conv, double charge_conv, double *flux, int flux_on_device)
{
    if (!ctx) return MDHIP_EINVAL;
    CallScope cs(ctx);
    MD_REQUIRE(n_frames >= 0 && n_atoms >= 0 && n_types >= 0, "negative sizes");
    MD_REQUIRE(n_frames < 65536LL * 65536LL, "too many frames");
    int rc = check_segments(ctx, n_atoms, n_seg, seg_off);
    if (rc) return rc;
    MD_REQUIRE(flux || n_frames == 0 || n_types == 0, "flux is NULL");
    if (n_frames == 0 || n_types == 0) return cs.end();
    MD_REQUIRE(vel && atom_mass && atom_q && seg_type, "NULL input");
    // molecules of one type must be contiguous (they are: ids are type-major, com_mols.py:31-42)
    std::vector<long long> type_off(n_types + 1, 0);
    {
        int64_t s = 0;
        for (int t = 0; t < n_types; ++t) {
            type_off[t] = s;
            while (s < n_seg && seg_type[s] == t) ++s;
        }
        type_off[n_types] = s;
        MD_REQUIRE(s == n_seg, "seg_type must be non-decreasing in 0..n_types-1");
    }
    MD_HIP(hipSetDevice(ctx->device));
    const double *d_vel = (const double *)mdhip_stage(ctx, WS_AUX0, vel, (size_t)n_frames * 3 * n_atoms * 8,
                                                      on_device, &rc);
    if (rc) return rc;
    MD_WS(d_mq, double, WS_AUX1, (size_t)n_atoms * 16);
    {
        // masses | charges and segment offsets | type offsets: pinned staging of the call, one copy each
        MD_PIN(h_mq, double, (size_t)n_atoms * 16);
        memcpy(h_mq, atom_mass, (size_t)n_atoms * 8);
        memcpy(h_mq + n_atoms, atom_q, (size_t)n_atoms * 8);
        MD_HIP(hipMemcpyAsync(d_mq, h_mq, (size_t)n_atoms * 16, hipMemcpyHostToDevice, ctx->stream));
    }
    MD_WS(d_off, long long, WS_AUX2, (size_t)(n_seg + 1 + n_types + 1) * 8);
    {
        MD_PIN(h_off, long long, (size_t)(n_seg + 1 + n_types + 1) * 8);
        memcpy(h_off, seg_off, (size_t)(n_seg + 1) * 8);
        memcpy(h_off + n_seg + 1, type_off.data(), (size_t)(n_types + 1) * 8);
        MD_HIP(hipMemcpyAsync(d_off, h_off, (size_t)(n_seg + 1 + n_types + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    MD_WS(d_tmp, double, WS_AUX3, (size_t)n_frames * 3 * n_seg * 8);
    const size_t flux_b = (size_t)3 * n_types * n_frames * 8;
    double *d_flux = flux;
    if (!flux_on_device) {
        d_flux = (double *)mdhip_ws(ctx, WS_OUT, flux_b);
        if (!d_flux) return MDHIP_ENOMEM;
    }
    std::vector<SegBlock> blocks;
    // default: four waves share a 1024-atom stage; seg_cap = 256 selects the wave-private form (one wave per block,
    // 256-atom stage, no block barrier that costs anything) — measured 0.61 against 0.635 of HBM spec at C4 shape, so
    // the two barriers per step are not what limits this kernel
    int cap = ctx->opt_seg_cap == 512 ? 512 : ctx->opt_seg_cap == 256 ? 256 : SC_CAP_MAX;
    bool staged;
    if (ctx->opt_seg_cap == 0 && ctx->opt_seg_frame != 0) {  // (as mdhip_segment_com)
        cap = pick_seg_cap(n_seg, seg_off, 3, blocks);
        staged = !blocks.empty();
    } else {
        staged = build_seg_blocks(n_seg, seg_off, blocks, cap, cap == 256 ? 64 : 256);
        if (!staged && cap != SC_CAP_MAX) {
            cap = SC_CAP_MAX;
            staged = build_seg_blocks(n_seg, seg_off, blocks, cap, 256);
        }
    }
    SegBlock *d_blocks = nullptr;
    if (staged) {
        d_blocks = (SegBlock *)mdhip_ws(ctx, WS_PART, blocks.size() * sizeof(SegBlock) +
                                                          (ctx->opt_seg_frame ? blocks.size() * sizeof(SegRun) + (size_t)n_seg * 16 : 0));
        if (!d_blocks) return MDHIP_ENOMEM;
        rc = mdhip_h2d_small(ctx, d_blocks, blocks.data(), blocks.size() * sizeof(SegBlock));
        if (rc) return rc;
    }
    const bool by_frame = staged && ctx->opt_seg_frame != 0 && (cap == SC_CAP_MAX || cap == 512);
    SegRun *d_runs = nullptr;
    double *d_msq = nullptr;  // per segment: mass sum, then (charge sum) x charge_conv — the host's index-order sums
    if (by_frame) {
        std::vector<SegRun> runs(blocks.size());
        for (size_t i = 0; i < blocks.size(); ++i)
            runs[i] = {blocks[i].s0, blocks[i].s1, (long long)seg_off[blocks[i].s0],
                       (int)(seg_off[blocks[i].s1] - seg_off[blocks[i].s0]), 0};
        std::vector<double> msq((size_t)n_seg * 2);
        for (int64_t sg = 0; sg < n_seg; ++sg) {
            double m = 0.0, q = 0.0;
            for (int64_t a = seg_off[sg]; a < seg_off[sg + 1]; ++a) {
                m += atom_mass[a];
                q += atom_q[a];
            }
            msq[sg] = m;
            msq[n_seg + sg] = q * charge_conv;  // _conductivity.py:25
        }
        d_runs = reinterpret_cast<SegRun *>(d_blocks + blocks.size());
        d_msq = reinterpret_cast<double *>(d_runs + runs.size());
        rc = mdhip_h2d_small(ctx, d_runs, runs.data(), runs.size() * sizeof(SegRun));  // (the tables are locals)
        if (rc) return rc;
        rc = mdhip_h2d_small(ctx, d_msq, msq.data(), msq.size() * 8);
        if (rc) return rc;
    }
    KernelTimer timer(ctx);
    if (by_frame) {
        const long long n_steps = (long long)n_frames;
        const unsigned gy = (unsigned)std::min<long long>(n_steps, 65535);
        if (cap == 512) {
            ctx->last_kernel = "segment_frame_kernel<true, 512>";
            hipLaunchKernelGGL((segment_frame_kernel<true, 512>), dim3((unsigned)blocks.size(), gy), dim3(256), 0,
                               ctx->stream, d_vel, d_mq, d_msq, d_msq + n_seg, d_off, d_runs, d_tmp, 3, (long long)n_atoms,
                               (long long)n_seg, n_steps, vel_conv);
        } else {
            ctx->last_kernel = "segment_frame_kernel<true, 1024>";
            hipLaunchKernelGGL((segment_frame_kernel<true, 1024>), dim3((unsigned)blocks.size(), gy), dim3(256), 0,
                               ctx->stream, d_vel, d_mq, d_msq, d_msq + n_seg, d_off, d_runs, d_tmp, 3, (long long)n_atoms,
                               (long long)n_seg, n_steps, vel_conv);
        }
    } else if (staged) {
        int64_t want = ((int64_t)ctx->cu_count * 16 + (int64_t)blocks.size() - 1) / (int64_t)blocks.size();
        want = std::max<int64_t>(want, n_frames / 10);
        const unsigned gy = (unsigned)std::max<int64_t>(1, std::min<int64_t>(n_frames, std::min<int64_t>(want, 65535)));
        ctx->last_kernel = cap == 512 ? "segment_staged_kernel<true, 512, 256>" : "segment_staged_kernel<true, 1024, 256>";
        if (cap == 256) {
            ctx->last_kernel = "segment_staged_kernel<true, 256, 64>";
            hipLaunchKernelGGL((segment_staged_kernel<true, 256, 64>), dim3((unsigned)blocks.size(), gy), dim3(64), 0,
                               ctx->stream, d_vel, d_mq, d_mq + n_atoms, d_off, d_blocks, d_tmp, 3, (long long)n_atoms,
                               (long long)n_seg, (long long)n_frames, vel_conv, charge_conv, ctx->opt_seg_vec);
        } else if (cap == 512)
            hipLaunchKernelGGL((segment_staged_kernel<true, 512, 256>), dim3((unsigned)blocks.size(), gy), dim3(256), 0,
                               ctx->stream, d_vel, d_mq, d_mq + n_atoms, d_off, d_blocks, d_tmp, 3, (long long)n_atoms,
                               (long long)n_seg, (long long)n_frames, vel_conv, charge_conv, ctx->opt_seg_vec);
        else
            hipLaunchKernelGGL((segment_staged_kernel<true, 1024, 256>), dim3((unsigned)blocks.size(), gy), dim3(256), 0,
                               ctx->stream, d_vel, d_mq, d_mq + n_atoms, d_off, d_blocks, d_tmp, 3, (long long)n_atoms,
                               (long long)n_seg, (long long)n_frames, vel_conv, charge_conv, ctx->opt_seg_vec);
    } else {
        const unsigned gy = (unsigned)std::min<int64_t>(n_frames, 4096);
        ctx->last_kernel = "mol_flux_kernel";
        hipLaunchKernelGGL(mol_flux_kernel, dim3((unsigned)((n_seg + 255) / 256), gy), dim3(256), 0,
                           ctx->stream, d_vel, d_mq, d_mq + n_atoms, d_off, d_tmp, (long long)n_atoms,
                           (long long)n_seg, (long long)n_frames, vel_conv, charge_conv);
    }
    timer.stop();
    MD_HIP(hipGetLastError());
    for (int64_t f0 = 0; f0 < n_frames; f0 += 65535) {
        const unsigned nf = (unsigned)std::min<int64_t>(65535, n_frames - f0);
        // frames beyond 65535 are handled by offsetting the pointers
        hipLaunchKernelGGL(type_sum_kernel, dim3((unsigned)n_types, 3, nf), dim3(256), 0, ctx->stream,
                           d_tmp + (size_t)f0 * 3 * n_seg, d_off + n_seg + 1, d_flux + f0,
                           (long long)n_seg, (long long)n_frames, n_types);
        MD_HIP(hipGetLastError());
    }
    if (!flux_on_device) {
        rc = mdhip_result(cs, flux, d_flux, flux_b, 0);
        if (rc) return rc;
    }
    cs.defer([timer]() {
        timer.collect();
        return MDHIP_OK;
    });
    return cs.end();
}

int mdhip_charge_flux(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *vel,
                      int on_device, const double *atom_mass, const double *atom_q, int64_t n_seg,
                      const int64_t *seg_off, const int32_t *seg_type, int n_types,
                      double vel_conv, double charge_conv, double *flux)
{
    return charge_flux_impl(ctx, n_frames, n_atoms, vel, on_device, atom_mass, atom_q, n_seg, seg_off, seg_type,
                            n_types, vel_conv, charge_conv, flux, 0);
}

int mdhip_charge_flux_dev(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *vel,
                          int on_device, const double *atom_mass, const double *atom_q, int64_t n_seg,
                          const int64_t *seg_off, const int32_t *seg_type, int n_types,
                          double vel_conv, double charge_conv, double *flux_dev)
{
    return charge_flux_impl(ctx, n_frames, n_atoms, vel, on_device, atom_mass, atom_q, n_seg, seg_off, seg_type,
                            n_types, vel_conv, charge_conv, flux_dev, 1);
}

int mdhip_charge_flux_async(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *vel, int on_device,
                            const double *atom_mass, const double *atom_q, int64_t n_seg, const int64_t *seg_off,
                            const int32_t *seg_type, int n_types, double vel_conv, double charge_conv, double *flux,
                            int flux_on_device)
{
    if (!ctx) return MDHIP_EINVAL;
    AsyncCall mark(ctx);
    return charge_flux_impl(ctx, n_frames, n_atoms, vel, on_device, atom_mass, atom_q, n_seg, seg_off, seg_type, n_types,
                            vel_conv, charge_conv, flux, flux_on_device ? 1 : 0);
}

int mdhip_segment_com_async(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, int n_attr, const double *attr,
                            int attr_on_device, const double *atom_mass, const double *atom_q, int64_t n_seg,
                            const int64_t *seg_off, double *out, int out_on_device, double *seg_mass, double *seg_q)
{
    if (!ctx) return MDHIP_EINVAL;
    AsyncCall mark(ctx);
    return mdhip_segment_com(ctx, n_frames, n_atoms, n_attr, attr, attr_on_device, atom_mass, atom_q, n_seg, seg_off, out,
                             out_on_device, seg_mass, seg_q);
}

}  // extern "C"
