// vertex_wg.h -- the WORKGROUP-COOPERATIVE program of the x-update (vertex step): one 256-thread workgroup solves
// one vertex sub-problem (reference: admm_solver_v3.py:352-466 built it, :490 SolveInParallel/MOSEK solved it).
//
// Same interior-point method as the wavefront program (vertex_program.inc) and the oracle -- Mehrotra predictor-
// corrector, Nesterov-Todd scaling of the one cone, arrow elimination blocks -> sides -> reduced border system,
// the numerical rules of DESIGN.md section 3 -- but a different mapping onto the machine:
//   * every matrix and vector of the sub-problem lives in LDS (layout WLay); nothing is held per lane;
//   * each step of the algorithm is a PARALLEL REGION: a loop over independent tasks (one facet row, one matrix
//     entry, one column of an inverse, ...) strided over the 256 threads, closed by a workgroup barrier;
//   * the only serial chains left are the pivots of the Cholesky factorisations (one barrier per column) and the
//     handful of cone scalars thread 0 computes.
// A vertex with d incident edges and m facets has (d+1) "units" (unit 0: the border rows on (z_v, y_v); unit e:
// the block (O_e, y_e)) of 4m facet rows each, so the facet-row passes run (d+1)*4m tasks wide instead of 2m
// iterations deep; the dimension-generic dense algebra (n = 2, 3, 6: blocks of 2n+1, border 4n+1) never touches a
// register array larger than one column.  This is the latency-optimal mapping: it is what small graphs
// (benchmark4: 42 vertices on a 256-CU chip), n = 3 / 6 and vertices of degree > 63 use; large n = 2 graphs keep
// the wavefront program, whose throughput per CU is higher.
//
// The file is host-compilable: with WG_FOR a plain loop and WG_SYNC a no-op the regions execute serially, which is
// exactly equivalent as long as the tasks of a region are independent (tests/hostemu/wg_emu.cpp runs them in
// ascending and in descending order and compares).
#pragma once
#include <type_traits>

#include "gcs_math.h"
#include "warm_start.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define WG_DEVICE 1
#else
#define WG_DEVICE 0
#endif

namespace gcs_wg {

using gcs_math::rcp;
using gcs_math::rcp1;
using gcs_math::rsqrt_nr;
using gcs_math::sqrt_nr;

#ifndef GCS_WG_THREADS
#define GCS_WG_THREADS 256
#endif
constexpr int WG_THREADS = GCS_WG_THREADS;        // threads per workgroup (a power of two, whole wavefronts)
constexpr int WG_WAVES = WG_THREADS / 64;
constexpr int WG_ITEM_WAVES = WG_WAVES > 1 ? WG_WAVES - 1 : 1;   // wavefronts that share the items of a wave-local pipeline
constexpr double CHOL_SKIP = 1e-12;
constexpr double REG_DELTA = 1e-7;   // Tikhonov term on every centred unknown except t (oracle/gcs_oracle.c REG_DELTA)

#if WG_DEVICE
// The thread id is made opaque at the head of every region: a thread's first task of a region (and its decode into
// unit / row / column, base pointers, ...) is invariant across the Newton loop, and hoisting all of that out of the loop
// costs hundreds of live registers (scratch spills) for no gain.
__device__ __forceinline__ int wg_tid()
{
    int t = (int)threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
#define WG_FOR(i, cnt) for (int i = gcs_wg::wg_tid(), i##_end = (cnt); i < i##_end; i += gcs_wg::WG_THREADS)
// the same with task 0 on thread `start`: a region with several KINDS of task starts each kind on its own wavefront
// (Place, below); with every kind starting at thread 0 the first wavefront runs all kinds back to back while the others idle
#define WG_FOR_AT(i, cnt, start) for (int i = (gcs_wg::wg_tid() - (start)) & (gcs_wg::WG_THREADS - 1), i##_end = (cnt); i < i##_end; i += gcs_wg::WG_THREADS)
#define WG_SYNC() __syncthreads()
#define WG_ONE() if (gcs_wg::wg_tid() == 0)
// the serial cone algebra runs on the LAST thread: wave 3 has the fewest row / entry tasks in every region
#define WG_CONE() if (gcs_wg::wg_tid() == gcs_wg::WG_THREADS - 1)
// the cone's two step bounds (slack side, dual side) are the SAME code on different data: two lanes of the last wavefront, l = 0, 1, run
// them side by side (one instruction stream) instead of one lane running them back to back
#define WG_CONE2(l) for (int l = gcs_wg::wg_tid() - (gcs_wg::WG_THREADS - 2); l >= 0 && l < 2; l = 2)
// work that one WAVEFRONT does with a matrix row per lane (wave_ldl below): no barrier inside, broadcasts by readlane
#define WG_WAVE0() if (gcs_wg::wg_tid() < 64)
// `cnt` pieces of such work, piece w on wavefront w (one matrix per wavefront; a workgroup with fewer wavefronts takes them in turns)
#define WG_FIRST_WAVES(w, cnt) for (int w = gcs_wg::wg_tid() >> 6; w < (cnt); w += gcs_wg::WG_WAVES)
// WAVE-LOCAL PIPELINE.  Tasks (q, i): `count` items (units) are dealt round-robin to the first WG_ITEM_WAVES wavefronts (item q
// belongs to wavefront q % WG_ITEM_WAVES; the last wavefront is left to the serial cone thread), i runs over the PER sub-tasks of
// an item, shared by the lanes of the owning wavefront.  Successive WG_ITEM_FOR loops over the SAME items exchange data through
// LDS with WG_WAVE_SYNC() only (a wavefront's LDS operations complete in order).  Measured (profiles/r02/README.md): this pays for
// chains of SHORT steps (d nu -> r_e -> d w_e of a solve: 2 060 -> 1 800 cycles); for steps with real work per task (block
// factor -> inverse -> products: 3 260 against 2 400 cycles; G'kappa -> t_e: 4 440 against 2 200; the head of the corrector
// solve on one wavefront) the work of four wavefronts lands on one and the pipeline is SLOWER than barrier-separated regions
// spread over all threads -- those stay regions.
#define WG_ITEM_FOR(q, i, count, PER)                                                                                          \
    for (int t_ = gcs_wg::wg_tid(), w_ = t_ >> 6, l_ = t_ & 63,                                                                \
             n_ = w_ < gcs_wg::WG_ITEM_WAVES ? (((count) - w_ + gcs_wg::WG_ITEM_WAVES - 1) / gcs_wg::WG_ITEM_WAVES) * (PER) : 0;  \
         l_ < n_; l_ += 64)                                                                                                    \
        if (const int qq_ = l_ / (PER), i = l_ - qq_ * (PER), q = w_ + gcs_wg::WG_ITEM_WAVES * qq_; true)
// a small step EVERY item wavefront repeats for itself (same values to the same words) instead of waiting at a barrier for one
#define WG_REPL_FOR(i, cnt) for (int i = (gcs_wg::wg_tid() >> 6) < gcs_wg::WG_ITEM_WAVES ? (gcs_wg::wg_tid() & 63) : (cnt), i##_end = (cnt); i < i##_end; i += 64)
#define WG_WAVE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define WG_FENCE() asm volatile("" ::: "memory")
#else
#define WG_FENCE() do { } while (0)
// host: one "thread" runs every task; GCS_WG_REVERSE flips the task order (independence check)
#ifdef GCS_WG_REVERSE
#define WG_FOR(i, cnt) for (int i = (cnt) - 1; i >= 0; --i)
#else
#define WG_FOR(i, cnt) for (int i = 0, i##_end = (cnt); i < i##_end; ++i)
#endif
#define WG_FOR_AT(i, cnt, start) WG_FOR(i, ((void)(start), (cnt)))
#define WG_SYNC() do { } while (0)
#define WG_ONE() if (true)
#define WG_CONE() if (true)
#define WG_CONE2(l) for (int l = 0; l < 2; ++l)
#define WG_WAVE0() if (true)
#define WG_FIRST_WAVES(w, cnt) for (int w = 0; w < (cnt); ++w)
#define WG_ITEM_FOR(q, i, count, PER) WG_FOR(t_, (count) * (PER)) if (const int q = t_ / (PER), i = t_ - q * (PER); true)
#define WG_REPL_FOR(i, cnt) WG_FOR(i, cnt)
#define WG_WAVE_SYNC() do { } while (0)
#endif

// diagnostic build (-DGCS_WG_TIMING, tools/wg_phase_timing.py): thread 0 of workgroup 0 accumulates the s_memtime ticks
// between consecutive stamps into g_wg_cycles[id]; nothing of this exists in the product build
#if defined(GCS_WG_TIMING) && defined(__HIPCC__)
__device__ unsigned long long g_wg_cycles[64];
__device__ unsigned long long g_wg_counts[64];
__device__ unsigned long long g_wg_wave_cycles[64 * 8];      // [region][wavefront]: ticks from the region's start to the wavefront's arrival at its closing barrier
#endif
#if defined(GCS_WG_TIMING) && WG_DEVICE
__device__ __forceinline__ void wg_stamp(int id, unsigned long long &last)
{
    // workgroup 0 only (a uniform branch).  Every wavefront takes the time right after the barrier that closed region `id` -- the
    // common origin of the next region; lane 0 of wavefront 0 adds the region's length with NON-RETURNING atomics (nothing waits for
    // them: the volatile read-modify-write of earlier rounds held wavefront 0 back by ~1 100 cycles per stamp, which the other
    // wavefronts then waited for at the next barrier).  (n = 6: a divergent branch here once made that instantiation spill beside
    // it, and this compiler stored the spill under the branch's partial EXEC mask; the timing build is a tool for n = 2, 3.)
    if (blockIdx.x == 0) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) {
            atomicAdd(&g_wg_cycles[id], t - last);
            atomicAdd(&g_wg_counts[id], 1ull);
        }
        last = t;
    }
}
// placed right BEFORE the barrier that closes region `id`: when did THIS wavefront get there (ticks since the region's start)?
__device__ __forceinline__ void wg_arrive(int id, unsigned long long last)
{
    if (blockIdx.x == 0) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_wg_wave_cycles[id * 8 + ((threadIdx.x >> 6) & 7)], t - last);
    }
}
#define WG_ARRIVE(id) gcs_wg::wg_arrive(id, wg_last_stamp)
#define WG_STAMP(id) gcs_wg::wg_stamp(id, wg_last_stamp)
#define WG_STAMP_INIT() unsigned long long wg_last_stamp = __builtin_amdgcn_s_memtime()
#else
#define WG_STAMP(id) do { } while (0)
#define WG_ARRIVE(id) do { } while (0)
#define WG_STAMP_INIT() do { } while (0)
#endif

// first thread of each kind of task inside one region: kinds are laid out one after the other, each starting on a
// wavefront boundary (wave-uniform arithmetic)
struct Place {
    int off = 0;
    GCS_HD int at(int cnt)
    {
        const int start = off & (WG_THREADS - 1);
        off = (off + cnt + 63) & ~63;
        return start;
    }
};

template <int N> struct WD {
    static constexpr int NW = 2 * N + 1, NX = 2 * N, NB1 = 4 * N + 1, Q = N + 1, NS = N * (N + 1) / 2;
    static constexpr int TA = 2 * NS + 2 * N + 1;   // Hessian assembly tasks per unit
};

// LDS layout of one vertex sub-problem, offsets in doubles.  Everything sits at a COMPILE-TIME offset except the
// stride between units (it holds the 16 m facet-row values of a unit) and the polytope at the end: with ~50 run-time
// offsets live across the Newton loop the compiler ran out of scalar registers and spilled.
//   [ fixed block | unit 0 | unit 1 | ... | unit d | A (m x n) | bc (m) ]
constexpr int pad2(int x) { return (x + 1) & ~1; }      // keep every array 16-byte aligned
template <int N> struct WSoc {   // cone block
    static constexpr int Q = N + 1;
    static constexpr int SS = 0, LS = Q, WB = 2 * Q, LT = 3 * Q, CV = LT + Q, SU = CV + N, DSSA = SU + N * N, DLSA = DSSA + Q,
                         DSS = DLSA + Q, DLS = DSS + Q, KS = DLS + Q, SIZE = KS + Q;
};
enum { SC_T = 0, SC_DT, SC_DTA, SC_ALPHA, SC_CONEFAIL, SC_C0, SC_ETA, SC_GT, SC_AMAXC, SC_C1C, SC_C2C, SC_AMAXC2, SC_N = 12 };
template <int N> struct WL {
    using D = WD<N>;
    static constexpr int NW = D::NW, NX = D::NX, NB1 = D::NB1;
    // fixed block
    static constexpr int CEN = 0, XV = CEN + pad2(N), DX = XV + pad2(NX), NU = DX + pad2(NX), DNU = NU + pad2(2 * NW),
                         BS = DNU + pad2(2 * NW), BSI = BS + pad2(2 * NW * NW), BXS = BSI + pad2(2 * NW * NW),
                         PIVS = BXS + pad2(2 * NW * NX), BG = PIVS + pad2(2 * NW), XBG = BG + pad2(2 * NW), XBX = XBG + pad2(2 * NX),
                         XS = XBX + pad2(2 * NX * NX), RP = XS + pad2(NX * N), VV = RP + pad2(2 * NW), WW = VV + pad2(2 * NW), M = WW + pad2(2 * NW),
                         PIVM = M + pad2(NB1 * NB1), RHS = PIVM + pad2(NB1), SOL = RHS + pad2(NB1),
                         SOC = SOL + pad2(NB1), PQ = SOC + pad2(WSoc<N>::SIZE), PC = PQ + pad2(NX + NW), SC = PC + pad2(NX + NW), RED = SC + pad2(SC_N), FIXED = RED + 3 * 3 * WG_WAVES;
    // per-unit block (offsets from the unit's base); the three facet-row arrays (4m each) follow at ROWS
    static constexpr int P = 0, DW = P + pad2(NW), TG = DW + pad2(NW), TF = TG + pad2(NW), LB = TF + pad2(N), KB = LB + 2, DLB = KB + 2,
                         PIV = DLB + 2, G0 = PIV + pad2(NW), GU = G0 + pad2(NW), GX = GU + pad2(NW), TE = GX + pad2(NX),
                         RV = TE + pad2(NW), K = RV + pad2(NW), X = K + pad2(NW * NW), B = X + pad2(NW * NX), ROWS = B + pad2(NW * NW);
    // ODD number of doubles: threads that read the same field of different units (most regions do) then hit different LDS
    // banks; with the natural even stride (e.g. 216 doubles at n = 2, m = 7) units 4 apart collide: 4-way conflicts
    static GCS_HD int unit_stride(int m) { return (ROWS + 12 * m) | 1; }
    static GCS_HD int total(int U, int m) { return FIXED + pad2(U * unit_stride(m)) + pad2(m * N) + pad2(m); }
};
// The same for the BOX instantiation (canonical axis-aligned boxes, wg_solve_vertex<N, T, true>): the fixed block is WL<N>'s; a unit
// holds K_e, X_e and B_e = K_e^{-1} in their STRUCTURED forms instead of three dense matrices (86 instead of 496 doubles at n = 6):
//   K_e = diag(KD) + the y column KY (both ways) + KYY,     X_e = diag(XD) on the x part + the y row XY,
//   B_e = diag(BD) + BRS * BW BW'   (BW = (w_1, w_2, -1): the closed form of the block inverse),   B_e X_e = diag(BD XD) + BRS * BW BQ'.
// That is what lets a CU hold three workgroups of BASELINE config 5 (47 KB each) instead of two (77 KB).
// (No K, X, B members: a generic access in a BOX path does not compile.)
template <int N> struct WLBox {
    using G = WL<N>;
    using D = WD<N>;
    static constexpr int NW = D::NW, NX = D::NX, NB1 = D::NB1;
    static constexpr int CEN = G::CEN, XV = G::XV, DX = G::DX, NU = G::NU, DNU = G::DNU, BS = G::BS, BSI = G::BSI, BXS = G::BXS, PIVS = G::PIVS,
                         BG = G::BG, XBG = G::XBG, XBX = G::XBX, XS = G::XS, RP = G::RP, VV = G::VV, WW = G::WW, M = G::M, PIVM = G::PIVM,
                         RHS = G::RHS, SOL = G::SOL, SOC = G::SOC, PQ = G::PQ, PC = G::PC, SC = G::SC, RED = G::RED, FIXED = G::FIXED;
    static constexpr int P = G::P, DW = G::DW, TG = G::TG, TF = G::TF, LB = G::LB, KB = G::KB, DLB = G::DLB, G0 = G::G0, GU = G::GU, GX = G::GX,
                         TE = G::TE, RV = G::RV;
    static constexpr int KD = G::K, KY = KD + pad2(NX), KYY = KY + pad2(NX), XD = KYY + 2, XY = XD + pad2(NX),
                         BD = XY + pad2(NX), BW = BD + pad2(NX), BRS = BW + pad2(NW), BQ = BRS + 2, ROWS = BQ + pad2(NX);
    static GCS_HD int unit_stride(int m) { return (ROWS + 12 * m) | 1; }
    static GCS_HD int total(int U, int m) { return FIXED + pad2(U * unit_stride(m)) + pad2(m * N) + pad2(m); }
};
template <int N> GCS_HD int wg_lds_doubles(int U, int m, bool box = false) { return box ? WLBox<N>::total(U, m) : WL<N>::total(U, m); }
// the program is dimension-generic (admm_solver_v3.py:363-377 takes any n): instantiated for n = 1 .. 6; the BOX instantiation and
// its layout exist for the two dimensions that are tuned for it, n = 3 and 6
constexpr int WG_MAX_N = 8;
inline bool wg_has_box(int n) { return n == 3 || n == 6; }
inline int wg_lds_doubles_n(int n, int U, int m, bool box = false)
{
    box = box && wg_has_box(n);
    switch (n) {
    case 1: return wg_lds_doubles<1>(U, m);
    case 2: return wg_lds_doubles<2>(U, m);
    case 3: return wg_lds_doubles<3>(U, m, box);
    case 4: return wg_lds_doubles<4>(U, m);
    case 5: return wg_lds_doubles<5>(U, m);
    case 6: return wg_lds_doubles<6>(U, m, box);
    case 7: return wg_lds_doubles<7>(U, m);
    default: return wg_lds_doubles<8>(U, m);
    }
}

struct alignas(8) WgF2 { float a, b; };      // two row duals of a warm-start record (warm_start.h: f32)

template <class T> struct WgArgs {
    int n_vtx;                  // generic vertices handled by this launch, one workgroup each
    const int *vtx;             // [n_vtx] vertex ids, heaviest first
    const int *inc_ptr;         // [V+1]
    const int *deg_in;          // [V]
    const int *inc_edge;        // [NI_owned]
    const int *poly_ptr;        // [V+1]
    const double *poly_A;       // [sum m][n]
    const double *poly_bc;      // [sum m] centred: b - A c
    const double *center;       // [V][n]
    int E, NI;
    const T *zedge, *mu;
    T *copy;
    double *xv, *zv, *yv;
    int *counters;              // [0] inner failures, [1] inner iterations
    double eps_edge, ipm_tol;
    int ipm_max_iter;
    // PROX configuration (the x-update of the reference's vertex-edge splits, admm_solver_v1.py:334-383):
    // no edge blocks; a separable quadratic 1/2 sum_k q_k (u_k - c_k)^2 on the border unknowns u = (x_v, z_v, y_v) instead of
    // the consensus penalty.  [V][4n+1] each, order x (2n), z (2n), y; nullptr = the ADMM vertex step of the v3 solver.
    const double *prox_q = nullptr, *prox_c = nullptr;
    int edge_major = 0;         // 1: state columns numbered by edge (tail side e, head side E + e) instead of by incidence
    // warm start (warm_start.h): the records of the handle's workspace, warm + warm_ptr[v]; nullptr = every solve starts cold
    double *warm = nullptr;
    const long long *warm_ptr = nullptr;
    // slowest-first dispatch (gcsadmm.hip reorder_kernel): workgroup b of the launch solves vtx[order[b]] and leaves its Newton
    // iteration count in unit_iters[order[b]] (both may be null: the static heaviest-first order of vtx, nothing recorded)
    const int *order = nullptr;
    int *unit_iters = nullptr;
};

// ---------------------------------------------------------------------------------------------------------------
// workgroup reductions: (min, sum, sum) of one value triple per thread -> the same result in every thread
// ---------------------------------------------------------------------------------------------------------------
struct Red3 { double mn, s1, s2; };

#if WG_DEVICE
template <int SH> __device__ __forceinline__ double dpp_row_shr(double x, double ident)
{
    // lane i reads lane i - SH of its 16-lane row; lanes without a source keep `ident`
    const int ilo = __double2loint(ident), ihi = __double2hiint(ident);
    const int lo = __builtin_amdgcn_update_dpp(ilo, __double2loint(x), 0x110 + SH, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(ihi, __double2hiint(x), 0x110 + SH, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_bcast(double x, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane), __builtin_amdgcn_readlane(__double2loint(x), lane));
}
#endif

// a workgroup-uniform value, moved to scalar registers on the device (it then costs no vector register while it stays live)
GCS_HD double wg_uniform(double x)
{
#if WG_DEVICE
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
#else
    return x;
#endif
}
GCS_HD int wg_uniform(int x)
{
#if WG_DEVICE
    return __builtin_amdgcn_readfirstlane(x);
#else
    return x;
#endif
}

GCS_HD Red3 wg_reduce(Red3 v, double *red, int &phase)
{
#if WG_DEVICE
    // rows of 16 lanes with DPP shifts (lane 15 of a row ends up with the row's result), then the four row results
    // through SGPRs, then the four wavefronts through LDS; `red` rotates over three buffers so that one barrier suffices
    v.mn = fmin(v.mn, dpp_row_shr<1>(v.mn, 1e300)); v.s1 += dpp_row_shr<1>(v.s1, 0.0); v.s2 += dpp_row_shr<1>(v.s2, 0.0);
    v.mn = fmin(v.mn, dpp_row_shr<2>(v.mn, 1e300)); v.s1 += dpp_row_shr<2>(v.s1, 0.0); v.s2 += dpp_row_shr<2>(v.s2, 0.0);
    v.mn = fmin(v.mn, dpp_row_shr<4>(v.mn, 1e300)); v.s1 += dpp_row_shr<4>(v.s1, 0.0); v.s2 += dpp_row_shr<4>(v.s2, 0.0);
    v.mn = fmin(v.mn, dpp_row_shr<8>(v.mn, 1e300)); v.s1 += dpp_row_shr<8>(v.s1, 0.0); v.s2 += dpp_row_shr<8>(v.s2, 0.0);
    Red3 w;
    w.mn = fmin(fmin(lane_bcast(v.mn, 15), lane_bcast(v.mn, 31)), fmin(lane_bcast(v.mn, 47), lane_bcast(v.mn, 63)));
    w.s1 = (lane_bcast(v.s1, 15) + lane_bcast(v.s1, 31)) + (lane_bcast(v.s1, 47) + lane_bcast(v.s1, 63));
    w.s2 = (lane_bcast(v.s2, 15) + lane_bcast(v.s2, 31)) + (lane_bcast(v.s2, 47) + lane_bcast(v.s2, 63));
    double *buf = red + phase * (3 * WG_WAVES);
    phase = phase == 2 ? 0 : phase + 1;
    const int wave = (int)threadIdx.x >> 6;
    if (((int)threadIdx.x & 63) == 0) { buf[wave * 3 + 0] = w.mn; buf[wave * 3 + 1] = w.s1; buf[wave * 3 + 2] = w.s2; }
    __syncthreads();
    Red3 r{buf[0], buf[1], buf[2]};
#pragma unroll
    for (int q = 1; q < WG_WAVES; ++q) { r.mn = fmin(r.mn, buf[3 * q]); r.s1 += buf[3 * q + 1]; r.s2 += buf[3 * q + 2]; }
    return r;
#else
    (void)red; (void)phase;
    return v;
#endif
}

// (Round 4, measured and rejected: FACET-PARALLEL SUMS -- the task kinds that sum over the facet rows of a unit (Hessian entries, G'kappa)
// with an aligned group of 8 lanes per output, lane l taking rows l, l + 8, ..., three DPP row shifts, the last lane storing.  The barrier-
// arrival probes of the timing build show the wavefronts of those kinds arriving last in their regions on benchmark4's heaviest vertex
// (7 facets, degree 8-10), yet the launch got slower, 9 900 -> 9 640 it/s: most vertices have 4-5 facets, where the loop is one batch of
// loads anyway, and the wider kinds need a second pass over the 512 threads on the heavy ones.  profiles/r04/README.md.)
// t = r (r + 1) / 2 + c with 0 <= c <= r  ->  (r, c), without a loop (exact for the sizes used here, t < 2^20)
GCS_HD void tri_decode(int t, int &r, int &c)
{
    r = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    if ((r + 1) * (r + 2) / 2 <= t) ++r;
    if (r * (r + 1) / 2 > t) --r;
    c = t - r * (r + 1) / 2;
}

// small-integer division by a run-time divisor without the ~40-instruction integer sequence (exact for the
// ranges used here: dividend < 2^20, divisor <= 1024; the margin (0.5/div) dwarfs the float rounding)
GCS_HD int fdiv(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

GCS_HD constexpr int pki(int i, int j) { return i * (i + 1) / 2 + j; }   // packed lower, i >= j

// ---------------------------------------------------------------------------------------------------------------
// The reduced border system inside ONE wavefront.  A = L D L' (L unit lower) of one SPD matrix of dimension DIM <= 64
// with one matrix ROW per lane in registers: the pivot and the pivot column reach the other lanes by readlane, so the
// DIM dependent column steps cost no barrier and no LDS round trip (the tile-blocked wg_chol pays three barriers per
// tile: 2 600 cycles at 9 x 9, 13 600 at 25 x 25, and the explicit inverse that followed it as much again; measured,
// profiles/r02).  The two solves of a Newton iteration are substitutions with the unit factor (wave_ldl_solve): the chain
// per step is one readlane + one FMA, there is no inverse.  Called by one wavefront (WG_WAVE0), the others wait at the
// region's barrier.
//   in : Mq  lower triangle of A (row-major, ld = DIM);   out: strictly lower part of Mq = L, rd[k] = 1 / D_k
// (the diagonal and the upper triangle of Mq are left undefined).  Pivot rule of oracle chol(): a pivot that has
// cancelled below CHOL_SKIP of its ORIGINAL diagonal entry is clamped there.
// (Sending the pivot column through LDS -- one write, broadcast ds_read_b128 back -- instead of two readlanes per entry was
// measured SLOWER at 25 x 25: 82 800 against 79 200 ticks per Newton iteration; the wait on the LDS round trip per column costs
// more than the readlanes it saves.)
// (Round 4, measured and rejected: the affine solve INSIDE this pass -- forward substitution riding in the elimination, one more
// readlane + FMA per column, backward substitution from the factor just stored -- saves a region and its barrier but lengthens the
// pivot chain by as much: benchmark4 9 555 it/s with it, 9 690 without, profiles/r04/README.md.)
// Host build: the same column-by-column elimination written serially.
// ---------------------------------------------------------------------------------------------------------------
template <int DIM> GCS_HD void wave_ldl(double *Mq, double *rd)
{
    static_assert(DIM <= 64, "one row per lane");
#if WG_DEVICE
    const int lane = wg_tid() & 63, row = lane < DIM ? lane : DIM - 1;      // spare lanes shadow the last row and store nothing
    double a[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) a[j] = Mq[row * DIM + j];
    const double od = Mq[row * DIM + row];
    double myr = 0.0;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
        double dk = lane_bcast(a[k], k);
        const double odk = lane_bcast(od, k);     // original diagonal entry (an LDS read here would sit on the pivot-to-pivot chain)
        if (!(dk > CHOL_SKIP * odk)) dk = odk > 0.0 ? CHOL_SKIP * odk : 1.0;
        const double rk = rcp(dk), col = a[k], l = col * rk;
        a[k] = l;
        if (lane == k) myr = rk;
#pragma unroll
        for (int j = k + 1; j < DIM; ++j) a[j] -= l * lane_bcast(col, j);     // (entries right of the diagonal: unused values)
    }
    if (lane < DIM) {
#pragma unroll
        for (int j = 0; j < DIM - 1; ++j) Mq[lane * DIM + j] = a[j];
        rd[lane] = myr;
    }
#else
    double od[DIM], col[DIM];
    for (int k = 0; k < DIM; ++k) od[k] = Mq[k * DIM + k];
    for (int k = 0; k < DIM; ++k) {
        double dk = Mq[k * DIM + k];
        if (!(dk > CHOL_SKIP * od[k])) dk = od[k] > 0.0 ? CHOL_SKIP * od[k] : 1.0;
        const double rk = rcp(dk);
        rd[k] = rk;
        for (int i = k + 1; i < DIM; ++i) col[i] = Mq[i * DIM + k];
        for (int i = k + 1; i < DIM; ++i) {
            const double l = col[i] * rk;
            Mq[i * DIM + k] = l;
            for (int j = k + 1; j <= i; ++j) Mq[i * DIM + j] -= l * col[j];
        }
    }
#endif
}

// x = A^{-1} b with the factor of wave_ldl: forward substitution with L, scaling by 1 / D, backward substitution with L'
// (lane i owns b_i, row i of L for the forward sweep and column i for the backward sweep)
template <int DIM> GCS_HD void wave_ldl_solve(const double *Mq, const double *rd, const double *b, double *x)
{
#if WG_DEVICE
    const int lane = wg_tid() & 63, row = lane < DIM ? lane : DIM - 1;
    double l[DIM], v = b[row];
#pragma unroll
    for (int k = 0; k < DIM - 1; ++k) l[k] = k < row ? Mq[row * DIM + k] : 0.0;
#pragma unroll
    for (int k = 0; k < DIM - 1; ++k) v -= l[k] * lane_bcast(v, k);
    v *= rd[row];
#pragma unroll
    for (int k = 1; k < DIM; ++k) l[k] = k > row ? Mq[k * DIM + row] : 0.0;
#pragma unroll
    for (int k = DIM - 1; k >= 1; --k) v -= l[k] * lane_bcast(v, k);
    if (lane < DIM) x[lane] = v;
#else
    double v[DIM];
    for (int i = 0; i < DIM; ++i) v[i] = b[i];
    for (int k = 0; k < DIM - 1; ++k)
        for (int i = k + 1; i < DIM; ++i) v[i] -= Mq[i * DIM + k] * v[k];
    for (int i = 0; i < DIM; ++i) v[i] *= rd[i];
    for (int k = DIM - 1; k >= 1; --k)
        for (int i = 0; i < k; ++i) v[i] -= Mq[k * DIM + i] * v[k];
    for (int i = 0; i < DIM; ++i) x[i] = v[i];
#endif
}

// column c of A^{-1} from the factor of wave_ldl (Lm: unit lower factor in the strictly lower part, rd: 1 / D): one thread per
// column, the column in registers (the side systems need their inverse as a matrix: it enters the border system)
// (WG_FENCE: a compiler-level fence per row on the device; without it the scheduler hoists every LDS load of the
//  factor to the top of the unrolled solve and the live ranges spill.  The scaling by 1 / D sits INSIDE the backward sweep: as a
//  loop of its own it cost the generic n = 6 kernel 90 registers and scratch; the stores stay a loop of their own: inside the
//  sweep they made the n = 6 iteration 6 % slower.)
// LOWREG: the stores inside the sweep as well (the generic n = 6 kernel, which otherwise spills)
template <int DIM, bool LOWREG> GCS_HD void ldl_inverse_col(const double *Lm, const double *rd, int c, double *out, int ldo)
{
    double x[DIM];
#pragma unroll
    for (int i = 0; i < DIM; ++i) {
        double s = (i == c) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) s -= Lm[i * DIM + k] * x[k];
        x[i] = s;
        if constexpr (DIM > 9) WG_FENCE();
    }
#pragma unroll
    for (int i = DIM - 1; i >= 0; --i) {
        double s = x[i] * rd[i];
#pragma unroll
        for (int k = i + 1; k < DIM; ++k) s -= Lm[k * DIM + i] * x[k];
        x[i] = s;
        if constexpr (LOWREG) out[i * ldo + c] = s;
        if constexpr (DIM > 9) WG_FENCE();
    }
    if constexpr (!LOWREG) {
#pragma unroll
        for (int i = 0; i < DIM; ++i) out[i * ldo + c] = x[i];
    }
}

// Nesterov-Todd scaling of the cone from (s, z): wb (unit hyperbolic vector), eta; false on a boundary point
template <int Q> GCS_HD bool soc_scaling_wb(const double *s, const double *z, double *wb, double &eta)
{
    const double ss = gcs_math::soc_det<Q>(s), zz = gcs_math::soc_det<Q>(z);
    if (!(ss > 0.0) || !(zz > 0.0)) return false;
    const double is = rsqrt_nr(ss), iz = rsqrt_nr(zz);
    double dot = 0;
    for (int k = 0; k < Q; ++k) dot += (s[k] * is) * (z[k] * iz);
    const double ig2 = 0.5 * rsqrt_nr(0.5 * (1.0 + dot));
    wb[0] = (s[0] * is + z[0] * iz) * ig2;
    for (int k = 1; k < Q; ++k) wb[k] = (s[k] * is - z[k] * iz) * ig2;
    eta = sqrt_nr((ss * is) * iz);
    return true;
}

// products with the Nesterov-Todd scaling W = eta * [wb0 wb1'; wb1 I + wb1 wb1'/(1+wb0)], its inverse and W^{-2} =
// eta^{-2}(2 v v' - J), v = (wb0, -wb1), applied from wb in O(Q) (the explicit Q x Q matrices are never formed)
template <int Q> GCS_HD void soc_apply_W(const double *wb, double eta, const double *x, double *y)
{
    double dd = 0;
#pragma unroll
    for (int k = 1; k < Q; ++k) dd += wb[k] * x[k];
    const double f = x[0] + dd * rcp(1.0 + wb[0]);
    y[0] = eta * (wb[0] * x[0] + dd);
#pragma unroll
    for (int k = 1; k < Q; ++k) y[k] = eta * (x[k] + wb[k] * f);
}
template <int Q> GCS_HD void soc_apply_Wi(const double *wb, double eta, const double *x, double *y)
{
    double dd = 0;
#pragma unroll
    for (int k = 1; k < Q; ++k) dd += wb[k] * x[k];
    const double ieta = rcp(eta), f = -x[0] + dd * rcp(1.0 + wb[0]);
    y[0] = ieta * (wb[0] * x[0] - dd);
#pragma unroll
    for (int k = 1; k < Q; ++k) y[k] = ieta * (x[k] + wb[k] * f);
}
template <int Q> GCS_HD void soc_apply_W2(const double *wb, double eta, const double *x, double *y)
{
    double vx = wb[0] * x[0];
#pragma unroll
    for (int k = 1; k < Q; ++k) vx -= wb[k] * x[k];
    const double ieta = rcp(eta), ie2 = ieta * ieta;
    y[0] = ie2 * (2.0 * wb[0] * vx - x[0]);
#pragma unroll
    for (int k = 1; k < Q; ++k) y[k] = ie2 * (x[k] - 2.0 * wb[k] * vx);
}

// ---------------------------------------------------------------------------------------------------------------
// one vertex sub-problem.  `sm` is the workgroup's LDS (wg_lds_doubles doubles).  Returns (to every thread) the
// solver status (0 = converged) and the number of interior-point iterations through status_out / iters_out.
// ---------------------------------------------------------------------------------------------------------------
// The facet rows of the sub-problem as tasks of a region (inside wg_solve_vertex: uses its locals).  A task takes the rows that SHARE
// their operands and runs them back to back (their reciprocal chains overlap):
//   generic: (unit, half, facet j): the two rows of the facet, type a (s = b y - a.p) and type b (s = b (1 - y) - a.(x - p)): one
//            pass over the facet's normal gives both dot products -- 2m(d+1) tasks instead of 4m(d+1) rows (benchmark4's heaviest
//            vertex: 154 tasks in one pass instead of 308 rows in two);
//   BOX:     (unit, half, coordinate k): the four rows on that coordinate, types a and b of the facets +e_k and -e_k -- 2n(d+1) tasks.
// Declares for the body: u, un, i, j, ty, ro (offset in the unit's row arrays), s (slack at the iterate) and, with DS, ds (slack
// direction for the direction DW of the unit / DX):  ds_a = b dy - a.dp ;  ds_b = -b dy - a.(dx - dp).
#define WG_ROWS_BEGIN_(pl, DS)                                                                                                 \
    WG_FOR_AT(rt_, (BOX ? U * 2 * N : U * m2), (pl).at(BOX ? U * 2 * N : U * m2)) {                                            \
        int u, i, jk_;                                                                                                         \
        if constexpr (BOX) { u = rt_ / (2 * N); const int ik_ = rt_ - u * (2 * N); i = ik_ / N; jk_ = ik_ - i * N; }           \
        else { u = fdiv(rt_, inv_m2); const int rem_ = rt_ - u * m2; i = rem_ >= m; jk_ = rem_ - i * m; }                      \
        double *un = UN(u);                                                                                                    \
        double ap_ = 0, ax_ = 0, adp_ = 0, adx_ = 0;                                                                           \
        if constexpr (BOX) {                                                                                                   \
            ap_ = un[W::P + i * N + jk_]; ax_ = sm[W::XV + i * N + jk_];                                                       \
            if (DS) { adp_ = un[W::DW + i * N + jk_]; adx_ = sm[W::DX + i * N + jk_]; }                                        \
        } else {                                                                                                               \
            _Pragma("unroll") for (int k_ = 0; k_ < N; ++k_) {                                                                 \
                const double a_ = A[jk_ * N + k_];                                                                             \
                ap_ += a_ * un[W::P + i * N + k_]; ax_ += a_ * sm[W::XV + i * N + k_];                                         \
                if (DS) { adp_ += a_ * un[W::DW + i * N + k_]; adx_ += a_ * sm[W::DX + i * N + k_]; }                          \
            }                                                                                                                  \
        }                                                                                                                      \
        const double yy_ = un[W::P + 2 * N], dy_ = DS ? un[W::DW + 2 * N] : 0.0;                                               \
        _Pragma("unroll") for (int rq_ = 0; rq_ < (BOX ? 4 : 2); ++rq_) {                                                      \
            const int ty = BOX ? rq_ >> 1 : rq_, j = BOX ? jk_ + (rq_ & 1) * N : jk_, ro = ty * m2 + i * m + j;                \
            const double sg_ = (BOX && (rq_ & 1)) ? -1.0 : 1.0, b_ = BC[j];                                                    \
            const double s = ty == 0 ? b_ * yy_ - sg_ * ap_ : b_ * (1.0 - yy_) - sg_ * (ax_ - ap_);                            \
            const double ds = ty == 0 ? b_ * dy_ - sg_ * adp_ : -(b_ * dy_) - sg_ * (adx_ - adp_);                             \
            (void)ds;
#define WG_ROWS_BEGIN(pl) WG_ROWS_BEGIN_(pl, false)
#define WG_ROWS_BEGIN_DS(pl) WG_ROWS_BEGIN_(pl, true)
#define WG_ROWS_END() }}

// BOX: the vertex's polytope is an axis-aligned box with its facets in the canonical order [+e_0 .. +e_{N-1}, -e_0 .. -e_{N-1}] (the
// lattice configurations; checked by the caller, canonical_box.h).  Facet j then has the single non-zero entry sg_j = +-1 at
// k_j = j mod N: every facet-row dot product is one term, K_h and the x-coupling X_e of a unit are DIAGONAL (plus the y row /
// column), and the loops over facets / coordinates below shrink accordingly.  Same algorithm, same operation order on the terms
// that remain (the dropped terms are exact zeros), so results agree with the generic instantiation to rounding of -0.0 + x.
template <int N, class T, bool BOX = false>
GCS_HD void wg_solve_vertex(const WgArgs<T> &a, int v, double rho, double mu_scale, double *sm, int &status_out, int &iters_out)
{
    using D = WD<N>;
    using SO = WSoc<N>;
    using W = std::conditional_t<BOX, WLBox<N>, WL<N>>;       // LDS layout: dense K_e / X_e / B_e per unit, or their structured forms
    constexpr int NW = D::NW, NX = D::NX, NB1 = D::NB1, Q = D::Q, NS = D::NS, TA = D::TA;
#ifndef GCS_WG_CONE_PAIR_MAXN
#define GCS_WG_CONE_PAIR_MAXN 3
#endif
    constexpr bool CONE_PAIR = N <= GCS_WG_CONE_PAIR_MAXN;      // the cone's two step bounds on two lanes side by side (solve_tail); at n = 6 the
                                                                // second lane costs the BOX instantiation two registers (167 -> 169: its third workgroup per CU)
    const bool prox = a.prox_q != nullptr;       // border-only problem with a separable quadratic (no blocks, no sides)
    const int lo = a.inc_ptr[v], d = prox ? 0 : a.inc_ptr[v + 1] - lo, d_in = prox ? 0 : a.deg_in[v], d_out = d - d_in;
    const bool sides = d > 0;
    const int p0 = a.poly_ptr[v], m = a.poly_ptr[v + 1] - p0;
    const int U = d + 1, R = 4 * m, m2 = 2 * m;
    const int US = W::unit_stride(m);
    const float inv_m2 = 1.0f / (float)m2;
    auto UN = [&](int u) -> double * { return sm + W::FIXED + u * US; };       // base of unit u
    const int oLAM = W::ROWS, oR1 = W::ROWS + R, oR2 = W::ROWS + 2 * R;   // facet-row arrays of a unit: duals, two work arrays
    double *const PA = sm + W::FIXED + pad2(U * US);
    const double *A = PA, *BC = PA + pad2(m * N), *CEN = sm + W::CEN;
    double *SOC = sm + W::SOC, *SC = sm + W::SC;
    int red_phase = 0;
    WG_STAMP_INIT();
    const int deg = (4 * m + 2) * (d + 1) + 1;
    const double inv_deg = 1.0 / (double)deg;
    auto side_of = [&](int u) { return (u - 1) >= d_in ? 1 : 0; };   // blocks 1..d_in incoming, the rest outgoing
    auto side_lo = [&](int s) { return s ? d_in + 1 : 1; };
    auto side_hi = [&](int s) { return s ? d : d_in; };              // inclusive

    // ---- load: polytope, targets; how far the targets moved since the vertex's warm-start record (warm_start.h) ----
    using WR = gcs_ws::WRec<N>;
    double *const wrec = (a.warm != nullptr && !prox) ? a.warm + a.warm_ptr[v] : nullptr;
    const int WUS = WR::unit_stride(m);
    double dtm = 0.0;
    Place pl0;
    WG_FOR_AT(t, d * NW, pl0.at(d * NW)) {
        const int e = t / NW, w = t - e * NW, edge = a.inc_edge[lo + e];
        const bool out = e >= d_in;
        const int inc = a.edge_major ? edge + (out ? 0 : a.E) : lo + e;      // state column of this incidence
        double *un = UN(e + 1);
        const double Tw = (double)a.zedge[(size_t)w * a.E + edge] - mu_scale * (double)a.mu[(size_t)w * a.NI + inc];
        // block targets: T1 (of O[:n]), T2 (of O[n:], outgoing only), Ty; the first word of an incoming edge is free
        int slot = -1;        // position of a PENALISED word in the unit's target array
        if (w == 2 * N) slot = 2 * N;
        else if (out) slot = w;
        else if (w < N) { un[W::TF + w] = Tw; un[W::TG + N + w] = 0.0; }
        else slot = w - N;
        if (slot >= 0) {
            un[W::TG + slot] = Tw;
            if (wrec) dtm = fmax(dtm, fabs(Tw - wrec[WR::UNITS + (e + 1) * WUS + WR::TG + slot]));
        }
    }
    WG_FOR_AT(t, m * N, pl0.at(m * N)) PA[t] = a.poly_A[(size_t)p0 * N + t];
    WG_FOR_AT(j, m + N, pl0.at(m + N)) {
        if (j < m) PA[pad2(m * N) + j] = a.poly_bc[p0 + j];
        else sm[W::CEN + (j - m)] = a.center[(size_t)v * N + (j - m)];
    }
    if (prox) {
        WG_FOR(k, NX + NW) {
            sm[W::PQ + k] = a.prox_q[(size_t)v * (NX + NW) + k];
            sm[W::PC + k] = a.prox_c[(size_t)v * (NX + NW) + k];
        }
    }
    bool use_warm = false;
    double mu_ref = gcs_ws::WS_COLD_REF, ws_dT = -1.0;      // ws_dT < 0: no comparable record
    if (wrec) {      // (workgroup-uniform; the reduction's barrier also publishes the loads above)
        const Red3 rt = wg_reduce(Red3{-dtm, 0.0, 0.0}, sm + W::RED, red_phase);
        const bool comparable = wg_uniform((int)(wrec[0] == 1.0 && wrec[1] == rho)) != 0;
        ws_dT = wg_uniform(comparable ? -rt.mn * rho : -1.0);
        use_warm = wg_uniform((int)(comparable && ws_dT <= gcs_ws::ws_theta(wrec))) != 0;
        if (use_warm) mu_ref = wg_uniform(fmax(gcs_ws::WS_MU_MIN, gcs_ws::WS_KAPPA * ws_dT));
    }
    const bool ws_started_warm = use_warm;
    const float inv_R2 = 1.0f / (float)(R / 2);

    WG_STAMP(0);
    // gradient entry k of unit u for the Newton right-hand side: smooth part G0, plus (corrector solve) G'kappa of the unit's
    // facet rows and, on unit 0, the cone's kappa (-ks on z1, +ks on z2)
    auto gval = [&](const double *un, int u, int k, bool wk) {
        double g = un[W::G0 + k];
        if (wk) {
            g += un[W::GU + k];
            if (u == 0 && k < 2 * N) g += k < N ? -SOC[SO::KS + 1 + k] : SOC[SO::KS + 1 + (k - N)];
        }
        return g;
    };
    // ---- Newton solve with the stored factors (oracle newton_solve): -> DX, DW of every unit, DNU, dt.  wk = corrector solve
    //      (the gradient carries the kappa terms); the gradient of the t row is SC_GT.  The HEAD of a solve (four kinds of
    //      task, one region each: t_e, their side sums, v_s, the right-hand side) does not need the border factor: for the
    //      affine solve the four ride in the regions of the side / border factorisation (below), for the corrector they are
    //      regions of their own (solve_head). ----
    auto te_tasks = [&](Place &pl, bool wk) {      // t_e = B_e (-g_e)
        WG_FOR_AT(t, d * NW, pl.at(d * NW)) {
            const int u = 1 + t / NW, i = t - (u - 1) * NW;
            double *un = UN(u);
            double s = 0;
            if constexpr (BOX) {       // B (-g) with B = diag(BD) + BRS BW BW'
                double wg = 0;
#pragma unroll
                for (int k = 0; k < NW; ++k) wg += un[W::BW + k] * gval(un, u, k, wk);
                s = -(un[W::BRS] * un[W::BW + i]) * wg;
                if (i < 2 * N) s -= un[W::BD + i] * gval(un, u, i, wk);
            } else {
#pragma unroll
                for (int k = 0; k < NW; ++k) s -= un[W::B + i * NW + k] * gval(un, u, k, wk);
            }
            un[W::TE + i] = s;
        }
    };
    auto bg_tasks = [&](Place &pl) {               // side sums of t_e; sum of X_e' t_e (the blocks in two halves)
        WG_FOR_AT(t, 2 * NW, pl.at(2 * NW)) {
            const int s = t / NW, i = t - s * NW;
            double acc = 0;
#pragma unroll 4
            for (int u = side_lo(s); u <= side_hi(s); ++u) acc += UN(u)[W::TE + i];
            sm[W::BG + t] = acc;
        }
        WG_FOR_AT(t, 2 * NX, pl.at(2 * NX)) {
            const int part = t / NX, c = t - part * NX, h = c / N;
            const int ulo = part ? 1 + d / 2 : 1, uhi = part ? d : d / 2;
            double acc = 0;
#pragma unroll 2
            for (int u = ulo; u <= uhi; ++u) {
                const double *un = UN(u);
                if constexpr (BOX) acc += un[W::XD + c] * un[W::TE + c] + un[W::XY + c] * un[W::TE + 2 * N];
                else {
#pragma unroll
                    for (int k = 0; k < N; ++k) acc += un[W::X + (h * N + k) * NX + c] * un[W::TE + h * N + k];
                    acc += un[W::X + 2 * N * NX + c] * un[W::TE + 2 * N];
                }
            }
            sm[W::XBG + t] = acc;
        }
    };
    auto v_tasks = [&](Place &pl) {                // v_s = Bs^{-1} (rp_s - Bg_s)
        WG_FOR_AT(t, 2 * NW, pl.at(2 * NW)) {
            const int s = t / NW, i = t - s * NW;
            const double *Bsi = sm + W::BSI + s * NW * NW;
            double acc = 0;
#pragma unroll
            for (int k = 0; k < NW; ++k) acc += Bsi[i * NW + k] * (sm[W::RP + s * NW + k] - sm[W::BG + s * NW + k]);
            sm[W::VV + t] = acc;
        }
    };
    auto rhs_tasks = [&](Place &pl, bool wk) {     // right-hand side in the (x, u = z1 - z2, z2, y_v) variables, t eliminated
        // four KINDS of row, each on its own wavefront: as branches of one loop the nine rows sat in one wavefront, which ran the four
        // bodies back to back (1 750 cycles for this region on benchmark4; the x rows alone are a 20-term sum)
        const double *u0 = UN(0);
        auto base_z = [&](int i) { return -gval(u0, 0, i, wk) - sm[W::VV + i] - sm[W::VV + NW + i]; };
        WG_FOR_AT(q, NX, pl.at(NX)) {             // x rows
            double r = -REG_DELTA * sm[W::XV + q] - sm[W::XBG + q] - sm[W::XBG + NX + q];
            if (prox) r -= sm[W::PQ + q] * (sm[W::XV + q] + CEN[q < N ? q : q - N] - sm[W::PC + q]);
            if (wk) r -= sm[W::SOL + q];       // sum over the units of G'kappa's x part (solve_head's first region put it there)
            for (int s = 0; s < 2; ++s) {
                const double *BXs = sm + W::BXS + s * NW * NX;
#pragma unroll
                for (int i = 0; i < NW; ++i) r -= BXs[i * NX + q] * sm[W::VV + s * NW + i];
            }
            sm[W::RHS + q] = r;
        }
        WG_FOR_AT(k, N, pl.at(N))                 // u rows: r_z1 + cv gt / c0
            sm[W::RHS + NX + k] = base_z(k) + SOC[SO::CV + k] * SC[SC_GT] * rcp(SC[SC_C0]);
        WG_FOR_AT(k, N, pl.at(N))                 // z2 rows: r_z1 + r_z2 (the cone terms cancel)
            sm[W::RHS + NX + N + k] = base_z(k) + base_z(N + k);
        WG_FOR_AT(k, 1, pl.at(1)) sm[W::RHS + NX + 2 * N + k] = base_z(2 * N);      // y_v row
    };
    auto solve_head = [&](bool wk) {
        {
            Place pl;
            te_tasks(pl, wk);
            if (wk) {      // rides here: the x part of G'kappa summed over the units, for the right-hand side (SOL is free until the solve)
                WG_FOR_AT(q, NX, pl.at(NX)) {
                    double acc = 0;
#pragma unroll 4
                    for (int u = 0; u <= d; ++u) acc += UN(u)[W::GX + q];
                    sm[W::SOL + q] = acc;
                }
            }
        }
        WG_ARRIVE(30);
        WG_SYNC();
        WG_STAMP(30);
        { Place pl; bg_tasks(pl); }
        WG_ARRIVE(31);
        WG_SYNC();
        WG_STAMP(31);
        { Place pl; v_tasks(pl); }
        WG_ARRIVE(32);
        WG_SYNC();
        WG_STAMP(32);
        { Place pl; rhs_tasks(pl, wk); }
        WG_ARRIVE(33);
        WG_SYNC();
        WG_STAMP(33);
    };
    // the cone's part of a direction, as soon as (d zeta, dt) exist: slack direction xs = (dt, d z1 - d z2) and dual direction
    // dl = kappa - lambda - W^{-2} xs (kappa = 0 for the affine direction)
    auto cone_dir = [&](int dt_slot, bool wk, double (&xs)[Q], double (&dl)[Q]) {
        const double *u0 = UN(0);
        double wb[Q], ys[Q];
        xs[0] = SC[dt_slot];
#pragma unroll
        for (int k = 0; k < N; ++k) xs[1 + k] = u0[W::DW + k] - u0[W::DW + N + k];
#pragma unroll
        for (int k = 0; k < Q; ++k) wb[k] = SOC[SO::WB + k];
        soc_apply_W2<Q>(wb, SC[SC_ETA], xs, ys);
#pragma unroll
        for (int k = 0; k < Q; ++k) dl[k] = (wk ? SOC[SO::KS + k] : 0.0) - SOC[SO::LS + k] - ys[k];
    };
    // The tail of a solve.  (1) one wavefront: substitutions with the border factor; (2) w_s, dx, d zeta, dt spread over the
    // threads; (3) a wave-local pipeline: every item wavefront repeats the 2 NW values of d nu_s for itself, then runs
    // r_e -> d w_e for ITS units, while the cone thread (last wavefront) does the cone's three parts.  (Merging (1) and (2) into
    // the one wavefront was measured slower: 2 250 against 1 830 cycles.)
    auto solve_tail = [&](int dt_slot, bool wk) {
        WG_WAVE0() wave_ldl_solve<NB1>(sm + W::M, sm + W::PIVM, sm + W::RHS, sm + W::SOL);
        WG_ARRIVE(34);
        WG_SYNC();
        WG_STAMP(34);
        // solution back in (x, z1, z2, y_v): dz1 = du + dz2
        auto dzeta = [&](int i) { return i < N ? sm[W::SOL + NX + i] + sm[W::SOL + NX + N + i] : sm[W::SOL + NX + i]; };
        Place plw;
        WG_FOR_AT(t, 2 * NW, plw.at(2 * NW)) {     // w_s = d zeta + (rp_s - Bg_s) + BXs dx
            const int s = t / NW, i = t - s * NW;
            const double *BXs = sm + W::BXS + s * NW * NX;
            double acc = dzeta(i) + (sm[W::RP + t] - sm[W::BG + t]);
#pragma unroll
            for (int c = 0; c < NX; ++c) acc += BXs[i * NX + c] * sm[W::SOL + c];
            sm[W::WW + t] = acc;
        }
        WG_FOR_AT(t, NX + NW + 1, plw.at(NX + NW + 1)) {
            if (t < NX) sm[W::DX + t] = sm[W::SOL + t];
            else if (t < NX + NW) UN(0)[W::DW + (t - NX)] = dzeta(t - NX);
            else {                            // t recovered: c0 dt + cv'(dz1 - dz2) = -gt
                double acc = -SC[SC_GT];
#pragma unroll
                for (int k = 0; k < N; ++k) acc -= SOC[SO::CV + k] * sm[W::SOL + NX + k];
                SC[dt_slot] = acc * rcp(SC[SC_C0]);
            }
        }
        WG_ARRIVE(36);
        WG_SYNC();
        WG_STAMP(36);
        // the cone, beside the pipeline below.  Two lanes of the last wavefront form the cone's directions (both the same values) and
        // then ONE step bound each, side by side: lane 0 the slack side, lane 1 the dual side (a square root and two reciprocal chains
        // each).  (Round 4, measured and rejected: a third lane on another wavefront forming the corrector's second-order cone term
        // here, so that the G'kappa region's cone thread has three multiplications left -- that wavefront also owns a unit of the
        // pipeline on vertices of degree >= 7 and became the last to arrive: benchmark4 9 555 it/s with it, 9 580 without.)
        auto cone_bounds = [&](int l) {      // l = 0: the slack side's bound, 1: the dual side's, 2: both (one lane, back to back)
            const int oDS = wk ? SO::DSS : SO::DSSA, oDL = wk ? SO::DLS : SO::DLSA;
            double xs[Q], dl[Q], bs[Q], dr[Q];
            cone_dir(dt_slot, wk, xs, dl);
            double c1c = 0, c2c = 0;
#pragma unroll
            for (int k = 0; k < Q; ++k) {
                const double sk = SOC[SO::SS + k], lk = SOC[SO::LS + k];
                SOC[oDS + k] = xs[k]; SOC[oDL + k] = dl[k];
                c1c += sk * dl[k] + lk * xs[k];
                c2c += xs[k] * dl[k];
                bs[k] = l == 1 ? lk : sk; dr[k] = l == 1 ? dl[k] : xs[k];
            }
            SC[SC_C1C] = c1c; SC[SC_C2C] = c2c;
            SC[l == 1 ? SC_AMAXC2 : SC_AMAXC] = gcs_math::soc_max_step<Q>(bs, dr);
            if (l == 2) SC[SC_AMAXC2] = gcs_math::soc_max_step<Q>(SOC + SO::LS, dl);
        };
        if constexpr (CONE_PAIR) { WG_CONE2(l) cone_bounds(l); }
        else { WG_CONE() cone_bounds(2); }
        WG_REPL_FOR(t, 2 * NW) {      // d nu_s = Bs^{-1} w_s
            const int s = t / NW, i = t - s * NW;
            const double *Bsi = sm + W::BSI + s * NW * NW;
            double acc = 0;
#pragma unroll
            for (int k = 0; k < NW; ++k) acc += Bsi[i * NW + k] * sm[W::WW + s * NW + k];
            sm[W::DNU + t] = acc;
        }
        WG_WAVE_SYNC();
        WG_ITEM_FOR(q, i, d, NW) {      // r_e = -g_e + d nu_side - X_e dx
            const int u = 1 + q;
            double *un = UN(u);
            double acc = -gval(un, u, i, wk) + sm[W::DNU + side_of(u) * NW + i];
            if constexpr (BOX) {
                if (i < 2 * N) acc -= un[W::XD + i] * sm[W::DX + i];
                else {
#pragma unroll
                    for (int c = 0; c < NX; ++c) acc -= un[W::XY + c] * sm[W::DX + c];
                }
            } else if (i < 2 * N) {
                const int h = i / N;
#pragma unroll
                for (int k = 0; k < N; ++k) acc -= un[W::X + i * NX + h * N + k] * sm[W::DX + h * N + k];
            } else {
#pragma unroll
                for (int c = 0; c < NX; ++c) acc -= un[W::X + i * NX + c] * sm[W::DX + c];
            }
            un[W::RV + i] = acc;
        }
        WG_WAVE_SYNC();
        WG_ITEM_FOR(q, i, d, NW) {      // d w_e = B_e r_e
            double *un = UN(1 + q);
            double acc = 0;
            if constexpr (BOX) {
                double wr = 0;
#pragma unroll
                for (int k = 0; k < NW; ++k) wr += un[W::BW + k] * un[W::RV + k];
                acc = (un[W::BRS] * un[W::BW + i]) * wr;
                if (i < 2 * N) acc += un[W::BD + i] * un[W::RV + i];
            } else {
#pragma unroll
                for (int k = 0; k < NW; ++k) acc += un[W::B + i * NW + k] * un[W::RV + k];
            }
            un[W::DW + i] = acc;
        }
        WG_ARRIVE(39);
        WG_SYNC();
        WG_STAMP(39);
    };

    int status = -1, it = 0, it_total = 0, stalled = 0;
    bool saved = false, init_pending = true;
    // a warm solve that fails is repeated cold: the same loop, started again from the fixed interior point (one loop, not two nested:
    // the nested form cost the n = 3 kernel 12 registers)
#define WG_FAIL_OR_RESTART()                                                                                                   \
    {                                                                                                                          \
        if (!use_warm) break;                                                                                                  \
        it_total += it; it = -1;                                                                                               \
        use_warm = false; mu_ref = gcs_ws::WS_COLD_REF; init_pending = true;                                                   \
        WG_SYNC();                                                                                                             \
        continue;                                                                                                              \
    }
    for (it = 0;; ++it) {
    if (init_pending) {
    // ---- start point: the record's iterate, or the strictly feasible point of oracle_solve_vertex ----
        init_pending = false; stalled = 0; saved = false;
        Place pls;
        if (use_warm) {
            const double *wu = wrec + WR::UNITS;
            WG_FOR_AT(t, U * NW, pls.at(U * NW)) {
                const int u = t / NW, k = t - u * NW;
                UN(u)[W::P + k] = wu[u * WUS + WR::P + k];
            }
            WG_FOR_AT(t, U * (R / 2), pls.at(U * (R / 2))) {      // row duals: f32 in the record, two per 8-byte word (R = 4m is even)
                const int u = fdiv(t, inv_R2), r2 = t - u * (R / 2);
                const WgF2 v = reinterpret_cast<const WgF2 *>(wu + u * WUS + WR::LAM)[r2];
                UN(u)[oLAM + 2 * r2] = (double)v.a; UN(u)[oLAM + 2 * r2 + 1] = (double)v.b;
            }
            WG_FOR_AT(t, 2 * U, pls.at(2 * U)) UN(t >> 1)[W::LB + (t & 1)] = wu[(t >> 1) * WUS + WR::LB + (t & 1)];
            WG_FOR_AT(k, NX + 2 * NW, pls.at(NX + 2 * NW)) {
                if (k < NX) sm[W::XV + k] = wrec[WR::XV + k];
                else sm[W::NU + (k - NX)] = wrec[WR::NU + (k - NX)];
            }
            // (t and the cone's dual: re-centred by the cone thread in the first rows region)
        } else {
            WG_FOR_AT(t, U * NW, pls.at(U * NW)) {
                const int u = t / NW, k = t - u * NW;
                double val = 0.0;
                if (k == 2 * N) val = u == 0 ? 0.5 : 0.5 / (double)(side_of(u) ? d_out : d_in);
                UN(u)[W::P + k] = val;
            }
            WG_FOR_AT(k, NX + 2 * NW, pls.at(NX + 2 * NW)) {
                if (k < NX) sm[W::XV + k] = 0.0;
                else sm[W::NU + (k - NX)] = 0.0;
            }
            WG_ONE() {
                SC[SC_T] = 1.0;
                SOC[SO::LS] = 1.0;
                for (int k = 1; k < Q; ++k) SOC[SO::LS + k] = 0.0;
            }
        }
        WG_SYNC();
    }
        const bool first_warm = use_warm && it == 0;      // the re-centring Newton step of a warm solve
        // ================= rows: slacks, duals at the start, D = l/s, complementarity =================
        double acc = 0.0; int bad = 0;
        Place plr;
        WG_ROWS_BEGIN(plr)
            const double is = rcp1(s);
            if (it == 0 && !use_warm) un[oLAM + ro] = is;
            const double l = un[oLAM + ro];
            un[oR1 + ro] = l * is;
            acc += s * l;
            if (!(s > 0.0)) bad = 1;
        WG_ROWS_END()
        WG_FOR_AT(u, U, plr.at(U)) {      // bounds 0 <= y <= 1 of every unit
            double *un = UN(u);
            const double yy = un[W::P + 2 * N], s5 = yy, s6 = 1.0 - yy;
            if (it == 0 && !use_warm) { un[W::LB] = rcp1(s5); un[W::LB + 1] = rcp1(s6); }
            acc += s5 * un[W::LB] + s6 * un[W::LB + 1];
            if (!(s5 > 0.0) || !(s6 > 0.0)) bad = 1;
        }
        WG_FOR_AT(tg, U * NW, plr.at(U * NW)) {
            // (rides in this region: it needs the primal iterate only)
            // gradient of the smooth objective + equality multipliers + Tikhonov term (no facet-row part):
            // blocks: consensus penalty (admm_solver_v3.py:392-413) and the edge cost 1e-4 y_e (:387-388)
            const int u = tg / NW, k = tg - u * NW;
            double *un = UN(u);
            const double *p = un + W::P;
            double g;
            if (u == 0) {
                g = sm[W::NU + k] + sm[W::NU + NW + k] + REG_DELTA * p[k];
                if (prox) {     // z = p + y cen (the border unknowns are centred like the blocks'): q_z (z - c_z), chain rule for y
                    const double *qz = sm + W::PQ + NX, *cz = sm + W::PC + NX;
                    const double yy = p[2 * N];
                    if (k < 2 * N) g += qz[k] * (p[k] + yy * CEN[k < N ? k : k - N] - cz[k]);
                    else {
                        g += qz[2 * N] * (yy - cz[2 * N]);
#pragma unroll
                        for (int c = 0; c < 2 * N; ++c) g += CEN[c < N ? c : c - N] * qz[c] * (p[c] + yy * CEN[c < N ? c : c - N] - cz[c]);
                    }
                }
            }
            else {
                const bool out = side_of(u);
                const double *tg_ = un + W::TG, *nu = sm + W::NU + (out ? NW : 0);
                const double yy = p[2 * N];
                if (k < N) g = rho * (p[k] + yy * CEN[k] - tg_[k]);
                else if (k < 2 * N) g = out ? rho * (p[k] + yy * CEN[k - N] - tg_[k]) : 0.0;
                else {
                    g = rho * (yy - tg_[2 * N]) + a.eps_edge;
#pragma unroll
                    for (int c = 0; c < N; ++c) {
                        g += CEN[c] * rho * (p[c] + yy * CEN[c] - tg_[c]);
                        if (out) g += CEN[c] * rho * (p[N + c] + yy * CEN[c] - tg_[N + c]);
                    }
                }
                g += REG_DELTA * p[k] - nu[k];
            }
            un[W::G0 + k] = g;
        }
        WG_CONE() {          // the cone: s = (t, z1 - z2)
            const double *u0 = UN(0);
            if (first_warm) {      // cone pair re-centred at mu_ref: t^2 - mu_ref t - |u|^2 = 0, lambda = (1, -u / t)
                double uu = 0;
                for (int k = 0; k < N; ++k) { const double uk = u0[W::P + k] - u0[W::P + N + k]; uu += uk * uk; }
                const double tn = 0.5 * (mu_ref + sqrt_nr(mu_ref * mu_ref + 4.0 * uu)), itn = rcp(tn);
                SC[SC_T] = tn;
                SOC[SO::LS] = 1.0;
                for (int k = 0; k < N; ++k) SOC[SO::LS + 1 + k] = -(u0[W::P + k] - u0[W::P + N + k]) * itn;
            }
            SOC[SO::SS] = SC[SC_T];
            for (int k = 0; k < N; ++k) SOC[SO::SS + 1 + k] = u0[W::P + k] - u0[W::P + N + k];
            for (int k = 0; k < Q; ++k) acc += SOC[SO::SS + k] * SOC[SO::LS + k];
            if (!gcs_math::soc_interior<Q>(SOC + SO::SS)) bad = 1;
        }
        WG_ARRIVE(1);
        const Red3 r0 = wg_reduce(Red3{bad ? -1.0 : 1.0, acc, 0.0}, sm + W::RED, red_phase);
        WG_STAMP(1);
        // (workgroup-uniform scalars go to scalar registers: the branches on them are then scalar branches, not EXEC-masked regions)
        const double gap = wg_uniform(r0.s1);
        const double mu = wg_uniform(r0.mn < 0.0 ? 0.0 / 0.0 : gap * inv_deg);
        WG_FENCE();
        if (wrec != nullptr && !saved && it >= 1 && mu <= gcs_ws::WS_SAVE * mu_ref) {
            // the record the next solve of this vertex restarts from (nothing below reads it; the iterate is stable until the update)
            saved = true;
            double *wu = wrec + WR::UNITS;
            Place plv;
            WG_FOR_AT(t, U * NW, plv.at(U * NW)) {
                const int u = t / NW, k = t - u * NW;
                wu[u * WUS + WR::P + k] = UN(u)[W::P + k];
                if (u > 0) wu[u * WUS + WR::TG + k] = UN(u)[W::TG + k];      // (unit 0, the border, has no targets)
            }
            WG_FOR_AT(t, U * (R / 2), plv.at(U * (R / 2))) {
                const int u = fdiv(t, inv_R2), r2 = t - u * (R / 2);
                WgF2 v; v.a = (float)UN(u)[oLAM + 2 * r2]; v.b = (float)UN(u)[oLAM + 2 * r2 + 1];
                reinterpret_cast<WgF2 *>(wu + u * WUS + WR::LAM)[r2] = v;
            }
            WG_FOR_AT(t, 2 * U, plv.at(2 * U)) wu[(t >> 1) * WUS + WR::LB + (t & 1)] = UN(t >> 1)[W::LB + (t & 1)];
            WG_FOR_AT(k, NX + 2 * NW, plv.at(NX + 2 * NW)) {
                if (k < NX) wrec[WR::XV + k] = sm[W::XV + k];
                else wrec[WR::NU + (k - NX)] = sm[W::NU + (k - NX)];
            }
            WG_ONE() { wrec[0] = 1.0; wrec[1] = rho; }
        }
        WG_FENCE();
        {   // stop on the barrier parameter alone (oracle/gcs_oracle.c); a vanishing step = precision exhausted
            const bool conv = !first_warm && (mu <= a.ipm_tol || (stalled && mu <= 1e3 * a.ipm_tol));
            bool stop = conv;
            // (a WARM solve does not leave through the precision-exhausted rule: it is repeated cold -- oracle/gcs_oracle.c)
            status = conv ? ((use_warm && !(mu <= a.ipm_tol)) ? -7 : 0) : -1;
            if (!(mu > 0.0)) { stop = true; status = -3; }
            if (!stop && it >= a.ipm_max_iter) { stop = true; status = -1; }
            if (stop) { if (status == 0) break; WG_FAIL_OR_RESTART(); }
        }

        // ================= Hessian pieces of every unit, objective gradient, cone scaling =================
        // K_u = [K1 0 k1y; 0 K2 k2y; . . kyy] (rows a+b), X_u = d(unit)/d(x) coupling (rows b); reference rows
        // admm_solver_v3.py:420-426 (unit 0) and :434-440 (blocks).  Tasks are ordered BY KIND (all matrix entries, then all
        // y-column entries, ...) so that the threads of a wavefront run the same loop: mixed kinds serialise per wavefront.
        WG_CONE() {
            // cone scaling: W^{-2} = eta^{-2}(2 v v' - J), v = (wb0, -wb1); t is eliminated in closed form (c0, cv, Su):
            // a numerical pivot on t cancels catastrophically once the cone is active
            double wb[Q], eta = 1.0;
            if (!soc_scaling_wb<Q>(SOC + SO::SS, SOC + SO::LS, wb, eta)) SC[SC_CONEFAIL] = 1.0;
            else {
                SC[SC_CONEFAIL] = 0.0; SC[SC_ETA] = eta;
                const double ieta = rcp(eta), ie2 = ieta * ieta;
#pragma unroll
                for (int i = 0; i < Q; ++i) SOC[SO::WB + i] = wb[i];
                {
                    double ls[Q], lt[Q];
#pragma unroll
                    for (int i = 0; i < Q; ++i) ls[i] = SOC[SO::LS + i];
                    soc_apply_W<Q>(wb, eta, ls, lt);
#pragma unroll
                    for (int i = 0; i < Q; ++i) SOC[SO::LT + i] = lt[i];
                }
                const double den = 2.0 * wb[0] * wb[0] - 1.0, g2 = 2.0 * rcp(den);
                SC[SC_C0] = ie2 * den;
                for (int k = 0; k < N; ++k) {
                    SOC[SO::CV + k] = -ie2 * 2.0 * wb[0] * wb[1 + k];
                    for (int l = 0; l < N; ++l) SOC[SO::SU + k * N + l] = ie2 * ((k == l ? 1.0 : 0.0) - g2 * wb[1 + k] * wb[1 + l]);
                }
            }
            SC[SC_GT] = 1.0;
        }
        // placement: every kind starts on its own wavefront boundary (a wavefront that holds tasks of two kinds runs both loops
        // back to back: with the K_yy tasks right behind the K entries the second wavefront was this region's critical path,
        // 3 900 cycles; measured by duplicating one kind at a time)
        Place pla;
        const int sK_ = pla.at(BOX ? U * 2 * N : U * 2 * NS), sKy_ = BOX ? 0 : pla.at(U * 2 * N), sKyy_ = pla.at(U);
        (void)sKy_;
        if constexpr (BOX) {
            // structured forms: one task per (unit, half, coordinate k) -- the diagonal entries of K_h and X_h and the k-th entries
            // of the y column / y row, from the two facets +-e_k
            WG_FOR_AT(t, U * 2 * N, sK_) {
                const int u = t / (2 * N), ik = t - u * (2 * N), i = ik / N, k = ik - i * N;
                double *un = UN(u);
                const double *Da = un + oR1, *Db = Da + m2;
                const bool blk = u > 0, out = blk && side_of(u);
                const double dp = Da[i * m + k] + Db[i * m + k], dn = Da[i * m + k + N] + Db[i * m + k + N];
                const double bp = BC[k], bn = -BC[k + N];       // b_j a_j[k] of the two facets on coordinate k
                double sk = dp + dn;
                sk += REG_DELTA; if (blk && (i == 0 || out)) sk += rho; if (prox) sk += sm[W::PQ + NX + i * N + k];
                un[W::KD + ik] = sk;
                un[W::XD + ik] = -(Db[i * m + k] + Db[i * m + k + N]);
                double sy = -(dp * bp) - dn * bn;
                if (blk && (i == 0 || out)) sy += rho * CEN[k];
                if (prox) sy += sm[W::PQ + NX + i * N + k] * CEN[k];
                un[W::KY + ik] = sy;
                un[W::XY + ik] = Db[i * m + k] * bp + Db[i * m + k + N] * bn;
            }
        } else {
            WG_FOR_AT(t, U * 2 * NS, sK_) {          // entries of K_i and X_i (packed lower, both halves)
                const int u = t / (2 * NS), q = t - u * (2 * NS), i = q / NS, pq = q - i * NS;
                double *un = UN(u);
                double *K = un + W::K, *X = un + W::X;
                const double *Da = un + oR1, *Db = Da + m2;
                const bool blk = u > 0, out = blk && side_of(u);
                int k, l;
                if constexpr (N == 2) { k = pq > 0; l = pq > 1; }      // packed lower index of a 2 x 2 block: (0,0) (1,0) (1,1)
                else tri_decode(pq, k, l);
                double sk = 0, sx = 0;
    #pragma unroll 4
                for (int j = 0; j < m; ++j) {
                    const double aa = A[j * N + k] * A[j * N + l];
                    sk += (Da[i * m + j] + Db[i * m + j]) * aa;
                    sx += Db[i * m + j] * aa;
                }
                if (k == l) { sk += REG_DELTA; if (blk && (i == 0 || out)) sk += rho; if (prox) sk += sm[W::PQ + NX + i * N + k]; }
                K[(i * N + k) * NW + i * N + l] = sk; K[(i * N + l) * NW + i * N + k] = sk;
                X[(i * N + k) * NX + i * N + l] = -sx; X[(i * N + l) * NX + i * N + k] = -sx;
                const int o = (1 - i) * N;      // the two halves are not coupled directly
                K[(i * N + k) * NW + o + l] = 0.0; K[(i * N + l) * NW + o + k] = 0.0;
                X[(i * N + k) * NX + o + l] = 0.0; X[(i * N + l) * NX + o + k] = 0.0;
            }
            WG_FOR_AT(t, U * 2 * N, sKy_) {           // y column of K, y row of X
                const int u = t / (2 * N), ik = t - u * (2 * N), i = ik / N, k = ik - i * N;
                double *un = UN(u);
                const double *Da = un + oR1, *Db = Da + m2;
                const bool blk = u > 0, out = blk && side_of(u);
                double sk = 0, sx = 0;
    #pragma unroll 4
                for (int j = 0; j < m; ++j) {
                    const double ba = BC[j] * A[j * N + k];
                    sk -= (Da[i * m + j] + Db[i * m + j]) * ba;
                    sx += Db[i * m + j] * ba;
                }
                if (blk && (i == 0 || out)) sk += rho * CEN[k];
                if (prox) sk += sm[W::PQ + NX + i * N + k] * CEN[k];
                un[W::K + (i * N + k) * NW + 2 * N] = sk; un[W::K + 2 * N * NW + i * N + k] = sk;
                un[W::X + 2 * N * NX + i * N + k] = sx;
            }
        }
        WG_FOR_AT(u, U, sKyy_) {                   // K_yy
            double *un = UN(u);
            const double *Da = un + oR1, *Db = Da + m2;
            const bool blk = u > 0, out = blk && side_of(u);
            double sk = 0;
#pragma unroll 4
            for (int j = 0; j < m2; ++j) { const double b = BC[j >= m ? j - m : j]; sk += (Da[j] + Db[j]) * b * b; }
            const double yy = un[W::P + 2 * N];
            sk += un[W::LB] * rcp1(yy) + un[W::LB + 1] * rcp1(1.0 - yy) + REG_DELTA;
            if (blk) {
                double cc = 0;
#pragma unroll
                for (int k = 0; k < N; ++k) cc += CEN[k] * CEN[k];
                sk += rho * (1.0 + (out ? 2.0 : 1.0) * cc);
            }
            if (prox) {
                sk += sm[W::PQ + NX + 2 * N];
#pragma unroll
                for (int k = 0; k < 2 * N; ++k) sk += sm[W::PQ + NX + k] * CEN[k < N ? k : k - N] * CEN[k < N ? k : k - N];
            }
            if constexpr (BOX) un[W::KYY] = sk; else un[W::K + 2 * N * NW + 2 * N] = sk;
        }
        WG_ARRIVE(2);
        WG_SYNC();
        WG_STAMP(2);
        if (wg_uniform(SC[SC_CONEFAIL]) != 0.0) {
            status = (mu <= 1e3 * a.ipm_tol && !use_warm) ? 0 : -4;
            if (status == 0) break;
            WG_FAIL_OR_RESTART();
        }

        // ================= blocks: explicit inverse B_e = K_e^{-1}, then B_e X_e =================
        // K_e = [K1 0 k1; 0 K2 k2; k1' k2' kappa]: the two halves of O_e are coupled only through y_e, so the Cholesky factor
        // in natural order is [L1 0 0; 0 L2 0; l1' l2' lambda] and the inverse has a closed form in the two N x N inverses:
        //   w_h = K_h^{-1} k_h,  s = kappa - k1'w1 - k2'w2,
        //   B = [K1^{-1} + w1 w1'/s   w1 w2'/s   -w1/s;   .   K2^{-1} + w2 w2'/s   -w2/s;   .   .   1/s].
        // Region 1, one thread per (block, half): Cholesky of K_h, K_h^{-1}, w_h, k_h'w_h, all in registers (the pivots are those
        // of the (2n+1)-dimensional factorisation: same clamp rule).  Region 2, one thread per entry: the rank-one term.
        // (The generic tiled Cholesky + column solves of the whole block took 2 000 / 8 300 cycles at n = 2 / 6.)
        WG_FOR(t, 2 * d) {
            const int u = 1 + (t >> 1), h = t & 1;
            double *un = UN(u);
            if constexpr (BOX) {        // K_h is diagonal: its inverse, w_h = K_h^{-1} k_h and k_h'w_h entry by entry
                double ch = 0;
#pragma unroll
                for (int r = 0; r < N; ++r) {
                    const double dd = un[W::KD + h * N + r], kr = un[W::KY + h * N + r];
                    const double inv = rsqrt_nr(dd), lr = kr * inv;       // (the pivot of a diagonal block is its entry: no clamp can fire)
                    un[W::BD + h * N + r] = inv * inv;
                    un[W::BW + h * N + r] = lr * inv;
                    ch += lr * lr;
                }
                un[W::RV + h] = ch;
            } else {
            double a[NS], kv[N], pv[N];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                kv[i] = un[W::K + (h * N + i) * NW + 2 * N];
#pragma unroll
                for (int j = 0; j <= i; ++j) a[pki(i, j)] = un[W::K + (h * N + i) * NW + h * N + j];
            }
#pragma unroll
            for (int j = 0; j < N; ++j) {           // Cholesky, packed lower; a(i,j) becomes L(i,j), pv the inverse pivots
                const double od = a[pki(j, j)];
                double dj = od;
#pragma unroll
                for (int k = 0; k < j; ++k) dj -= a[pki(j, k)] * a[pki(j, k)];
                if (!(dj > CHOL_SKIP * od)) dj = od > 0.0 ? CHOL_SKIP * od : 1.0;
                const double inv = rsqrt_nr(dj);
                pv[j] = inv;
#pragma unroll
                for (int i = j + 1; i < N; ++i) {
                    double sij = a[pki(i, j)];
#pragma unroll
                    for (int k = 0; k < j; ++k) sij -= a[pki(i, k)] * a[pki(j, k)];
                    a[pki(i, j)] = sij * inv;
                }
            }
            double (&li)[NS] = a;                  // L^{-1} (lower) IN PLACE of L: column c reads columns >= c of L only, and an entry
#pragma unroll                                  // of column c is overwritten after its last use (21 doubles fewer at n = 6)
            for (int c = 0; c < N; ++c)
#pragma unroll
                for (int i = c; i < N; ++i) {
                    double sx = (i == c) ? 1.0 : 0.0;
#pragma unroll
                    for (int k = c; k < i; ++k) sx -= a[pki(i, k)] * li[pki(k, c)];
                    li[pki(i, c)] = sx * pv[i];
                }
#pragma unroll
            for (int r = 0; r < N; ++r)             // K_h^{-1} = L^{-T} L^{-1}, stored (both triangles) into the half's block of B
#pragma unroll
                for (int c = 0; c <= r; ++c) {
                    double acc = 0;
#pragma unroll
                    for (int i = r; i < N; ++i) acc += li[pki(i, r)] * li[pki(i, c)];
                    un[W::B + (h * N + r) * NW + h * N + c] = acc;
                    un[W::B + (h * N + c) * NW + h * N + r] = acc;
                }
            // l = L^{-1} k_h (the y row of the factor), k_h' K_h^{-1} k_h = |l|^2 exactly as the Cholesky pivot forms it (through the
            // explicit inverse the Schur complement s loses cond(K_h) instead of its square root), w_h = L^{-T} l
            double lv[N], ch = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double acc = 0;
#pragma unroll
                for (int c = 0; c <= i; ++c) acc += li[pki(i, c)] * kv[c];
                lv[i] = acc;
                ch += acc * acc;
            }
#pragma unroll
            for (int r = 0; r < N; ++r) {
                double acc = 0;
#pragma unroll
                for (int i = r; i < N; ++i) acc += li[pki(i, r)] * lv[i];
                un[W::PIV + h * N + r] = acc;
            }
            un[W::RV + h] = ch;
            }
        }
        WG_ARRIVE(3);
        WG_SYNC();
        WG_STAMP(3);
        if constexpr (BOX) {
            // 1 / s of the y pivot (every thread of the unit forms it: no region of its own), the last entry of BW, and
            // BQ = X_e' BW (B_e X_e = diag(BD XD) + BRS BW BQ' is never formed)
            WG_FOR(t, d * NW) {
                const int u = 1 + t / NW, c = t - (u - 1) * NW;
                double *un = UN(u);
                if (c < 2 * N) un[W::BQ + c] = un[W::BW + c] * un[W::XD + c] - un[W::XY + c];
                else {
                    const double kap = un[W::KYY];
                    double sy = kap - un[W::RV] - un[W::RV + 1];
                    if (!(sy > CHOL_SKIP * kap)) sy = kap > 0.0 ? CHOL_SKIP * kap : 1.0;
                    un[W::BRS] = rcp(sy);
                    un[W::BW + 2 * N] = -1.0;
                }
            }
            WG_ARRIVE(4);
            WG_SYNC();
            WG_STAMP(4);
            if (!first_warm) {      // affine solve, head 1/4 (a re-centring iteration has no affine solve: none of its four head steps)
                { Place plx; te_tasks(plx, false); }
                WG_SYNC();
            }
            WG_STAMP(5);
        } else {
        WG_FOR(t, d * NW * NW) {
            const int u = 1 + t / (NW * NW), ij = t - (u - 1) * (NW * NW), i = ij / NW, j = ij - i * NW;
            double *un = UN(u);
            const double kap = un[W::K + 2 * N * NW + 2 * N];
            double sy = kap - un[W::RV] - un[W::RV + 1];
            if (!(sy > CHOL_SKIP * kap)) sy = kap > 0.0 ? CHOL_SKIP * kap : 1.0;
            const double rs = rcp(sy);
            const double wi = i < 2 * N ? un[W::PIV + i] : -1.0, wj = j < 2 * N ? un[W::PIV + j] : -1.0;
            const double base = (i < 2 * N && j < 2 * N && i / N == j / N) ? un[W::B + ij] : 0.0;
            un[W::B + ij] = base + wi * wj * rs;
        }
        WG_ARRIVE(4);
        WG_SYNC();
        WG_STAMP(4);
        Place plx;
        WG_FOR_AT(t, d * NW * NX, plx.at(d * NW * NX)) {     // B_e X_e, written over the (now dead) factor of the block
            const int u = 1 + t / (NW * NX), ic = t - (u - 1) * (NW * NX), i = ic / NX, c = ic - i * NX, h = c / N;
            double *un = UN(u);
            double s = un[W::B + i * NW + 2 * N] * un[W::X + 2 * N * NX + c];
#pragma unroll
            for (int k = 0; k < N; ++k) s += un[W::B + i * NW + h * N + k] * un[W::X + (h * N + k) * NX + c];
            un[W::K + ic] = s;
        }
        if (!first_warm) te_tasks(plx, false);            // affine solve, head 1/4
        WG_ARRIVE(5);
        WG_SYNC();
        WG_STAMP(5);
        }
        // ================= side sums: Bs, BXs, X'BX, equality residuals =================
        Place plq;
        // (B_s and X'BX are symmetric and only their lower triangles are read -- by the side factorisation and by the border assembly)
        constexpr int NWT = NW * (NW + 1) / 2, NXT = NX * (NX + 1) / 2;
        WG_FOR_AT(t0, 2 * NWT, plq.at(2 * NWT)) {
            const int s = t0 / NWT;
            int i, j;
            tri_decode(t0 - s * NWT, i, j);
            const int q = i * NW + j, t = s * NW * NW + q;
            double acc2 = 0;
            if constexpr (BOX) {       // B_e = diag(BD) + BRS BW BW'
#pragma unroll 4
                for (int u = side_lo(s); u <= side_hi(s); ++u) {
                    const double *un = UN(u);
                    acc2 += (un[W::BRS] * un[W::BW + i]) * un[W::BW + j];
                    if (i == j && i < 2 * N) acc2 += un[W::BD + i];
                }
            } else {
#pragma unroll 4
                for (int u = side_lo(s); u <= side_hi(s); ++u) acc2 += UN(u)[W::B + q];
            }
            sm[W::BS + t] = acc2;
        }
        WG_FOR_AT(tt, 2 * NW * NX, plq.at(2 * NW * NX)) {
            const int s = tt / (NW * NX), q = tt - s * NW * NX;
            double acc2 = 0;
            if constexpr (BOX) {       // B_e X_e = diag(BD XD) + BRS BW BQ'
                const int i = q / NX, c = q - i * NX;
#pragma unroll 4
                for (int u = side_lo(s); u <= side_hi(s); ++u) {
                    const double *un = UN(u);
                    acc2 += (un[W::BRS] * un[W::BW + i]) * un[W::BQ + c];
                    if (i == c) acc2 += un[W::BD + c] * un[W::XD + c];
                }
            } else {
#pragma unroll 4
                for (int u = side_lo(s); u <= side_hi(s); ++u) acc2 += UN(u)[W::K + q];
            }
            sm[W::BXS + tt] = acc2;
        }
        WG_FOR_AT(tp0, 2 * NXT, plq.at(2 * NXT)) {      // sum_e X_e' (B_e X_e), the blocks in two halves (shorter chains)
            const int part = tp0 / NXT;
            int r, c;
            tri_decode(tp0 - part * NXT, r, c);
            const int tp = part * NX * NX + r * NX + c, h = r / N;
            (void)h;
            const int ulo = part ? 1 + d / 2 : 1, uhi = part ? d : d / 2;
            double acc2 = 0;
#pragma unroll 2
            for (int u = ulo; u <= uhi; ++u) {
                const double *un = UN(u);
                if constexpr (BOX) {       // X_e' B_e X_e = diag(XD^2 BD) + BRS BQ BQ'
                    acc2 += (un[W::BRS] * un[W::BQ + r]) * un[W::BQ + c];
                    if (r == c) acc2 += (un[W::XD + r] * un[W::XD + r]) * un[W::BD + r];
                } else {
#pragma unroll
                    for (int k = 0; k < N; ++k) acc2 += un[W::X + (h * N + k) * NX + r] * un[W::K + (h * N + k) * NX + c];
                    acc2 += un[W::X + 2 * N * NX + r] * un[W::K + 2 * N * NX + c];
                }
            }
            sm[W::XBX + tp] = acc2;
        }
        WG_FOR_AT(tt, NX * N, plq.at(NX * N)) {                 // sum over ALL units of the x-x coupling (same half only)
            const int r = tt / N, c = (r / N) * N + (tt - r * N);
            double acc2 = 0;
            if constexpr (BOX) {
                if (c == r) {
#pragma unroll 4
                    for (int u = 0; u <= d; ++u) acc2 += UN(u)[W::XD + r];
                }
            } else {
#pragma unroll 4
                for (int u = 0; u <= d; ++u) acc2 += UN(u)[W::X + r * NX + c];
            }
            sm[W::XS + tt] = acc2;
        }
        WG_FOR_AT(tt, 2 * NW, plq.at(2 * NW)) {
            const int s = tt / NW, k = tt - s * NW;
            double acc2 = UN(0)[W::P + k];
#pragma unroll 4
            for (int u = side_lo(s); u <= side_hi(s); ++u) acc2 -= UN(u)[W::P + k];
            sm[W::RP + tt] = acc2;
        }
        if (!first_warm) bg_tasks(plq);                   // affine solve, head 2/4
        WG_ARRIVE(6);
        WG_SYNC();
        WG_STAMP(6);
        // ================= sides: factor, invert, Y_s = Bs^{-1} BXs (over the dead factor) =================
        Place ply;
        if (sides) {
            // B_in and B_out = L D L' concurrently, one wavefront each, one matrix row per lane (one call site: one copy of the code)
            WG_FIRST_WAVES(sd, 2) wave_ldl<NW>(sm + W::BS + sd * NW * NW, sm + W::PIVS + sd * NW);
            WG_ARRIVE(7);
            WG_SYNC();
            WG_STAMP(7);
            WG_FOR(t, 2 * NW) {
                const int s = t / NW, c = t - s * NW;
                ldl_inverse_col<NW, (!BOX && NW > 9)>(sm + W::BS + s * NW * NW, sm + W::PIVS + s * NW, c, sm + W::BSI + s * NW * NW, NW);
            }
            WG_SYNC();
            WG_FOR_AT(t, 2 * NW * NX, ply.at(2 * NW * NX)) {
                const int s = t / (NW * NX), ic = t - s * NW * NX, i = ic / NX, c = ic - i * NX;
                const double *Bsi = sm + W::BSI + s * NW * NW, *BXs = sm + W::BXS + s * NW * NX;
                double acc2 = 0;
#pragma unroll
                for (int k = 0; k < NW; ++k) acc2 += Bsi[i * NW + k] * BXs[k * NX + c];
                sm[W::BS + s * NW * NW + ic] = acc2;     // Y_s
            }
        } else {
            // no blocks (prox configuration): the side matrices and everything derived from them are zero -- Bs and BXs
            // already are (empty sums), the formulas below then need Bs^{-1} := 0 and Y_s := 0 (= the Bs buffer as it stands)
            WG_FOR(t, 2 * NW * NW) { sm[W::BSI + t] = 0.0; sm[W::BS + t] = 0.0; }      // (Bs: only its lower triangle was written)
            WG_SYNC();
        }
        if (!first_warm) v_tasks(ply);                    // affine solve, head 3/4
        WG_ARRIVE(9);
        WG_SYNC();
        WG_STAMP(9);
        // ================= reduced border matrix in the (x, u, z2, y_v) variables =================
        // One region: every entry of the lower triangle directly in the new variables.  The change of variables (u, z2) =
        // (z1 - z2, z2) -- columns / rows of z2 gain those of z1; the cone term Su then sits on u alone (with the cone inactive
        // Su ~ 1/mu would cancel in the (z1, z2) form) -- touches the zeta rows only, whose entries in the old variables are
        // three-term sums (zx, zz below): an entry of the new matrix is at most four of them, summed in the order a two-step
        // assembly through a scratch matrix would use (that was one more region: 2 200 cycles against ~1 400).
        {
            const double *YS0 = sm + W::BS, *YS1 = sm + W::BS + NW * NW, *u0 = UN(0);
            auto x0 = [&](int i, int c) {       // X of the border unit
                if constexpr (BOX) return i == c ? u0[W::XD + c] : (i == 2 * N ? u0[W::XY + c] : 0.0);
                else return u0[W::X + i * NX + c];
            };
            auto k0 = [&](int hi, int lo) {     // K of the border unit, lower triangle
                if constexpr (BOX) return hi == lo ? (hi < 2 * N ? u0[W::KD + hi] : u0[W::KYY]) : (hi == 2 * N ? u0[W::KY + lo] : 0.0);
                else return u0[W::K + hi * NW + lo];
            };
            auto zx = [&](int i, int c) { return x0(i, c) + YS0[i * NX + c] + YS1[i * NX + c]; };
            auto zz = [&](int i, int k) {       // symmetric; the lower-triangle copy is the one used
                const int hi = i > k ? i : k, lo = i > k ? k : i;
                return k0(hi, lo) + sm[W::BSI + hi * NW + lo] + sm[W::BSI + NW * NW + hi * NW + lo];
            };
            Place plm;
            WG_FOR_AT(t, NX * (NX + 1) / 2, plm.at(NX * (NX + 1) / 2)) {        // x-x
                int r, c;
                tri_decode(t, r, c);
                double val = (r == c ? REG_DELTA + (prox ? sm[W::PQ + r] : 0.0) : 0.0) - sm[W::XBX + r * NX + c] - sm[W::XBX + NX * NX + r * NX + c];
                if (r / N == c / N) val -= sm[W::XS + r * N + (c - (c / N) * N)];
#pragma unroll
                for (int k = 0; k < NW; ++k)
                    val += sm[W::BXS + k * NX + r] * YS0[k * NX + c] + sm[W::BXS + NW * NX + k * NX + r] * YS1[k * NX + c];
                sm[W::M + r * NB1 + c] = val;
            }
            WG_FOR_AT(t, NW * NX, plm.at(NW * NX)) {                            // zeta-x
                const int i = t / NX, c = t - i * NX;
                double val = zx(i, c);
                if (i >= N && i < 2 * N) val += zx(i - N, c);
                sm[W::M + (NX + i) * NB1 + c] = val;
            }
            WG_FOR_AT(t, NW * (NW + 1) / 2, plm.at(NW * (NW + 1) / 2)) {        // zeta-zeta
                int i, k;
                tri_decode(t, i, k);
                const bool rz2 = i >= N && i < 2 * N, cz2 = k >= N && k < 2 * N;
                double val = zz(i, k);
                if (cz2) val += zz(i, k - N);
                if (rz2) val += zz(i - N, k);
                if (rz2 && cz2) val += zz(i - N, k - N);
                if (i < N && k < N) val += SOC[SO::SU + i * N + k];
                sm[W::M + (NX + i) * NB1 + NX + k] = val;
            }
            if (!first_warm) rhs_tasks(plm, false);       // affine solve, head 4/4
        }
        WG_ARRIVE(10);
        WG_SYNC();
        WG_STAMP(10);
        WG_WAVE0() wave_ldl<NB1>(sm + W::M, sm + W::PIVM);      // L D L' inside one wavefront; no explicit inverse
        WG_ARRIVE(11);
        WG_SYNC();
        WG_STAMP(11);
        WG_STAMP(12);
        double rmax = 0.0, c1 = 0.0, c2 = 0.0, amax_cone = 1e300;
        double sigmu;
        if (first_warm) {
            // ================= re-centring step of a warm solve: no predictor =================
            // The affine direction would only feed sigma (fixed here: sigma mu = mu_ref) and the second-order term (dropped): its
            // solve and the reduction of its step statistics are skipped; the rows keep 1 / s for the pass over G'kappa.
            Place plb;
            WG_ROWS_BEGIN(plb)
                un[oR1 + ro] = 0.0;
                un[oR2 + ro] = rcp1(s);
            WG_ROWS_END()
            WG_FOR_AT(u, U, plb.at(U)) { double *un = UN(u); un[W::KB] = 0.0; un[W::KB + 1] = 0.0; }
            sigmu = mu_ref;
            WG_ARRIVE(14);
            WG_SYNC();
            WG_STAMP(14);
        } else {
        // ================= affine direction (kappa = 0) =================
        solve_tail(SC_DTA, false);      // the head of the affine solve rode in the factorisation regions above
        WG_STAMP(13);
        // rows: step bound, mu_aff sums, ds_a dl_a
        Place plb;
        WG_ROWS_BEGIN_DS(plb)
            const double l = un[oLAM + ro], is = rcp1(s);
            const double q = ds * is, dl = -l - l * q;          // dl / l = -1 - ds / s
            rmax = fmax(rmax, fmax(-q, 1.0 + q));
            c1 += s * dl + l * ds; c2 += ds * dl;
            un[oR1 + ro] = ds * dl;
            un[oR2 + ro] = is;           // kappa = (sigma mu - ds_a dl_a) / s is formed where it is used (G'kappa, final direction)
        WG_ROWS_END()
        WG_FOR_AT(u, U, plb.at(U)) {
            double *un = UN(u);
            const double yy = un[W::P + 2 * N], dy = un[W::DW + 2 * N];
            const double s5 = yy, s6 = 1.0 - yy, l5 = un[W::LB], l6 = un[W::LB + 1];
            const double q5 = dy * rcp1(s5), q6 = -dy * rcp1(s6);
            const double dl5 = -l5 - l5 * q5, dl6 = -l6 - l6 * q6;
            rmax = fmax(rmax, fmax(fmax(-q5, 1.0 + q5), fmax(-q6, 1.0 + q6)));
            c1 += s5 * dl5 + l5 * dy + s6 * dl6 - l6 * dy; c2 += dy * dl5 - dy * dl6;
            un[W::KB] = dy * dl5; un[W::KB + 1] = -dy * dl6;
        }
        // the cone's share (step bounds, mu_aff sums) was computed by the cone lanes inside the solve (solve_tail)
        WG_CONE() { amax_cone = fmin(SC[SC_AMAXC], SC[SC_AMAXC2]); c1 += SC[SC_C1C]; c2 += SC[SC_C2C]; }
        WG_ARRIVE(14);
        const Red3 rb = wg_reduce(Red3{fmin(rmax > 0.0 ? rcp1(rmax) : 1e300, amax_cone), c1, c2}, sm + W::RED, red_phase);
        WG_STAMP(14);
        {
            const double al = fmin(1.0, rb.mn);
            const double mu_aff = (gap + al * rb.s1 + al * al * rb.s2) * inv_deg;
            double sig = mu_aff * rcp(mu);
            sig = sig < 0 ? 0 : (sig > 1 ? 1 : sig);
            sig = sig * sig * sig;
            sigmu = wg_uniform(sig * mu);
        }
        }
        // ================= corrector: kappa = (sigma mu - ds_a dl_a) / s per row; cone part by the cone thread =================
        // (no region of its own for the row kappas: both factors are in the row arrays since the pass above -- the reduction's
        //  barrier published them -- and sigma mu is known to every thread)
        WG_CONE() {   // (beside the G'kappa tasks: the last wavefront has none) kappa_soc = sigma mu s^{-1} - W^{-1}( lt \ ((W^{-1} ds_a) o (W dl_a)) )
            double wb[Q], a1[Q], a2[Q], pr[Q], qv[Q], xs[Q], lt[Q], ss[Q];
            const double eta = SC[SC_ETA];
#pragma unroll
            for (int k = 0; k < Q; ++k) { wb[k] = SOC[SO::WB + k]; xs[k] = SOC[SO::DSSA + k]; lt[k] = SOC[SO::LT + k]; ss[k] = SOC[SO::SS + k]; }
            soc_apply_Wi<Q>(wb, eta, xs, a1);
#pragma unroll
            for (int k = 0; k < Q; ++k) xs[k] = SOC[SO::DLSA + k];
            soc_apply_W<Q>(wb, eta, xs, a2);
            double dsum = 0;
#pragma unroll
            for (int k = 0; k < Q; ++k) dsum += a1[k] * a2[k];
            pr[0] = dsum;
#pragma unroll
            for (int k = 1; k < Q; ++k) pr[k] = a1[0] * a2[k] + a2[0] * a1[k];
            const double det = gcs_math::soc_det<Q>(lt);
            double ld1 = 0;
#pragma unroll
            for (int k = 1; k < Q; ++k) ld1 += lt[k] * pr[k];
            qv[0] = (lt[0] * pr[0] - ld1) * rcp(det);
            const double ilt0 = rcp(lt[0]);
#pragma unroll
            for (int k = 1; k < Q; ++k) qv[k] = (pr[k] - qv[0] * lt[k]) * ilt0;
            const double smd = sigmu * rcp(gcs_math::soc_det<Q>(ss));
            soc_apply_Wi<Q>(wb, eta, qv, a1);
#pragma unroll
            for (int i = 0; i < Q; ++i) SOC[SO::KS + i] = smd * (i == 0 ? ss[0] : -ss[i]) - (first_warm ? 0.0 : a1[i]);
            SC[SC_GT] = 1.0 - SOC[SO::KS];
        }
        // G' kappa per unit: own unknowns (GU) and the x part (GX).  One task per (unit, half, coordinate) forms both entries: they
        // run over the same rows (kappa_a and kappa_b of every facet), so one pass over the row arrays serves both
        Place plg;
        WG_FOR_AT(t, U * 2 * N, plg.at(U * 2 * N)) {
            const int u = t / (2 * N), q = t - u * (2 * N), i = q / N, k = q - i * N;
            double *un = UN(u);
            const double *ea = un + oR1, *eb = ea + m2, *ia = un + oR2, *ib = ia + m2;
            double su = 0, sx = 0;
            if constexpr (BOX) {
                const int jp = i * m + k, jn = jp + N;
                const double kbp = (sigmu - eb[jp]) * ib[jp], kbn = (sigmu - eb[jn]) * ib[jn];
                su = ((sigmu - ea[jp]) * ia[jp] - kbp) - ((sigmu - ea[jn]) * ia[jn] - kbn);
                sx = kbp - kbn;
            } else {
#pragma unroll 4
                for (int j = 0; j < m; ++j) {
                    const double kb = (sigmu - eb[i * m + j]) * ib[i * m + j], aj = A[j * N + k];
                    su += aj * ((sigmu - ea[i * m + j]) * ia[i * m + j] - kb);
                    sx += aj * kb;
                }
            }
            un[W::GU + q] = su;
            un[W::GX + q] = sx;
        }
        WG_FOR_AT(u, U, plg.at(U)) {
            double *un = UN(u);
            const double *ea = un + oR1, *eb = ea + m2, *ia = un + oR2, *ib = ia + m2;
            double s = 0;
#pragma unroll 4
            for (int j = 0; j < m2; ++j) s += BC[j >= m ? j - m : j] * ((sigmu - eb[j]) * ib[j] - (sigmu - ea[j]) * ia[j]);
            const double yy = un[W::P + 2 * N];
            un[W::GU + 2 * N] = s - (sigmu - un[W::KB]) * rcp1(yy) + (sigmu - un[W::KB + 1]) * rcp1(1.0 - yy);
        }
        WG_ARRIVE(16);
        WG_SYNC();
        WG_STAMP(16);
        solve_head(true);
        solve_tail(SC_DT, true);
        WG_STAMP(18);
        // ================= final direction: dual directions, step bound =================
        rmax = 0.0;
        Place pld;
        WG_ROWS_BEGIN_DS(pld)
            const double l = un[oLAM + ro];
            const double ip = rcp1(s * l), is = l * ip, il = s * ip;      // 1/s and 1/l from one reciprocal
            const double dl = (sigmu - un[oR1 + ro]) * un[oR2 + ro] - l - (l * is) * ds;      // kappa - l - (l / s) ds
            un[oR2 + ro] = dl;
            rmax = fmax(rmax, fmax(-ds * is, -dl * il));
        WG_ROWS_END()
        WG_FOR_AT(u, U, pld.at(U)) {
            double *un = UN(u);
            const double yy = un[W::P + 2 * N], dy = un[W::DW + 2 * N];
            const double s5 = yy, s6 = 1.0 - yy, l5 = un[W::LB], l6 = un[W::LB + 1];
            const double i5 = rcp1(s5), i6 = rcp1(s6);
            const double dl5 = (sigmu - un[W::KB]) * i5 - l5 - l5 * i5 * dy, dl6 = (sigmu - un[W::KB + 1]) * i6 - l6 + l6 * i6 * dy;
            un[W::DLB] = dl5; un[W::DLB + 1] = dl6;
            rmax = fmax(rmax, fmax(fmax(-dy * i5, -dl5 * rcp1(l5)), fmax(dy * i6, -dl6 * rcp1(l6))));
        }
        amax_cone = 1e300;
        WG_CONE() amax_cone = fmin(SC[SC_AMAXC], SC[SC_AMAXC2]);
        WG_ARRIVE(19);
        const Red3 rd = wg_reduce(Red3{fmin(rmax > 0.0 ? rcp1(rmax) : 1e300, amax_cone), 0.0, 0.0}, sm + W::RED, red_phase);
        WG_STAMP(19);
        WG_CONE() {      // step length with the cone guard (round-off must not push either cone point outside)
            double al = fmin(1.0, 0.99 * rd.mn);
            for (int tries = 0; tries < 40; ++tries) {
                double s2[Q], l2[Q];
#pragma unroll
                for (int k = 0; k < Q; ++k) { s2[k] = SOC[SO::SS + k] + al * SOC[SO::DSS + k]; l2[k] = SOC[SO::LS + k] + al * SOC[SO::DLS + k]; }
                if (gcs_math::soc_interior<Q>(s2) && gcs_math::soc_interior<Q>(l2)) break;
                al *= 0.7;
            }
            SC[SC_ALPHA] = al;
        }
        WG_ARRIVE(20);
        WG_SYNC();
        WG_STAMP(20);
        const double alpha = wg_uniform(SC[SC_ALPHA]);
        stalled = alpha < 1e-3;
        // ================= update =================
        Place plu;
        WG_ROWS_BEGIN(plu)       // (the slack the macro offers is unused here: the compiler drops it)
            (void)s;
            un[oLAM + ro] += alpha * un[oR2 + ro];
        WG_ROWS_END()
        WG_FOR_AT(t, U * NW, plu.at(U * NW)) {
            const int u = t / NW, k = t - u * NW;
            double *un = UN(u);
            un[W::P + k] += alpha * un[W::DW + k];
        }
        WG_FOR_AT(t, 2 * U, plu.at(2 * U)) {
            double *un = UN(t >> 1);
            un[W::LB + (t & 1)] += alpha * un[W::DLB + (t & 1)];
        }
        WG_FOR_AT(t, NX + 2 * NW, plu.at(NX + 2 * NW)) {
            if (t < NX) sm[W::XV + t] += alpha * sm[W::DX + t];
            else sm[W::NU + (t - NX)] += alpha * sm[W::DNU + (t - NX)];
        }
        WG_CONE() {
            SC[SC_T] += alpha * SC[SC_DT];
            for (int k = 0; k < Q; ++k) SOC[SO::LS + k] += alpha * SOC[SO::DLS + k];
        }
        WG_ARRIVE(21);
        WG_SYNC();
        WG_STAMP(21);
    }
#undef WG_FAIL_OR_RESTART
    it_total += it;
    if (wrec != nullptr) {
        WG_ONE() {
            if (status != 0) wrec[0] = 0.0;      // no restart from a solve that failed
            else gcs_ws::ws_learn(wrec, use_warm, ws_started_warm && !use_warm, ws_dT, it);
        }
    }
    status_out = status;
    iters_out = it_total;
    WG_STAMP(22);
    WG_SYNC();
    if (status != 0) return;      // inner failure: the previous copy columns stay (admm_solver_v3.py:524-538 intent)
    // ---- un-centre and write out ----
    Place plo;
    WG_FOR_AT(t, d * NW, plo.at(d * NW)) {
        const int e = t / NW, w = t - e * NW;
        const bool out = e >= d_in;
        const int inc = a.edge_major ? a.inc_edge[lo + e] + (out ? 0 : a.E) : lo + e;
        const double *un = UN(e + 1), *p = un + W::P;
        const double yy = p[2 * N];
        double val;
        if (w == 2 * N) val = yy;
        else if (w < N) val = out ? p[w] + yy * CEN[w] : un[W::TF + w];     // incoming: the free word keeps its target
        else val = out ? p[w] + yy * CEN[w - N] : p[w - N] + yy * CEN[w - N];
        a.copy[(size_t)w * a.NI + inc] = (T)val;
    }
    WG_FOR_AT(k, NX, plo.at(NX)) {
        const int c = k < N ? k : k - N;
        const double *u0 = UN(0);
        a.xv[(size_t)v * NX + k] = sm[W::XV + k] + CEN[c];
        a.zv[(size_t)v * NX + k] = u0[W::P + k] + u0[W::P + 2 * N] * CEN[c];
    }
    WG_ONE() a.yv[v] = UN(0)[W::P + 2 * N];
}

} // namespace gcs_wg
