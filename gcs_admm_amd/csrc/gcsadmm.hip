// gcsadmm.hip -- gfx950 kernels and C ABI (include/gcsadmm.h) of the ADMM iteration loop.
//
// Kernels (one HIP stream, launched back to back, no host round trip inside the loop):
//   vertex_kernel<2,T>   x-update, wavefront program (n = 2, degree <= 63): one wavefront per workgroup, several vertices
//                        per wavefront, program in vertex_program.h     (admm_solver_v3.py:352-540)
//   vertex_wg_kernel<N,T> (vertex_wg.hip) x-update, workgroup program: one 256-thread workgroup per vertex, any n / degree
//                        trailing workgroups of either launch: x-update of s, t (closed form) and of vertices no flow can cross
//   edge_kernel<T,MODE,C> z-update, dual update, five partial norms     (admm_solver_v3.py:543-614); in gcsadmm_run the last
//                        workgroup to finish also does the final reduction and the control step (one launch per edge step)
//   finalize_kernel / control_kernel   deterministic final reduction; residuals, rho adaptation, stop test,
//                        trace record (admm_solver_v3.py:697-733): the separate steps of the partitioned loop
//   halo_pack / halo_unpack_kernel     messages of the cut edges' copies between vertex partitions (RCCL); in the overlapped
//                        partitioned loop they and the transfer run on a second stream while the interior wavefronts are solved
//   cost_kernel<T>       GCS_utils.py:184-211
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "gcsadmm.h"
#include "terminal_launch.h"
#ifdef GCS_PHASE_TIMING
// diagnostic build: sub-phase stamps inside the border factorisation (lane 0 of each wavefront)
__device__ unsigned long long g_sub_cycles[16];
__device__ __forceinline__ void gcs_stamp(int k)
{
    static __shared__ unsigned long long last;
    __builtin_amdgcn_sched_barrier(0);
    if (threadIdx.x == 0) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (k > 0) atomicAdd(&g_sub_cycles[k], t - last);
        last = t;
    }
    __builtin_amdgcn_sched_barrier(0);
}
#define GCS_STAMP(k) gcs_stamp(k)
#endif
#include "vertex_kernel.h"
#include "vertex_wg_launch.h"
#include "warm_start.h"
#include "canonical_box.h"

namespace {

using namespace gcs;
using namespace gcsadmm_k;

constexpr int EDGE_BLOCK = 256;

// -------------------------------------------------------------------------------------------------
// edge kernel: one thread per directed edge, all c coupled words
// -------------------------------------------------------------------------------------------------
template <class T> struct EdgeArgs {
    int E, NI, c;
    const int *edge_inc_tail, *edge_inc_head;
    const uint8_t *inc_counted, *edge_counted;   // may be null
    const T *copy;
    T *zedge, *mu;
    double *partials;    // [gridDim.x][5]
};

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

struct ControlParams {
    double tau_incr, tau_decr, nu, eps_abs, eps_rel, nx, nmu;
    int it_rho_limit, max_it;
};

// admm_solver_v3.py:697-733 on the five (globally reduced) sums; one thread
// global_fails: the inner-failure count comes with the (all-reduced) sums as sums[5] instead of from this handle's counter
__device__ void control_body(gcsadmm_control_block *cb, const double *sums, const ControlParams &p, int *counters, double *trace,
                             bool global_fails = false)
{
    if (cb->status != GCSADMM_RUNNING) return;
    double s[5];
    for (int k = 0; k < 5; ++k) { s[k] = sums[k]; cb->sums[k] = s[k]; }
    const int it = cb->it;
    double rho = cb->rho;
    const int fails = global_fails ? (int)(sums[5] + 0.5) : counters[0], iters = counters[1];
    counters[0] = 0; counters[1] = 0;
    cb->inner_failures = fails; cb->inner_iters = iters;
    const double tot = s[0] + s[1] + s[2] + s[3] + s[4];
    if (!(tot == tot) || fabs(tot) > 1.7e308) {   // non-finite iterate: admm_solver_v3.py:662-664, 679-681
        cb->status = GCSADMM_DIVERGED;
        return;
    }
    const double pri = sqrt(s[0]), dual = rho * sqrt(2.0 * s[1]);
    double mu_scale = 1.0;
    if (pri >= p.nu * dual && it < p.it_rho_limit) { rho *= p.tau_incr; mu_scale = 1.0 / p.tau_incr; }
    else if (dual >= p.nu * pri && it < p.it_rho_limit) { rho *= 1.0 / p.tau_decr; mu_scale = p.tau_incr; }
    const double eps_pri = sqrt(p.nx) * p.eps_abs + p.eps_rel * fmax(sqrt(s[2]), sqrt(2.0 * s[3]));
    const double eps_dual = sqrt(p.nmu) * p.eps_abs + p.eps_rel * mu_scale * sqrt(s[4]);
    cb->rho = rho; cb->mu_scale = mu_scale;
    cb->pri = pri; cb->dual = dual; cb->eps_pri = eps_pri; cb->eps_dual = eps_dual;
    if (trace) {
        double *tr = trace + (size_t)(it - 1) * 6;
        tr[0] = rho; tr[1] = pri; tr[2] = dual; tr[3] = eps_pri; tr[4] = eps_dual; tr[5] = (double)fails;
    }
    if (pri < eps_pri && dual < eps_dual) { cb->status = GCSADMM_CONVERGED; return; }
    cb->it = it + 1;
    if (it + 1 > p.max_it) cb->status = GCSADMM_MAX_IT;
}

__global__ void control_kernel(gcsadmm_control_block *cb, const double *sums, ControlParams p, int *counters, double *trace, bool global_fails)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    control_body(cb, sums, p, counters, trace, global_fails);
}

// edges a thread of the edge kernel has in flight at once on LARGE graphs (registers: U x 5C words); graphs that would not fill
// the chip with such tiles (fewer than 256 workgroups: one per CU) keep one edge per thread
template <class T, int C> __host__ __device__ constexpr int edge_unroll() { return sizeof(T) == 4 ? (C <= 7 ? 4 : 2) : (C <= 7 ? 2 : 1); }
static int edge_unroll_rt(int dtype, int c, int E)
{
    const int u = dtype == GCSADMM_F32 ? (c <= 7 ? 4 : 2) : (c <= 7 ? 2 : 1);
    return (E + EDGE_BLOCK * u - 1) / (EDGE_BLOCK * u) >= 256 ? u : 1;
}

// MODE 0: partial sums per workgroup only (gcsadmm_edge_step: the caller all-reduces / finalizes);
// MODE 1: single workgroup (at most EDGE_BLOCK edges, gcsadmm_run on small graphs): the workgroup also does the final
//         reduction and the control step;
// MODE 2: any grid (gcsadmm_run): the LAST workgroup to finish -- told by an agent-scope ticket counter -- reduces all the
//         partials in the fixed order of finalize_kernel and runs the control step: one launch per edge step instead of two;
// MODE 3: as MODE 2 without the control step (gcsadmm_run_partitioned): the last workgroup leaves the five sums and, in
//         sums[5], this partition's inner-failure count for the all-reduce that follows.
// C = coupled words per copy (2n+1), compile-time so that all C x 5 loads of an edge are in flight at once.
template <class T, int MODE, int C, int U>
__global__ __launch_bounds__(EDGE_BLOCK) void edge_kernel(EdgeArgs<T> a, gcsadmm_control_block *cb, double *sums, ControlParams cp,
                                                          int *counters, double *trace, unsigned *ticket)
{
    if (cb->status != GCSADMM_RUNNING) return;
    const double mu_scale = cb->mu_scale;
    double s[5] = {0, 0, 0, 0, 0};
    // a workgroup takes tiles of U x EDGE_BLOCK consecutive edges; a thread handles U edges of the tile, EDGE_BLOCK apart, and issues
    // the loads of all of them before the first use: U x 5C coalesced 4/8-byte loads in flight per thread (one edge per thread left
    // the stream latency-bound: 64 MB in 36 us on the 100k lattice)
    for (int base = blockIdx.x * (U * EDGE_BLOCK); base < a.E; base += gridDim.x * (U * EDGE_BLOCK)) {
        int it[U], ih[U];
        T cu_[U][C], cw_[U][C], zo_[U][C], mu_[U][C], mw_[U][C];
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int e = base + q * EDGE_BLOCK + (int)threadIdx.x, ee = e < a.E ? e : a.E - 1;     // (tail of the last tile: a valid edge, result unused)
            // edge-major columns (null index arrays): the two columns of edge e are e and E + e, every access below is a stream
            it[q] = a.edge_inc_tail ? a.edge_inc_tail[ee] : ee; ih[q] = a.edge_inc_head ? a.edge_inc_head[ee] : a.E + ee;
#pragma unroll
            for (int w = 0; w < C; ++w) {
                cu_[q][w] = a.copy[(size_t)w * a.NI + it[q]]; cw_[q][w] = a.copy[(size_t)w * a.NI + ih[q]];
                zo_[q][w] = a.zedge[(size_t)w * a.E + ee];
                mu_[q][w] = a.mu[(size_t)w * a.NI + it[q]]; mw_[q][w] = a.mu[(size_t)w * a.NI + ih[q]];
            }
        }
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int e = base + q * EDGE_BLOCK + (int)threadIdx.x;
            if (e >= a.E) break;
            const double we = a.edge_counted ? (double)a.edge_counted[e] : 1.0;
            const double wt = a.inc_counted ? (double)a.inc_counted[it[q]] : 1.0;
            const double wh = a.inc_counted ? (double)a.inc_counted[ih[q]] : 1.0;
#pragma unroll
            for (int w = 0; w < C; ++w) {
                const double cu = (double)cu_[q][w], cw = (double)cw_[q][w], zo = (double)zo_[q][w];
                const T zn_t = (T)(0.5 * (cu + cw));
                const double zn = (double)zn_t;
                const double ru = cu - zn, rw = cw - zn;
                const T mu_u_t = (T)(mu_scale * (double)mu_[q][w] + ru);
                const T mu_w_t = (T)(mu_scale * (double)mw_[q][w] + rw);
                a.mu[(size_t)w * a.NI + it[q]] = mu_u_t;
                a.mu[(size_t)w * a.NI + ih[q]] = mu_w_t;
                a.zedge[(size_t)w * a.E + e] = zn_t;
                const double mu_u = (double)mu_u_t, mu_w = (double)mu_w_t;
                s[0] += wt * ru * ru + wh * rw * rw;
                s[1] += we * (zn - zo) * (zn - zo);
                s[2] += wt * cu * cu + wh * cw * cw;
                s[3] += we * zn * zn;
                s[4] += wt * mu_u * mu_u + wh * mu_w * mu_w;
            }
        }
    }
    __shared__ double red[EDGE_BLOCK][5];
    __shared__ int is_last;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const double t = wave_sum(s[k]);
        if (lane == 0) red[wv][k] = t;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        double t = 0;
        for (int q = 0; q < EDGE_BLOCK / WAVE; ++q) t += red[q][threadIdx.x];
        if (MODE >= 2) __hip_atomic_store(&a.partials[(size_t)blockIdx.x * 5 + threadIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else a.partials[(size_t)blockIdx.x * 5 + threadIdx.x] = t;
        if (MODE == 1) sums[threadIdx.x] = t;      // one workgroup: its partial is the sum (what finalize_kernel would produce)
    }
    if (MODE == 1) {
        __syncthreads();
        if (threadIdx.x == 0) control_body(cb, sums, cp, counters, trace);
    }
    if (MODE >= 2) {
        // hand-off of the partials to the last workgroup (MI355X_MICROARCH.md, inter-workgroup visibility): write-through (sc1)
        // stores by the first wavefront, drained, then ONE agent-scope ticket add by a lane of that same wavefront; the
        // workgroup whose add returns gridDim.x - 1 came last and reads every partial with sc1 loads.  (Measured alternative: an
        // agent-scope ACQ_REL ticket add instead of the drain -- the release writes back the L2 of the XCD, which holds this
        // kernel's own 24 MB of stores: edge step 23.6 -> 33.4 us on the 100k lattice.  Only the five partials need to cross.)
        if (threadIdx.x < WAVE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) {
            const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            is_last = (t == gridDim.x - 1);
        }
        __syncthreads();
        if (!is_last) return;
        const int nblocks = gridDim.x;
        double acc[5] = {0, 0, 0, 0, 0};
        for (int b = threadIdx.x; b < nblocks; b += EDGE_BLOCK)
            for (int k = 0; k < 5; ++k) acc[k] += __hip_atomic_load(&a.partials[(size_t)b * 5 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();       // red[][] above has been consumed by every thread
        for (int k = 0; k < 5; ++k) red[threadIdx.x][k] = acc[k];
        __syncthreads();
        for (int off = EDGE_BLOCK / 2; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off)
                for (int k = 0; k < 5; ++k) red[threadIdx.x][k] += red[threadIdx.x + off][k];
            __syncthreads();
        }
        if (threadIdx.x < 5) sums[threadIdx.x] = red[0][threadIdx.x];
        __syncthreads();
        if (threadIdx.x == 0) {
            if (MODE == 2) control_body(cb, sums, cp, counters, trace);
            else sums[5] = (double)counters[0];
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
        }
    }
}

// SLOWEST-FIRST DISPATCH.  A vertex-step launch ends when its slowest workgroup does, and with the warm start most solves take 3-5
// Newton iterations while a few take 15: on the 10k lattice (1 654 wavefronts on 1 024 one-wavefront-per-SIMD slots) a slow
// wavefront that happens to start in the second round ends the launch at 21 iteration times instead of 17.  Every few ADMM
// iterations the units (wavefronts / workgroups) are re-ordered by the Newton iterations of their last launch, descending: a
// counting sort by one workgroup.  The order among equal counts is arbitrary (atomics); it affects scheduling only, never results.
constexpr int REORDER_BINS = 64, REORDER_THREADS = 1024, REORDER_EVERY = 8, REORDER_MIN_UNITS = 512;
// ids (may be null): the units to order are ids[0 .. n) instead of 0 .. n (the boundary / interior subsets of a partition's overlapped loop)
__global__ __launch_bounds__(REORDER_THREADS) void reorder_kernel(int n, const int *iters, int *order, const gcsadmm_control_block *cb, const int *ids = nullptr)
{
    if (cb->status != GCSADMM_RUNNING) return;
    __shared__ int cnt[REORDER_BINS], off[REORDER_BINS];
    if (threadIdx.x < REORDER_BINS) cnt[threadIdx.x] = 0;
    __syncthreads();
    auto unit = [&](int i) { return ids ? ids[i] : i; };
    auto key = [&](int i) { const int k = iters[unit(i)]; return k < 0 ? 0 : (k >= REORDER_BINS ? REORDER_BINS - 1 : k); };
    for (int i = threadIdx.x; i < n; i += REORDER_THREADS) atomicAdd(&cnt[key(i)], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int b = REORDER_BINS - 1; b >= 0; --b) { off[b] = run; run += cnt[b]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += REORDER_THREADS) order[atomicAdd(&off[key(i)], 1)] = unit(i);
}

// fixed-order reduction of the per-workgroup partials -> sums[5]
__global__ __launch_bounds__(256) void finalize_kernel(const double *partials, int nblocks, double *sums,
                                                      const gcsadmm_control_block *cb)
{
    if (cb->status != GCSADMM_RUNNING) return;
    __shared__ double red[256][5];
    double s[5] = {0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < nblocks; b += 256)
        for (int k = 0; k < 5; ++k) s[k] += partials[(size_t)b * 5 + k];
    for (int k = 0; k < 5; ++k) red[threadIdx.x][k] = s[k];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int k = 0; k < 5; ++k) red[threadIdx.x][k] += red[threadIdx.x + off][k];
        __syncthreads();
    }
    if (threadIdx.x < 5) sums[threadIdx.x] = red[0][threadIdx.x];
}

// halo of a vertex partition: copies of the cut edges' coupled words, packed per neighbour as [c][columns of that peer]
// (one contiguous message per peer).  base[j] / stride[j]: where column j of the flat send (recv) list sits in the buffer.
template <class T>
__global__ __launch_bounds__(256) void halo_pack_kernel(int c, int ncols, int NI, const int *cols, const int *base, const int *stride,
                                                        const T *copy, T *buf, const gcsadmm_control_block *cb)
{
    if (cb->status != GCSADMM_RUNNING) return;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= c * ncols) return;
    const int w = t / ncols, j = t - w * ncols;
    buf[base[j] + w * stride[j]] = copy[(size_t)w * NI + cols[j]];
}
template <class T>
__global__ __launch_bounds__(256) void halo_unpack_kernel(int c, int ncols, int NI, const int *cols, const int *base, const int *stride,
                                                          const T *buf, T *copy, const gcsadmm_control_block *cb)
{
    if (cb->status != GCSADMM_RUNNING) return;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= c * ncols) return;
    const int w = t / ncols, j = t - w * ncols;
    copy[(size_t)w * NI + cols[j]] = buf[base[j] + w * stride[j]];
}
template <class T>
__global__ __launch_bounds__(256) void cost_kernel(int V, int E, int n, const double *zv, const T *zedge,
                                                   const uint8_t *edge_counted, double eps_edge, double *cost)
{
    // single workgroup, fixed order: this runs once after the loop
    __shared__ double red[256];
    double s = 0;
    for (int v = threadIdx.x; v < V; v += 256) {
        double q = 0;
        for (int k = 0; k < n; ++k) { const double dlt = zv[(size_t)v * 2 * n + k] - zv[(size_t)v * 2 * n + n + k]; q += dlt * dlt; }
        s += sqrt(q);
    }
    for (int e = threadIdx.x; e < E; e += 256)
        s += eps_edge * (edge_counted ? (double)edge_counted[e] : 1.0) * (double)zedge[(size_t)(2 * n) * E + e];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) cost[0] = red[0];
}


} // namespace

// =================================================================================================
// host side
// =================================================================================================
struct gcsadmm_handle_s {
    int n = 0, V = 0, E = 0, NI = 0, NI_owned = 0, c = 0, MM = 0, dtype = 0, device = 0;
    int n_waves = 0, n_special = 0, slots_cap = 0, lds_bytes = 0, edge_blocks = 0;
    int n_wg = 0, wg_lds_bytes = 0;   // vertices solved by the workgroup program (vertex_wg.hip), LDS per workgroup
    int wg_box = 0;           // every workgroup-program vertex is a canonical box: BOX instantiation of that program
    int wg_t512 = 0;          // the launch uses the 512-thread build of the workgroup program (at most one workgroup per CU)
    int edge_unroll = 1;      // edges in flight per thread of the edge kernel
    int edge_major = 0;       // state columns numbered by edge (gcsadmm_graph_desc.edge_major_columns)
    std::vector<char> col_owned;   // [NI] 1: the column of an incidence of this handle's vertices, 0: a ghost column
    int all_m4 = 0;           // 2: every wavefront-program vertex is a canonical box (4 facets) -> the box instantiation; 0: the generic one
    int align_rows = 0;       // group placement rule (group_base)
    int store_dl = 0;         // LDS holds the final dual directions of the facet rows (kernel template SDL)
    double nx = 0, nmu = 0;
    gcsadmm_params params{};
    bool params_set = false;
    // device buffers
    int *d_inc_ptr = nullptr, *d_deg_in = nullptr, *d_inc_edge = nullptr, *d_poly_ptr = nullptr;
    int *d_edge_inc_tail = nullptr, *d_edge_inc_head = nullptr;
    int *d_wave_slot_ptr = nullptr, *d_wave_vtx = nullptr, *d_special_vtx = nullptr, *d_special_kind = nullptr, *d_wg_vtx = nullptr;
    double *d_poly_A = nullptr, *d_poly_bc = nullptr, *d_center = nullptr;
    uint8_t *d_inc_counted = nullptr, *d_edge_counted = nullptr;
    gcsadmm_control_block *d_cb = nullptr;
    int *d_counters = nullptr;
    double *d_partials = nullptr, *d_sums = nullptr;
    unsigned *d_ticket = nullptr;
    // prox configuration (gcsadmm_vertex_prox): every non-terminal vertex, LDS of the border-only problem, own counters
    int *d_prox_vtx = nullptr, *d_prox_counters = nullptr, n_prox = 0, prox_lds_bytes = 0, src = -1, dst = -1;     // arrival counter of the single-launch edge step (edge_kernel MODE 2)
    std::vector<hipEvent_t> events;
    std::string err;
    // vertex partition across GPUs (gcsadmm_attach_comm): RCCL communicator, halo index lists and message buffers
    void *comm = nullptr;             // ncclComm_t
    int rank = 0, world = 1;
    std::vector<int> peers, peer_cnt, peer_off;   // neighbour ranks; columns per peer; first column of each peer's block
    int n_send = 0, n_recv = 0;       // halo columns sent / received per iteration
    int *d_send_cols = nullptr, *d_send_base = nullptr, *d_send_stride = nullptr;
    int *d_recv_cols = nullptr, *d_recv_base = nullptr, *d_recv_stride = nullptr;
    void *d_sendbuf = nullptr, *d_recvbuf = nullptr;
    double *d_sums6 = nullptr;        // the five norms + the inner-failure count, all-reduced together
    // warm start of the vertex solves (warm_start.h): one record per generic vertex, d_warm + d_warm_ptr[v]
    double *d_warm = nullptr;
    long long *d_warm_ptr = nullptr;
    size_t warm_doubles = 0;
    // slowest-first dispatch (reorder_kernel): per wavefront / per workgroup-program vertex, last Newton iteration count and launch order
    int *d_wave_iters = nullptr, *d_wave_order = nullptr, *d_wg_iters = nullptr, *d_wg_order = nullptr;
    int vertex_steps = 0;     // vertex steps enqueued since the last reset
    // overlapped partitioned loop (boundary vertices first, the halo exchange behind them while the interior is solved):
    // host copies of the wavefront packing (which wavefront holds which vertex) and of the column -> vertex map, the split of the
    // wavefronts into boundary (holding a vertex with a cut edge) and interior, a second stream for the exchange and two events
    std::vector<int> h_wave_slot_ptr, h_wave_vtx, col_vertex;
    int overlap_mode = 0;     // gcsadmm_set_overlap: 0 automatic, 1 forced (tests: works without peers), 2 off
    int n_wave_b = 0;         // boundary wavefronts (0: no split)
    int *d_split_ids = nullptr, *d_split_order = nullptr;    // [n_waves] static ids / launch order: boundary wavefronts first, then interior
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_boundary = nullptr, ev_halo = nullptr;
    // terminals that are regions (terminal_region.h): at most two, one workgroup each, on an auxiliary stream beside the vertex-step launch
    int n_term = 0, term_vtx[2] = {-1, -1}, term_is_src[2] = {0, 0};
    long long term_ws_off[2] = {0, 0}, term_rec_off[2] = {0, 0};
    double *d_term_ws = nullptr, *d_term_rec = nullptr;      // work arrays (when they do not fit LDS); warm-start records
    size_t term_rec_doubles = 0;
    int term_threads = 256, term_lds_doubles = 0;     // launch shape: one wavefront and LDS work arrays for small terminals
    hipStream_t term_stream = nullptr;
    hipEvent_t ev_term_fork = nullptr, ev_term_join = nullptr;
};

// ---- RCCL, bound at run time ----
// The library is not linked against librccl: a process that already carries an RCCL (PyTorch-ROCm ships its own copy with the
// SONAME of /opt/rocm's) must not end up with two, and a single-GPU user needs none.  dlopen returns the copy that is already
// loaded, or loads the system one.
namespace {
struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    std::string err;
    bool ok() const { return lib && GetUniqueId && CommInitRank && CommDestroy && GroupStart && GroupEnd && Send && Recv && AllReduce; }
};
RcclApi &rccl()
{
    static RcclApi api = [] {
        RcclApi a;
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (a.lib) break;
        }
        if (!a.lib) {
            const char *why = dlerror();      // (a second call would return NULL: the message is consumed)
            a.err = std::string("dlopen(librccl): ") + (why ? why : "not found");
            return a;
        }
#define RCCL_SYM(f) a.f = (decltype(a.f))dlsym(a.lib, "nccl" #f)
        RCCL_SYM(GetUniqueId); RCCL_SYM(CommInitRank); RCCL_SYM(CommDestroy); RCCL_SYM(GroupStart); RCCL_SYM(GroupEnd);
        RCCL_SYM(Send); RCCL_SYM(Recv); RCCL_SYM(AllReduce); RCCL_SYM(GetErrorString); RCCL_SYM(CommCount);
#undef RCCL_SYM
        if (!a.ok()) a.err = "librccl lacks an expected symbol";
        return a;
    }();
    return api;
}
}  // namespace
#define NCCLCHK(h, call)                                                                             \
    do {                                                                                             \
        ncclResult_t r_ = (call);                                                                    \
        if (r_ != ncclSuccess) {                                                                     \
            (h)->err = std::string(#call) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r_) : "RCCL error"); \
            return GCSADMM_ERR_HIP;                                                                  \
        }                                                                                            \
    } while (0)


static std::string g_create_error;

// Entry points work on the handle's device and leave the caller's current device as they found it (a process may hold
// handles on several devices, and PyTorch tracks "its" current device on its own).
struct DeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev)
    {
        int cur = -1;
        err = hipGetDevice(&cur);
        if (err == hipSuccess && cur != dev) { err = hipSetDevice(dev); if (err == hipSuccess) prev = cur; }
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define USE_DEVICE(h)                                                                                \
    DeviceGuard device_guard_((h)->device);                                                          \
    do {                                                                                             \
        if (device_guard_.err != hipSuccess) {                                                       \
            (h)->err = std::string("hipSetDevice: ") + hipGetErrorString(device_guard_.err);        \
            return GCSADMM_ERR_HIP;                                                                  \
        }                                                                                            \
    } while (0)

#define HIPCHK(h, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                            \
            return GCSADMM_ERR_HIP;                                                                  \
        }                                                                                            \
    } while (0)

template <class U> static hipError_t upload(U **dst, const U *src, size_t count)
{
    *dst = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc((void **)dst, count * sizeof(U));
    if (e != hipSuccess) return e;
    if (src) return hipMemcpy(*dst, src, count * sizeof(U), hipMemcpyHostToDevice);
    return hipMemset(*dst, 0, count * sizeof(U));
}

static VertexLaunchDesc make_launch_desc(gcsadmm_handle h, const gcsadmm_state *st);
static void halo_free(gcsadmm_handle h);

// The generic vertices of a handle are split at create between the wavefront program (n = 2, degree <= 63: vertex_kernel.h)
// and the workgroup program (everything else, and all vertices of small graphs: vertex_wg.hip); the closed-form vertices
// ride in the trailing workgroups of whichever launch exists.
static WgLaunchDesc make_wg_desc(gcsadmm_handle h, const gcsadmm_state *st, bool with_special)
{
    WgLaunchDesc d;
    d.n = h->n; d.dtype = h->dtype; d.n_vtx = h->n_wg; d.n_special = with_special ? h->n_special : 0; d.lds_bytes = h->wg_lds_bytes;
    d.vtx = h->d_wg_vtx; d.special_vtx = h->d_special_vtx; d.special_kind = h->d_special_kind;
    d.inc_ptr = h->d_inc_ptr; d.deg_in = h->d_deg_in; d.inc_edge = h->d_inc_edge; d.poly_ptr = h->d_poly_ptr;
    d.poly_A = h->d_poly_A; d.poly_bc = h->d_poly_bc; d.center = h->d_center;
    d.E = h->E; d.NI = h->NI; d.edge_major = h->edge_major; d.box = h->wg_box;
    d.zedge = st->zedge; d.mu = st->mu; d.copy = st->copy; d.xv = st->xv; d.zv = st->zv; d.yv = st->yv;
    d.counters = h->d_counters; d.cb = h->d_cb;
    d.eps_edge = h->params.eps_edge; d.ipm_tol = h->params.ipm_tol; d.ipm_max_iter = h->params.ipm_max_iter;
    d.warm = h->params.cold_start ? nullptr : h->d_warm; d.warm_ptr = h->d_warm_ptr;
    d.order = h->d_wg_order; d.unit_iters = h->d_wg_iters;
    return d;
}

// part: -1 the whole vertex step; 0 / 1 the boundary / interior wavefronts of the overlapped partitioned loop (handles whose generic
// vertices are all on the wavefront program; the closed-form vertices ride with the boundary part)
static gcsadmm_k::TermLaunchDesc make_term_desc(gcsadmm_handle h, const gcsadmm_state *st)
{
    gcsadmm_k::TermLaunchDesc d;
    d.n = h->n; d.dtype = h->dtype; d.count = h->n_term;
    for (int i = 0; i < 2; ++i) { d.vtx[i] = h->term_vtx[i]; d.is_src[i] = h->term_is_src[i]; d.ws_off[i] = h->term_ws_off[i]; }
    d.ws = h->d_term_ws; d.threads = h->term_threads; d.lds_doubles = h->term_lds_doubles;
    d.rec = h->params.cold_start ? nullptr : h->d_term_rec;
    for (int i = 0; i < 2; ++i) d.rec_off[i] = h->term_rec_off[i];
    d.inc_ptr = h->d_inc_ptr; d.deg_in = h->d_deg_in; d.inc_edge = h->d_inc_edge; d.poly_ptr = h->d_poly_ptr;
    d.poly_A = h->d_poly_A; d.poly_bc = h->d_poly_bc; d.center = h->d_center;
    d.E = h->E; d.NI = h->NI; d.edge_major = h->edge_major;
    d.zedge = st->zedge; d.mu = st->mu; d.copy = st->copy; d.xv = st->xv; d.zv = st->zv; d.yv = st->yv;
    d.counters = h->d_counters; d.cb = h->d_cb;
    d.eps_edge = h->params.eps_edge; d.ipm_tol = h->params.ipm_tol; d.ipm_max_iter = h->params.ipm_max_iter;
    return d;
}

template <class T> static gcsadmm_status launch_vertex(gcsadmm_handle h, const gcsadmm_state *st, hipStream_t s, int part = -1, bool reorder = false)
{
    // terminals that are regions: their kernel runs on the auxiliary stream beside the launches below (it is a latency-bound solve in
    // one or two workgroups; the vertex launches do not wait for it, the caller's stream does at the end)
    const bool with_term = h->n_term > 0 && part <= 0;
    if (with_term) {
        HIPCHK(h, hipEventRecord(h->ev_term_fork, s));
        HIPCHK(h, hipStreamWaitEvent(h->term_stream, h->ev_term_fork, 0));
        gcsadmm_terminal_launch(make_term_desc(h, st), h->term_stream);
        HIPCHK(h, hipEventRecord(h->ev_term_join, h->term_stream));
    }
    struct Join {       // (every return path below joins)
        gcsadmm_handle h; hipStream_t s; bool on;
        ~Join() { if (on) (void)hipStreamWaitEvent(s, h->ev_term_join, 0); }
    } join_{h, s, with_term};
    if (part >= 0) {
        VertexLaunchDesc d = make_launch_desc(h, st);
        const int off = part ? h->n_wave_b : 0, cnt = part ? h->n_waves - h->n_wave_b : h->n_wave_b;
        d.wave_order = h->d_split_order + off;
        d.n_waves = cnt;
        if (part) d.n_special = 0;
        launch_vertex_dim<2, T>(d, s);
        if (reorder && h->d_wave_iters)      // slowest first inside the part, on the part's own stream (its next launch reads the order)
            hipLaunchKernelGGL(reorder_kernel, dim3(1), dim3(REORDER_THREADS), 0, s, cnt, h->d_wave_iters, h->d_split_order + off, h->d_cb, h->d_split_ids + off);
        HIPCHK(h, hipGetLastError());
        return GCSADMM_OK;
    }
    const bool special_on_wave = h->n_waves > 0;
    if (h->n_waves > 0) {
        VertexLaunchDesc d = make_launch_desc(h, st);
        launch_vertex_dim<2, T>(d, s);
    }
    if (h->n_wg > 0 || (!special_on_wave && h->n_special > 0)) {
        if (h->wg_t512) gcsadmm_wg_launch_t512(make_wg_desc(h, st, !special_on_wave), s);
        else gcsadmm_wg_launch(make_wg_desc(h, st, !special_on_wave), s);
    }
    if (++h->vertex_steps % REORDER_EVERY == 0) {      // slowest-first dispatch of the following launches (graphs that need more than one round)
        if (h->d_wave_order) hipLaunchKernelGGL(reorder_kernel, dim3(1), dim3(REORDER_THREADS), 0, s, h->n_waves, h->d_wave_iters, h->d_wave_order, h->d_cb);
        if (h->d_wg_order) hipLaunchKernelGGL(reorder_kernel, dim3(1), dim3(REORDER_THREADS), 0, s, h->n_wg, h->d_wg_iters, h->d_wg_order, h->d_cb);
    }
    HIPCHK(h, hipGetLastError());
    return GCSADMM_OK;
}

// with_control: the control step rides in the same launches (gcsadmm_run); trace may be null
template <class T> static gcsadmm_status launch_edge(gcsadmm_handle h, const gcsadmm_state *st, double *sums, hipStream_t s,
                                                     bool with_control = false, double *trace = nullptr, bool sums6 = false)
{
    const gcsadmm_params &pp = h->params;
    const ControlParams cp{pp.tau_incr, pp.tau_decr, pp.nu, pp.eps_abs, pp.eps_rel, h->nx, h->nmu, pp.it_rho_limit, pp.max_it};
    EdgeArgs<T> a;
    a.E = h->E; a.NI = h->NI; a.c = h->c;
    a.edge_inc_tail = h->edge_major ? nullptr : h->d_edge_inc_tail; a.edge_inc_head = h->edge_major ? nullptr : h->d_edge_inc_head;
    a.inc_counted = h->d_inc_counted; a.edge_counted = h->d_edge_counted;
    a.copy = (const T *)st->copy; a.zedge = (T *)st->zedge; a.mu = (T *)st->mu; a.partials = h->d_partials;
    // one kernel instantiation per (state type, mode, words per copy)
    auto go = [&](auto mode, int blocks) {
        constexpr int M = decltype(mode)::value;
#define GCS_EDGE_U(CC, UU) hipLaunchKernelGGL((edge_kernel<T, M, CC, UU>), dim3(blocks), dim3(EDGE_BLOCK), 0, s, a, h->d_cb, sums, cp, h->d_counters, trace, h->d_ticket)
#define GCS_EDGE(CC) do { if (h->edge_unroll > 1) GCS_EDGE_U(CC, (edge_unroll<T, CC>())); else GCS_EDGE_U(CC, 1); } while (0)
        switch (h->c) {      // c = 2n + 1
        case 3: GCS_EDGE(3); break;
        case 5: GCS_EDGE(5); break;
        case 7: GCS_EDGE(7); break;
        case 9: GCS_EDGE(9); break;
        case 11: GCS_EDGE(11); break;
        case 13: GCS_EDGE(13); break;
        case 15: GCS_EDGE(15); break;
        default: GCS_EDGE(17);          // n = 8 (gcsadmm_create admits n = 1 .. 8)
        }
#undef GCS_EDGE_U
#undef GCS_EDGE
    };
    if (sums6) go(std::integral_constant<int, 3>(), h->edge_blocks);
    else if (with_control && h->edge_blocks == 1) go(std::integral_constant<int, 1>(), 1);
    else if (with_control) go(std::integral_constant<int, 2>(), h->edge_blocks);
    else {
        go(std::integral_constant<int, 0>(), h->edge_blocks);
        hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, s, h->d_partials, h->edge_blocks, sums, h->d_cb);
    }
    HIPCHK(h, hipGetLastError());
    return GCSADMM_OK;
}

static bool state_ok(gcsadmm_handle h, const gcsadmm_state *st)
{
    if (!h || !st || !st->copy || !st->mu || !st->zedge || !st->xv || !st->zv || !st->yv) { if (h) h->err = "null state pointer"; return false; }
    if (!h->params_set) { h->err = "gcsadmm_reset has not been called"; return false; }
    return true;
}

static VertexLaunchDesc make_launch_desc(gcsadmm_handle h, const gcsadmm_state *st)
{
    VertexLaunchDesc d;
    d.n_waves = h->n_waves; d.n_special = h->n_special; d.all_m4 = h->all_m4; d.lds_bytes = h->lds_bytes; d.align_rows = h->align_rows; d.store_dl = h->store_dl;
    d.wave_slot_ptr = h->d_wave_slot_ptr; d.wave_vtx = h->d_wave_vtx; d.special_vtx = h->d_special_vtx; d.special_kind = h->d_special_kind;
    d.inc_ptr = h->d_inc_ptr; d.deg_in = h->d_deg_in; d.inc_edge = h->d_inc_edge; d.poly_ptr = h->d_poly_ptr;
    d.poly_A = h->d_poly_A; d.poly_bc = h->d_poly_bc; d.center = h->d_center;
    d.E = h->E; d.NI = h->NI; d.MM = h->MM; d.edge_major = h->edge_major;
    d.zedge = st->zedge; d.mu = st->mu; d.copy = st->copy; d.xv = st->xv; d.zv = st->zv; d.yv = st->yv;
    d.counters = h->d_counters; d.cb = h->d_cb;
    d.eps_edge = h->params.eps_edge; d.ipm_tol = h->params.ipm_tol; d.ipm_max_iter = h->params.ipm_max_iter;
    d.warm = h->params.cold_start ? nullptr : h->d_warm; d.warm_ptr = h->d_warm_ptr;
    d.wave_order = h->d_wave_order; d.wave_iters = h->d_wave_iters;
    return d;
}

// ---- halo of a vertex partition: host-side helpers (C++ linkage) ----
// the checks of the halo lists, on the host alone: no allocation, no collective (gcsadmm_check_halo, and the head of attach_comm)
static gcsadmm_status halo_validate(gcsadmm_handle h, int rank, int world, const gcsadmm_halo_desc *hd)
{
    if (!hd || world < 1 || rank < 0 || rank >= world) { h->err = "bad communicator arguments"; return GCSADMM_ERR_BAD_ARG; }
    const int P = hd->num_peers;
    if (P < 0 || (P > 0 && (!hd->peer_rank || !hd->send_ptr || !hd->recv_ptr || !hd->send_cols || !hd->recv_cols))) { h->err = "null halo array"; return GCSADMM_ERR_BAD_ARG; }
    const int n_send = P ? hd->send_ptr[P] : 0, n_recv = P ? hd->recv_ptr[P] : 0;
    if (n_send != n_recv) { h->err = "halo lists: a partition sends and receives one column per cut edge and neighbour"; return GCSADMM_ERR_BAD_ARG; }
    for (int p = 0; p < P; ++p) {
        const int lo = hd->send_ptr[p], cnt = hd->send_ptr[p + 1] - lo;
        if (cnt < 0 || hd->recv_ptr[p + 1] - hd->recv_ptr[p] != cnt || hd->recv_ptr[p] != lo) { h->err = "halo lists: send and receive counts per peer must agree"; return GCSADMM_ERR_BAD_ARG; }
        if (hd->peer_rank[p] < 0 || hd->peer_rank[p] >= world || hd->peer_rank[p] == rank) { h->err = "halo lists: bad peer rank"; return GCSADMM_ERR_BAD_ARG; }
    }
    for (int j = 0; j < n_send; ++j) {
        if (hd->send_cols[j] < 0 || hd->send_cols[j] >= h->NI || hd->recv_cols[j] < 0 || hd->recv_cols[j] >= h->NI) { h->err = "halo lists: column out of range"; return GCSADMM_ERR_BAD_ARG; }
        if (!h->col_owned[hd->send_cols[j]]) { h->err = "halo lists: send column is not an owned incidence"; return GCSADMM_ERR_BAD_ARG; }
        if (h->col_owned[hd->recv_cols[j]]) { h->err = "halo lists: receive column is not a ghost column"; return GCSADMM_ERR_BAD_ARG; }
    }
    return GCSADMM_OK;
}

static void overlap_free(gcsadmm_handle h)
{
    for (void **p : {(void **)&h->d_split_ids, (void **)&h->d_split_order})
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (h->ev_boundary) { (void)hipEventDestroy(h->ev_boundary); h->ev_boundary = nullptr; }
    if (h->ev_halo) { (void)hipEventDestroy(h->ev_halo); h->ev_halo = nullptr; }
    if (h->comm_stream) { (void)hipStreamDestroy(h->comm_stream); h->comm_stream = nullptr; }
    h->n_wave_b = 0;
}

// The split of the wavefronts for the overlapped loop: boundary = holds a vertex one of whose columns is sent to a neighbour.  Only for
// handles whose generic vertices all run the wavefront program (config 4's strips); mode 1 (tests) splits even without neighbours --
// the first quarter of the wavefronts plays the boundary -- so that the two launches, the second stream and the events can be
// exercised on one GPU.  Leaves n_wave_b = 0 when there is nothing to split.
static gcsadmm_status overlap_setup(gcsadmm_handle h, const gcsadmm_halo_desc *hd)
{
    overlap_free(h);
    if (h->overlap_mode == 2 || h->n_waves < 2 || h->n_wg > 0) return GCSADMM_OK;
    const int n_send = (hd && hd->num_peers > 0) ? hd->send_ptr[hd->num_peers] : 0;
    if (n_send == 0 && h->overlap_mode != 1) return GCSADMM_OK;
    std::vector<char> vb((size_t)std::max(h->V, 1), 0);
    for (int j = 0; j < n_send; ++j) {
        const int v = h->col_vertex[hd->send_cols[j]];
        if (v >= 0) vb[v] = 1;
    }
    std::vector<int> ids_b, ids_i;
    for (int w = 0; w < h->n_waves; ++w) {
        bool b = (n_send == 0) && w < std::max(1, h->n_waves / 4);
        for (int q = h->h_wave_slot_ptr[w]; q < h->h_wave_slot_ptr[w + 1] && !b; ++q) b = vb[h->h_wave_vtx[q]] != 0;
        (b ? ids_b : ids_i).push_back(w);
    }
    if (ids_b.empty() || ids_i.empty()) return GCSADMM_OK;      // nothing to overlap with
    h->n_wave_b = (int)ids_b.size();
    ids_b.insert(ids_b.end(), ids_i.begin(), ids_i.end());
    HIPCHK(h, upload(&h->d_split_ids, ids_b.data(), ids_b.size()));
    HIPCHK(h, upload(&h->d_split_order, ids_b.data(), ids_b.size()));
    if (!h->d_wave_iters) HIPCHK(h, upload(&h->d_wave_iters, (const int *)nullptr, (size_t)h->n_waves));
    HIPCHK(h, hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_boundary, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_halo, hipEventDisableTiming));
    return GCSADMM_OK;
}

static void halo_free(gcsadmm_handle h)
{
    for (void **p : {(void **)&h->d_send_cols, (void **)&h->d_send_base, (void **)&h->d_send_stride, (void **)&h->d_recv_cols, (void **)&h->d_recv_base,
                     (void **)&h->d_recv_stride, &h->d_sendbuf, &h->d_recvbuf, (void **)&h->d_sums6})
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    h->peers.clear(); h->peer_cnt.clear(); h->peer_off.clear(); h->n_send = h->n_recv = 0;
}

static gcsadmm_status halo_upload(gcsadmm_handle h, const gcsadmm_halo_desc *hd)
{
    const int P = hd->num_peers, c = h->c;
    h->peers.assign(hd->peer_rank, hd->peer_rank + P);
    h->peer_cnt.resize(P); h->peer_off.resize(P);
    h->n_send = P ? hd->send_ptr[P] : 0; h->n_recv = P ? hd->recv_ptr[P] : 0;
    std::vector<int> sbase(std::max(h->n_send, 1)), sstride(std::max(h->n_send, 1));
    for (int p = 0; p < P; ++p) {
        const int lo = hd->send_ptr[p], cnt = hd->send_ptr[p + 1] - lo;
        h->peer_cnt[p] = cnt; h->peer_off[p] = lo;
        for (int j = 0; j < cnt; ++j) { sbase[lo + j] = lo * c + j; sstride[lo + j] = cnt; }     // block of peer p: [c][cnt] at lo * c
    }
    const size_t esz = h->dtype == GCSADMM_F64 ? 8 : 4;
    HIPCHK(h, upload(&h->d_send_cols, hd->send_cols, (size_t)h->n_send));
    HIPCHK(h, upload(&h->d_recv_cols, hd->recv_cols, (size_t)h->n_recv));
    HIPCHK(h, upload(&h->d_send_base, sbase.data(), (size_t)h->n_send));
    HIPCHK(h, upload(&h->d_send_stride, sstride.data(), (size_t)h->n_send));
    HIPCHK(h, upload(&h->d_recv_base, sbase.data(), (size_t)h->n_recv));        // same block layout on the receiving side
    HIPCHK(h, upload(&h->d_recv_stride, sstride.data(), (size_t)h->n_recv));
    HIPCHK(h, hipMalloc(&h->d_sendbuf, std::max<size_t>((size_t)h->n_send * c * esz, 16)));
    HIPCHK(h, hipMalloc(&h->d_recvbuf, std::max<size_t>((size_t)h->n_recv * c * esz, 16)));
    // [0..6): this partition's five norms + inner failures, written by the edge step; [6..12): their sum over the ranks.  (Out of
    // place: after the stop test has fired the edge step no longer writes, and an in-place all-reduce would multiply the stale
    // values by `world` with every further iteration that was enqueued.)
    HIPCHK(h, upload(&h->d_sums6, (const double *)nullptr, 12));
    return GCSADMM_OK;
}

template <class T> static gcsadmm_status halo_pack(gcsadmm_handle h, const gcsadmm_state *st, hipStream_t s)
{
    if (h->n_send == 0) return GCSADMM_OK;
    const int tot = h->c * h->n_send;
    hipLaunchKernelGGL((halo_pack_kernel<T>), dim3((tot + 255) / 256), dim3(256), 0, s, h->c, h->n_send, h->NI, h->d_send_cols,
                       h->d_send_base, h->d_send_stride, (const T *)st->copy, (T *)h->d_sendbuf, h->d_cb);
    HIPCHK(h, hipGetLastError());
    return GCSADMM_OK;
}
template <class T> static gcsadmm_status halo_unpack(gcsadmm_handle h, const gcsadmm_state *st, hipStream_t s)
{
    if (h->n_recv == 0) return GCSADMM_OK;
    const int tot = h->c * h->n_recv;
    hipLaunchKernelGGL((halo_unpack_kernel<T>), dim3((tot + 255) / 256), dim3(256), 0, s, h->c, h->n_recv, h->NI, h->d_recv_cols,
                       h->d_recv_base, h->d_recv_stride, (const T *)h->d_recvbuf, (T *)st->copy, h->d_cb);
    HIPCHK(h, hipGetLastError());
    return GCSADMM_OK;
}

// the grouped point-to-point exchange of the packed halo (one message per neighbour and direction)
static gcsadmm_status halo_transfer(gcsadmm_handle h, hipStream_t s)
{
    if (h->peers.empty()) return GCSADMM_OK;
    if (!h->comm) { h->err = "halo exchange needs a communicator (gcsadmm_attach_comm with an id)"; return GCSADMM_ERR_BAD_ARG; }
    const ncclDataType_t dt = h->dtype == GCSADMM_F64 ? ncclFloat64 : ncclFloat32;
    const size_t esz = h->dtype == GCSADMM_F64 ? 8 : 4;
    NCCLCHK(h, rccl().GroupStart());
    for (size_t p = 0; p < h->peers.size(); ++p) {
        const size_t off = (size_t)h->peer_off[p] * h->c * esz, cnt = (size_t)h->peer_cnt[p] * h->c;
        NCCLCHK(h, rccl().Send((const char *)h->d_sendbuf + off, cnt, dt, h->peers[p], (ncclComm_t)h->comm, s));
        NCCLCHK(h, rccl().Recv((char *)h->d_recvbuf + off, cnt, dt, h->peers[p], (ncclComm_t)h->comm, s));
    }
    NCCLCHK(h, rccl().GroupEnd());
    return GCSADMM_OK;
}

extern "C" {

const char *gcsadmm_last_error(gcsadmm_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

void gcsadmm_destroy(gcsadmm_handle h)
{
    if (!h) return;
    DeviceGuard device_guard_(h->device);
    void *ptrs[] = {h->d_inc_ptr, h->d_deg_in, h->d_inc_edge, h->d_poly_ptr, h->d_edge_inc_tail, h->d_edge_inc_head,
                    h->d_wave_slot_ptr, h->d_wave_vtx, h->d_special_vtx, h->d_special_kind, h->d_wg_vtx, h->d_poly_A, h->d_poly_bc,
                    h->d_center, h->d_inc_counted, h->d_edge_counted, h->d_cb, h->d_counters, h->d_partials, h->d_sums, h->d_ticket, h->d_prox_vtx, h->d_prox_counters,
                    h->d_warm, h->d_warm_ptr, h->d_wave_iters, h->d_wave_order, h->d_wg_iters, h->d_wg_order};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    halo_free(h);
    overlap_free(h);
    if (h->d_term_ws) (void)hipFree(h->d_term_ws);
    if (h->d_term_rec) (void)hipFree(h->d_term_rec);
    if (h->ev_term_fork) (void)hipEventDestroy(h->ev_term_fork);
    if (h->ev_term_join) (void)hipEventDestroy(h->ev_term_join);
    if (h->term_stream) (void)hipStreamDestroy(h->term_stream);
    if (h->comm && rccl().ok()) (void)rccl().CommDestroy((ncclComm_t)h->comm);
    for (auto ev : h->events) (void)hipEventDestroy(ev);
    delete h;
}

gcsadmm_status gcsadmm_create(const gcsadmm_graph_desc *g, gcsadmm_handle *out)
{
    if (out) *out = nullptr;
    auto fail = [&](gcsadmm_status st, const std::string &msg) { g_create_error = msg; return st; };
    if (!g || !out) return fail(GCSADMM_ERR_BAD_ARG, "null descriptor or output pointer");
    if (g->n < 1 || g->n > 8) return fail(GCSADMM_ERR_UNSUPPORTED, "the vertex kernels are instantiated for n = 1 .. 8");
    if (g->num_vertices < 0 || g->num_edges < 0) return fail(GCSADMM_ERR_BAD_ARG, "negative size");
    if (!g->inc_ptr || !g->poly_ptr || (g->num_edges > 0 && (!g->inc_edge || !g->inc_out || !g->edge_inc_tail || !g->edge_inc_head)) ||
        (g->num_vertices > 0 && (!g->poly_A || !g->poly_b || !g->center)))
        return fail(GCSADMM_ERR_BAD_ARG, "null graph array");
    if (g->state_dtype != GCSADMM_F64 && g->state_dtype != GCSADMM_F32) return fail(GCSADMM_ERR_BAD_ARG, "bad state_dtype");
    const int V = g->num_vertices, E = g->num_edges, n = g->n;
    const int NIo = g->inc_ptr[V];
    if (g->inc_ptr[0] != 0 || NIo < 0 || g->num_incidences < NIo) return fail(GCSADMM_ERR_BAD_ARG, "inconsistent incidence CSR");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(GCSADMM_ERR_NO_DEVICE, "no HIP device");
    if (g->device < 0 || g->device >= ndev) return fail(GCSADMM_ERR_BAD_ARG, "device ordinal out of range");

    // validate CSR, derive deg_in, facet maximum
    std::vector<int> deg_in(V, 0);
    int MM = 1;
    for (int v = 0; v < V; ++v) {
        const int lo = g->inc_ptr[v], hi = g->inc_ptr[v + 1];
        if (hi < lo) return fail(GCSADMM_ERR_BAD_ARG, "inc_ptr not monotone");
        bool seen_out = false;
        for (int k = lo; k < hi; ++k) {
            if (g->inc_edge[k] < 0 || g->inc_edge[k] >= E) return fail(GCSADMM_ERR_BAD_ARG, "inc_edge out of range");
            if (g->inc_out[k]) seen_out = true;
            else { if (seen_out) return fail(GCSADMM_ERR_BAD_ARG, "incoming incidences must precede outgoing ones"); deg_in[v]++; }
        }
        const int m = g->poly_ptr[v + 1] - g->poly_ptr[v];
        if (m < n + 1) return fail(GCSADMM_ERR_BAD_ARG, "polytope with fewer than n+1 facets cannot be bounded");
        MM = std::max(MM, m);
    }
    for (int e = 0; e < E; ++e)
        if (g->edge_inc_tail[e] < 0 || g->edge_inc_tail[e] >= g->num_incidences || g->edge_inc_head[e] < 0 ||
            g->edge_inc_head[e] >= g->num_incidences)
            return fail(GCSADMM_ERR_BAD_ARG, "edge incidence slot out of range");
    if (g->edge_major_columns != 0 && g->edge_major_columns != 1) return fail(GCSADMM_ERR_BAD_ARG, "edge_major_columns must be 0 or 1");
    if (g->edge_major_columns) {
        if (g->num_incidences != 2 * (int64_t)E) return fail(GCSADMM_ERR_BAD_ARG, "edge-major columns: num_incidences must be 2 num_edges");
        for (int e = 0; e < E; ++e)
            if (g->edge_inc_tail[e] != e || g->edge_inc_head[e] != E + e)
                return fail(GCSADMM_ERR_BAD_ARG, "edge-major columns: edge_inc_tail[e] must be e and edge_inc_head[e] num_edges + e");
    }

    // centred right-hand sides b - A c
    const int MT = g->poly_ptr[V];
    std::vector<double> bc(MT > 0 ? MT : 1);
    for (int v = 0; v < V; ++v)
        for (int j = g->poly_ptr[v]; j < g->poly_ptr[v + 1]; ++j) {
            double s = g->poly_b[j];
            for (int k = 0; k < n; ++k) s -= g->poly_A[(size_t)j * n + k] * g->center[(size_t)v * n + k];
            bc[j] = s;
            if (v != g->src && v != g->dst && !(s > 0.0)) return fail(GCSADMM_ERR_BAD_ARG, "center is not strictly inside its polytope");
        }

    // ---- classify the vertices ----
    // closed form: s, t (points) and vertices no flow can cross; generic: an interior-point solve each.  A generic vertex
    // goes to the WORKGROUP program (vertex_wg.hip) when the wavefront program cannot take it (n != 2, more than 63
    // incident edges) or when the graph is small enough that latency, not throughput, decides (or on request).
    // s / t: the reference builds them as points (utils.py:12-28, boxes of half-width 1e-6) -> closed form (special_vertex.h).  A terminal
    // whose polytope has an extent is a REGION: its sub-problem is the reference's with delta_sv / delta_tv (admm_solver_v3.py:450-464),
    // solved by terminal_region.h.  Extent = the widest distance from `center` to a facet; the rule the oracle uses (terminal_extent).
    int term_vtx[2] = {-1, -1}, term_is_src[2] = {0, 0}, n_term = 0;
    for (int term : {g->src, g->dst}) {
        if (term < 0 || term >= V) continue;
        double ext = 0;
        for (int j = g->poly_ptr[term]; j < g->poly_ptr[term + 1]; ++j) {
            double nrm = 0;
            for (int k = 0; k < n; ++k) nrm += g->poly_A[(size_t)j * n + k] * g->poly_A[(size_t)j * n + k];
            ext = std::max(ext, std::fabs(bc[j]) / std::sqrt(nrm > 0 ? nrm : 1.0));
        }
        if (ext > 1e-5) {
            const int d = g->inc_ptr[term + 1] - g->inc_ptr[term], din = deg_in[term];
            if (g->src == g->dst) return fail(GCSADMM_ERR_UNSUPPORTED, "source and target are the same region");
            if ((term == g->src ? d - din : din) < 1)
                return fail(GCSADMM_ERR_UNSUPPORTED, "a terminal that is a region needs an edge on its live side (outgoing for the source, incoming for the target)");
            term_vtx[n_term] = term; term_is_src[n_term] = term == g->src; ++n_term;
        }
    }
    auto is_region_terminal = [&](int v) { return (n_term > 0 && v == term_vtx[0]) || (n_term > 1 && v == term_vtx[1]); };
    auto is_special = [&](int v) {
        if (is_region_terminal(v)) return false;
        const int d = g->inc_ptr[v + 1] - g->inc_ptr[v], din = deg_in[v];
        return v == g->src || v == g->dst || din == 0 || d - din == 0;
    };
    int n_generic = 0;
    for (int v = 0; v < V; ++v) n_generic += !is_special(v) && !is_region_terminal(v);
    // Crossover of the two programs on n = 2 (measured on box lattices, profiles/r02/README.md: 1 024 vertices 3 570 vs 3 030 it/s,
    // 1 444 vertices 2 410 vs 3 060): the workgroup program holds 4 workgroups per CU (102 registers), i.e. 1 024 vertices in one
    // round of ~0.28 ms; the wavefront program packs up to 7 vertices per wavefront and serves up to ~7 000 in one round of 0.33 ms.
    constexpr int WG_AUTO_MAX = 1024;
    if (g->vertex_program < 0 || g->vertex_program > 3) return fail(GCSADMM_ERR_BAD_ARG, "vertex_program must be 0, 1, 2 or 3");
    const bool prefer_wg = g->vertex_program >= 2 || (g->vertex_program == 0 && n_generic <= WG_AUTO_MAX);
    std::vector<int> special_vtx, special_kind, wave_slot_ptr{0}, wave_vtx, wg_vtx;
    std::vector<char> on_wave(V, 0);
    int wg_lds = 0, MMw = 1;
    int wg_lds_box = 0;       // LDS per workgroup under the BOX instantiation's layout (used when every vertex turns out to be a box)
    bool wg_all_box = g->wave_generic_rows == 0 && gcsadmm_wg_has_box(n);      // (the knob that forces the generic wavefront variants forces this one too)
    bool all_m4 = (n == 2) && g->wave_generic_rows != 1, all_box = all_m4 && g->wave_generic_rows != 2;
    for (int v = 0; v < V; ++v) {
        const int d = g->inc_ptr[v + 1] - g->inc_ptr[v];
        const int m = g->poly_ptr[v + 1] - g->poly_ptr[v];
        if (is_region_terminal(v)) continue;      // its own kernel
        if (is_special(v)) {
            if (d > MAX_SPECIAL_DEG) return fail(GCSADMM_ERR_UNSUPPORTED, "terminal vertex degree above 256");
            special_vtx.push_back(v);
            special_kind.push_back(v == g->src ? 1 : (v == g->dst ? 2 : 0));
        } else if (n != 2 || d + 1 > WAVE || prefer_wg) {
            wg_vtx.push_back(v);
            wg_lds = std::max(wg_lds, gcsadmm_wg_lds_bytes(n, d + 1, m));
            wg_lds_box = std::max(wg_lds_box, gcsadmm_wg_lds_bytes(n, d + 1, m, true));
            if (wg_all_box && !canonical_box(n, m, g->poly_A + (size_t)g->poly_ptr[v] * n)) wg_all_box = false;
        } else {
            on_wave[v] = 1;
            MMw = std::max(MMw, m);
            if (m != 4) all_m4 = false;
            if (all_m4 && all_box && !canonical_box(2, m, g->poly_A + (size_t)g->poly_ptr[v] * 2)) all_box = false;   // [+e0, +e1, -e0, -e1]
        }
    }
    // threads per workgroup: 512 while every workgroup of the launch has a CU to itself (vertex_wg_launch.h), 256 otherwise
    const bool wg_t512 = !wg_vtx.empty() && g->vertex_program != 3 && wg_vtx.size() + 1 <= 256;
    if (wg_t512) {
        wg_lds = wg_lds_box = 0;
        for (int v : wg_vtx) {
            const int d = g->inc_ptr[v + 1] - g->inc_ptr[v], m = g->poly_ptr[v + 1] - g->poly_ptr[v];
            wg_lds = std::max(wg_lds, gcsadmm_wg_lds_bytes_t512(n, d + 1, m));
            wg_lds_box = std::max(wg_lds_box, gcsadmm_wg_lds_bytes_t512(n, d + 1, m, true));
        }
    }
    if (wg_all_box && n > 2 && !wg_vtx.empty()) wg_lds = wg_lds_box;      // the BOX instantiation (n > 2) and its structured unit layout
    if (wg_lds > 160 * 1024) return fail(GCSADMM_ERR_UNSUPPORTED, "a vertex sub-problem (degree x facets) does not fit the 160 KB of LDS of a CU");
    // heaviest sub-problems first: the launch ends when its slowest workgroup does
    std::stable_sort(wg_vtx.begin(), wg_vtx.end(), [&](int a, int b) {
        const long ca = (long)(g->inc_ptr[a + 1] - g->inc_ptr[a] + 1) * (g->poly_ptr[a + 1] - g->poly_ptr[a]);
        const long cb = (long)(g->inc_ptr[b + 1] - g->inc_ptr[b] + 1) * (g->poly_ptr[b + 1] - g->poly_ptr[b]);
        return ca > cb;
    });
    MM = MMw;   // facet maximum over the wavefront program's vertices only (sizes its LDS)

    // ---- wavefront program: pack its vertices into wavefronts, d+1 lanes each ----
    // LDS per wavefront with / without room for the final dual directions (kernel template SDL): they save the
    // update pass its facet rows (10k lattice +5 %) but must not cost a resident wavefront: kept only while four
    // wavefronts still fit a CU's 160 KB
    int store_dl = 0;
    auto lds_need = [&](int slots) { return (size_t)((all_m4 && all_box) ? gcs_box::lds_doubles(n, MM, slots, store_dl) : gcs::lds_doubles(n, MM, slots, store_dl)) * 8; };
    int slots_cap = MAX_SLOTS;
    int n_on_wave = 0;
    for (int v = 0; v < V; ++v) n_on_wave += on_wave[v];
    if (n_on_wave > 0) {
        while (slots_cap > 1 && lds_need(slots_cap) > 160 * 1024) --slots_cap;
        if (lds_need(slots_cap) > 160 * 1024) return fail(GCSADMM_ERR_UNSUPPORTED, "facet count too large for LDS");
        const int lds_cap = slots_cap;
        // fewer vertices than wave slots: one vertex per wavefront (a wavefront runs as long as its slowest vertex)
        const int want = std::max(1, (n_on_wave + 1023) / 1024);
        slots_cap = std::min(slots_cap, want);
        if (g->wave_slots > 0) slots_cap = std::max(1, std::min(lds_cap, (int)g->wave_slots));
    }
    // Group placement (vertex_program.inc group_base).  Aligned: no side segment straddles a 16-lane row, the
    // reductions use DPP row shifts (kernel RMODE 0).  Dense: groups back to back, more vertices per wavefront,
    // reductions by chained wave shifts (RMODE 1).  Aligned wins while the wavefronts fit the chip in
    // two rounds (2 x 1024 one-wave-per-SIMD slots); beyond that throughput is per wavefront and dense wins
    // (10k lattice: 1 490 vs 1 440 it/s; 100k lattice: 241 vs 264 it/s).
    int max_slots_used = 0;
    auto pack = [&](int align) {
        wave_slot_ptr.assign(1, 0); wave_vtx.clear(); max_slots_used = 0;
        int lanes = 0, slots = 0;
        for (int v = 0; v < V; ++v) {
            if (!on_wave[v]) continue;
            const int d = g->inc_ptr[v + 1] - g->inc_ptr[v], din = deg_in[v];
            int base = gcs::group_base(lanes, d, din, align);
            if (base < 0 || slots + 1 > slots_cap) {
                wave_slot_ptr.push_back((int)wave_vtx.size());
                slots = 0;
                base = gcs::group_base(0, d, din, align);
            }
            wave_vtx.push_back(v);
            lanes = base + d + 1; slots += 1;
            max_slots_used = std::max(max_slots_used, slots);
        }
        if ((int)wave_vtx.size() > wave_slot_ptr.back()) wave_slot_ptr.push_back((int)wave_vtx.size());
    };
    int align_rows = 1;
    pack(1);
    if (g->wave_align == 1) align_rows = 1;
    else if (g->wave_align == 2) align_rows = 0;
    else if ((int)wave_slot_ptr.size() - 1 > 2048) align_rows = 0;
    if (!align_rows) pack(0);
    const int n_waves = (int)wave_slot_ptr.size() - 1;

    auto *h = new (std::nothrow) gcsadmm_handle_s;
    if (!h) return fail(GCSADMM_ERR_HIP, "out of host memory");
    h->n = n; h->V = V; h->E = E; h->NI = g->num_incidences; h->NI_owned = NIo; h->c = 2 * n + 1; h->MM = MM;
    h->edge_major = g->edge_major_columns;
    h->col_owned.assign((size_t)std::max<int64_t>(g->num_incidences, 1), 0);
    for (int v = 0; v < V; ++v)
        for (int k = g->inc_ptr[v]; k < g->inc_ptr[v + 1]; ++k)
            h->col_owned[h->edge_major ? g->inc_edge[k] + (g->inc_out[k] ? 0 : E) : k] = 1;
    h->col_vertex.assign((size_t)std::max<int64_t>(g->num_incidences, 1), -1);
    for (int v = 0; v < V; ++v)
        for (int k = g->inc_ptr[v]; k < g->inc_ptr[v + 1]; ++k)
            h->col_vertex[h->edge_major ? g->inc_edge[k] + (g->inc_out[k] ? 0 : E) : k] = v;
    h->h_wave_slot_ptr = wave_slot_ptr; h->h_wave_vtx = wave_vtx;
    h->dtype = g->state_dtype; h->device = g->device;
    h->n_waves = n_waves; h->n_special = (int)special_vtx.size();
    h->slots_cap = std::max(1, max_slots_used);
    h->all_m4 = (all_m4 && all_box) ? 2 : 0; h->align_rows = align_rows;
    if (n_waves > 0) {
        store_dl = 1;
        if (lds_need(h->slots_cap) > 40 * 1024) store_dl = 0;
        if (g->wave_store_dl == 1) store_dl = 1;
        else if (g->wave_store_dl == 2) store_dl = 0;
        if (store_dl && lds_need(h->slots_cap) > 160 * 1024) store_dl = 0;
    }
    h->store_dl = store_dl;
    h->lds_bytes = n_waves > 0 ? (int)lds_need(h->slots_cap) : 0;
    h->n_wg = (int)wg_vtx.size(); h->wg_lds_bytes = wg_lds; h->wg_box = (wg_all_box && !wg_vtx.empty()) ? 1 : 0; h->wg_t512 = wg_t512 ? 1 : 0;
    h->nx = g->nx_global > 0 ? g->nx_global : (4.0 * n + 1) * (V + 2.0 * E);
    h->nmu = g->nmu_global > 0 ? g->nmu_global : (4.0 * n + 2) * E;
    {
        h->edge_unroll = edge_unroll_rt(h->dtype, h->c, E);
        const int tile = EDGE_BLOCK * h->edge_unroll;     // edges per workgroup and pass
        h->edge_blocks = std::max(1, std::min((E + tile - 1) / tile, 2048));
    }
    auto bail = [&](hipError_t e, const char *what) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(e);
        gcsadmm_destroy(h);
        return GCSADMM_ERR_HIP;
    };
    hipError_t e;
    DeviceGuard device_guard_(g->device);
    if ((e = device_guard_.err) != hipSuccess) return bail(e, "hipSetDevice");
#define UP(dst, src, cnt) if ((e = upload(&h->dst, src, (size_t)(cnt))) != hipSuccess) return bail(e, "upload " #dst)
    UP(d_inc_ptr, g->inc_ptr, V + 1);
    UP(d_deg_in, deg_in.data(), V);
    UP(d_inc_edge, g->inc_edge, NIo);
    UP(d_poly_ptr, g->poly_ptr, V + 1);
    UP(d_edge_inc_tail, g->edge_inc_tail, E);
    UP(d_edge_inc_head, g->edge_inc_head, E);
    UP(d_wave_slot_ptr, wave_slot_ptr.data(), wave_slot_ptr.size());
    UP(d_wave_vtx, wave_vtx.data(), wave_vtx.size());
    UP(d_special_vtx, special_vtx.data(), special_vtx.size());
    UP(d_special_kind, special_kind.data(), special_kind.size());
    UP(d_wg_vtx, wg_vtx.data(), wg_vtx.size());
    UP(d_poly_A, g->poly_A, (size_t)MT * n);
    UP(d_poly_bc, bc.data(), MT);
    UP(d_center, g->center, (size_t)V * n);
    if (g->inc_counted) UP(d_inc_counted, g->inc_counted, g->num_incidences);
    if (g->edge_counted) UP(d_edge_counted, g->edge_counted, E);
    UP(d_cb, (const gcsadmm_control_block *)nullptr, 1);
    UP(d_counters, (const int *)nullptr, 2);
    UP(d_partials, (const double *)nullptr, (size_t)h->edge_blocks * 5);
    UP(d_sums, (const double *)nullptr, 5);
    UP(d_ticket, (const unsigned *)nullptr, 1);
    if (n_term > 0) {       // region terminals: workspace, an auxiliary stream and the fork / join events
        long long off = 0, largest = 0, roff = 0;
        int rows = 0;
        for (int i = 0; i < n_term; ++i) {
            const int v = term_vtx[i], d = g->inc_ptr[v + 1] - g->inc_ptr[v], din = deg_in[v], live = term_is_src[i] ? d - din : din;
            const long long need = gcsadmm_terminal_ws_doubles(n, g->poly_ptr[v + 1] - g->poly_ptr[v], live);
            h->term_vtx[i] = v; h->term_is_src[i] = term_is_src[i]; h->term_ws_off[i] = off;
            off += need; largest = std::max(largest, need);
            h->term_rec_off[i] = roff; roff += gcsadmm_terminal_record_doubles(n, g->poly_ptr[v + 1] - g->poly_ptr[v], live);
            rows = std::max(rows, live * 2 * (g->poly_ptr[v + 1] - g->poly_ptr[v]));
        }
        h->n_term = n_term;
        // the solve is latency-bound: work arrays in LDS while they fit 48 KB, one wavefront (barriers and reductions stay inside it)
        // while no phase has more than four passes over its rows
        h->term_lds_doubles = largest * 8 <= 48 * 1024 ? (int)largest : 0;
        h->term_threads = rows <= 256 ? 64 : 256;
        UP(d_term_ws, (const double *)nullptr, (size_t)off);
        h->term_rec_doubles = (size_t)roff;
        UP(d_term_rec, (const double *)nullptr, (size_t)roff);
        if ((e = hipStreamCreateWithFlags(&h->term_stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
        if ((e = hipEventCreateWithFlags(&h->ev_term_fork, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
        if ((e = hipEventCreateWithFlags(&h->ev_term_join, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    }
    {   // warm-start workspace: one record per generic vertex (either program), none for the closed-form ones; zero = no record
        std::vector<long long> wp(V + 1, 0);
        for (int v = 0; v < V; ++v)
            wp[v + 1] = wp[v] + ((is_special(v) || is_region_terminal(v)) ? 0 : gcs_ws::warm_record_doubles(n, g->poly_ptr[v + 1] - g->poly_ptr[v], g->inc_ptr[v + 1] - g->inc_ptr[v]));
        h->warm_doubles = (size_t)wp[V];
        UP(d_warm_ptr, wp.data(), V + 1);
        UP(d_warm, (const double *)nullptr, h->warm_doubles);
    }
    {   // slowest-first dispatch: only where a launch needs more than one round of the chip (small graphs run all at once)
        std::vector<int> iota(std::max(std::max(n_waves, (int)wg_vtx.size()), 1));
        for (size_t i = 0; i < iota.size(); ++i) iota[i] = (int)i;
        if (n_waves >= REORDER_MIN_UNITS) { UP(d_wave_iters, (const int *)nullptr, n_waves); UP(d_wave_order, iota.data(), n_waves); }
        if ((int)wg_vtx.size() >= REORDER_MIN_UNITS) { UP(d_wg_iters, (const int *)nullptr, wg_vtx.size()); UP(d_wg_order, iota.data(), wg_vtx.size()); }
    }
    {
        std::vector<int> pv;
        int mm_all = 1;
        for (int v = 0; v < V; ++v) {
            if (v != g->src && v != g->dst) pv.push_back(v);
            mm_all = std::max(mm_all, g->poly_ptr[v + 1] - g->poly_ptr[v]);
        }
        h->n_prox = (int)pv.size(); h->src = g->src; h->dst = g->dst;
        h->prox_lds_bytes = gcsadmm_wg_lds_bytes(n, 1, mm_all);
        UP(d_prox_vtx, pv.data(), pv.size());
        UP(d_prox_counters, (const int *)nullptr, 2);
    }
#undef UP
    if (h->lds_bytes > 48 * 1024) {
        e = h->dtype == GCSADMM_F64 ? set_lds_attr<2, double>(h->all_m4, h->lds_bytes) : set_lds_attr<2, float>(h->all_m4, h->lds_bytes);
        if (e != hipSuccess) return bail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    if (h->wg_lds_bytes > 48 * 1024 && (e = h->wg_t512 ? gcsadmm_wg_set_lds_t512(h->n, h->dtype, h->wg_lds_bytes)
                                                        : gcsadmm_wg_set_lds(h->n, h->dtype, h->wg_lds_bytes)) != hipSuccess)
        return bail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize, workgroup program)");
    *out = h;
    return GCSADMM_OK;
}

gcsadmm_status gcsadmm_reset(gcsadmm_handle h, const gcsadmm_params *p, void *stream)
{
    if (!h || !p) return GCSADMM_ERR_BAD_ARG;
    if (!(p->rho > 0) || p->max_it < 1 || !(p->ipm_tol > 0) || p->ipm_max_iter < 1) { h->err = "bad parameter"; return GCSADMM_ERR_BAD_ARG; }
    h->params = *p; h->params_set = true;
    gcsadmm_control_block cb{};
    cb.rho = p->rho; cb.mu_scale = 1.0; cb.it = 1; cb.status = GCSADMM_RUNNING;
    USE_DEVICE(h);
    HIPCHK(h, hipMemcpyAsync(h->d_cb, &cb, sizeof(cb), hipMemcpyHostToDevice, (hipStream_t)stream));
    HIPCHK(h, hipMemsetAsync(h->d_counters, 0, 2 * sizeof(int), (hipStream_t)stream));
    // a new run starts without warm-start records (runs from the same state are then identical, whatever ran before)
    if (h->warm_doubles > 0) HIPCHK(h, hipMemsetAsync(h->d_warm, 0, h->warm_doubles * sizeof(double), (hipStream_t)stream));
    if (h->term_rec_doubles > 0) HIPCHK(h, hipMemsetAsync(h->d_term_rec, 0, h->term_rec_doubles * sizeof(double), (hipStream_t)stream));
    h->vertex_steps = 0;
    if (h->d_split_order) HIPCHK(h, hipMemcpyAsync(h->d_split_order, h->d_split_ids, sizeof(int) * (size_t)h->n_waves, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    for (auto po : {std::make_pair(h->d_wave_order, h->n_waves), std::make_pair(h->d_wg_order, h->n_wg)})
        if (po.first) {
            std::vector<int> iota(po.second);
            for (int i = 0; i < po.second; ++i) iota[i] = i;
            HIPCHK(h, hipMemcpyAsync(po.first, iota.data(), sizeof(int) * po.second, hipMemcpyHostToDevice, (hipStream_t)stream));
            HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));      // iota is a stack object
        }
    HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));   // cb is a stack object
    return GCSADMM_OK;
}

gcsadmm_status gcsadmm_vertex_step(gcsadmm_handle h, const gcsadmm_state *st, void *stream)
{
    if (!state_ok(h, st)) return GCSADMM_ERR_BAD_ARG;
    USE_DEVICE(h);
    return h->dtype == GCSADMM_F64 ? launch_vertex<double>(h, st, (hipStream_t)stream) : launch_vertex<float>(h, st, (hipStream_t)stream);
}

gcsadmm_status gcsadmm_edge_step(gcsadmm_handle h, const gcsadmm_state *st, double *sums_dev, void *stream)
{
    if (!state_ok(h, st) || !sums_dev) return GCSADMM_ERR_BAD_ARG;
    USE_DEVICE(h);
    return h->dtype == GCSADMM_F64 ? launch_edge<double>(h, st, sums_dev, (hipStream_t)stream) : launch_edge<float>(h, st, sums_dev, (hipStream_t)stream);
}

gcsadmm_status gcsadmm_control(gcsadmm_handle h, const double *sums_dev, double *trace_dev, void *stream)
{
    if (!h || !sums_dev || !h->params_set) return GCSADMM_ERR_BAD_ARG;
    USE_DEVICE(h);
    const gcsadmm_params &p = h->params;
    ControlParams cp{p.tau_incr, p.tau_decr, p.nu, p.eps_abs, p.eps_rel, h->nx, h->nmu, p.it_rho_limit, p.max_it};
    hipLaunchKernelGGL(control_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, h->d_cb, sums_dev, cp, h->d_counters, trace_dev, false);
    HIPCHK(h, hipGetLastError());
    return GCSADMM_OK;
}

gcsadmm_status gcsadmm_run(gcsadmm_handle h, const gcsadmm_state *st, int32_t k, double *trace_dev, void *stream)
{
    if (!state_ok(h, st) || k < 0) return GCSADMM_ERR_BAD_ARG;
    USE_DEVICE(h);      // (the edge launches below are issued from here, not through an entry point that guards for itself)
    for (int i = 0; i < k; ++i) {
        gcsadmm_status s;
        if ((s = gcsadmm_vertex_step(h, st, stream)) != GCSADMM_OK) return s;
        // edge step and control step in two launches (one when all edges fit a single workgroup)
        s = h->dtype == GCSADMM_F64 ? launch_edge<double>(h, st, h->d_sums, (hipStream_t)stream, true, trace_dev)
                                    : launch_edge<float>(h, st, h->d_sums, (hipStream_t)stream, true, trace_dev);
        if (s != GCSADMM_OK) return s;
    }
    return GCSADMM_OK;
}

// =================================================================================================
// vertex partitions across GPUs: one handle per rank, RCCL over xGMI.
// Per iteration, all on one stream with no host synchronisation:
//   vertex step -> pack the cut edges' copies per neighbour -> grouped ncclSend / ncclRecv -> unpack into the ghost columns
//   -> edge step on local + ghost columns -> ncclAllReduce(sum) of the five norms and the inner-failure count (6 doubles)
//   -> control (every rank takes the same decision from the same numbers).
// Both messages are latency-bound at every configuration of BASELINE.json (about 25 KB per boundary of the 100k lattice,
// 48 bytes for the all-reduce).
// =================================================================================================
gcsadmm_status gcsadmm_comm_unique_id(void *id128)
{
    if (!id128) return GCSADMM_ERR_BAD_ARG;
    if (!rccl().ok()) { g_create_error = rccl().err; return GCSADMM_ERR_HIP; }
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) { g_create_error = "ncclGetUniqueId failed"; return GCSADMM_ERR_HIP; }
    std::memcpy(id128, &id, sizeof(id));
    return GCSADMM_OK;
}

gcsadmm_status gcsadmm_check_halo(gcsadmm_handle h, int32_t rank, int32_t world, const gcsadmm_halo_desc *halo)
{
    if (!h) return GCSADMM_ERR_BAD_ARG;
    if (h->d_sums6) { h->err = "a communicator is already attached"; return GCSADMM_ERR_BAD_ARG; }
    return halo_validate(h, rank, world, halo);
}

gcsadmm_status gcsadmm_attach_comm(gcsadmm_handle h, int32_t rank, int32_t world, const void *id128, const gcsadmm_halo_desc *halo)
{
    // everything that can fail on this rank alone comes first: a rank must not return an error while its peers wait in the collective
    gcsadmm_status st = gcsadmm_check_halo(h, rank, world, halo);
    if (st != GCSADMM_OK) return st;
    if (id128 && !rccl().ok()) { h->err = rccl().err; return GCSADMM_ERR_HIP; }
    USE_DEVICE(h);
    h->rank = rank; h->world = world;
    if ((st = halo_upload(h, halo)) != GCSADMM_OK) { halo_free(h); return st; }
    if (id128) {      // id128 == NULL: no communicator (the host moves the packed buffers itself; gcsadmm_run_partitioned needs one)
        ncclUniqueId id;
        std::memcpy(&id, id128, sizeof(id));
        ncclComm_t comm = nullptr;
        const ncclResult_t r = rccl().CommInitRank(&comm, world, id, rank);
        if (r != ncclSuccess) {      // not attached: the handle can be attached again
            h->err = std::string("ncclCommInitRank: ") + (rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error");
            halo_free(h);
            return GCSADMM_ERR_HIP;
        }
        h->comm = comm;
    }
    return overlap_setup(h, halo);
}

gcsadmm_status gcsadmm_halo_pack(gcsadmm_handle h, const gcsadmm_state *st, void *stream)
{
    if (!state_ok(h, st) || !h->d_sums6) { if (h) h->err = "no halo attached"; return GCSADMM_ERR_BAD_ARG; }
    USE_DEVICE(h);
    return h->dtype == GCSADMM_F64 ? halo_pack<double>(h, st, (hipStream_t)stream) : halo_pack<float>(h, st, (hipStream_t)stream);
}
gcsadmm_status gcsadmm_halo_unpack(gcsadmm_handle h, const gcsadmm_state *st, void *stream)
{
    if (!state_ok(h, st) || !h->d_sums6) { if (h) h->err = "no halo attached"; return GCSADMM_ERR_BAD_ARG; }
    USE_DEVICE(h);
    return h->dtype == GCSADMM_F64 ? halo_unpack<double>(h, st, (hipStream_t)stream) : halo_unpack<float>(h, st, (hipStream_t)stream);
}
gcsadmm_status gcsadmm_halo_buffers(gcsadmm_handle h, void **send_buf, void **recv_buf, int64_t *num_elements)
{
    if (!h || !h->d_sums6) { if (h) h->err = "no halo attached"; return GCSADMM_ERR_BAD_ARG; }
    if (send_buf) *send_buf = h->d_sendbuf;
    if (recv_buf) *recv_buf = h->d_recvbuf;
    if (num_elements) *num_elements = (int64_t)h->c * h->n_send;
    return GCSADMM_OK;
}

gcsadmm_status gcsadmm_halo_exchange(gcsadmm_handle h, const gcsadmm_state *st, void *stream)
{
    if (!state_ok(h, st) || !h->d_sums6) { if (h) h->err = "no halo attached"; return GCSADMM_ERR_BAD_ARG; }
    USE_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    gcsadmm_status r;
    if ((r = h->dtype == GCSADMM_F64 ? halo_pack<double>(h, st, s) : halo_pack<float>(h, st, s)) != GCSADMM_OK) return r;
    if ((r = halo_transfer(h, s)) != GCSADMM_OK) return r;
    return h->dtype == GCSADMM_F64 ? halo_unpack<double>(h, st, s) : halo_unpack<float>(h, st, s);
}

// one loop for gcsadmm_run_partitioned and its event-bracketed twin: ev != nullptr records 6 events per iteration on the stream
// (before / after the vertex step, after the halo exchange, after the edge step, after the all-reduce, after the control step)
static gcsadmm_status run_partitioned_loop(gcsadmm_handle h, const gcsadmm_state *st, int k, double *trace_dev, hipStream_t s, hipEvent_t *ev)
{
    const gcsadmm_params &pp = h->params;
    const ControlParams cp{pp.tau_incr, pp.tau_decr, pp.nu, pp.eps_abs, pp.eps_rel, h->nx, h->nmu, pp.it_rho_limit, pp.max_it};
    if (!h->comm && h->world > 1) { h->err = "gcsadmm_run_partitioned needs a communicator (gcsadmm_attach_comm with an id)"; return GCSADMM_ERR_BAD_ARG; }
    // OVERLAPPED form (not for the stage-timed twin, whose events want one stream): the wavefronts that hold a vertex with a
    // cut edge are launched FIRST and on a second stream, with pack, grouped send / recv and unpack of the halo behind them; the interior
    // wavefronts are launched on the caller's stream at the same time, and the edge step waits for both.  Same kernels, same numbers: the
    // split only changes what runs when.
    //   s : [ev_boundary = previous iteration done] -> interior launch -> wait [ev_halo] -> edge step -> all-reduce -> control
    //   sc: wait [ev_boundary] -> boundary launch (+ closed-form vertices) -> pack -> send / recv -> unpack -> [ev_halo]
    // (The two launches must be CONCURRENT: one after the other on one stream each waits for its own slowest wavefront -- measured on a
    // strip of 12.6 k vertices, 348 -> 509 us per iteration.  RCCL orders the operations of one communicator across streams itself;
    // every rank issues them in the same order.)
    const bool overlap = !ev && h->n_wave_b > 0 && h->overlap_mode != 2;
    for (int i = 0; i < k && overlap; ++i) {
        gcsadmm_status r;
        const bool f64 = h->dtype == GCSADMM_F64;
        const bool reorder = ++h->vertex_steps % REORDER_EVERY == 0;
        HIPCHK(h, hipEventRecord(h->ev_boundary, s));
        HIPCHK(h, hipStreamWaitEvent(h->comm_stream, h->ev_boundary, 0));
        if ((r = f64 ? launch_vertex<double>(h, st, h->comm_stream, 0, reorder) : launch_vertex<float>(h, st, h->comm_stream, 0, reorder)) != GCSADMM_OK) return r;
        if ((r = gcsadmm_halo_exchange(h, st, (void *)h->comm_stream)) != GCSADMM_OK) return r;
        HIPCHK(h, hipEventRecord(h->ev_halo, h->comm_stream));
        if ((r = f64 ? launch_vertex<double>(h, st, s, 1, reorder) : launch_vertex<float>(h, st, s, 1, reorder)) != GCSADMM_OK) return r;
        HIPCHK(h, hipStreamWaitEvent(s, h->ev_halo, 0));
        r = f64 ? launch_edge<double>(h, st, h->d_sums6, s, false, nullptr, true) : launch_edge<float>(h, st, h->d_sums6, s, false, nullptr, true);
        if (r != GCSADMM_OK) return r;
        const double *reduced = h->d_sums6;
        if (h->comm) {
            NCCLCHK(h, rccl().AllReduce(h->d_sums6, h->d_sums6 + 6, 6, ncclFloat64, ncclSum, (ncclComm_t)h->comm, s));
            reduced = h->d_sums6 + 6;
        }
        hipLaunchKernelGGL(control_kernel, dim3(1), dim3(1), 0, s, h->d_cb, reduced, cp, h->d_counters, trace_dev, true);
        HIPCHK(h, hipGetLastError());
    }
    if (overlap) {      // the caller's stream is the one the caller synchronises: nothing of this call may still run on the other
        HIPCHK(h, hipEventRecord(h->ev_halo, h->comm_stream));
        HIPCHK(h, hipStreamWaitEvent(s, h->ev_halo, 0));
        return GCSADMM_OK;
    }
    for (int i = 0; i < k; ++i) {
        gcsadmm_status r;
        if (ev) HIPCHK(h, hipEventRecord(ev[6 * i + 0], s));
        if ((r = gcsadmm_vertex_step(h, st, (void *)s)) != GCSADMM_OK) return r;
        if (ev) HIPCHK(h, hipEventRecord(ev[6 * i + 1], s));
        if ((r = gcsadmm_halo_exchange(h, st, (void *)s)) != GCSADMM_OK) return r;
        if (ev) HIPCHK(h, hipEventRecord(ev[6 * i + 2], s));
        r = h->dtype == GCSADMM_F64 ? launch_edge<double>(h, st, h->d_sums6, s, false, nullptr, true)
                                    : launch_edge<float>(h, st, h->d_sums6, s, false, nullptr, true);      // sums + failure count, one launch
        if (r != GCSADMM_OK) return r;
        if (ev) HIPCHK(h, hipEventRecord(ev[6 * i + 3], s));
        const double *reduced = h->d_sums6;
        if (h->comm) {
            NCCLCHK(h, rccl().AllReduce(h->d_sums6, h->d_sums6 + 6, 6, ncclFloat64, ncclSum, (ncclComm_t)h->comm, s));
            reduced = h->d_sums6 + 6;
        }
        if (ev) HIPCHK(h, hipEventRecord(ev[6 * i + 4], s));
        hipLaunchKernelGGL(control_kernel, dim3(1), dim3(1), 0, s, h->d_cb, reduced, cp, h->d_counters, trace_dev, true);
        HIPCHK(h, hipGetLastError());
        if (ev) HIPCHK(h, hipEventRecord(ev[6 * i + 5], s));
    }
    return GCSADMM_OK;
}

gcsadmm_status gcsadmm_run_partitioned(gcsadmm_handle h, const gcsadmm_state *st, int32_t k, double *trace_dev, void *stream)
{
    if (!state_ok(h, st) || k < 0) return GCSADMM_ERR_BAD_ARG;
    if (!h->d_sums6) { h->err = "gcsadmm_attach_comm has not been called"; return GCSADMM_ERR_BAD_ARG; }
    USE_DEVICE(h);
    return run_partitioned_loop(h, st, k, trace_dev, (hipStream_t)stream, nullptr);
}

gcsadmm_status gcsadmm_run_partitioned_timed(gcsadmm_handle h, const gcsadmm_state *st, int32_t k, double *trace_dev, void *stream,
                                             float *vertex_ms, float *halo_ms, float *edge_ms, float *reduce_ms)
{
    if (!state_ok(h, st) || k < 0 || !vertex_ms || !halo_ms || !edge_ms || !reduce_ms) return GCSADMM_ERR_BAD_ARG;
    if (!h->d_sums6) { h->err = "gcsadmm_attach_comm has not been called"; return GCSADMM_ERR_BAD_ARG; }
    USE_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    while (h->events.size() < (size_t)6 * k) {
        hipEvent_t e;
        HIPCHK(h, hipEventCreate(&e));
        h->events.push_back(e);
    }
    gcsadmm_status r = run_partitioned_loop(h, st, k, trace_dev, s, h->events.data());
    if (r != GCSADMM_OK) return r;
    HIPCHK(h, hipStreamSynchronize(s));
    double acc[4] = {0, 0, 0, 0};
    for (int i = 0; i < k; ++i) {
        float t = 0;
        HIPCHK(h, hipEventElapsedTime(&t, h->events[6 * i + 0], h->events[6 * i + 1])); acc[0] += t;
        HIPCHK(h, hipEventElapsedTime(&t, h->events[6 * i + 1], h->events[6 * i + 2])); acc[1] += t;
        HIPCHK(h, hipEventElapsedTime(&t, h->events[6 * i + 2], h->events[6 * i + 3])); acc[2] += t;
        HIPCHK(h, hipEventElapsedTime(&t, h->events[6 * i + 3], h->events[6 * i + 5])); acc[3] += t;
    }
    *vertex_ms = (float)acc[0]; *halo_ms = (float)acc[1]; *edge_ms = (float)acc[2]; *reduce_ms = (float)acc[3];
    return GCSADMM_OK;
}

gcsadmm_status gcsadmm_set_overlap(gcsadmm_handle h, int32_t mode, int32_t *boundary_units)
{
    if (!h || mode < 0 || mode > 2) return GCSADMM_ERR_BAD_ARG;
    if (boundary_units) *boundary_units = 0;
    if (!h->d_sums6) { h->err = "gcsadmm_attach_comm has not been called"; return GCSADMM_ERR_BAD_ARG; }
    USE_DEVICE(h);
    h->overlap_mode = mode;
    // the split is derived from the halo lists uploaded at attach: rebuild them as a descriptor of host arrays
    std::vector<int> send_cols((size_t)std::max(h->n_send, 1)), ptr(h->peers.size() + 1, 0);
    if (h->n_send > 0) HIPCHK(h, hipMemcpy(send_cols.data(), h->d_send_cols, sizeof(int) * (size_t)h->n_send, hipMemcpyDeviceToHost));
    for (size_t p = 0; p < h->peers.size(); ++p) ptr[p + 1] = ptr[p] + h->peer_cnt[p];
    gcsadmm_halo_desc hd{};
    hd.num_peers = (int)h->peers.size(); hd.peer_rank = h->peers.data(); hd.send_ptr = ptr.data(); hd.send_cols = send_cols.data();
    hd.recv_ptr = ptr.data(); hd.recv_cols = send_cols.data();
    const gcsadmm_status r = overlap_setup(h, &hd);
    if (boundary_units) *boundary_units = h->n_wave_b;
    return r;
}

gcsadmm_status gcsadmm_comm_count(gcsadmm_handle h, int32_t *count)
{
    if (!h || !count) return GCSADMM_ERR_BAD_ARG;
    *count = 0;
    if (!h->comm) return GCSADMM_OK;        // no communicator attached (single handle, or a host-side transport)
    if (!rccl().CommCount) { h->err = "librccl lacks ncclCommCount"; return GCSADMM_ERR_HIP; }
    USE_DEVICE(h);
    int n = 0;
    NCCLCHK(h, rccl().CommCount((ncclComm_t)h->comm, &n));
    *count = n;
    return GCSADMM_OK;
}

gcsadmm_status gcsadmm_vertex_prox(gcsadmm_handle h, const double *q_dev, const double *c_dev, double *xv_dev, double *zv_dev,
                                   double *yv_dev, double ipm_tol, int32_t ipm_max_iter, int32_t *failures_host, void *stream)
{
    if (!h || !q_dev || !c_dev || !xv_dev || !zv_dev || !yv_dev || !(ipm_tol > 0) || ipm_max_iter < 1) { if (h) h->err = "bad prox argument"; return GCSADMM_ERR_BAD_ARG; }
    if (h->prox_lds_bytes > 160 * 1024) { h->err = "facet count too large for LDS"; return GCSADMM_ERR_UNSUPPORTED; }
    if (h->n_term > 0) { h->err = "the prox kernel (v1 x-update) takes its terminals as points; this graph has a terminal that is a region"; return GCSADMM_ERR_UNSUPPORTED; }
    USE_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(h, hipMemsetAsync(h->d_prox_counters, 0, 2 * sizeof(int), s));
    WgLaunchDesc d{};
    d.n = h->n; d.dtype = GCSADMM_F64; d.n_vtx = h->n_prox; d.lds_bytes = h->prox_lds_bytes; d.vtx = h->d_prox_vtx;
    d.inc_ptr = h->d_inc_ptr; d.deg_in = h->d_deg_in; d.inc_edge = h->d_inc_edge; d.poly_ptr = h->d_poly_ptr;
    d.poly_A = h->d_poly_A; d.poly_bc = h->d_poly_bc; d.center = h->d_center;
    d.xv = xv_dev; d.zv = zv_dev; d.yv = yv_dev; d.counters = h->d_prox_counters; d.ipm_tol = ipm_tol; d.ipm_max_iter = ipm_max_iter;
    gcsadmm_wg_launch_prox(d, q_dev, c_dev, h->src, h->dst, s);
    HIPCHK(h, hipGetLastError());
    if (failures_host) {      // optional: synchronises the stream
        int cnt[2] = {0, 0};
        HIPCHK(h, hipMemcpyAsync(cnt, h->d_prox_counters, sizeof(cnt), hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        *failures_host = cnt[0];
    }
    return GCSADMM_OK;
}

gcsadmm_status gcsadmm_run_timed(gcsadmm_handle h, const gcsadmm_state *st, int32_t k, double *trace_dev, void *stream,
                                 float *vertex_ms, int32_t *vertex_launches, float *edge_ms, int32_t *edge_launches)
{
    if (!state_ok(h, st) || k < 0 || !vertex_ms || !edge_ms || !vertex_launches || !edge_launches) return GCSADMM_ERR_BAD_ARG;
    USE_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    const size_t need = (size_t)4 * k;
    while (h->events.size() < need) {
        hipEvent_t ev;
        HIPCHK(h, hipEventCreate(&ev));
        h->events.push_back(ev);
    }
    for (int i = 0; i < k; ++i) {
        gcsadmm_status r;
        HIPCHK(h, hipEventRecord(h->events[4 * i + 0], s));
        if ((r = gcsadmm_vertex_step(h, st, stream)) != GCSADMM_OK) return r;
        HIPCHK(h, hipEventRecord(h->events[4 * i + 1], s));
        HIPCHK(h, hipEventRecord(h->events[4 * i + 2], s));
        r = h->dtype == GCSADMM_F64 ? launch_edge<double>(h, st, h->d_sums, s, true, trace_dev) : launch_edge<float>(h, st, h->d_sums, s, true, trace_dev);
        if (r != GCSADMM_OK) return r;
        HIPCHK(h, hipEventRecord(h->events[4 * i + 3], s));
    }
    HIPCHK(h, hipStreamSynchronize(s));
    double vm = 0, em = 0;
    for (int i = 0; i < k; ++i) {
        float t = 0;
        HIPCHK(h, hipEventElapsedTime(&t, h->events[4 * i + 0], h->events[4 * i + 1])); vm += t;
        HIPCHK(h, hipEventElapsedTime(&t, h->events[4 * i + 2], h->events[4 * i + 3])); em += t;
    }
    *vertex_ms = (float)vm; *edge_ms = (float)em; *vertex_launches = k; *edge_launches = k;
    return GCSADMM_OK;
}

gcsadmm_status gcsadmm_read_control(gcsadmm_handle h, gcsadmm_control_block *out, void *stream)
{
    if (!h || !out) return GCSADMM_ERR_BAD_ARG;
    USE_DEVICE(h);
    HIPCHK(h, hipMemcpyAsync(out, h->d_cb, sizeof(*out), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
    return GCSADMM_OK;
}

gcsadmm_status gcsadmm_cost(gcsadmm_handle h, const gcsadmm_state *st, double eps_edge, double *cost_dev, void *stream)
{
    if (!h || !st || !st->zv || !st->zedge || !cost_dev) return GCSADMM_ERR_BAD_ARG;
    USE_DEVICE(h);
    if (h->dtype == GCSADMM_F64)
        hipLaunchKernelGGL((cost_kernel<double>), dim3(1), dim3(256), 0, (hipStream_t)stream, h->V, h->E, h->n, st->zv,
                           (const double *)st->zedge, h->d_edge_counted, eps_edge, cost_dev);
    else
        hipLaunchKernelGGL((cost_kernel<float>), dim3(1), dim3(256), 0, (hipStream_t)stream, h->V, h->E, h->n, st->zv,
                           (const float *)st->zedge, h->d_edge_counted, eps_edge, cost_dev);
    HIPCHK(h, hipGetLastError());
    return GCSADMM_OK;
}

#ifdef GCS_PHASE_TIMING
int gcsadmm_debug_sub_cycles(unsigned long long *out16)
{
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_sub_cycles), 16 * sizeof(unsigned long long));
}

int gcsadmm_debug_phase_cycles(unsigned long long *out64)
{
    return (int)hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_phase_cycles), 64 * sizeof(unsigned long long));
}
#endif

gcsadmm_status gcsadmm_query(gcsadmm_handle h, int32_t *num_waves, int32_t *lds_bytes, int32_t *num_special,
                             int32_t *num_workgroup_vertices, int32_t *workgroup_lds_bytes)
{
    if (!h) return GCSADMM_ERR_BAD_ARG;
    if (num_waves) *num_waves = h->n_waves;
    if (lds_bytes) *lds_bytes = h->lds_bytes;
    if (num_special) *num_special = h->n_special;
    if (num_workgroup_vertices) *num_workgroup_vertices = h->n_wg;
    if (workgroup_lds_bytes) *workgroup_lds_bytes = h->wg_lds_bytes;
    return GCSADMM_OK;
}

gcsadmm_status gcsadmm_unit_iterations(gcsadmm_handle h, int32_t *out, int32_t capacity, int32_t *count, void *stream)
{
    if (!h || !count) return GCSADMM_ERR_BAD_ARG;
    USE_DEVICE(h);
    const int *src = h->d_wave_iters ? h->d_wave_iters : h->d_wg_iters;
    const int n = h->d_wave_iters ? h->n_waves : (h->d_wg_iters ? h->n_wg : 0);
    *count = n;
    if (n == 0 || !out) return GCSADMM_OK;
    HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
    HIPCHK(h, hipMemcpy(out, src, sizeof(int) * (size_t)(n < capacity ? n : capacity), hipMemcpyDeviceToHost));
    return GCSADMM_OK;
}

} // extern "C"
