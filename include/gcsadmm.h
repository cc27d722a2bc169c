/*
 * gcsadmm.h -- C ABI of the MI355X (gfx950) implementation of the per-iteration
 * loop of the reference's "full vertex split" ADMM solver for shortest paths in
 * graphs of convex sets.
 *
 * The reference (Python) has no FFI seam; the call sites this library replaces
 * are, in /root/reference/admm_solver_v3.py:
 *     :469-540  parallel_vertex_update()  -- SolveInParallel(progs, MOSEK) + scatter   -> gcsadmm_vertex_step
 *     :543-587  parallel_edge_update()    -- per-edge average of the two vertex copies \
 *     :590-594  dual_update()             -- mu += A x + B z - c                        > gcsadmm_edge_step
 *     :597-614  residuals, eps_pri/dual   -- the five norms                            /
 *     :655-733  the while loop            -- order, rho adaptation, stop test          -> gcsadmm_control / gcsadmm_run
 * and GCS_utils.py:184-211 compute_cost -> gcsadmm_cost.
 *
 * Conventions
 *   - plain C, no exceptions, no exit(): every entry point returns a gcsadmm_status code;
 *     gcsadmm_last_error() gives the text of the last failure on that handle.
 *   - graph description pointers are HOST pointers (copied at create); state, trace and
 *     scalar buffers are DEVICE pointers owned by the caller (e.g. PyTorch-ROCm tensors).
 *   - one handle <-> one device <-> one stream at a time; calls on a handle are not re-entrant.
 *     `stream` is a hipStream_t passed as void* (NULL = default stream).  All entry points
 *     that take a stream only enqueue work; they never synchronise.
 *
 * State layout (shared with the CPU oracle): for every directed edge e=(u,w) the
 * coupled words are [ z_{e,u}[0:n], z_{e,w}[0:n], y_e ]  (c = 2n+1 words).
 *   copy [c][NI]   vertex copies, incidence-major (word-major rows of NI = incidences)
 *   mu   [c][NI]   scaled duals of the consensus rows, same indexing
 *   zedge[c][E]    edge copies
 *   xv [V][2n], zv [V][2n], yv [V]   per-vertex outputs of the last vertex step
 * Element type of copy/mu/zedge is f64 or f32 (desc.state_dtype); xv/zv/yv are f64.
 * All interior-point arithmetic is f64 regardless.
 */
#ifndef GCSADMM_H
#define GCSADMM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gcsadmm_status {
    GCSADMM_OK = 0,
    GCSADMM_ERR_BAD_ARG = 1,        /* null pointer, negative size, inconsistent CSR, unsupported n */
    GCSADMM_ERR_UNSUPPORTED = 2,    /* a vertex whose sub-problem does not fit the CU's LDS; 's' / 't' that are not points */
    GCSADMM_ERR_HIP = 3,            /* a HIP runtime call failed */
    GCSADMM_ERR_NO_DEVICE = 4
} gcsadmm_status;

enum { GCSADMM_F64 = 0, GCSADMM_F32 = 1 };

/* loop exit reasons written to gcsadmm_control_block.status */
enum { GCSADMM_RUNNING = -1, GCSADMM_CONVERGED = 0, GCSADMM_MAX_IT = 1, GCSADMM_DIVERGED = 2 };

typedef struct gcsadmm_graph_desc {
    int32_t n;                       /* space dimension: 1 .. 8 */
    int32_t num_vertices;            /* vertices whose sub-problem this handle solves */
    int32_t num_edges;               /* directed edges this handle updates */
    int32_t num_incidences;          /* NI: columns of copy/mu; >= inc_ptr[num_vertices]; the surplus are
                                        ghost slots filled by the caller (halo of a vertex partition) */
    const int32_t *inc_ptr;          /* [V+1] CSR over vertices; incoming edges first, then outgoing */
    const int32_t *inc_edge;         /* [inc_ptr[V]] directed edge id of each incidence */
    const int32_t *inc_out;          /* [inc_ptr[V]] 1 if the vertex is the tail of that edge */
    const int32_t *edge_inc_tail;    /* [E] incidence slot holding the tail's copy of edge e */
    const int32_t *edge_inc_head;    /* [E] incidence slot holding the head's copy */
    const int32_t *poly_ptr;         /* [V+1] CSR over facets */
    const double *poly_A;            /* [poly_ptr[V]][n] facet normals, A x <= b */
    const double *poly_b;            /* [poly_ptr[V]] */
    const double *center;            /* [V][n] a strictly interior point of each polytope */
    int32_t src, dst;                /* local vertex ids of 's' and 't' (-1 if not in this partition).  A terminal whose polytope lies within
                                        1e-5 of its `center` is a point (utils.py:12-28; closed form); one with an extent is a region and is
                                        constrained like any set (admm_solver_v3.py:415-464): own kernel, needs an edge on its live side */
    int32_t state_dtype;             /* GCSADMM_F64 / GCSADMM_F32 */
    int32_t device;                  /* HIP device ordinal */
    const uint8_t *inc_counted;      /* [NI] or NULL(=all 1): this handle owns the copy (counts it in the norms) */
    const uint8_t *edge_counted;     /* [E]  or NULL(=all 1): this handle counts the edge in the norms */
    double nx_global, nmu_global;    /* lengths of the reference's x / mu vectors for eps_pri / eps_dual
                                        (admm_solver_v3.py:605-614); 0 = derive from this handle's sizes */
    /* Schedule of the vertex step (all 0 = automatic).  Two programs solve the same sub-problem with the same
     * interior-point method: the WAVEFRONT program (n = 2, degree <= 63; several vertices per 64-lane wavefront,
     * highest throughput on large graphs) and the WORKGROUP program (any n, degree and facet count that fits LDS;
     * one 256-thread workgroup per vertex, lowest latency: small graphs).  These fields replace what used to be
     * process-global environment knobs; they never change WHAT is computed, only how it is laid out on the chip. */
    int32_t vertex_program;          /* 0 auto, 1 wavefront (where it applies), 2 workgroup, 3 workgroup with 256 threads per workgroup even
                                        where the automatic choice is 512 (launches of at most one workgroup per CU) */
    int32_t wave_slots;              /* wavefront program: vertices per wavefront (0 auto) */
    int32_t wave_align;              /* wavefront program: 0 auto, 1 row-aligned groups, 2 dense packing */
    int32_t wave_store_dl;           /* wavefront program: 0 auto, 1 keep the facet-row dual directions in LDS, 2 recompute */
    int32_t wave_generic_rows;       /* 1 (or 2) = generic facet rows everywhere: the any-facet-count instantiation of the wavefront
                                        program and the generic instantiation of the workgroup program even when every polytope is a
                                        canonical axis-aligned box (tuning / tests) */
    /* Numbering of the state columns (copy / mu are [c][num_incidences]).  0: INCIDENCE-major, column k = position k of the
     * vertex CSR (a vertex's columns are contiguous; edge_inc_tail / edge_inc_head give the two columns of an edge; ghost
     * columns follow the owned ones).  1: EDGE-major, the tail-side column of edge e is e and the head-side column is
     * num_edges + e (num_incidences must be 2 num_edges; edge_inc_tail / edge_inc_head must say exactly that; on a partition
     * the remote side of a cut edge is the ghost).  Edge-major makes the edge step a pure stream (measured 2-3x less HBM
     * traffic) and turns the vertex step's column accesses into gathers, which that latency-bound step does not feel:
     * the layout for large graphs.  Same numbers either way. */
    int32_t edge_major_columns;
} gcsadmm_graph_desc;

typedef struct gcsadmm_params {
    double rho;          /* initial penalty (1)                     admm_solver_v3.py:621 */
    double tau_incr;     /* 2                                        :641 */
    double tau_decr;     /* 2                                        :642 */
    double nu;           /* 10                                       :643 */
    int32_t it_rho_limit;/* rho adapts while it < this (0.1*1000)   :644,703 */
    int32_t max_it;      /* 1000                                     :651 */
    double eps_abs;      /* 1e-4                                     :647 */
    double eps_rel;      /* 1e-3                                     :648 */
    double eps_edge;     /* 1e-4 edge activation penalty             :388 */
    double ipm_tol;      /* barrier parameter at which a vertex solve stops (3e-9: the host default, gcs_admm_amd.IPM_TOL) */
    int32_t ipm_max_iter;/* 60 */
    int32_t cold_start;  /* 0 (default): a vertex solve restarts from the record its previous solve left in the handle's workspace
                            (csrc/warm_start.h; MOSEK at admm_solver_v3.py:490 starts cold -- the minimiser is the same);
                            1: every solve starts from the fixed interior point */
} gcsadmm_params;

typedef struct gcsadmm_state {
    void *copy;   /* [c][NI] */
    void *mu;     /* [c][NI] */
    void *zedge;  /* [c][E]  */
    double *xv;   /* [V][2n] */
    double *zv;   /* [V][2n] */
    double *yv;   /* [V]     */
} gcsadmm_state;

/* Loop state kept on the device so that iterations can be enqueued back to back
 * without host round trips.  Lives in device memory owned by the handle;
 * gcsadmm_read_control copies it out (synchronising the given stream). */
typedef struct gcsadmm_control_block {
    double rho;           /* penalty used by the next vertex step */
    double mu_scale;      /* pending rescale of mu (rho change), applied lazily by the next steps */
    double sums[5];       /* |r|^2, |dz|^2, |copy|^2, |zedge|^2, |mu|^2 of the last edge step */
    double pri, dual, eps_pri, eps_dual;
    int32_t it;           /* iteration counter as the reference reports it (starts at 1) */
    int32_t status;       /* GCSADMM_RUNNING / CONVERGED / MAX_IT / DIVERGED */
    int32_t inner_failures; /* vertex solves of the last step that hit ipm_max_iter */
    int32_t inner_iters;    /* interior-point iterations of the last step, summed over vertices */
} gcsadmm_control_block;

typedef struct gcsadmm_handle_s *gcsadmm_handle;

gcsadmm_status gcsadmm_create(const gcsadmm_graph_desc *desc, gcsadmm_handle *out);
void gcsadmm_destroy(gcsadmm_handle h);
const char *gcsadmm_last_error(gcsadmm_handle h);   /* h may be NULL: error of the last failed create */

/* (Re)start the loop: control block := {rho, mu_scale 1, it 1, RUNNING}.  Does not touch the state. */
gcsadmm_status gcsadmm_reset(gcsadmm_handle h, const gcsadmm_params *p, void *stream);

/* x-update: one convex sub-problem per vertex, all vertices of the handle.  Reads zedge, mu and
 * the control block's rho / mu_scale; writes copy (owned incidences only), xv, zv, yv. */
gcsadmm_status gcsadmm_vertex_step(gcsadmm_handle h, const gcsadmm_state *st, void *stream);

/* z-update + dual update + the five partial norms over this handle's edges.  Reads copy (all NI
 * columns: ghost slots must have been filled), updates zedge and mu in place, applies and clears
 * the pending mu rescale, writes sums[5] (f64, device) -- the caller all-reduces them across
 * partitions before gcsadmm_control when the graph is partitioned. */
gcsadmm_status gcsadmm_edge_step(gcsadmm_handle h, const gcsadmm_state *st, double *sums_dev, void *stream);

/* residuals, rho adaptation, stop test, it += 1 (admm_solver_v3.py:697-733) from sums_dev[5];
 * appends {rho, pri, dual, eps_pri, eps_dual, inner_failures} to trace_dev[(it-1)*6 ..] if non-NULL. */
gcsadmm_status gcsadmm_control(gcsadmm_handle h, const double *sums_dev, double *trace_dev, void *stream);

/* Enqueue up to k full iterations (vertex, edge, control) with no host synchronisation; once the
 * stop test fires the remaining enqueued kernels exit immediately. */
gcsadmm_status gcsadmm_run(gcsadmm_handle h, const gcsadmm_state *st, int32_t k, double *trace_dev, void *stream);

/* Copy the control block to the host (synchronises `stream`). */
gcsadmm_status gcsadmm_read_control(gcsadmm_handle h, gcsadmm_control_block *out, void *stream);

/* sum_v |z_v[:n] - z_v[n:]| + eps_edge * sum_e y_e over counted edges -> cost_dev[0] (f64, device). */
gcsadmm_status gcsadmm_cost(gcsadmm_handle h, const gcsadmm_state *st, double eps_edge, double *cost_dev, void *stream);

/* Kernel-side facts for benchmarking (any pointer may be NULL): workgroups of the wavefront program and their LDS
 * bytes, closed-form vertices, workgroups (= vertices) of the workgroup program and their LDS bytes. */
gcsadmm_status gcsadmm_query(gcsadmm_handle h, int32_t *num_waves, int32_t *lds_bytes, int32_t *num_special,
                             int32_t *num_workgroup_vertices, int32_t *workgroup_lds_bytes);

/* Diagnostics: Newton iterations of the last vertex step per dispatch unit (wavefronts of the wavefront program: the slowest vertex
 * of each; otherwise workgroups = vertices of the workgroup program), for handles large enough to keep them (>= 512 units: the
 * slowest-first dispatch sorts by them; *count = 0 otherwise).  Synchronises `stream`; copies min(*count, capacity) entries. */
gcsadmm_status gcsadmm_unit_iterations(gcsadmm_handle h, int32_t *out, int32_t capacity, int32_t *count, void *stream);

/* As gcsadmm_run, but every kernel launch is bracketed by HIP events recorded on `stream`; after the
 * k iterations the call synchronises and returns the summed device time (ms) and launch count of the
 * vertex-step kernel(s) and of the edge-step kernel.  For measurement (bench.py roofline). */
gcsadmm_status gcsadmm_run_timed(gcsadmm_handle h, const gcsadmm_state *st, int32_t k, double *trace_dev,
                                 void *stream, float *vertex_ms, int32_t *vertex_launches,
                                 float *edge_ms, int32_t *edge_launches);

/* ---------------------------------------------------------------------------------------------
 * The x-update of the reference's OTHER splittings: admm_solver_v1.py:334-383 (shared by v2) solves,
 * per vertex, the border-only problem -- no edge blocks -- whose consensus penalty (:350-367) is a separable quadratic in the
 * vertex's own unknowns:
 *     min |z_v[:n] - z_v[n:]| + 1/2 sum_k q_k (u_k - c_k)^2,  u = (x_v [2n], z_v [2n], y_v),
 *     s.t. A z_i <= y_v b,  A (x_i - z_i) <= (1 - y_v) b  (:371-381),  0 <= y_v <= 1  (:346).
 * Same interior-point method and kernel as the v3 vertex step (workgroup program, "prox" configuration).  q_dev, c_dev:
 * [V][4n+1] f64 device arrays (weights >= 0, not all zero per vertex; centres); outputs [V][2n], [V][2n], [V] f64 device arrays.
 * The two terminals are points (closed form; a graph with a terminal that is a region is refused here).  failures_host (may be NULL; non-NULL synchronises the stream) receives the
 * number of inner solves that did not converge (their outputs are left untouched).  The edge side of those splittings -- one
 * monolithic conic program over all edges (v1) or a sequential sweep (v2) -- is out of scope.
 */
gcsadmm_status gcsadmm_vertex_prox(gcsadmm_handle h, const double *q_dev, const double *c_dev, double *xv_dev, double *zv_dev,
                                   double *yv_dev, double ipm_tol, int32_t ipm_max_iter, int32_t *failures_host, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Vertex partitions across GPUs.  The reference's fan-out point is admm_solver_v3.py:490
 * (SolveInParallel over all vertices); here each rank holds one handle built from its part of the graph (ghost incidence
 * columns for the remote endpoint of every cut edge, ownership masks, global nx / nmu: see gcsadmm_graph_desc) and the ranks
 * meet twice per iteration: the halo exchange of the cut edges' copies and one all-reduce of six doubles.  Both run on the
 * caller's stream through RCCL (bound at run time with dlopen: the library has no link dependency on it).
 */
typedef struct gcsadmm_halo_desc {
    int32_t num_peers;               /* neighbour partitions (<= 2 for strip partitions) */
    const int32_t *peer_rank;        /* [num_peers] */
    const int32_t *send_ptr;         /* [num_peers+1] CSR into send_cols */
    const int32_t *send_cols;        /* owned incidence columns whose copies go to that peer, in (global edge, side) order */
    const int32_t *recv_ptr;         /* [num_peers+1] CSR into recv_cols; recv_ptr == send_ptr entry by entry */
    const int32_t *recv_cols;        /* ghost columns filled from that peer, same canonical order */
} gcsadmm_halo_desc;

/* 128 bytes identifying a new communicator (ncclGetUniqueId); rank 0 calls it, the host distributes the bytes (any transport). */
gcsadmm_status gcsadmm_comm_unique_id(void *id128);

/* Attach rank `rank` of `world` to the communicator `id128` (collective: every rank calls it; blocks until all have) and
 * upload the halo lists.  id128 == NULL: no communicator is created -- for world == 1 the exchange and the all-reduce are
 * no-ops; for world > 1 the host moves the packed halo itself (gcsadmm_halo_pack / _buffers / _unpack) and drives the steps. */
gcsadmm_status gcsadmm_attach_comm(gcsadmm_handle h, int32_t rank, int32_t world, const void *id128, const gcsadmm_halo_desc *halo);
/* The checks gcsadmm_attach_comm makes on the halo lists, alone: host only, no allocation, no collective.  Ranks agree on the outcome
 * (e.g. one all-reduce of an ok flag in the host's own transport) BEFORE any of them enters gcsadmm_attach_comm, whose
 * ncclCommInitRank is collective: a rank that returned an error there would leave its peers waiting. */
gcsadmm_status gcsadmm_check_halo(gcsadmm_handle h, int32_t rank, int32_t world, const gcsadmm_halo_desc *halo);

/* Enqueue up to k full iterations of the partitioned loop (vertex step, halo exchange, edge step, all-reduce, control) with
 * no host synchronisation; every rank must enqueue the same k.  trace_dev as for gcsadmm_run (identical on every rank). */
gcsadmm_status gcsadmm_run_partitioned(gcsadmm_handle h, const gcsadmm_state *st, int32_t k, double *trace_dev, void *stream);

/* As gcsadmm_run_partitioned with every stage bracketed by HIP events on `stream` (collective: every rank calls it with the same k);
 * synchronises and returns the summed device time (ms) of the vertex step, the halo exchange (pack, RCCL send/recv, unpack), the
 * edge step, and the all-reduce + control step.  For measurement (bench.py at N > 1). */
gcsadmm_status gcsadmm_run_partitioned_timed(gcsadmm_handle h, const gcsadmm_state *st, int32_t k, double *trace_dev, void *stream,
                                             float *vertex_ms, float *halo_ms, float *edge_ms, float *reduce_ms);

/* Schedule of gcsadmm_run_partitioned (never its numbers).  The overlapped form ("boundary vertices first") solves the
 * wavefronts that hold a vertex with a cut edge first and on a second stream, with the pack, exchange and unpack of the halo behind them,
 * WHILE the interior wavefronts are solved on the caller's stream; the edge step waits for both.  It exists for handles whose generic vertices all run the
 * wavefront program (the strips of a large n = 2 graph).  mode 0 (default, set at attach): overlapped when the partition has neighbours;
 * 1: overlapped even without neighbours (the first quarter of the wavefronts plays the boundary: tests of the schedule on one GPU);
 * 2: the serial form (vertex step, exchange, edge step on one stream).  boundary_units (may be NULL): wavefronts of the boundary part
 * (0: no split -- the serial form runs).  Call after gcsadmm_attach_comm. */
gcsadmm_status gcsadmm_set_overlap(gcsadmm_handle h, int32_t mode, int32_t *boundary_units);

/* Ranks of the attached RCCL communicator as RCCL itself reports them (ncclCommCount); 0 when none is attached. */
gcsadmm_status gcsadmm_comm_count(gcsadmm_handle h, int32_t *count);

/* The pieces of the halo exchange, for hosts that bring their own transport and for tests: pack the copies to send into the
 * send buffer ([c][columns of peer] per peer, in peer order) / scatter the receive buffer into the ghost columns / both with
 * the RCCL transfer in between / the two device buffers and their length in elements of the state type. */
gcsadmm_status gcsadmm_halo_pack(gcsadmm_handle h, const gcsadmm_state *st, void *stream);
gcsadmm_status gcsadmm_halo_unpack(gcsadmm_handle h, const gcsadmm_state *st, void *stream);
gcsadmm_status gcsadmm_halo_exchange(gcsadmm_handle h, const gcsadmm_state *st, void *stream);
gcsadmm_status gcsadmm_halo_buffers(gcsadmm_handle h, void **send_buf, void **recv_buf, int64_t *num_elements);

/* ---------------------------------------------------------------------------------------------
 * Graph construction at scale.  The reference decides every ordered pair of
 * regions with one LP feasibility solve through Drake/MOSEK (utils.py:31-82 build_graph, :49-65
 * check_overlap); these entry points run the same decisions as batches of tiny LPs on the device, one LP
 * per lane (polytope_lp.hip).  All pointers are HOST pointers (set-up code, called once per scene); the
 * polytope CSR is the one of gcsadmm_graph_desc (rows of region p: poly_ptr[p] .. poly_ptr[p+1]).
 * n = 1..6.  Return value: gcsadmm_status; text of the last failure: gcsadmm_polytope_last_error().
 * Optional `status` arrays receive the LP status per problem: 0 converged, 1 / 2 decided early
 * (overlap / separation proven), -1 iteration limit.
 */
const char *gcsadmm_polytope_last_error(void);

/* Chebyshev centre (centre of the largest inscribed ball) and its radius for every region: the strictly
 * interior point the vertex kernel centres a sub-problem on (graph.py chebyshev_center on the host).
 * centers[P][n], radii[P] (may be NULL; capped at 1e6 for unbounded sets; <= 0: no interior). */
int gcsadmm_polytope_centers(int n, int num_polytopes, const int *poly_ptr, const double *poly_A, const double *poly_b,
                             int device, double *centers, double *radii, int *status);

/* Axis-aligned bounding box of every region (2n LPs each, started from `centers`); lo[P][n], hi[P][n];
 * unbounded directions come back near +-1e8. */
int gcsadmm_polytope_bounds(int n, int num_polytopes, const int *poly_ptr, const double *poly_A, const double *poly_b,
                            const double *centers, int device, double *lo, double *hi, int *status);

/* overlap[t] = 1 iff regions pair_a[t] and pair_b[t] share a point (closed sets; inscribed radius of the
 * intersection >= -tol), the decision of utils.py:49-65.  `centers` (may be NULL) only supplies start points. */
int gcsadmm_polytope_overlaps(int n, int num_polytopes, const int *poly_ptr, const double *poly_A, const double *poly_b,
                              const double *centers, long num_pairs, const int *pair_a, const int *pair_b, double tol,
                              int device, unsigned char *overlap, int *status);

#ifdef __cplusplus
}
#endif
#endif /* GCSADMM_H */
