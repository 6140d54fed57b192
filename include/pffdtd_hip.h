/*
 * pffdtd_hip.h -- C ABI of the MI355X-native FDTD time-step engine (libpffdtd_hip.so).
 *
 * This is the drop-in boundary for the reference's in-process engine seam
 *
 *     double run_sim(struct SimData *sd);          c_cuda/cpu_engine.h:52, c_cuda/gpu_engine.h:665
 *
 * called from c_cuda/fdtd_main.c:44-53 (load_sim_data -> scale_input -> run_sim ->
 * rescale_output -> write_outputs).  `pf_simdata` is a field-for-field POD mirror of
 * `struct SimData` (c_cuda/fdtd_data.h:38-76); the only additions are `real_bytes`
 * (the reference selects `Real` at compile time with -DPRECISION, c_cuda/fdtd_common.h:44-71;
 * this library carries both precisions and selects at run time) and the widened copies of the
 * `Real` scalars.  Everything is plain pointers and sizes: no C++ or torch types cross this line.
 *
 * All array pointers in pf_simdata are HOST pointers owned by the caller (as in the reference,
 * where load_sim_data mallocs and free_sim_data frees, fdtd_data.h:99,721).  The engine owns
 * only its device state.  Linear indices are in the file layout  ii = ix*Ny*Nz + iy*Nz + iz
 * (cpu_engine.h:180); the engine re-bases them onto its own padded HBM layout internally.
 *
 * Error behaviour: the reference asserts/exits (helper_funcs.h:86-94, gpu_engine.h:192-200);
 * this ABI returns a non-zero pf_status and keeps a message retrievable with pf_last_error().
 */
#ifndef PFFDTD_HIP_H
#define PFFDTD_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PF_MMB 12 /* max RLC branches per material: MMb, fdtd_data.h:33 */
#define PF_MNM 64 /* max number of materials:      MNm, fdtd_data.h:35 */

/* struct MatQuad (fdtd_data.h:79-84) for Real=float / Real=double */
typedef struct pf_matquad_f32 { float  b, bd, bDh, bFh; } pf_matquad_f32;
typedef struct pf_matquad_f64 { double b, bd, bDh, bFh; } pf_matquad_f64;

/* Mirror of struct SimData (fdtd_data.h:38-76), same field order. */
typedef struct pf_simdata {
   int64_t  *bn_ixyz;     /* [Nb]  boundary node indices */
   int64_t  *bnl_ixyz;    /* [Nbl] lossy boundary node indices */
   int64_t  *bna_ixyz;    /* [Nba] absorbing (ABC) node indices */
   int8_t   *Q_bna;       /* [Nba] 1 face, 2 edge, 3 corner */
   int64_t  *in_ixyz;     /* [Ns]  source nodes */
   int64_t  *out_ixyz;    /* [Nr]  receiver nodes (duplicates allowed) */
   int64_t  *out_reorder; /* [Nr]  row permutation used by write_outputs (fdtd_data.h:941-945) */
   uint16_t *adj_bn;      /* [Nb]  adjacency bits, bit j = neighbour j (fdtd_data.h:532-538) */
   void     *ssaf_bnl;    /* [Nbl] Real: scaled surface-area factors (fdtd_data.h:283-289,608) */
   uint8_t  *bn_mask;     /* [(Npts-1)/8+1] bit ii%8 of byte ii>>3 (fdtd_data.h:567-572); may be NULL:
                             the engine rebuilds its own mask from bn_ixyz (as gpu_engine.h:791 does) */
   int8_t   *mat_bnl;     /* [Nbl] material index of lossy nodes */
   int8_t   *K_bn;        /* [Nb]  popcount(adj_bn); may be NULL (cpu_engine.h:238-240 recomputes it) */
   double   *in_sigs;     /* [Ns*Nt] input signals, row-major (already scaled by scale_input) */
   double   *u_out;       /* [Nr*Nt] receiver outputs, engine row order, written by the engine */
   int64_t   Ns, Nr, Nt, Npts, Nx, Ny, Nz, Nb, Nbl, Nba;
   double    l, l2;
   int8_t    fcc_flag;    /* 0 Cartesian, 1 FCC checkerboard, 2 FCC folded */
   int8_t    NN;          /* 6 | 12 */
   int8_t    Nm;          /* number of materials */
   int8_t   *Mb;          /* [Nm] branches per material */
   void     *mat_quads;   /* [Nm*PF_MMB] pf_matquad_f32 | pf_matquad_f64 */
   void     *mat_beta;    /* [Nm] Real */
   double    infac;       /* input rescaling (scale_input, fdtd_data.h:879-909); not used by the engine */
   double    sl2, lo2, a2, a1; /* the Real-rounded coefficients of fdtd_data.h:186-194, widened exactly */
   int32_t   real_bytes;  /* 4 = float (PRECISION=1), 8 = double (PRECISION=2) */
} pf_simdata;

typedef enum pf_status {
   PF_OK = 0,
   PF_ERR_ARG = 1,      /* bad argument / unsupported configuration */
   PF_ERR_HIP = 2,      /* a HIP runtime call failed */
   PF_ERR_NODEV = 3,    /* no HIP device visible */
   PF_ERR_STATE = 4     /* call sequence violated */
} pf_status;

/* numerics modes */
#define PF_NUM_CPU_EXACT       0 /* operation order and rounding of the reference C CPU engine (cpu_engine.h:174-301,363-405), no
                                    contraction: results are bit-identical to it (the parity target) */
#define PF_NUM_GPU_SAFEGUARDED 2 /* the reference GPU engine's arithmetic for the air and rigid-node updates (fdtd_common.h:44-71,
                                    gpu_engine.h:220-274,288-348): pairwise neighbour sums -- in fp32 rounded TOWARDS ZERO, its
                                    long-run stability safeguard -- then two round-to-nearest FMAs; every kernel family has it: single steps,
                                    7-point blocked pairs with their wall regions, 13-point blocked pairs (round 5) */

typedef struct pf_opts {
   int32_t device;        /* HIP device ordinal */
   int32_t numerics;      /* PF_NUM_* */
   int32_t slab_first;    /* 1 if this grid holds the global ix=0 ghost plane (gpu_engine.h:1030-1032) */
   int32_t slab_last;     /* 1 if this grid holds the global ix=Nx-1 ghost plane (gpu_engine.h:1033-1035) */
   int32_t readout_chunk; /* receiver ring depth in steps before a D2H flush (0 = default) */
   int32_t air_variant;   /* 0 = automatic (measured at creation); 3 = the reference's kernel sequence (flips, air, ABC lists); 4 / 7 =
                             barrier-free marching kernel with virtual ghosts / in-kernel ABC; 25 = lean fused kernel (7-point); 40 =
                             temporally blocked pairs forced (41: their driver only); | 256 = separate rigid / branch-ODE kernels */
   int32_t air_chunk;     /* planes marched per workgroup (0 = auto, <0 = that many equal chunks) */
   int32_t timing;        /* 1 = bracket the air kernel with HIP events every step (pf_engine_timing) */
   void   *ext_u0;        /* optional caller-owned DEVICE buffers for the two state grids, each of */
   void   *ext_u1;        /*   pf_grid_bytes() bytes, zero-filled by the caller; NULL = engine allocates */
   int32_t x_global0;     /* global ix of this grid's plane 0 (slabs): only its parity matters, for the FCC
                             checkerboard form (fcc_flag 1) whose existing nodes have even ix+iy+iz */
   int32_t layout;        /* PF_LAYOUT_*: how the engine stores the grid (pf_engine_layout reports it) */
   int32_t energy;        /* 1 = keep what the energy diagnostic needs (explicit Laplacian grid, unfused kernel
                             sequence); then use pf_engine_energy_cfg + pf_engine_run_energy */
   int32_t multi_flags;   /* pf_run_sim_devices / pf_multi_create only: PF_MULTI_* */
   int32_t transport;     /* pf_run_sim_devices / pf_multi_create only: PF_TRANSPORT_* (how ghost planes travel between devices) */
   int32_t verify_exchange; /* same: the first n exchanges are checksummed on both sides (pf_multi_info.exchange_verified) */
   int32_t only_slab;     /* pf_multi_create only: 1 + g = cost model of ONE rank -- the chain is cut as usual but slab g alone is
                             instantiated and receives its own edge planes as ghost planes, through the chosen transport (the
                             physics is wrong, the work and the launches are a rank's); 0 = the whole chain */
   double  wall_scale;    /* pf_multi_create / pf_run_sim_devices: > 0 = cut the chain with the wall planes' weights times this factor (a host that
                             measured it once -- pf_slab_wall_scale -- and wants the same cut in every process); 0 = the compiled-in weights, or the
                             library's own measurement under PF_MULTI_MEASURE_WEIGHTS */
} pf_opts;

#define PF_LAYOUT_AUTO      0 /* decided per scene: rooms whose large surfaces are normal to file z are stored with the file's x and z axes
                               exchanged (single domains: +11-15 % on the two reference rooms; chains: PF_MULTI_CUT_Z) */
#define PF_LAYOUT_EXCHANGED 1 /* always exchanged (a slab of a chain cut along file z: pffdtd_amd/dist.py; no energy diagnostic) */
#define PF_LAYOUT_FILE      2 /* never: the file's own order (x slowest, z unit stride) */

#define PF_TRANSPORT_AUTO 0 /* peer copies where hipDeviceCanAccessPeer says yes for every neighbouring pair, else RCCL, else -- librccl
                               missing, its communicators failing or not returning within PFFDTD_RCCL_INIT_TIMEOUT_S (60) seconds --
                               host-staged copies (pf_multi_info.transport / transport_note say which and why) */
#define PF_TRANSPORT_PEER 1 /* each slab pulls its ghost planes with hipMemcpyPeerAsync on its edge stream (gpu_engine.h:1086-1126
                               uses cudaMemcpyPeerAsync after a full sync); an error without peer access when asked for explicitly */
#define PF_TRANSPORT_RCCL 2 /* ncclSend / ncclRecv of both planes, grouped per slab on its edge stream, one single-process
                               communicator clique over the chain (librccl is loaded at run time); environment PFFDTD_TRANSPORT=
                               peer|rccl|host|auto overrides pf_opts.transport */
#define PF_TRANSPORT_HOST 3 /* the last resort: every slab copies its two edge planes into a pinned bounce buffer on its edge stream and its
                               neighbours copy them out on theirs; the two sides meet on the host (hipEventSynchronize).  Needs nothing
                               from the driver beyond device <-> pinned-host copies.  No counterpart in the reference */

#define PF_MULTI_EVEN_SPLIT  1 /* the reference's Nx/G planes per slab (gpu_engine.h:532-550) instead of the cost-balanced cut */
#define PF_MULTI_ONE_THREAD  2 /* one host thread drives every slab (the reference's arrangement) instead of one thread per slab */
#define PF_MULTI_NO_PAIRS    4 /* never step slabs in temporally blocked pairs */
#define PF_MULTI_FORCE_PAIRS 8 /* ask every slab engine for pairs regardless of its thickness (tests) */
#define PF_MULTI_CUT_Z      16 /* cut the chain along FILE Z instead of x: the slab engines store the grid with the x and z axes
                                  exchanged (pf_engine_layout), what rooms whose large surfaces are normal to z gain 11-15 % from;
                                  chosen automatically for such rooms */
#define PF_MULTI_CUT_X      32 /* never (the reference's arrangement, gpu_engine.h:516-662) */
#define PF_MULTI_MEASURE_WEIGHTS 128 /* measure the wall planes' weights on the scene when the chain is created (pf_slab_wall_scale) instead of
                                  cutting with the compiled-in ones (23 / 5 interior planes per full plane of lossy / rigid nodes).  Opt-in:
                                  on MI355X the measurement reproduces the compiled-in figures where it is clean (factors 0.99-1.05,
                                  profiles/r05_partition_weights.txt) and its own noise -- two engines of one geometry differ by 2-3 %
                                  with the luck of their grid placement -- is larger than their error */
#define PF_MULTI_NO_TRIPLES 64 /* slabs step in pairs at most (round 5: by default a slab whose wall regions fit steps THREE steps per
                                  pass across three split-phase steps, pf_engine_place_grids5) */

typedef struct pf_timing {
   double  air_ms_total;    /* sum of HIP-event durations of the air kernel launches */
   int64_t air_launches;
   double  step_ms_total;   /* sum of HIP-event durations of whole steps (pre .. readout) */
   int64_t steps;
   double  tb2_ms_total;    /* temporal blocking: sum of the two-steps-per-pass kernel's launch durations (included in air_ms_total) */
   int64_t tb2_launches;    /* kernel launches behind tb2_ms_total; 0 when the engine steps one step per pass */
   int64_t tb2_cells;       /* cells one such launch advances by two steps (average over the x ranges of the box) */
   double  tune_ms[3];      /* creation-time measurement of the interior update of one step, ms: lean fused kernel,
                               barrier-free kernel, temporally blocked pair / 2 (0 = not measured) */
   int64_t air_path;        /* what the engine runs: 0 lean, 1 barrier-free (virtual ghosts), 2 blocked pairs, -1 other */
   int64_t tb2_lw;          /* blocked pairs: lanes per row segment of the two-steps-per-pass kernel (64 | 32 | 16), else 0 */
   int64_t tb2_dirty_tiles; /* blocked pairs: tiles of the box that step singly (geometry or a source inside) */
   int64_t place_candidates;/* blocked pairs: grid placements timed at creation (0: none), and the two-steps-per-pass kernel's */
   double  place_ms[3];     /*   ms per launch on the first (as allocated), the chosen (fastest) and the slowest of them */
   int64_t wall_blocks[2];  /* blocked pairs with the shell in pairs too (wall regions, pf_wall.h): blocks of the launches whose pencils
                               are all alike / generic blocks (edges, corners); 0, 0: the shell takes single steps */
   int64_t tb_steps_per_pass; /* steps one launch behind tb2_ms_total advances its cells by: 2 (pairs), 3 (k_tb3, pf_tb3.h), 0: none */
   int64_t wall_bricks;     /* wall regions: bricks of the frame (edges and corners of the shell, stepped in LDS: pf_brick.h); 0: generic blocks */
   int64_t wall_three_steps;/* triples: which wall regions take all three steps in ONE pass (k_wall2<..., NS = 3>): bit 0 the x / y regions, bit 3 the
                               column strips; 0: two steps + one.  Slabs of a chain: bits 4 / 5 -- the slab's low / high x side is the grid's own wall
                               and a region's and the bricks' too (no single steps of its planes) */
} pf_timing;

typedef struct pf_engine pf_engine;

/* ---- library-level ---- */
const char *pf_last_error(void);
const char *pf_version(void);
int         pf_device_count(void);
/* bytes of one state grid in the engine's padded HBM layout, and its z pitch in elements */
size_t      pf_grid_bytes(int64_t Nx, int64_t Ny, int64_t Nz, int32_t real_bytes);
int64_t     pf_grid_pitch(int64_t Nz, int32_t real_bytes);
void        pf_opts_default(pf_opts *o);

/* ---- the reference seam: double run_sim(struct SimData*) ---- */
/* Runs all sd->Nt steps, fills sd->u_out (engine row order), returns elapsed seconds (<0 on error; pf_last_error()).
 * Like the reference's GPU engine (gpu_engine.h:680-682) it spreads the grid over EVERY visible device as a chain of
 * Z-slabs along Nx (environment: PFFDTD_NGPUS=n limits it to the first n devices, PFFDTD_DEVICES=0,1,... names the chain). */
double      pf_run_sim(pf_simdata *sd);
/* The same on an explicit chain: slab g on device devices[g]; an id may repeat ("virtual slabs": the whole multi-device
 * code path on one GPU).  One host thread per slab, split-phase steps, ghost planes pulled from the neighbours with
 * peer copies on the edge stream while the interior planes run (replaces gpu_engine.h:516-662,739-823,993-1145).
 * base: options common to all slabs (numerics, air_variant, readout_chunk, multi_flags, transport, ...); NULL = defaults. */
double      pf_run_sim_devices(pf_simdata *sd, int32_t nslabs, const int32_t *devices, const pf_opts *base);
/* The owned plane ranges such a run uses when cut along x: cuts[0..nslabs], slab g owns global planes [cuts[g], cuts[g+1]).
 * pf_slab_partition: with the compiled-in wall-plane weights; _w: those weights times wall_scale. */
int         pf_slab_partition(const pf_simdata *sd, int32_t nslabs, int32_t even_split, int64_t *cuts);
int         pf_slab_partition_w(const pf_simdata *sd, int32_t nslabs, int32_t even_split, double wall_scale, int64_t *cuts);
/* ... and the cut along FILE Z (along_z = 1: what a chain of slabs stored with exchanged axes uses, PF_MULTI_CUT_Z) -- the ONE implementation of
 * the cut: pffdtd_amd/slab.py (one process per GPU under torch.distributed) calls this too.  No device needed. */
int         pf_slab_partition_axis(const pf_simdata *sd, int32_t nslabs, int32_t even_split, double wall_scale, int32_t along_z, int64_t *cuts);
/* Measures that factor for this scene on `device` (round 5): three short one-rank cost models -- an interior rank with Nx / nslabs planes,
 * one with a few planes more, the first rank with its x wall -- give the cost of an interior plane and of the wall; the ratio to what
 * the compiled-in weights predict is returned (<= 0: not measured -- scene too small, fewer than 63 steps --: use 1).  pf_multi_create
 * does this by itself under PF_MULTI_MEASURE_WEIGHTS; hosts with one process per device measure on rank 0 and hand the factor to every
 * rank (pf_opts.wall_scale). */
double      pf_slab_wall_scale(pf_simdata *sd, int32_t nslabs, int32_t device, const pf_opts *base);

/* ---- the same chain as an object (what pf_run_sim_devices does inside): lets a host warm up, time and inspect a
 * multi-device run -- bench.py --gpus N drives this from one plain process, as the reference drives all its GPUs from one
 * (gpu_engine.h:680-682, 993-1145).  One persistent host thread per slab.  sd must outlive the object; receivers land in
 * sd->u_out after every pf_multi_run. */
typedef struct pf_multi pf_multi;
typedef struct pf_multi_info {
   int32_t nslabs;
   int32_t transport;          /* PF_TRANSPORT_PEER | PF_TRANSPORT_RCCL (0: a single slab, nothing travels) */
   int32_t rccl_self;          /* 1: RCCL with one 1-rank communicator per slab (all slabs on one device: tests) */
   int32_t exchange_verified;  /* 1: every checked exchange delivered the senders' planes bit for bit; 0: one did not; -1: none checked */
   int64_t exchanges_checked;  /* exchanges (steps) checksummed so far */
   int32_t exchange_nonzero;   /* 1: at least one checked ghost plane was not all zeros (the check was not vacuous) */
   int32_t cut_along_z;        /* 1: the chain is cut along FILE Z (PF_MULTI_CUT_Z or chosen for the scene); pf_multi_get_slab's ranges are then z ranges */
   int64_t plane_bytes;        /* bytes of one exchanged plane */
   double  last_run_seconds;   /* wall time of the last pf_multi_run */
   char    transport_name[64];
   char    transport_note[256]; /* why this transport: the fallbacks PF_TRANSPORT_AUTO took ("" = its first choice) */
   double  wall_scale;         /* the factor on the wall planes' weights the chain was cut with (1: the compiled-in weights) */
   int32_t wall_measured;      /* 1: that factor was measured at creation (pf_slab_wall_scale) */
} pf_multi_info;
int  pf_multi_create(pf_simdata *sd, int32_t nslabs, const int32_t *devices, const pf_opts *base, pf_multi **out);
/* steps n0 .. n0+nsteps-1 on every slab; returns when all streams have drained and the receiver rows are in sd->u_out.
 * A slab whose host thread does not reach a step barrier within PFFDTD_BARRIER_TIMEOUT_S seconds (a stuck device, a collective that never
 * completes) turns the call into PF_ERR_HIP, pf_last_error() = "slab chain hung ...".  After THAT error the object is beyond repair: the
 * stuck thread may still be inside a driver / RCCL call and cannot be joined, so pf_multi_destroy only detaches the threads and frees
 * nothing -- device grids, pinned buffers and communicators of every slab stay allocated until the process ends.  A host that catches
 * the error should report it and EXIT; retrying in the same process can run out of device memory. */
int  pf_multi_run(pf_multi *m, int64_t n0, int64_t nsteps);
int  pf_multi_get_info(pf_multi *m, pf_multi_info *info);
/* slab g: owned global planes [x0, x1), device, whether it steps in temporally blocked pairs (1) or triples (3), and its engine (for
 * pf_engine_state_grids / pf_engine_timing between runs; do not step it directly) */
int  pf_multi_get_slab(pf_multi *m, int32_t g, int64_t *x0, int64_t *x1, int32_t *device, int32_t *paired, pf_engine **engine);
void pf_multi_destroy(pf_multi *m);

/* ---- one process per device: the plane exchange of a slab engine by native RCCL (replaces cudaMemcpyPeerAsync of gpu_engine.h:1086-1126
 * for a host whose ranks are PROCESSES, e.g. under torch.distributed.run; the chain object above is the one-process form).  Rank 0 obtains
 * the 128-byte id and hands it to every rank by any channel; all ranks then create their communicator together (returns PF_ERR_HIP with
 * pf_last_error() if the rendezvous fails or does not complete within PFFDTD_RCCL_INIT_TIMEOUT_S seconds: fall back to the host's own p2p).
 * pf_rccl_exchange goes between pf_engine_step_begin and pf_engine_step_end: first / last updated plane to rank peer_lo / peer_hi, theirs
 * into the ghost planes (peer < 0: no neighbour there), one ncclGroup on the engine's edge stream; asynchronous. */
typedef struct pf_rccl_comm pf_rccl_comm;
int  pf_rccl_unique_id(void *id128);
int  pf_rccl_comm_create(const void *id128, int32_t nranks, int32_t rank, int32_t device, pf_rccl_comm **out);
int  pf_rccl_exchange(pf_rccl_comm *c, pf_engine *e, int32_t peer_lo, int32_t peer_hi);
void pf_rccl_comm_destroy(pf_rccl_comm *c);

/* ---- engine object (what run_sim does inside, exposed for the Python host, slabs and tests) ---- */
int  pf_engine_create(const pf_simdata *sd, const pf_opts *opts, pf_engine **out);
void pf_engine_destroy(pf_engine *e);
/* steps n0 .. n0+nsteps-1 with no halo exchange (single slab); receivers land in sd->u_out[r*Nt+n] */
int  pf_engine_run(pf_engine *e, int64_t n0, int64_t nsteps);
/* split-phase step for Z-slab runs (gpu_engine.h:993-1145 re-thought): begin enqueues the edge planes +
 * all boundary-node work on the edge stream and the interior planes on the main stream; the caller then
 * exchanges the planes returned by pf_engine_halo_ptrs() (ordered after the edge stream), and end joins
 * the streams and rotates the state pointers. */
int  pf_engine_step_begin(pf_engine *e, int64_t n);
int  pf_engine_halo_ptrs(pf_engine *e, void **send_lo, void **send_hi, void **recv_lo, void **recv_hi,
                         size_t *plane_bytes);
int  pf_engine_step_end(pf_engine *e, int64_t n);
/* Slab engines created on caller-owned grids (pf_opts.ext_u0/ext_u1): hand over two more grids of the same size, so
 * that the engine may advance in temporally blocked pairs (two split-phase steps per pair, the state then cycles
 * through the four grids: pf_engine_halo_ptrs always names the grid being written).  Returns 0 when pairs are on,
 * 1 when this engine keeps stepping singly (scene without a boundary-free box, small y-z cross-section, 13-point,
 * single-domain engine, ...); other values
 * are pf_status errors.  No counterpart in the reference. */
int  pf_engine_set_spares(pf_engine *e, void *grid2, void *grid3);
/* The same with a choice: before its first step a slab engine is offered a pool of n >= 4 zero-filled caller-owned grids
 * (pool[0], pool[1] may be the ext_u0 / ext_u1 it was created on).  The speed of the two-steps-per-pass kernel depends on
 * where its four grids lie relative to each other in physical memory ("grid placement"), so the engine times
 * assignments of pool members to its four roles and adopts the fastest: idx[0], idx[1] = the state grids from now on,
 * idx[2], idx[3] = the spares; the caller may free the others.  An engine that keeps stepping singly (cf.
 * pf_engine_set_spares) picks the pair of the pool its single step is fastest on and says idx = i, j, -1, -1.  Returns a pf_status.
 * THE OFFERED GRIDS MUST BE ALL ZEROS AND ARE ZEROED AGAIN: the search runs real step kernels on them.  Initialise the field
 * (pf_engine_set_grid, device writes) only afterwards; after pf_engine_set_grid the call is refused with PF_ERR_STATE.
 * No counterpart in the reference. */
int  pf_engine_place_grids(pf_engine *e, void *const *pool, int32_t n, int32_t idx[4]);
/* The same with room for TRIPLES (round 5): a slab engine offered n >= 5 grids whose wall regions fit steps three steps per pass
 * across three split-phase steps (k_tb3, pf_tb3.h): idx[0], idx[1] = the state grids, idx[2], idx[3] = the two grids its blocked
 * kernel writes, idx[4] = the grid that holds u^{n+1} where somebody needs it in memory.  idx[4] = -1: pairs (idx[2], idx[3] the
 * spares) or single steps (idx[2] = idx[3] = -1), exactly as pf_engine_place_grids reports them.  No counterpart in the reference. */
int  pf_engine_place_grids5(pf_engine *e, void *const *pool, int32_t n, int32_t idx[5]);
/* Device pointers of the two state grids as they stand between runs (u_prev = u^{n-1}, overwritten by the next step;
 * u_cur = u^n), each pf_grid_bytes() long: the engine's own allocations unless pf_opts.ext_u0 / ext_u1 were given.
 * Lets a host that left the allocation to the engine (which then also chooses WHERE the grids live, see DESIGN.md
 * "placement") initialise or inspect the field on the device.  No counterpart in the reference. */
int  pf_engine_state_grids(pf_engine *e, void **u_prev, void **u_cur);
/* How those grids are laid out: dims[0..2] = planes, rows, columns as STORED, pitch = elements per row (a grid holds
 * dims[0] * dims[1] * pitch elements), exchanged = 1 when the engine stores the file's x and z axes exchanged (unit stride along
 * file x; only engines that own their grids may choose to, see DESIGN.md 5: rooms whose large surfaces are normal to file z).
 * pf_engine_get_grid / _set_grid and every index in pf_simdata stay in file order whatever the storage. */
int  pf_engine_layout(pf_engine *e, int64_t *dims, int64_t *pitch, int32_t *exchanged);
void *pf_engine_stream(pf_engine *e, int32_t which); /* 0 main, 1 edge: hipStream_t */
int  pf_engine_sync(pf_engine *e);
int  pf_engine_flush_outputs(pf_engine *e);          /* ring -> sd->u_out */
/* copy a state grid to/from a host array in FILE layout [Nx*Ny*Nz] Real; which: 0 = u0, 1 = u1.
 * pf_engine_set_grid: the field must be FINITE everywhere, the cells inside the walls included.  In the CPU-exact arithmetic the
 * boundary pass does not fetch a neighbour whose adjacency bit is clear (a cell inside the wall): the reference adds (a2 * 0) * u1
 * there, which is +-0 for a finite u1 and leaves the sum as it is (the sums of the time loop never hold -0), but NaN for an Inf
 * or NaN -- the internal switch PF_DBG_BND_FETCH_ALL (csrc/pf_debug.h) fetches every neighbour, the exact reference behaviour for such fields too. */
int  pf_engine_get_grid(pf_engine *e, int32_t which, void *host);
int  pf_engine_set_grid(pf_engine *e, int32_t which, const void *host);
int  pf_engine_timing(pf_engine *e, pf_timing *t, int32_t reset);
/* switch the per-launch HIP events of pf_opts.timing on / off between runs (a host times its headline region without them
 * and collects kernel durations in a region of its own) */
int  pf_engine_set_timing(pf_engine *e, int32_t on);

/* ---- energy-conservation diagnostic of the reference Python engine (python/fdtd/sim_fdtd.py:587-620,671-678) ----
 * Needs pf_opts.energy=1 at creation.  DEF = the materials' (D,E,F) triplets, double[Nm*PF_MMB*3] (zero padded),
 * h = grid spacing, c = speed of sound, Ts = time step (sim_consts.h5).  pf_engine_run_energy runs steps like
 * pf_engine_run and fills H_tot[n], E_lost[n+1], E_in[n+1] (arrays of Nt, Nt+1, Nt+1 doubles; E_*[0] must be
 * initialised by the caller).  fcc_flag 0 and 1 only, like the reference. */
int  pf_engine_energy_cfg(pf_engine *e, double h, double c, double Ts, const double *DEF);
int  pf_engine_run_energy(pf_engine *e, int64_t n0, int64_t nsteps, double *H_tot, double *E_lost, double *E_in);

#ifdef __cplusplus
}
#endif
#endif /* PFFDTD_HIP_H */
