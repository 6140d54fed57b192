// pf_debug.h -- development and test switches, 0 in production.  Each PF_DBG_* bit forces an alternative arrangement the engine also
// contains -- an older kernel family, a fallback, one stream instead of two -- so that the tests can pin every path to the oracle and A/B
// measurements can be taken on one box.  INTERNAL: none of this is part of the drop-in boundary.  Round 6: the switches left the public
// pf_opts (include/pffdtd_hip.h); they reach the library through pf_internal_hooks below (exported, declared only here; pffdtd_amd/engine.py
// calls it for its `debug=` / `test_*=` keyword arguments) or the environment variable PFFDTD_DEBUG (a number, OR-ed in).
#pragma once
#include "pffdtd_hip.h"
// the options as the engines and the chain see them: the public ones + the switches
struct pf_opts_x : pf_opts {
   int32_t debug;              // a mask of PF_DBG_* bits
   int32_t test_drop_exchange; // pf_multi_create: 1 + n = slab 1 misses the ghost planes of step n (the exchange self-check, which then always
                               // covers that step, must notice)
   int32_t test_faults;        // pf_multi_create: fault injection for the first-contact paths of a multi-device box.  1: no peer access between
                               // any two devices; 2: RCCL unusable; 4: the host thread of slab 1 stalls before the barrier of its fourth step
                               // (the watchdog of the others must turn the hang into an error)
};
// The switches of the NEXT pf_engine_create / pf_multi_create / pf_run_sim_devices / pf_slab_wall_scale call of the calling thread (that call
// clears them again).
extern "C" void pf_internal_hooks(int32_t debug, int32_t test_drop_exchange, int32_t test_faults);
pf_opts_x pf__take_hooks(const pf_opts *o); // *o (NULL: the defaults) + the calling thread's pending switches + PFFDTD_DEBUG
int pf__engine_create_x(const pf_simdata *sd, const pf_opts_x *o, pf_engine **out);
enum : int {
   PF_DBG_LW32 = 0x100,               // 32-lane row segments (0x200: 16-lane, 0x400: 64-lane) instead of the measured choice
   PF_DBG_LW16 = 0x200,
   PF_DBG_LW64 = 0x400,
   PF_DBG_SRC_TILES_SINGLE = 0x40,    // triples: the tiles within reach of a source (and a receiver's tile) step singly, the sources added by k_io (until round 6)
   PF_DBG_EDGE_SEPARATE = 0x80,       // slab triples: the edge planes of the two sides by separate launches (lean kernel, boundary kernel), as in round 5
   PF_DBG_RUNTIME_GEOMETRY = 0x800,   // three-step wall regions: the bodies with run-time pencil geometry, never the ones with it compiled in
   PF_DBG_SWZ_ON = 0x1000,            // store the grid with the file's x and z axes exchanged (0x2000: never; default: decided per scene)
   PF_DBG_SWZ_OFF = 0x2000,
   PF_DBG_SINGLE_STEPS = 0x4000,      // single steps only (no blocked pairs / triples)
   PF_DBG_NO_AUTOTUNE = 0x8000,       // no creation-time measurement: static rules choose the kernel, grids stay as allocated
   PF_DBG_WALLS_BESIDE_BOX = 0x10000, // experiment: a triple's alike wall launches beside k_tb3 instead of before it
   PF_DBG_NO_TRIPLES = 0x20000,       // never three steps per pass (pairs as in round 4)
   PF_DBG_FCC_PAIR_R4 = 0x40000,      // 13-point pairs by the round-4 kernel k_tb2_fcc_x (A/B measurements)
   PF_DBG_THIRD_STEP_LISTS = 0x80000, // the third step of a single domain's triple by the list kernels (round-5 start); no bricks
   PF_DBG_BND_PLAIN_ORDER = 0x100000, // boundary pass in plain workgroup order (no XCD-aware runs)
   PF_DBG_BND_FETCH_ALL = 0x200000,   // ... fetches the neighbours inside the wall too
   PF_DBG_FRAME_GENERIC = 0x400000,   // the frame of the shell (edges, corners) as generic blocks of k_wall2 (round 5) instead of bricks (pf_brick.h)
   PF_DBG_GRAPH = 0x800000,           // replay the single-step loop from a hipGraph (six steps per graph; no faster on this stack)
   PF_DBG_XY_TWO_PLUS_ONE = 0x1000000,// a triple's x / y wall regions take two steps + one instead of three in one pass (k_wall2 NS = 3)
   PF_DBG_STRIPS_SPLIT = 0x2000000,   // wide column strips cut in two (12- + 16-cell pencils)
   PF_DBG_WALLS_ONE_STREAM = 0x4000000, // every launch of a blocked pass on the main stream
   PF_DBG_WALLS_ALL_GENERIC = 0x8000000, // wall regions: every block generic (no alike fast path, no bricks)
   PF_DBG_NO_WALL_REGIONS = 0x10000000, // blocked pairs keep the single-step shell
   PF_DBG_BND_ALL_NODES = 0x20000000, // the boundary-list kernel visits every node (none left to the column-strip kernel)
   PF_DBG_STRIPS_TWO_PLUS_ONE = 0x40000000, // a triple's column strips take two steps + one instead of three in one pass
};
