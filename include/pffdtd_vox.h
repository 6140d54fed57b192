/*
 * pffdtd_vox.h -- C ABI of the MI355X voxelizer in libpffdtd_hip.so.
 *
 * Replaces the ray-triangle stage of the reference's Python voxelizer
 *
 *     VoxGrid.fill()      python/voxelizer/vox_grid_base.py:64-199   (triangle -> voxel binning)
 *     VoxScene.calc_adj() python/voxelizer/vox_scene.py:99-391       (per grid point: which of the NN legs to the
 *                                                                     neighbours are cut by a surface)
 *
 * which the reference runs as N Python processes over a voxel hierarchy with temporary .h5 files.  Here one call
 * bins the triangles into fixed cells of grid points on the device and tests every (grid point, candidate
 * triangle, leg) in one kernel, with the reference's arithmetic operation by operation (IEEE double, no
 * contraction), so the boundary-node set, adjacency bits and nearest-triangle choice are identical.
 *
 * The caller (pffdtd_amd/voxelizer.py) prepares the per-triangle records with numpy exactly as
 * python/common/tris_precompute.py:21-123 does and derives materials and surface-area factors from the result
 * (vox_scene.py:393-424).  Plain pointers and sizes only; all pointers are host pointers owned by the caller.
 * Errors: NULL / non-zero return; the message is kept for the pf_last_error call of pffdtd_hip.h.
 */
#ifndef PFFDTD_VOX_H
#define PFFDTD_VOX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PF_VOX_TRI_DOUBLES 30 /* one triangle record, see pf_vox_desc.tris */

typedef struct pf_vox_desc {
   int64_t Nx, Ny, Nz;      /* grid points (cart_grid.py:31) */
   const double *xv, *yv, *zv; /* grid vectors [Nx],[Ny],[Nz] (cart_grid.py:39-43) */
   int32_t NN;              /* legs per point: 6 Cartesian, 12 FCC (vox_scene.py:69-84) */
   int32_t fcc;             /* 1: only points with (ix+iy+iz) even exist (vox_scene.py:157-158) */
   const double *vvh;       /* [NN][3] leg vectors h*VV (vox_scene.py:86) */
   const double *ray_un;    /* [NN][3] ray directions as normalise() returns them (tri_ray_intersection.py:78) */
   double h;                /* Cartesian grid spacing */
   double hf;               /* leg length: h, or sqrt(2) h on the FCC subgrid (vox_scene.py:71,80) */
   double hfe;              /* hf*(1+R_EPS): bounding-box / plane-distance slack (vox_scene.py:170-178) */
   double hf1;              /* (1+R_EPS)*hf: a hit up to here cuts the leg (vox_scene.py:220-222) */
   double nb_eps;           /* R_EPS*hf: |hit| below this = point lies on a surface (vox_scene.py:212-214) */
   double d_eps;            /* edge-function slack 1e-3 h (vox_scene.py:207) */
   double cp_eps;           /* coplanarity threshold 1e-6 (tri_ray_intersection.py:67) */
   int64_t Ntris;
   const double *tris;      /* [Ntris][30]: cent, unor, (a+b)/2, (b+c)/2, (c+a)/2, eab_unor, ebc_unor, eca_unor,
                               bmin-hfe, bmax+hfe  (3 doubles each) */
   int32_t device;
   int32_t reserved;
} pf_vox_desc;

typedef struct pf_vox_stats {
   double ms_bin;           /* triangle -> cell binning kernels */
   double ms_vox;           /* ray-triangle kernel(s) */
   double ms_total;         /* whole call incl. transfers */
   int64_t ncells, ncells_nonempty, npairs; /* cells, cells with candidates, (cell, triangle) pairs */
   int64_t npoints_tested;  /* grid points in non-empty cells */
} pf_vox_stats;

typedef struct pf_vox_job pf_vox_job;

/* Runs the whole voxelization; the result stays in the job until pf_vox_free. NULL on error. */
pf_vox_job *pf_vox_run(const pf_vox_desc *desc);
/* Number of boundary nodes found (grid points with at least one cut leg). */
int64_t pf_vox_count(const pf_vox_job *job);
/* Copies the result, in no particular order: linear index ix*Ny*Nz+iy*Nz+iz, bit j set = leg j is CUT
   (the complement of the reference's adj_bn row), nearest triangle and its distance (vox_scene.py:224-232). */
int pf_vox_fetch(const pf_vox_job *job, int64_t *bn_ixyz, uint16_t *cut_bits, int32_t *tidx, double *ndist);
int pf_vox_get_stats(const pf_vox_job *job, pf_vox_stats *st);
void pf_vox_free(pf_vox_job *job);

#ifdef __cplusplus
}
#endif
#endif
