/* c_cuda/hip_engine.h -- binds libpffdtd_hip.so (include/pffdtd_hip.h) */
#include "pffdtd_hip.h"
double run_sim(struct SimData *sd) {            /* same signature as cpu_engine.h:52 / gpu_engine.h:665 */
   pf_simdata p;
   memset(&p, 0, sizeof p);
   p.bn_ixyz = sd->bn_ixyz;   p.bnl_ixyz = sd->bnl_ixyz; p.bna_ixyz = sd->bna_ixyz; p.Q_bna = sd->Q_bna;
   p.in_ixyz = sd->in_ixyz;   p.out_ixyz = sd->out_ixyz; p.out_reorder = sd->out_reorder;
   p.adj_bn  = sd->adj_bn;    p.ssaf_bnl = sd->ssaf_bnl; p.bn_mask = sd->bn_mask; p.mat_bnl = sd->mat_bnl;
   p.K_bn    = sd->K_bn;      p.in_sigs  = sd->in_sigs;  p.u_out   = sd->u_out;
   p.Ns = sd->Ns; p.Nr = sd->Nr; p.Nt = sd->Nt; p.Npts = sd->Npts; p.Nx = sd->Nx; p.Ny = sd->Ny; p.Nz = sd->Nz;
   p.Nb = sd->Nb; p.Nbl = sd->Nbl; p.Nba = sd->Nba; p.l = sd->l; p.l2 = sd->l2;
   p.fcc_flag = sd->fcc_flag; p.NN = sd->NN; p.Nm = sd->Nm; p.Mb = sd->Mb;
   p.mat_quads = sd->mat_quads; p.mat_beta = sd->mat_beta; p.infac = sd->infac;
   p.sl2 = sd->sl2; p.lo2 = sd->lo2; p.a2 = sd->a2; p.a1 = sd->a1;   /* Real -> double is exact */
   p.real_bytes = (int32_t)sizeof(Real);
   double t = pf_run_sim(&p);
   if (t < 0) { printf("pffdtd_hip: %s\n", pf_last_error()); assert(true==false); }
   return t;
}
