"""GPU: transport smoke test of the exchange path with a 1-rank RCCL process group: the same
`batch_isend_irecv` + edge-stream ordering the N>1 runner uses, sending a halo plane to ourselves.
(Real multi-rank exchange needs >1 GPU: bench.py --gpus N on the multi-GPU node.)
"""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

import cases
from pffdtd_amd import dist as pdist

pytestmark = pytest.mark.gpu


def test_rccl_self_exchange_ordered_after_edge_stream():
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        sd = cases.make_sd("cart_outside", "single")
        runner, loc, info = pdist.make_hip_runner(sd, 0, 1, 0)
        st = runner.st
        scratch = torch.zeros_like(st.grids[0][0])
        for n in range(20):
            st.step_begin(n)
            s_lo, s_hi, r_lo, r_hi = st.halo_tensors()
            ops = [dist.P2POp(dist.isend, s_lo, 0), dist.P2POp(dist.irecv, scratch, 0)]
            with st.comm_context():
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            st.step_end(n)
            st.sync()
            # the received plane must be the freshly computed plane 1 of the NEW state (u1 after the rotation)
            new = st.grids[(st.k - 1) % 2]
            assert torch.equal(scratch, new[1]), f"step {n}: exchange saw stale data"
        assert float(scratch.abs().max()) > 0
        st.close()
    finally:
        dist.destroy_process_group()
