"""GPU: the driver's own command lines for bench.py.  `python bench.py --gpus N --steps K --warmup W` with NO process
launcher must return rc 0 and one JSON line for N > 1 as well (round-2 verdict: it exited with "launch with torchrun"):
one process then drives N slabs through the C seam's chain object -- on this 1-GPU box as virtual slabs sharing device 0."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _bench(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], capture_output=True, text=True, cwd=ROOT, env=e, timeout=1500)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("transport", ["auto", "rccl"])
def test_plain_command_line_with_two_gpus(transport):
    extra = [] if transport == "auto" else ["--transport", transport]
    res = _bench("--gpus", "2", "--steps", "6", "--warmup", "3", *extra)
    assert res["n_gpus"] == 2 and res["steps"] == 6 and res["warmup"] == 3
    assert res["value"] > 0 and res["metric"] == "Gvoxel-updates/s" and res["config"]["grid"] == [1024, 1024, 1024]
    assert res["exchange_verified"] is True and res["exchange"]["checked_steps"] == 3 and res["exchange"]["nonzero_planes"]
    assert ("rccl" in res["exchange"]["backend"]) == (transport == "rccl")
    assert len(res["slabs"]) == 2 and res["slabs"][0]["planes"][1] == res["slabs"][1]["planes"][0]
    assert res["roofline"]["kernel_ms_per_launch"] > 0


def test_default_command_line_single_gpu_line_has_the_contract_fields():
    res = _bench("--steps", "8", "--warmup", "4", "--repeats", "3", "--size", "512", "--no-rigid-run")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "selfcheck"):
        assert k in res, k
    assert res["n_gpus"] == 1 and res["selfcheck"]["family_agreement"] is True and res["selfcheck"]["max_abs_sample"] > 0
    assert res["roofline"]["bound"] == "hbm" and 0 < res["roofline"]["frac"] < 1.5
    rl = res["roofline"]
    if rl.get("steps_per_launch", 1) > 1 and rl.get("measured_traffic_frac") is not None:
        # a multi-step kernel's headline fraction is the bounded one (PMC bytes / time / peak); SURVEY 8d's figure stays beside it
        assert rl["frac"] == rl["measured_traffic_frac"] < 1.0 and rl["frac_algorithmic_8d"] > rl["frac"]
    # the reference's own CPU binary where oracle/_ref shipped with the snapshot, else the bit-pinned port
    assert res["cpu_baseline"]["kind"] in ("reference", "port") and res["cpu_baseline"]["cores"] >= 1 and res["cpu_baseline"]["value"] > 0
    from pathlib import Path
    if (Path(__file__).resolve().parents[1] / "oracle" / "_ref" / "fdtd_main_cpu_single.x").exists():
        assert res["cpu_baseline"]["kind"] == "reference", res["cpu_baseline"]


def test_plain_command_line_with_eight_virtual_gpus():
    """The driver's N = 8 command on this 1-GPU box: eight virtual slabs through the C chain, exchange checksummed."""
    res = _bench("--gpus", "8", "--steps", "6", "--warmup", "3", "--repeats", "2")
    assert res["n_gpus"] == 8 and len(res["slabs"]) == 8 and res["virtual_slabs"] is True
    assert res["exchange_verified"] is True and res["exchange"]["checked_steps"] == 3 and res["exchange"]["nonzero_planes"]
    assert [s["planes"][1] for s in res["slabs"][:-1]] == [s["planes"][0] for s in res["slabs"][1:]]
    assert res["slabs"][0]["planes"][0] == 0 and res["slabs"][-1]["planes"][1] == 1024 and res["value"] > 0


@pytest.mark.parametrize("transport", ["copy", "rccl"])
def test_cost_model_of_one_rank_through_the_chain(transport):
    """--emulate-slab r/N: slab r of an N-rank chain alone on the device (pf_opts.only_slab), its own edge planes as ghost planes."""
    res = _bench("--emulate-slab", "3/8", "--emulate-transport", transport, "--steps", "6", "--warmup", "4", "--repeats", "2")
    em = res["emulated_slab"]
    assert em["rank"] == 3 and em["of"] == 8 and 120 <= em["planes"][1] - em["planes"][0] <= 136 and em["ms_per_step"] > 0
    assert ("rccl" in res["exchange"]["backend"]) == (transport == "rccl")
    assert res["n_gpus"] == 8 and "COST MODEL" in res["config"]["parallelism"]
