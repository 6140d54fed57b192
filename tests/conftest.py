import shutil
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_hipcc():
    return Path("/opt/rocm/bin/hipcc").exists() or shutil.which("hipcc") is not None


def _device_count():
    from pffdtd_amd import build, engine
    if _have_hipcc():
        build.build_hip()
    return engine.device_count()


def pytest_collection_modifyitems(config, items):
    """`gpu` tests are skipped (not failed) on a machine without a HIP device."""
    if not any("gpu" in it.keywords for it in items):
        return
    if Path("/dev/kfd").exists():
        return  # a GPU box: nothing is skipped -- a missing library or an invisible device must fail loudly there
    try:
        if _device_count() > 0:
            return
    except Exception:
        pass
    skip = pytest.mark.skip(reason="no HIP device visible (the HIP engine has no CPU fallback)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the native pieces exist (cheap no-op when up to date; the HIP library only where hipcc is)."""
    from pffdtd_amd import build
    build.build_h5()
    build.build_oracle()
    if _have_hipcc():
        build.build_hip()
