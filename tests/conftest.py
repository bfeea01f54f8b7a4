import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


# ---- oracle trajectories for the non-toy GPU parity tests (tests/oracle_jobs.py): started once, when the first GPU test runs, in spawned
# worker processes, so that the float64 oracle (13 s per iteration at 128 x 128 x 3000) works under the rest of the GPU suite
_observed = {}


@pytest.fixture(scope="session")
def observed():
    """error magnitudes the parity tests measured (written to gpurun_out/parity_observed.json at the end of the session)"""
    return _observed


@pytest.fixture(scope="session", autouse=True)
def oracle_jobs(request):
    if not _has_gpu() or not any("gpu" in it.keywords for it in request.session.items):
        yield {}
        return
    import concurrent.futures as cf
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_jobs as oj
    wanted = [n for n in oj.JOBS if any(n in it.name for it in request.session.items)]
    ex = cf.ProcessPoolExecutor(max_workers=max(1, min(len(wanted), 10)), mp_context=mp.get_context("spawn"))
    futs = {n: ex.submit(oj.trajectory, n) for n in wanted}
    yield futs
    ex.shutdown(wait=False, cancel_futures=True)


def pytest_sessionfinish(session, exitstatus):
    out = os.path.join(ROOT, "gpurun_out")
    try:
        import parity_util
        if parity_util.RECORD:
            _observed["rel_by_call_site"] = parity_util.RECORD
    except Exception:
        pass
    gpu_run = any(item.get_closest_marker("gpu") for item in getattr(session, "items", []))
    if _observed and os.path.isdir(out):
        import json
        # (a CPU-only session must not overwrite the record of the last GPU session)
        with open(os.path.join(out, "parity_observed.json" if gpu_run else "parity_observed_cpu.json"), "w") as fh:
            json.dump(_observed, fh, indent=1, sort_keys=True)
