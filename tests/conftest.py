import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

_PER_TEST = {}  # test id -> (n comparisons, worst scale_err, worst rms_err, worst elem_err)
NOTES = []      # free-form measured-error lines the parity tests want in the terminal summary


def _usable_cores():
    """CPU threads this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU boxes expose 256
    hardware threads under a 16-CPU quota; torch's default of one thread per visible core then throttles the CPU oracle)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:
        import torch
        torch.set_num_threads(min(16, _usable_cores()))
    except Exception:  # noqa: BLE001
        pass


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped automatically where no GPU is visible (the CPU container)."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _collect_errors(request):
    import _util

    start = len(_util.REPORT)
    yield
    rows = _util.REPORT[start:]
    if rows:
        _PER_TEST[request.node.nodeid] = (len(rows), max(r[1] for r in rows), max(r[2] for r in rows), max(r[3] for r in rows))


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Observed parity errors (max|a-b|/max|b|, rms, element-relative 99.9 %) of every test that compared tensors."""
    if not _PER_TEST and not NOTES:
        return
    tr = terminalreporter
    tr.section("observed parity errors")
    for line in NOTES:
        tr.write_line(line)
    import _util
    if _util.SLOPES:
        sl = sorted(_util.SLOPES, key=lambda r: -abs(r[1]))
        tr.write_line(f"{len(sl)} gradient tensors checked for a systematic error (least-squares slope against the reference); the 8 largest |slope - 1|:")
        for what, ds, cos, n in sl[:8]:
            tr.write_line(f"  slope - 1 = {ds:+.2e}  cosine {cos:.5f}  ({n} elements)  {what}")
    worst = sorted(_PER_TEST.items(), key=lambda kv: -kv[1][1])
    tr.write_line(f"{len(_PER_TEST)} tests compared tensors; the 25 largest max|a-b|/max|b| (scale / rms / elem99.9, #comparisons):")
    for nodeid, (n, s, r, e) in worst[:25]:
        tr.write_line(f"  {s:.2e} / {r:.2e} / {e:.2e}  ({n:4d})  {nodeid.split('/')[-1]}")
