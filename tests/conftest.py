import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


os.environ.setdefault("PHONIC_DEBUG_HOOKS", "1")   # arms pg_debug_fail_launch_round (the fault-injection test); a production process leaves it unset


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle

    return oracle.lib()


# ---- fuzz bookkeeping --------------------------------------------------------------------------------
# Every run that collects tests/test_gpu_fuzz.py ends with ONE machine-readable line — cases requested / run / passed / skipped by the
# discontinuity classifier / skipped otherwise / failed, per family and in total, with the campaign's seed settings — also when a time limit
# cuts the run short (a SIGTERM from `timeout` is turned into an orderly session end first): appended to gpurun_out/fuzz_summary.jsonl and
# printed. PHONIC_FUZZ_TIME_LIMIT=seconds ends a campaign from inside (the remaining cases count as `not_run`).
_FUZZ = {"collected": 0, "by_family": {}, "start": None, "cut": None}


def _fam(nodeid):
    name = nodeid.split("::")[-1]
    return name.split("[")[0]


def pytest_collection_modifyitems(session, config, items):
    for it in items:
        if "test_gpu_fuzz.py" in it.nodeid:
            _FUZZ["collected"] += 1
            _FUZZ["by_family"].setdefault(_fam(it.nodeid), {"requested": 0, "passed": 0, "failed": 0, "skipped_by_classifier": 0, "skipped_other": 0})["requested"] += 1


def pytest_deselected(items):   # (-k / -m deselect behind the collection hook above: those cases were never requested)
    for it in items:
        if "test_gpu_fuzz.py" in it.nodeid:
            f = _FUZZ["by_family"].get(_fam(it.nodeid))
            if f and f["requested"] > 0:
                f["requested"] -= 1
                _FUZZ["collected"] -= 1


def pytest_sessionstart(session):
    import signal
    import time

    _FUZZ["start"] = time.time()

    def _term(signum, frame):   # `timeout` sends SIGTERM: end the session in order, the summary line still goes out
        _FUZZ["cut"] = "SIGTERM"
        session.shouldstop = "time limit (SIGTERM)"

    try:
        signal.signal(signal.SIGTERM, _term)
    except (ValueError, OSError):
        pass


def pytest_runtest_logreport(report):
    if "test_gpu_fuzz.py" not in report.nodeid:
        return
    f = _FUZZ["by_family"].setdefault(_fam(report.nodeid), {"requested": 0, "passed": 0, "failed": 0, "skipped_by_classifier": 0, "skipped_other": 0})
    if report.when == "call":
        if report.passed:
            f["passed"] += 1
        elif report.failed:
            f["failed"] += 1
        elif report.skipped:
            text = str(report.longrepr)
            f["skipped_by_classifier" if "discontinuous" in text else "skipped_other"] += 1
    elif report.when == "setup" and report.skipped:
        f["skipped_other"] += 1
    elif report.when == "setup" and report.failed:
        f["failed"] += 1


def pytest_runtest_setup(item):
    import time

    lim = float(os.environ.get("PHONIC_FUZZ_TIME_LIMIT", "0") or 0)
    if lim > 0 and "test_gpu_fuzz.py" in item.nodeid and _FUZZ["start"] and time.time() - _FUZZ["start"] > lim:
        _FUZZ["cut"] = f"PHONIC_FUZZ_TIME_LIMIT={lim:g}"
        item.session.shouldstop = "fuzz time limit"


def pytest_sessionfinish(session, exitstatus):
    import json
    import time

    if not _FUZZ["collected"]:
        return
    fams = _FUZZ["by_family"]
    tot = {k: sum(f[k] for f in fams.values()) for k in ("requested", "passed", "failed", "skipped_by_classifier", "skipped_other")}
    tot["run"] = tot["passed"] + tot["failed"] + tot["skipped_by_classifier"] + tot["skipped_other"]
    tot["not_run"] = tot["requested"] - tot["run"]
    if tot["run"] == 0 and not _FUZZ["cut"]:
        return   # (the fuzz file was collected but deselected: a CPU run)
    line = {"fuzz_summary": True, "seeds": int(os.environ.get("PHONIC_FUZZ_SEEDS", "0") or 0), "base": int(os.environ.get("PHONIC_FUZZ_BASE", "0") or 0),
            "seconds": round(time.time() - (_FUZZ["start"] or time.time()), 1), "cut_by": _FUZZ["cut"], "exitstatus": int(exitstatus), **tot, "families": fams}
    try:
        from phonic_amd import _capi

        line["source_hash"] = _capi.source_hash()
    except Exception:
        pass
    text = json.dumps(line)
    print("\n" + text)
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "fuzz_summary.jsonl"), "a") as fh:
            fh.write(text + "\n")
    except OSError:
        pass
