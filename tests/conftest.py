import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


_FAULT_LOG = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Round 6: ANY run of the suite -- the driver's too -- leaves a Python traceback of every thread if the process dies on a signal
    # (SIGSEGV / SIGABRT / SIGBUS / SIGFPE) or hangs: round 5 saw ONE core dump in eleven GPU-suite runs and kept six lines of it.
    # The log goes to gpurun_out/ (copied back from the GPU box) when that directory exists; where it does not (the driver's box: the
    # snapshot leaves gpurun_out/ behind) pytest's own faulthandler plugin stays in charge and the traceback goes to stderr, which is
    # what that run keeps.  faulthandler has ONE output file per process, so it is one or the other.
    global _FAULT_LOG
    import faulthandler
    d = os.path.join(ROOT, "gpurun_out")
    if not os.path.isdir(d) or not os.access(d, os.W_OK):
        return
    try:
        _FAULT_LOG = open(os.path.join(d, "pytest_faulthandler_%d.log" % os.getpid()), "w")
        faulthandler.enable(file=_FAULT_LOG, all_threads=True)
        # a hang (a kernel that never ends, a lost RCCL rendezvous) gets a traceback every 10 minutes instead of silence
        faulthandler.dump_traceback_later(600, repeat=True, file=_FAULT_LOG)
    except OSError:
        _FAULT_LOG = None


def pytest_unconfigure(config):
    global _FAULT_LOG
    if _FAULT_LOG is not None:
        import faulthandler
        faulthandler.cancel_dump_traceback_later()
        faulthandler.disable()
        name = _FAULT_LOG.name
        _FAULT_LOG.close()
        _FAULT_LOG = None
        try:
            if os.path.getsize(name) == 0:
                os.remove(name)          # a clean run leaves nothing behind
        except OSError:
            pass


def gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def has_gpu():
    return gpu_available()


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need a HIP device: without one they are skipped, not failed (plain `pytest tests` on a CPU box)"""
    if gpu_available():
        return
    skip = pytest.mark.skip(reason="no HIP device (run with -m gpu on an MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
