import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


_FAULT_LOG = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.hookimpl(trylast=True)
def pytest_sessionstart(session):
    # Round 6: ANY run of the suite -- the driver's too -- leaves a Python traceback of every thread if the process dies on a signal
    # (SIGSEGV / SIGABRT / SIGBUS / SIGFPE) or hangs: round 5 saw ONE core dump in eleven GPU-suite runs and kept six lines of it.
    # The log goes to gpurun_out/ (copied back from the GPU box) when that directory exists; where it does not (the driver's box: the
    # snapshot leaves gpurun_out/ behind) pytest's own faulthandler plugin stays in charge and the traceback goes to stderr, which is
    # what that run keeps.  faulthandler has ONE output file per process, so it is one or the other -- and the last enable() wins: this
    # runs at session start, after every plugin's pytest_configure (as a pytest_configure hook it lost to pytest's own plugin, and the
    # one abort round 6 did catch left its traceback on stderr and an empty file here).
    # The other half is pytest.ini's --capture=sys: with the default fd-level capture the HIP / HSA runtime's own last words ("Memory
    # access fault by GPU node ...", a queue error, glibc's heap-check message) go to a temporary file that dies with the process.
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


# Round 6 crash hunt (profiles/r06_crash_hunt.txt, part 2): DART_TRACE_BLOCKS=1 logs, on the real stderr, every output block a HipStepper page-locks
# (host range, the device address the runtime maps it at, the heap's current end) and every handle's close -- so that the address in the runtime's
# "Memory access fault by GPU node ... on address ..." line can be placed.  Test infrastructure only: it wraps two methods of the Python mirror.
if os.environ.get("DART_TRACE_BLOCKS"):
    import ctypes as _C

    def _install_block_trace():
        from dart_env_amd import stepper as _st
        libc = _C.CDLL(None); libc.sbrk.restype = _C.c_void_p; libc.sbrk.argtypes = [_C.c_long]
        hip = _C.CDLL("libamdhip64.so")
        hip.hipHostGetDevicePointer.argtypes = [_C.POINTER(_C.c_void_p), _C.c_void_p, _C.c_uint]
        free0, close0 = _st.HipStepper._free_block, _st.HipStepper.close

        def _free_block(self):
            before = len(self.__dict__.get("_blocks", []))
            blk = free0(self)
            pool = self.__dict__.get("_blocks", [])
            if len(pool) > before:
                a = pool[-1][2].value; d = _C.c_void_p()
                rc = hip.hipHostGetDevicePointer(_C.byref(d), _C.c_void_p(a), 0)
                os.write(2, ("[blocks] handle %#x registers %#x .. %#x (%d bytes), device address %#x (rc %d), brk %#x\n" % (
                    self.h.value or 0, a, a + self._layout()[0], self._layout()[0], d.value or 0, rc, libc.sbrk(0) or 0)).encode())
            return blk

        def close(self):
            if getattr(self, "h", None):
                os.write(2, ("[blocks] handle %#x closes, blocks %s\n" % (self.h.value or 0, " ".join("%#x" % e[2].value for e in self.__dict__.get("_blocks", [])))).encode())
            return close0(self)

        _st.HipStepper._free_block, _st.HipStepper.close = _free_block, close

    _install_block_trace()

    def pytest_runtest_logstart(nodeid, location):
        os.write(2, ("[blocks] test %s\n" % nodeid).encode())

    def pytest_runtest_logfinish(nodeid, location):
        # DART_STOP_AFTER=<part of a node id>: end the session behind that test (the hunt loops the suite's first quarter with the WHOLE suite collected)
        stop = os.environ.get("DART_STOP_AFTER")
        if stop and stop in nodeid:
            pytest.exit("DART_STOP_AFTER", returncode=0)
