import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "scalable-ccd_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


# ---- safety net ---------------------------------------------------------------------------------
# A checker that takes the host's memory, or a kernel that never returns, costs a GPU box (round 1 lost
# two that way).  (RLIMIT_AS is not usable: the HIP runtime reserves terabytes of address space.)
#  * a watchdog thread ends the process (exit code 86) when its resident set passes SCCD_TEST_RSS_GB
#    (default 24 GB; the largest legitimate test needs about 6);
#  * every test has a wall-clock limit (pytest-timeout, thread method: it can end a process whose main
#    thread is stuck inside a C call), SCCD_TEST_TIMEOUT seconds, default 900.
def _start_rss_watchdog():
    import threading
    import time

    limit = float(os.environ.get("SCCD_TEST_RSS_GB", "24")) * (1 << 30)

    def rss():
        with open("/proc/self/statm") as f:
            return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE")

    def watch():
        while True:
            try:
                if rss() > limit:
                    sys.stderr.write("\n[conftest] resident set above %.0f GB: aborting the test run\n" % (limit / (1 << 30)))
                    sys.stderr.flush()
                    os._exit(86)
            except OSError:
                pass
            time.sleep(0.25)

    threading.Thread(target=watch, name="rss-watchdog", daemon=True).start()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _start_rss_watchdog()
    if config.pluginmanager.hasplugin("timeout"):
        if not getattr(config.option, "timeout", None):
            config.option.timeout = float(os.environ.get("SCCD_TEST_TIMEOUT", "900"))
        config.option.timeout_method = "thread"


def pytest_collection_modifyitems(config, items):
    """SCCD_TEST_ORDER=reverse | shuffle:<seed>: the tests in another order -- the GPU tests share one context, and what a test
    leaves there (buffer sizes, a BroadPhase's guess of its next build) must not matter to the next."""
    order = os.environ.get("SCCD_TEST_ORDER", "")
    if order == "reverse":
        items.reverse()
    elif order.startswith("shuffle:"):
        import random

        random.Random(int(order.split(":", 1)[1])).shuffle(items)


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (oracle/sccd_oracle.c) -- the checker, never the product."""
    import orc as _orc

    _orc.lib()
    return _orc


@pytest.fixture(scope="session")
def sccd():
    import sccd as _sccd

    return _sccd


@pytest.fixture(scope="session")
def ctx(sccd):
    """A GPU context; the HIP extension must be the thing that runs (no fallback)."""
    c = sccd.Context(0)
    yield c
    c.close()
