"""Not collected by `pytest tests/` (the file name does not match test_*.py): tests/test_host_logic.py runs it in a child pytest process to show
what the `isolated` marker of tests/conftest.py does with a body that aborts."""
import os

import pytest

SEEN = []


@pytest.mark.isolated
def test_first_passes():
    assert os.environ.get("BOT_TEST_ISOLATED_CHILD") == "1"       # the body only ever runs in the child
    SEEN.append(os.getpid())


@pytest.mark.isolated
def test_second_aborts():
    import ctypes
    import threading
    t = threading.Thread(target=ctypes.CDLL(None).abort)           # abort() on a thread without a Python frame of its own, like a runtime's watchdog
    t.start()
    t.join()


@pytest.mark.isolated
@pytest.mark.parametrize("kind", ["a-b", "c"])
def test_third_still_runs(kind):
    assert kind in ("a-b", "c") and os.environ.get("BOT_TEST_ISOLATED_CHILD") == "1"


def test_parent_never_ran_a_body():
    assert SEEN == [] and not os.environ.get("BOT_TEST_ISOLATED_CHILD")
