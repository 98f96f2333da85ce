"""AddressSanitizer + UndefinedBehaviorSanitizer run of the HOST-side code of the path on the CPU (the
reference has no sanitizer runs; GPU ASan is not available on the test pool): the host builder of the derived-graph handle
(csrc/graph.hip, compiled as plain C++ against tests/asan/shim) and the C restatement oracle/ngpde_oracle.c, driven by
tests/asan/graph_host_asan.cpp over the reference's fixture graph, empty / edgeless graphs, duplicates, self loops, a hub row,
batches and out-of-range indices."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ADIR = os.path.join(HERE, "asan")


@pytest.mark.skipif(shutil.which("g++") is None or shutil.which("make") is None, reason="needs g++ and make")
def test_host_graph_builder_and_c_oracle_are_clean_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", ADIR], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(ADIR, "graph_host_asan")], capture_output=True, text=True, timeout=300, env=env)
    report = r.stdout + r.stderr
    assert r.returncode == 0, report[-3000:]
    assert "AddressSanitizer" not in report and "runtime error" not in report and "LeakSanitizer" not in report, report[-3000:]
    assert "0 failed checks" in r.stdout
