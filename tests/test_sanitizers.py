"""The host-side code under AddressSanitizer + UndefinedBehaviorSanitizer (`make asan`, CPU only):
the oracle (tools/asan/asan_oracle.c) and the library's scene parser, built-in scenes, camera
functions, BVH builder and checkpoint files (tools/asan/asan_host.cpp).  The reference compiles
its own asserts out (globals.h:25-34) and has no sanitizer run; these are this build's."""
import os
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def built():
    p = subprocess.run(["make", "-C", ROOT, "asan"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    return os.path.join(ROOT, "tools", "asan")


def run_clean(cmd):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    for needle in ("AddressSanitizer", "LeakSanitizer", "runtime error"):
        assert needle not in p.stderr, p.stderr
    return p.stdout


def test_oracle_under_sanitizers_prints_what_the_plain_build_prints(built):
    sanitized = run_clean([os.path.join(built, "asan_oracle")])
    plain = subprocess.run([os.path.join(built, "plain_oracle")], capture_output=True, text=True, timeout=600)
    assert plain.returncode == 0
    assert sanitized == plain.stdout
    assert sanitized.count("rc=0") == 10


def test_host_code_under_sanitizers(built, tmp_path):
    out = run_clean([os.path.join(built, "asan_host"), str(tmp_path)])
    assert "asan_host: 0 failure(s)" in out
