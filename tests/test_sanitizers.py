"""Sanitized CPU builds, the counterpart of the reference's Debug build with
-fsanitize=undefined (CMakeLists.txt:214-219): the oracle (C) and the
library's host-only code (scene import, .hdr decode; C++ inside the HIP
sources) are built with AddressSanitizer and UndefinedBehaviorSanitizer and
driven through their main paths, including truncated and corrupted input
files. A sanitizer report fails the make target. No GPU is touched."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make(directory):
    r = subprocess.run(["make", "-C", directory, "SAN=1", "check", "-j8"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def test_oracle_under_asan_and_ubsan():
    assert "san_check ok" in _make(os.path.join(ROOT, "oracle"))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_library_host_code_under_asan_and_ubsan():
    assert "ptc_host_san_check ok" in _make(os.path.join(ROOT, "path_tracer_amd", "csrc"))
