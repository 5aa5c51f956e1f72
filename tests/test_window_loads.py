"""The predict kernels' input-window loads (weath3rb0i_amd/csrc/w3_window.h) on the CPU, between two inaccessible pages: every small
block size, buffer flush with the page start and with the page end — values against the definition, and no read outside the buffer.
(On the GPU an out-of-buffer read only faults when the neighbouring page happens to be unmapped: one aborted suite run in nine is how
the reads before the buffer at block sizes under 7 bytes were found in round 4.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host", "window_loads.cpp")


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not found")
def test_window_loads_stay_inside_the_buffer(tmp_path):
    exe = str(tmp_path / "window_loads")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-o", exe, SRC])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout[-400:], r.stderr[-400:])   # (-11: a read outside the buffer)
    assert "window loads ok" in r.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not found")
def test_the_harness_traps_the_round_3_behaviour(tmp_path):
    """The same harness with "the block is the input's first one" as the only special case (what the kernels did until round 4) must die
    on its first read before the buffer — otherwise the test above proves nothing."""
    src = open(SRC, encoding="utf-8").read()
    old = src.replace("w3::window_head(off, 3u)", "(off == 0 ? 0u : W3_NO_HEAD)").replace("w3::window_head(off, 7u)", "(off == 0 ? 0u : W3_NO_HEAD)")
    old = old.replace('"../../weath3rb0i_amd/csrc/w3_window.h"', '"%s"' % os.path.join(ROOT, "weath3rb0i_amd", "csrc", "w3_window.h"))
    assert old != src
    p = tmp_path / "window_old.cpp"
    p.write_text(old, encoding="utf-8")
    exe = str(tmp_path / "window_old")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, str(p)])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
