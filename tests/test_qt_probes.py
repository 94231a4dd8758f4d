"""
The render_human rasterizer (procgen_amd/csrc/pg_human.h) restates three pieces of Qt 5.9's raster engine; tests/tools/qt_smooth_aa_probe.py
holds the same restatement in Python and compares it with the real thing, PyQt5 5.9.7 -- the Qt the compiled reference links.  This
test runs a short probe where that interpreter exists (the build container's /opt/conda); the kernels themselves are compared with the
compiled reference's frames in tests/test_render_human.py.
"""
import os
import subprocess

import pytest

CONDA_PY = "/opt/conda/bin/python3.9"
PROBE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "qt_smooth_aa_probe.py")


def _has_pyqt():
    if not os.path.exists(CONDA_PY):
        return False
    r = subprocess.run([CONDA_PY, "-c", "import PyQt5.QtCore as c; print(c.QT_VERSION_STR)"], capture_output=True, text=True)
    return r.returncode == 0 and r.stdout.strip().startswith("5.9")


@pytest.mark.skipif(not _has_pyqt(), reason="needs PyQt5 5.9 (the build container's /opt/conda)")
def test_antialiased_smooth_transform_restatement_equals_qt():
    r = subprocess.run([CONDA_PY, PROBE, "100", "2", "all"], capture_output=True, text=True, cwd="/tmp", timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-2000:]
    assert "100 cases, 0 with differences" in out, out[-2000:]                 # untransformed drawImage / fillRect
    assert "turned fills: 100 cases, 0 with differences" in out, out[-2000:]   # the antialiased trapezoid walker
    assert "turned images: 100 cases, 0 with differences" in out, out[-2000:]  # rotation branch of the bilinear fetch


GRAY = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "qt_gray_raster_probe.py")


@pytest.mark.skipif(not _has_pyqt(), reason="needs PyQt5 5.9 (the build container's /opt/conda)")
def test_antialiased_path_restatement_equals_qt():
    """jumper's compass under render_human (procgen_amd/csrc/pg_aapath.h): gray raster fill, antialiased cosmetic outline; and the
    square-cap needle (pg_human.h aa_wide_line)"""
    r = subprocess.run([CONDA_PY, GRAY, "80", "3"], capture_output=True, text=True, cwd="/tmp", timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-2000:]
    assert "filled antialiased ellipses: 80 cases, 0 with differences" in out, out[-2000:]
    assert "outlined antialiased ellipses (pen width 1): 80 cases, 0 with differences" in out, out[-2000:]
    r = subprocess.run([CONDA_PY, PROBE, "120", "2", "lines"], capture_output=True, text=True, cwd="/tmp", timeout=600)
    out = r.stdout + r.stderr
    assert "wide lines: 120 cases, 0 with differences" in out, out[-2000:]
    r = subprocess.run([CONDA_PY, PROBE, "60", "4", "generic"], capture_output=True, text=True, cwd="/tmp", timeout=600)
    assert "generic (ARGB32) sources: 60 cases, 0 with differences" in (r.stdout + r.stderr), (r.stdout + r.stderr)[-2000:]
