"""The C-ABI used from plain C (no Python, no torch types in between): tests/c/cabi_example.c is compiled with gcc against
include/hipdrt.h + libhipdrt.so and its printed results are compared with the CPU checker."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "cabi_example.c")
LIBDIR = os.path.join(ROOT, "hybrid-drt_amd")


def _compile(tmp_path):
    exe = os.path.join(tmp_path, "cabi_example")
    cmd = ["gcc", "-std=c99", "-Wall", SRC, "-I", os.path.join(ROOT, "include"), "-L", LIBDIR, "-lhipdrt",
           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_c_example_compiles_against_the_header(tmp_path):
    """CPU: the public header is valid C99 and every entry point the example uses links."""
    assert os.path.exists(_compile(str(tmp_path)))


@pytest.mark.gpu
def test_c_example_runs_and_matches_the_checker(tmp_path):
    from oracle import drt_oracle as orc
    from oracle.coneqp import coneqp_boxlow
    exe = _compile(str(tmp_path))
    env = dict(os.environ, LD_LIBRARY_PATH=LIBDIR + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe], check=True, capture_output=True, text=True, env=env).stdout.splitlines()
    assert out[0].split()[1].startswith("gfx950")
    P = [np.array([[4, 1, 0], [1, 3, 1], [0, 1, 2]], float), 2.0 * np.eye(3)]
    q = [np.array([1, -2, 1], float), np.array([-1, 1, -3], float)]
    for b in range(2):
        tok = out[1 + b].split()
        ref = coneqp_boxlow(P[b], q[b], np.zeros(3))
        assert int(tok[3]) == 0 and int(tok[5]) == ref["iterations"]
        np.testing.assert_allclose([float(t) for t in tok[7:10]], ref["x"], rtol=1e-9, atol=1e-12)
    freq, tau = np.array([1e3, 1e2, 1e1, 1e0]), np.array([1e-4, 1e-3, 1e-2, 1e-1, 1e0, 1e1])
    a_re = orc.construct_impedance_matrix(freq, 'real', tau, 0.43429448190325176, 'trapz')
    a_im = orc.construct_impedance_matrix(freq, 'imag', tau, 0.43429448190325176, 'trapz')
    got = np.array([[float(v) for v in line.split()[3:5]] for line in out[3:27]]).reshape(4, 6, 2)
    np.testing.assert_allclose(got[..., 0], a_re, rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(got[..., 1], a_im, rtol=1e-12, atol=1e-300)
    # potentiostatic response (mat1d.py:114-118): exp(-(t - t_k) / tau) * unit_step(t, t_k) * size_k summed over the steps
    times, ptau, st, sa = np.array([0.5, 1.5, 3.0]), np.array([0.5, 2.0]), (1.0, 2.0), (1e-3, -2e-3)
    want = sum(np.where(times[:, None] >= t_k, np.exp(-(times[:, None] - t_k) / ptau[None, :]) * s_k, 0.0) for t_k, s_k in zip(st, sa))
    pot = np.array([[float(v) for v in line.split()[2:4]] for line in out[27:30]])
    assert not pot[0].any()
    np.testing.assert_allclose(pot, want, rtol=1e-13, atol=0)
    assert out[30].split()[0] == "bytes_per_spectrum" and 4.7e6 < int(out[30].split()[1]) < 5.1e6
