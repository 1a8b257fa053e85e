"""The per-correspondence terms of the pass kernels (kicp_kernels.hpp::correspondence_terms, to_fixed, basis_of) without a GPU.

The reference forms J = [R UnitX | R (-s.y, s.x, 0)] and r = T s - t per correspondence and adds J^T J and J^T r
(/root/reference/cpp/kinematic_icp/registration/Registration.cpp:86-93,108-113).  The kernels evaluate the closed form of SURVEY.md
App. B.1 instead - JTJ = [[1, -s.y], [-s.y, s.x^2 + s.y^2]], JTr = [c0 . r, s.x (c1 . r) - s.y (c0 . r)] with c0 = R UnitX, c1 = R UnitY -
while the oracle keeps the literal products.  Here: the two agree to 1e-12 (relative to the magnitudes involved) on random poses and
points, in the arithmetic both sides use (fp64, the quaternion rotation of kicp_se3.hpp); and the host build of to_fixed / basis_of
passes its own checks (tests/cpp/fixed_point_test.cpp).  The GPU parity suite then holds the kernels' sums to the oracle's."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def quat_rotate(q, p):
    """p + w (2 v x p) + v x (2 v x p): kicp_se3.hpp::quat_rotate, the form Sophus evaluates"""
    v, w = q[:3], q[3]
    u = 2.0 * np.cross(v, p)
    return p + w * u + np.cross(v, u)


def test_closed_form_equals_the_literal_products_to_1e_12():
    rng = np.random.default_rng(17)
    worst = 0.0
    for _ in range(2000):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        t = rng.normal(size=3) * 50.0
        s = rng.normal(size=3) * np.array([60.0, 60.0, 3.0])     # source point, base frame
        tgt = quat_rotate(q, s) + t + rng.normal(size=3) * 0.3    # a target within a few decimetres of T s
        # literal (Registration.cpp:86-93,108-113)
        r = quat_rotate(q, s) + t - tgt
        j0 = quat_rotate(q, np.array([1.0, 0.0, 0.0]))
        j1 = quat_rotate(q, np.array([-s[1], s[0], 0.0]))
        lit = np.array([j0 @ j0, j0 @ j1, j1 @ j1, j0 @ r, j1 @ r])
        # closed form (kicp_kernels.hpp::correspondence_terms)
        c0, c1 = j0, quat_rotate(q, np.array([0.0, 1.0, 0.0]))
        a, b = c0 @ r, c1 @ r
        closed = np.array([1.0, -s[1], s[0] * s[0] + s[1] * s[1], a, s[0] * b - s[1] * a])
        scale = np.array([1.0, np.hypot(s[0], s[1]), s[0] * s[0] + s[1] * s[1], np.linalg.norm(r), np.hypot(s[0], s[1]) * np.linalg.norm(r)])
        worst = max(worst, float(np.max(np.abs(lit - closed) / scale)))
    assert worst < 1e-12, worst


def test_fixed_point_split_and_basis_on_the_host(tmp_path):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not present")
    exe = str(tmp_path / "fixed_point_test")
    subprocess.check_call([hipcc, "-x", "hip", "--cuda-host-only", "-std=c++17", "-O1", "-ffp-contract=off", "-I", os.path.join(ROOT, "kinematic_icp_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "fixed_point_test.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert " 0 bad" in out.stdout
