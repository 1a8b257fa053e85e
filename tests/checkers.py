"""The two CPU checkers the parity tests compare against.

  okicp : oracle/kicp_oracle.cpp, the dependency-free restatement (always available).
  rkicp : oracle/_ref/libkicp_ref.so, the reference's own Registration.cpp / CorrespondenceThreshold.cpp /
          KinematicICP.cpp compiled unmodified against oracle/ref_shim (built where /root/reference exists, i.e. in the
          build container; the prebuilt library travels to the GPU box).
`ref()` returns the rkicp module, or skips the calling test with a clear message when the library is neither present
nor buildable.  tests/golden/ref_outputs.npz carries outputs of that library, so the reference's results are pinned
even where it is absent.
"""
import os

import numpy as np
import pytest

from oracle import okicp, rkicp  # noqa: F401

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def ref_available():
    return rkicp.available()


def ref():
    if not rkicp.available():
        pytest.skip("oracle/_ref/libkicp_ref.so missing and /root/reference not present to build it")
    return rkicp


def ref_map_like(omap):
    """A reference-build VoxelHashMap holding the same points as the oracle map `omap`, inserted in an order that
    reproduces every voxel's internal point order (Pointcloud() lists voxels one after another)."""
    m = ref().VoxelHashMap(omap.voxel_size_, omap.max_distance_, omap.max_points_per_voxel_)
    m.AddPoints(omap.Pointcloud())
    return m


def assert_pose_close(a, b, atol):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=0, atol=atol)
