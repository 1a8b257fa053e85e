"""VoxelHashMap::Update(points, pose) on the GPU (kicp_map_update_pose_device) against the sequential oracle AND the reference
build's own kiss_icp::VoxelHashMap (oracle/_ref, where present): a drive through a scene with pruning and bucket re-use; after
every frame the device-maintained map must hold exactly their points (per voxel, in the same order), satisfy the table
invariants, and give the same registration results."""
import numpy as np
import pytest

import kinematic_icp_amd as K
from conftest import sort_rows
from kinematic_icp_amd import synthetic as syn
from checkers import okicp, ref_available, rkicp

pytestmark = pytest.mark.gpu


class Checkers:
    """The oracle's map and - where oracle/_ref is present - the reference build's, driven in lock step: every query is answered
    by both and must agree bit for bit before the device result is compared with it."""

    def __init__(self, vs, max_range, cap):
        self.o = okicp.VoxelHashMap(vs, max_range, cap)
        self.r = rkicp.VoxelHashMap(vs, max_range, cap) if ref_available() else None

    def AddPoints(self, pts):
        self.o.AddPoints(pts)
        if self.r is not None:
            self.r.AddPoints(pts)

    def Update(self, pts, pose_or_origin):
        self.o.Update(pts, pose_or_origin)
        if self.r is not None:
            self.r.Update(pts, pose_or_origin)

    def Clear(self):
        self.o.Clear()
        if self.r is not None:
            self.r.Clear()

    def num_points(self):
        n = self.o.num_points()
        assert self.r is None or self.r.num_points() == n
        return n

    def num_voxels(self):
        n = self.o.num_voxels()
        assert self.r is None or self.r.num_voxels() == n
        return n

    def Pointcloud(self):
        pc = self.o.Pointcloud()
        if self.r is not None:  # (the two tables iterate in different orders: compare as sets)
            np.testing.assert_array_equal(sort_rows(pc), sort_rows(self.r.Pointcloud()))
        return pc

    def GetClosestNeighbor(self, q):
        nn, d = self.o.GetClosestNeighbor(q)
        if self.r is not None:
            nn_r, d_r = self.r.GetClosestNeighbor(q)
            assert np.array_equal(nn, nn_r) and np.array_equal(d, d_r)
        return nn, d


def first_seen_downsample(pts, vs):
    keys = np.floor(pts / vs).astype(np.int64)
    _, first = np.unique(keys, axis=0, return_index=True)
    return pts[np.sort(first)]


@pytest.mark.parametrize("vs,max_range", [(0.5, 12.0), (1.0, 25.0)])
def test_device_update_equals_sequential_reference(vs, max_range):
    rng = np.random.Generator(np.random.PCG64(7))
    scene = syn.make_scene(rng, half=30.0, height=5.0, n_boxes=14, box_xy=(2.0, 6.0), box_z=(1.5, 4.0), keep_clear=3.0)
    dirs = syn.beam_directions(16, 512, (-22.0, 6.0))
    gmap, omap = K.VoxelHashMap(vs, max_range, 20), Checkers(vs, max_range, 20)
    reg, oreg = K.KinematicRegistration(), okicp.KinematicRegistration()
    pose = syn.planar_pose(-18.0, -15.0, 0.6)
    on_device = 0
    for k in range(16):
        step = syn.planar_pose(1.8, 0.0, np.deg2rad(3.0))
        true_next = syn.pose_mul(pose, step)
        scan = syn.make_scan(scene, true_next, dirs, 1.0, rng)
        scan = first_seen_downsample(scan[np.linalg.norm(scan, axis=1) < max_range], 0.5 * vs)  # what the pipeline feeds
        if k > 0:
            rel = syn.pose_mul(step, syn.planar_pose(0.05, 0.0, np.deg2rad(0.4)))
            a = reg.ComputeRobotMotion(scan, gmap, pose, rel, 3 * vs / np.sqrt(20))
            b = oreg.ComputeRobotMotion(scan, omap.o, pose, rel, 3 * vs / np.sqrt(20))
            assert reg.last_stats.iterations == oreg.last_stats.iterations
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-9)
            if omap.r is not None:  # the reference's Registration.cpp on the reference build's map
                np.testing.assert_allclose(a, rkicp.KinematicRegistration().ComputeRobotMotion(scan, omap.r, pose, rel, 3 * vs / np.sqrt(20)), rtol=0, atol=1e-9)
        on_device += int(gmap.UpdateDevice(K.DeviceFrame(scan), true_next))
        omap.Update(scan, true_next)
        assert (gmap.num_points(), gmap.num_voxels()) == (omap.num_points(), omap.num_voxels()), "frame %d" % k
        if k % 5 == 4 or k == 15:
            pc_device = gmap.Pointcloud()  # gathered on the GPU while the HBM copy is the newer one (no table download)
            assert len(pc_device) == omap.num_points()
            np.testing.assert_array_equal(sort_rows(pc_device), sort_rows(omap.Pointcloud()))
            assert gmap.check() == 0, "frame %d" % k  # a host-side view: forces the download of the device state
            np.testing.assert_array_equal(gmap.Pointcloud(), pc_device)  # host copy, same table -> same order
            assert np.array_equal(gmap.Pointcloud()[:7], pc_device[:7])
            q = scan[:500] + rng.normal(0, 0.2, (min(500, len(scan)), 3))
            q = okicp.se3_act(true_next, q)
            nn_g, d_g = gmap.GetClosestNeighbor(q)
            nn_o, d_o = omap.GetClosestNeighbor(q)
            assert np.array_equal(d_g, d_o) and np.array_equal(nn_g, nn_o)  # incl. the order inside every bucket (tie rule)
        pose = true_next
    assert on_device >= 10  # after the first growth steps the updates stay on the GPU
    assert omap.num_points() < 16 * len(scan)  # the sliding window dropped old voxels: freed buckets were re-used


def test_mixed_host_and_device_updates():
    rng = np.random.default_rng(5)
    g, o = K.VoxelHashMap(1.0, 40.0, 20), Checkers(1.0, 40.0, 20)
    pts = rng.normal(0, 8, (30000, 3)) * np.array([1, 1, 0.2])
    g.AddPoints(pts[:20000]), o.AddPoints(pts[:20000])
    for k in range(6):
        chunk = pts[20000 + 1500 * k: 21500 + 1500 * k]
        pose = syn.planar_pose(1.0 * k, -0.5 * k, 0.2 * k)
        if k % 2 == 0:
            g.UpdateDevice(K.DeviceFrame(chunk), pose)
        else:
            g.Update(chunk, pose)  # host path after a device update: needs the download first
        o.Update(chunk, pose)
        assert (g.num_points(), g.num_voxels()) == (o.num_points(), o.num_voxels())
    assert g.check() == 0
    np.testing.assert_array_equal(sort_rows(g.Pointcloud()), sort_rows(o.Pointcloud()))
    g.Clear()
    assert g.Empty() and g.num_points() == 0
    assert g.UpdateDevice(K.DeviceFrame(pts[:100]), syn.IDENTITY) in (True, False)
    o.Clear(), o.Update(pts[:100], syn.IDENTITY)
    np.testing.assert_array_equal(sort_rows(g.Pointcloud()), sort_rows(o.Pointcloud()))


def test_bulk_host_calls_insert_on_the_device():
    """kicp_map_set_device: AddPoints / Update(points, origin) / Update(points, pose) from host arrays with many points go
    through HBM; the map must equal the sequential reference's point by point (per voxel, in order), small calls stay on
    the host, and host / bulk calls mix freely."""
    rng = np.random.default_rng(11)
    pts = rng.normal(0, 9, (60000, 3)) * np.array([1, 1, 0.15])
    g, h, o = K.VoxelHashMap(1.0, 30.0, 20, device=0), K.VoxelHashMap(1.0, 30.0, 20), Checkers(1.0, 30.0, 20)
    g.AddPoints(pts[:30000]), h.AddPoints(pts[:30000]), o.AddPoints(pts[:30000])      # bulk / host / oracle
    assert lib_last_on_device(g) and not lib_last_on_device(h)
    g.AddPoints(pts[30000:30100]), h.AddPoints(pts[30000:30100]), o.AddPoints(pts[30000:30100])  # small: host path (after a download)
    g.Update(pts[30100:45000], np.array([4.0, -3.0, 0.0])), h.Update(pts[30100:45000], np.array([4.0, -3.0, 0.0]))
    o.Update(pts[30100:45000], np.array([4.0, -3.0, 0.0]))                              # AddPoints + pruning around an origin
    pose = syn.planar_pose(2.0, 1.0, 0.3)
    g.Update(pts[45000:], pose), h.Update(pts[45000:], pose), o.Update(pts[45000:], pose)
    for m in (g, h):
        assert (m.num_points(), m.num_voxels()) == (o.num_points(), o.num_voxels())
        assert m.check() == 0
        np.testing.assert_array_equal(sort_rows(m.Pointcloud()), sort_rows(o.Pointcloud()))
    q = pts[::37] + rng.normal(0, 0.2, (len(pts[::37]), 3))
    nn_g, d_g = g.GetClosestNeighbor(q)
    nn_o, d_o = o.GetClosestNeighbor(q)
    assert np.array_equal(d_g, d_o) and np.array_equal(nn_g, nn_o)                      # incl. the order inside every bucket


def lib_last_on_device(m):
    return bool(K.lib().kicp_map_last_update_on_device(m._h))


def test_both_apply_kernels_build_the_reference_map():
    """The insertion step has two kernels (kicp_mapdev.hpp 4 / 4b): a wave per touched voxel for frame-sized updates, a
    thread per voxel beyond 16384 touched voxels.  One large update (> 16384 voxels, many points per voxel, buckets that
    fill up) and a series of small ones on top, against the sequential oracle - points in bucket order, not just as a set."""
    rng = np.random.default_rng(31)
    big = rng.uniform(-40, 40, (400000, 3)) * np.array([1, 1, 0.005])     # ~26k-50k voxels of 0.5 m, ~10 points offered to each
    g, o = K.VoxelHashMap(0.5, 200.0, 12, device=0), Checkers(0.5, 200.0, 12)
    ident = np.array([0.0, 0, 0, 1, 0, 0, 0])
    assert g.UpdateDevice(K.DeviceFrame(big), ident)
    o.Update(big, ident)
    assert g.num_voxels() == o.num_voxels() > 16384 and g.num_points() == o.num_points()
    for k in range(6):                                                    # frame-sized updates: the wave-per-voxel kernel
        pose = syn.planar_pose(0.7 * k, -0.3 * k, 0.05 * k)
        small = rng.uniform(-25, 25, (6000, 3)) * np.array([1, 1, 0.005])
        assert g.UpdateDevice(K.DeviceFrame(small), pose)
        o.Update(small, pose)
        assert (g.num_points(), g.num_voxels()) == (o.num_points(), o.num_voxels()), k
    q = rng.uniform(-30, 30, (4000, 3)) * np.array([1, 1, 0.005])
    nn_g, d_g = g.GetClosestNeighbor(q)
    nn_o, d_o = o.GetClosestNeighbor(q)
    assert np.array_equal(nn_g, nn_o) and np.array_equal(d_g, d_o)   # the tie rule sees the buckets' internal order
    assert np.array_equal(sort_rows(g.Pointcloud()), sort_rows(o.Pointcloud())) and g.check() == 0


def test_update_in_two_halves_equals_the_one_call_update():
    """kicp_map_update_pose_device_begin / kicp_map_update_finish (what the drop-in RegisterFrame uses to overlap the map update with
    collecting its results): the same map as the one-call update and as the sequential reference, whether the caller finishes it
    explicitly, forgets to (the next call on the map collects it), registers against the map or clears it in between; a far point
    met by a deferred update still ends in the host map's result."""
    rng = np.random.Generator(np.random.PCG64(11))
    scene = syn.make_scene(rng, half=30.0, height=5.0, n_boxes=14, box_xy=(2.0, 6.0), box_z=(1.5, 4.0), keep_clear=3.0)
    dirs = syn.beam_directions(16, 512, (-22.0, 6.0))
    vs, max_range = 1.0, 25.0
    two, one, omap = K.VoxelHashMap(vs, max_range, 20), K.VoxelHashMap(vs, max_range, 20), Checkers(vs, max_range, 20)
    pose = syn.planar_pose(-18.0, -15.0, 0.6)
    reg, oreg = K.KinematicRegistration(), okicp.KinematicRegistration()
    deferred = 0
    for k in range(14):
        true_next = syn.pose_mul(pose, syn.planar_pose(1.8, 0.0, np.deg2rad(3.0)))
        scan = syn.make_scan(scene, true_next, dirs, 1.0, rng)
        scan = first_seen_downsample(scan[np.linalg.norm(scan, axis=1) < max_range], 0.5 * vs)
        frame = K.DeviceFrame(scan)
        two.UpdateDeviceBegin(frame, true_next)
        if k % 3 == 0:
            deferred += int(two.UpdateFinish())
        elif k % 3 == 1 and k > 1:  # a registration right behind the begin: it must see the finished update
            rel = syn.planar_pose(0.03, 0.0, np.deg2rad(0.3))
            a = reg.ComputeRobotMotion(scan, two, true_next, rel, 3 * vs / np.sqrt(20))
        one.UpdateDevice(frame, true_next)
        omap.Update(scan, true_next)
        if k % 3 == 1 and k > 1:
            b = oreg.ComputeRobotMotion(scan, omap.o, true_next, rel, 3 * vs / np.sqrt(20))
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-9)
        assert (two.num_points(), two.num_voxels()) == (one.num_points(), one.num_voxels()) == (omap.num_points(), omap.num_voxels()), "frame %d" % k
        pose = true_next
    np.testing.assert_array_equal(sort_rows(two.Pointcloud()), sort_rows(omap.Pointcloud()))
    assert two.check() == 0 and deferred >= 3
    # cleared while an update is pending: the map is empty afterwards, and usable
    two.UpdateDeviceBegin(frame, true_next)
    two.Clear()
    assert two.Empty() and two.num_points() == 0
    two.UpdateDeviceBegin(frame, true_next)
    assert two.UpdateFinish() and two.num_points() > 0
    # a point beyond the packed keys' range in a deferred update: the host map takes it over at the finish
    far = np.concatenate([scan[:200], [[2.0e6, 0.0, 0.0]]])
    h, ho = K.VoxelHashMap(vs, 1e7, 20), okicp.VoxelHashMap(vs, 1e7, 20)
    h.UpdateDevice(K.DeviceFrame(scan), okicp.IDENTITY), ho.Update(scan, okicp.IDENTITY)
    ff = K.DeviceFrame(far)
    h.UpdateDeviceBegin(ff, okicp.IDENTITY)
    assert not h.UpdateFinish()
    ho.Update(far, okicp.IDENTITY)
    assert (h.num_points(), h.num_voxels()) == (ho.num_points(), ho.num_voxels())
