"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol include/kicp.h declares,
the host voxel map (kinematic_icp_amd/csrc/kicp_host_map.hpp, behind kicp_map_*) reproduces the oracle's
kiss_icp::VoxelHashMap semantics, and the device entry points fail LOUDLY (no CPU fallback) without a GPU.
No kernel is launched here."""
import os
import re
import subprocess

import numpy as np
import pytest

import kinematic_icp_amd as K
from conftest import ROOT, sort_rows
from kinematic_icp_amd import synthetic as syn
from oracle import okicp


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "kicp.h")).read()
    declared = set(re.findall(r"\b(kicp_[a-z0-9_]+)\s*\(", hdr)) - {"kicp_allreduce_fn"}
    assert len(declared) >= 30
    out = subprocess.check_output(["nm", "-D", "--defined-only", K.LIB_PATH], text=True)
    exported = set(re.findall(r"\b(kicp_[a-z0-9_]+)\b", out))
    assert declared <= exported, sorted(declared - exported)
    assert declared == set(K._SIGNATURES), sorted(declared ^ set(K._SIGNATURES))
    lib = K.lib()
    assert lib.kicp_version() == 100


def test_library_contains_gfx950_code_and_no_oracle():
    out = subprocess.run(["strings", "-n", "6", K.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out
    assert "okicp_" not in out  # the product never links the oracle
    src = "".join(open(os.path.join(ROOT, "kinematic_icp_amd", f)).read() for f in ("__init__.py", "synthetic.py"))
    assert "oracle" not in src.replace("the oracle", "").replace("oracle's", "") or "import oracle" not in src
    for f in (x for x in os.listdir(os.path.join(ROOT, "kinematic_icp_amd", "csrc")) if x.endswith((".hpp", ".hip", "Makefile"))):
        assert "oracle/" not in open(os.path.join(ROOT, "kinematic_icp_amd", "csrc", f)).read()


def test_every_aql_kernel_name_exists_in_the_embedded_code_object():
    """The direct-dispatch path resolves its kernels by demangled name in the code object embedded in the library
    (kicp_aql.hpp).  Every name the host code can ask for must be a kernel of build/kicp_reg.hsaco - and must need no
    scratch memory, which the path refuses - so a compiler or signature change fails HERE instead of silently turning
    every dispatch into a HIP launch."""
    import ctypes as C
    lib = K.lib()
    need = lib.kicp_aql_kernel_names(None, 0)
    buf = C.create_string_buffer(need)
    lib.kicp_aql_kernel_names(buf, need)
    wanted = [n for n in buf.value.decode().split("\n") if n]
    assert len(wanted) >= 11 and any("k_pass_small" in n for n in wanted)
    hsaco = os.path.join(ROOT, "kinematic_icp_amd", "csrc", "build", "kicp_reg.hsaco")
    assert os.path.exists(hsaco), "build/kicp_reg.hsaco missing: run __graft_entry__.build()"
    llvm = "/opt/rocm/lib/llvm/bin"
    syms = subprocess.check_output([os.path.join(llvm, "llvm-readelf"), "--symbols", "--wide", hsaco], text=True)
    descriptors = [ln.split()[-1] for ln in syms.splitlines() if ln.strip().endswith(".kd")]
    assert descriptors
    demangled = sorted(set(subprocess.check_output(["c++filt"], input="\n".join(d[:-3] for d in descriptors), text=True).splitlines()))  # (.dynsym and .symtab list each)
    for name in wanted:
        assert sum(d.startswith(name) for d in demangled) == 1, name
    # no kernel the path dispatches may use scratch: .private_segment_fixed_size == 0 in the code object's metadata
    notes = subprocess.check_output([os.path.join(llvm, "llvm-readelf"), "--notes", hsaco], text=True)
    blocks = notes.split("- .agpr_count")[1:]
    checked = 0
    for b in blocks:
        m = re.search(r"\.name:\s+(\S+)", b)
        # (only what the path can dispatch: the EXPORT instantiations - kicp_pass_correspondences - are launched through HIP and may spill)
        if m and any(subprocess.check_output(["c++filt", m.group(1)], text=True).strip().startswith(w) for w in wanted):
            assert re.search(r"\.private_segment_fixed_size:\s+0\b", b), m.group(1)
            checked += 1
    assert checked >= len(wanted)


def test_no_cpu_fallback_without_a_gpu():
    if K.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(K.KicpError) as e:
        K.KinematicRegistration()
    assert e.value.code == K.KICP_ERR_HIP and "no CPU fallback" in str(e.value)
    m = K.VoxelHashMap(1.0, 100.0, 20)
    m.AddPoints(np.random.default_rng(0).uniform(-3, 3, (50, 3)))
    with pytest.raises(K.KicpError):
        m.GetClosestNeighbor(np.zeros((1, 3)))  # the search runs on the device or not at all


def test_argument_and_capacity_errors():
    with pytest.raises(K.KicpError) as e:
        K.VoxelHashMap(1.0, 100.0, 65536)
    assert e.value.code == K.KICP_ERR_CAPACITY
    K.VoxelHashMap(1.0, 100.0, 256), K.VoxelHashMap(1.0, 100.0, 65535)  # (the reference's field is a plain unsigned int)
    with pytest.raises(K.KicpError) as e:
        K.VoxelHashMap(0.0, 100.0, 20)
    assert e.value.code == K.KICP_ERR_ARG


def test_host_map_equals_oracle_map():
    rng = np.random.default_rng(21)
    pts = rng.normal(0, 6, (40000, 3)) * np.array([1, 1, 0.2])
    for vs, cap, md in ((1.0, 20, 100.0), (0.3, 7, 12.0), (2.5, 1, 9.0)):
        g, o = K.VoxelHashMap(vs, md, cap), okicp.VoxelHashMap(vs, md, cap)
        assert g.Empty() and o.Empty()
        g.AddPoints(pts[:15000]), o.AddPoints(pts[:15000])
        assert (g.num_points(), g.num_voxels()) == (o.num_points(), o.num_voxels())
        np.testing.assert_array_equal(sort_rows(g.Pointcloud()), sort_rows(o.Pointcloud()))
        for k in range(5):  # a short trajectory of Update(points, pose): transform + add + prune
            pose = syn.planar_pose(2.0 * k, -1.0 * k, 0.3 * k)
            chunk = pts[15000 + 5000 * k: 20000 + 5000 * k]
            g.Update(chunk, pose), o.Update(chunk, pose)
            assert (g.num_points(), g.num_voxels()) == (o.num_points(), o.num_voxels())
        np.testing.assert_array_equal(sort_rows(g.Pointcloud()), sort_rows(o.Pointcloud()))
        g.Update(pts[:100], np.array([50.0, 50.0, 0.0])), o.Update(pts[:100], np.array([50.0, 50.0, 0.0]))  # origin overload
        np.testing.assert_array_equal(sort_rows(g.Pointcloud()), sort_rows(o.Pointcloud()))
        g.Clear()
        assert g.Empty() and g.num_points() == 0 and g.Pointcloud().shape == (0, 3)
        g.AddPoints(pts[:10])  # usable after Clear (SetPose path, KinematicICP.hpp:86-90)
        assert g.num_points() > 0


def test_map_copy_is_deep():
    # the reference's VoxelHashMap is copyable: kicp_map_clone / the drop-in's copy constructor
    rng = np.random.default_rng(22)
    pts = rng.normal(0, 6, (6000, 3)) * np.array([1, 1, 0.2])
    g = K.VoxelHashMap(0.8, 50.0, 11)
    g.AddPoints(pts[:3000])
    c = g.copy()
    assert (c.voxel_size_, c.max_distance_, c.max_points_per_voxel_) == (0.8, 50.0, 11)
    np.testing.assert_array_equal(c.Pointcloud(), g.Pointcloud())  # same points in the same table order
    assert (c.num_points(), c.num_voxels(), c.check()) == (g.num_points(), g.num_voxels(), 0)
    before = g.Pointcloud()
    c.Update(pts[3000:], syn.planar_pose(1.0, 0.5, 0.2))  # the copy moves on ...
    np.testing.assert_array_equal(g.Pointcloud(), before)   # ... the original does not
    o = okicp.VoxelHashMap(0.8, 50.0, 11)
    o.AddPoints(pts[:3000]), o.Update(pts[3000:], syn.planar_pose(1.0, 0.5, 0.2))
    np.testing.assert_array_equal(sort_rows(c.Pointcloud()), sort_rows(o.Pointcloud()))
    g.Clear()
    assert c.num_points() == o.num_points() and g.Empty()


@pytest.mark.parametrize("cap", [255, 256, 1000, 5000])
def test_host_map_with_deep_buckets_equals_oracle(cap):
    """max_points_per_voxel beyond 255 (the count then takes 12 or 16 bits of the slot's value word): same points per voxel, in the
    same order, as the oracle's map; invariants hold."""
    rng = np.random.default_rng(cap)
    pts = rng.uniform(-1.5, 1.5, (30000, 3))
    g, o = K.VoxelHashMap(1.0, 100.0, cap), okicp.VoxelHashMap(1.0, 100.0, cap)
    g.AddPoints(pts), o.AddPoints(pts)
    assert (g.num_points(), g.num_voxels()) == (o.num_points(), o.num_voxels()) and g.num_points() > 27 * min(cap, 600)
    assert g.check() == 0
    np.testing.assert_array_equal(sort_rows(g.Pointcloud()), sort_rows(o.Pointcloud()))
    g.Update(pts[:100] + 500.0, np.array([500.0, 500.0, 500.0])), o.Update(pts[:100] + 500.0, np.array([500.0, 500.0, 500.0]))  # prunes the old voxels
    assert (g.num_points(), g.num_voxels()) == (o.num_points(), o.num_voxels()) and g.check() == 0


def test_host_map_insertion_order_inside_a_bucket():
    # the query's tie rule depends on insertion order, so Pointcloud must list a voxel's points in that order
    g = K.VoxelHashMap(1.0, 100.0, 20)
    p = np.array([[0.9, 0.1, 0.1], [0.1, 0.9, 0.1], [0.5, 0.5, 0.9], [0.1, 0.1, 0.5]])
    g.AddPoints(p)
    np.testing.assert_array_equal(g.Pointcloud(), p)


def test_empty_map_registration_needs_no_gpu():
    # Registration.cpp:157: the early-out is host arithmetic; it works even where no device exists
    lib = K.lib()
    m = K.VoxelHashMap(1.0, 100.0, 20)
    last, rel = syn.planar_pose(1.0, 2.0, 0.3), syn.planar_pose(0.5, 0.0, 0.1)
    expect = okicp.se3_mul(last, rel)
    if K.device_count() > 0:
        reg = K.KinematicRegistration()
        np.testing.assert_allclose(reg.ComputeRobotMotion(np.zeros((4, 3)), m, last, rel, 1.0), expect, atol=1e-15)
    else:
        # product pose composition (kicp_se3.hpp) == oracle's, checked through the Update(points, pose) path instead
        g, o = K.VoxelHashMap(1.0, 100.0, 20), okicp.VoxelHashMap(1.0, 100.0, 20)
        pts = np.random.default_rng(3).uniform(-5, 5, (300, 3))
        g.Update(pts, expect), o.Update(pts, expect)
        np.testing.assert_array_equal(sort_rows(g.Pointcloud()), sort_rows(o.Pointcloud()))
    assert lib.kicp_map_empty(m._h) == 1


def test_synthetic_generator_is_deterministic_and_sized():
    c1 = syn.make_case("cfg1", n_scans=2)
    c2 = syn.make_case("cfg1", n_scans=2)
    assert c1[2][0]["frame"].shape == (16384, 3)
    for a, b in zip(c1[2], c2[2]):
        np.testing.assert_array_equal(a["frame"], b["frame"])
        np.testing.assert_array_equal(a["last_pose"], b["last_pose"])
    assert syn.CONFIGS["cfg2"].n_points == 131072 and syn.CONFIGS["cfg4"].n_points == 1080 and syn.CONFIGS["cfg5"].n_points == 500000
    np.testing.assert_allclose(syn.CONFIGS["cfg2"].first_frame_tau(), 0.6708203932499369)
    # every ray returns (closed scene) and stays inside max_range
    r = np.linalg.norm(c1[2][0]["frame"], axis=1)
    assert np.isfinite(r).all() and r.max() < syn.CONFIGS["cfg1"].max_range
    # the initial guess is never exactly the truth (reference quirk F9) and the pose helpers agree with the oracle
    s = c1[2][0]
    guess = syn.pose_mul(s["last_pose"], s["rel_odom"])
    assert np.abs(guess - s["true_pose"]).max() > 1e-4
    np.testing.assert_allclose(guess, okicp.se3_mul(s["last_pose"], s["rel_odom"]), atol=1e-14)
    np.testing.assert_allclose(syn.pose_act(guess, s["frame"][:50]), okicp.se3_act(guess, s["frame"][:50]), atol=1e-12)


def test_table_invariants_under_random_updates():
    """Neighbour masks, bucket records, halo entries, the fp32 mirror and the counters stay consistent through random
    AddPoints / Update(pose) / RemovePointsFarFromLocation / Clear sequences, incl. re-hashes and bucket re-use."""
    rng = np.random.default_rng(123)
    for vs, cap, md in ((1.0, 20, 15.0), (0.25, 3, 6.0), (2.0, 1, 30.0)):
        g, o = K.VoxelHashMap(vs, md, cap), okicp.VoxelHashMap(vs, md, cap)
        assert g.check() == 0
        centre = np.zeros(3)
        for step in range(40):
            op = rng.integers(0, 10)
            pts = centre + rng.normal(0, md / 2, (int(rng.integers(1, 1500)), 3)) * np.array([1, 1, 0.15])
            if op < 5:
                g.AddPoints(pts), o.AddPoints(pts)
            elif op < 8:
                centre = centre + rng.normal(0, md / 4, 3) * np.array([1, 1, 0])
                pose = syn.planar_pose(centre[0], centre[1], rng.uniform(-3, 3))
                g.Update(pts - centre, pose), o.Update(pts - centre, pose)
            elif op < 9:
                g.RemovePointsFarFromLocation(centre), o.RemovePointsFarFromLocation(centre)
            else:
                g.Clear(), o.Clear()
            assert g.check() == 0, "step %d op %d" % (step, op)
            assert (g.num_points(), g.num_voxels()) == (o.num_points(), o.num_voxels())
        np.testing.assert_array_equal(sort_rows(g.Pointcloud()), sort_rows(o.Pointcloud()))


def test_bridge_parameter_order_and_layout(tmp_path):
    """The drop-in headers' Sophus::SE3d <-> C-ABI conversion (kicp_bridge.hpp), compiled against cpp/compat: one code path
    for the real libraries and the stand-ins (tests/cpp/bridge_test.cpp)."""
    import subprocess
    cpp = os.path.join(ROOT, "kinematic_icp_amd", "cpp")
    exe = str(tmp_path / "bridge_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I", cpp, "-I", os.path.join(cpp, "compat"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "bridge_test.cpp"), "-o", exe, "-Wl,--unresolved-symbols=ignore-all"])
    assert subprocess.run([exe], capture_output=True, text=True).stdout.strip() == "OK"


def test_compat_sophus_and_bridge_math_against_scipy(tmp_path):
    """The ONE set of Eigen / Sophus stand-ins in the tree (kinematic_icp_amd/cpp/compat: what the drop-in headers compile against
    where the real libraries are absent, and - since round 3 - also what the reference build of oracle/_ref is compiled against)
    pinned to an INDEPENDENT implementation: SE3 product, inverse, action, exp and log through kicp_bridge and the compat Sophus
    types against scipy.spatial.transform.Rotation and the closed-form V matrices of tests/ref_numpy.py - so an error in those
    formulas cannot hide behind the checker sharing them."""
    import subprocess
    import ref_numpy as rn
    cpp = os.path.join(ROOT, "kinematic_icp_amd", "cpp")
    exe = str(tmp_path / "bridge_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I", cpp, "-I", os.path.join(cpp, "compat"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "bridge_test.cpp"), "-o", exe, "-Wl,--unresolved-symbols=ignore-all"])
    rng = np.random.default_rng(12)

    def rand_pose():
        q = rng.normal(size=4)
        return np.concatenate([q / np.linalg.norm(q), rng.normal(0, 10, 3)])

    rows = []
    for scale in (1.0, 1.0, 1.0, 1e-3, 1e-9, 1e-12, 0.0):
        for _ in range(12):
            xi = rng.normal(size=6) * np.array([1, 1, 1, scale, scale, scale])
            xi[3:] *= min(1.0, 2.5 / max(np.linalg.norm(xi[3:]), 1e-300))
            rows.append(np.concatenate([rand_pose(), rand_pose(), xi, rng.normal(0, 5, 3)]))
    rows = np.array(rows)
    rows.tofile(tmp_path / "in.bin")
    out = np.frombuffer(subprocess.check_output([exe, "math", str(tmp_path / "in.bin")]), dtype=np.float64).reshape(len(rows), 36)

    def same_pose(p, T, tol):  # (a quaternion and its negative are the same rotation)
        want = rn.to_qt(T)
        if np.dot(p[:4], want[:4]) < 0:
            want = np.concatenate([-want[:4], want[4:]])
        np.testing.assert_allclose(p, want, rtol=0, atol=tol)

    for r, o in zip(rows, out):
        a, b, xi, pt = rn.from_qt(r[:7]), rn.from_qt(r[7:14]), r[14:20], r[20:23]
        np.testing.assert_allclose(o[0:3], rn.act(a, pt[None])[0], rtol=0, atol=1e-12)
        same_pose(o[3:10], rn.mul(a, b), 1e-12)
        same_pose(o[10:17], rn.inv(a), 1e-12)
        # (below its 1e-10 switch Sophus takes V = R, the closed form above it: either is within |omega| |upsilon| of the series)
        same_pose(o[17:24], rn.se3_exp(xi), 1e-12 if np.linalg.norm(xi[3:]) > 1e-6 else 1e-9)
        # log(exp(xi)) == xi (Sophus' closed-form V loses digits to cancellation for angles just above its 1e-10 Taylor switch: 5e-9)
        np.testing.assert_allclose(o[24:30], xi, rtol=0, atol=5e-9)
        np.testing.assert_allclose(o[30:36], rn.se3_log(a), rtol=0, atol=5e-9)  # (|omega| < pi: scipy's rotvec is the principal one too)


def test_drop_in_voxel_map_exposes_a_read_only_map_view(tmp_path):
    """kiss_icp::VoxelHashMap::map_ (the reference's public member) as a read-only view over the backend:
    tests/cpp/map_view_test.cpp (host map only: no GPU needed)."""
    import subprocess
    cpp = os.path.join(ROOT, "kinematic_icp_amd", "cpp")
    exe = str(tmp_path / "map_view_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I", cpp, "-I", os.path.join(cpp, "compat"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "map_view_test.cpp"), "-o", exe, K.LIB_PATH, "-Wl,-rpath," + os.path.dirname(K.LIB_PATH)])
    assert subprocess.run([exe], capture_output=True, text=True).stdout.strip().splitlines()[-1] == "OK"


@pytest.mark.parametrize("mutation", ["m.map_.clear();", "m.map_.erase(kiss_icp::Voxel(0, 0, 0));", "m.map_[kiss_icp::Voxel(0, 0, 0)].clear();",
                                      "m.map_.insert(std::make_pair(kiss_icp::Voxel(0, 0, 0), std::vector<Eigen::Vector3d>{}));", "m.map_.emplace(kiss_icp::Voxel(0, 0, 0), std::vector<Eigen::Vector3d>{});",
                                      "m.map_.reserve(10);"])
def test_writing_through_the_map_view_is_a_compile_time_error_that_names_the_way_out(tmp_path, mutation):
    """the reference's `map_` is a writable tsl::robin_map (KinematicICP.hpp:94-95 hands it out); here it is a read-only view, and a
    caller that writes it is told so by the compiler - not by a missing-member error, and not at run time (missing 4)"""
    import subprocess
    cpp = os.path.join(ROOT, "kinematic_icp_amd", "cpp")
    src = tmp_path / "mutate.cpp"
    src.write_text("#include <kiss_icp/core/VoxelHashMap.hpp>\nvoid f(kiss_icp::VoxelHashMap &m) { %s }\n" % mutation)
    flags = ["g++", "-std=c++17", "-fsyntax-only", "-I", cpp, "-I", os.path.join(cpp, "compat"), "-I", os.path.join(ROOT, "include")]
    r = subprocess.run(flags + [str(src)], capture_output=True, text=True)
    assert r.returncode != 0 and "read-only view in this backend" in r.stderr and "AddPoints / Update" in r.stderr, r.stderr[-2000:]
    src.write_text("#include <kiss_icp/core/VoxelHashMap.hpp>\nsize_t f(const kiss_icp::VoxelHashMap &m) { return m.map_.size() + m.map_.count(kiss_icp::Voxel(0, 0, 0)); }\n")
    subprocess.check_call(flags + [str(src)])  # (reading compiles as before)
