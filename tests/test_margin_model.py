"""The fp32 pre-selection of the default pass kernel (kicp_kernels.hpp, k_pass_gather32) is only allowed to decide what it
can decide safely: two candidates are told apart in fp32 only if their fp32 keys differ by more than `margin`.  That is
sound iff every fp32 key is within margin/2 of the true squared distance.  This test re-enacts the kernel's fp32 arithmetic
in numpy (mirror offsets from the voxel corner, query offset seen from the neighbour's corner, fma-accumulated squares,
5 mantissa bits dropped for the integer tournament) on random queries / map points over the whole 27-voxel neighbourhood
and checks the documented error model - including voxel sizes that are not representable in fp32 and thresholds larger
than a voxel."""
import numpy as np
import pytest


def f32(x):
    return np.asarray(x, dtype=np.float64).astype(np.float32)


def fma32(a, b, c):
    """round_to_f32(a * b + c): the product of two floats is exact in float64; the sum's float64 rounding is far below
    the float32 rounding that follows."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def kernel_key_value(q, p, vs):
    """fp32 squared distance as the kernel's tournament sees it, for query q and map point p (fp64 world coordinates)."""
    qv, pv = np.floor(q / vs), np.floor(p / vs)
    fvs = np.float32(vs)
    l = f32(q - qv * vs)                                  # lx, ly, lz
    d = (pv - qv).astype(np.float32)                      # shift components in {-1, 0, 1}
    qrel = (l - d * fvs).astype(np.float32)               # the query as seen from that voxel's corner
    c = f32(p - pv * vs)                                  # mirror offset (kicp_host_map.hpp store32 / k_up_apply)
    dd = (c - qrel).astype(np.float32)
    d2 = fma32(dd[:, 2], dd[:, 2], fma32(dd[:, 1], dd[:, 1], (dd[:, 0] * dd[:, 0]).astype(np.float32)))
    bits = d2.view(np.uint32) & np.uint32(0xFFFFFFE0)     # 5 mantissa bits make room for the position
    return bits.view(np.float32).astype(np.float64)


@pytest.mark.parametrize("vs", [1.0, 0.5, 0.2, 0.1, 0.37, 2.5])
def test_fp32_keys_stay_within_the_documented_error_model(vs):
    rng = np.random.Generator(np.random.PCG64(int(vs * 1000)))
    n = 400_000
    base = rng.integers(-2000, 2000, (n, 3)).astype(np.float64)           # voxels up to 2000 voxel sizes from the origin
    q = (base + rng.uniform(0, 1, (n, 3))) * vs
    shift = rng.integers(-1, 2, (n, 3)).astype(np.float64)
    p = (np.floor(q / vs) + shift + rng.uniform(0, 1, (n, 3))) * vs         # anywhere in the 27-voxel neighbourhood
    same = np.all(np.floor(p / vs) == np.floor(q / vs) + shift, axis=1)     # (guard against rounding across a voxel border)
    q, p = q[same], p[same]
    D = np.sum((p - q) ** 2, axis=1)
    key = kernel_key_value(q, p, vs)
    err = np.abs(key - D)
    model = 1.25e-6 * np.sqrt(D) * vs + 4.2e-6 * D                          # per-candidate share of the kernel's margin
    assert np.all(err <= model + 1e-300), "worst ratio %.3f" % float(np.max(err / np.maximum(model, 1e-300)))
    # the kernel's margin for an acceptance bound B covers two such errors with 10 % to spare, for every D <= B
    for tau in (0.3 * vs, 0.6708 * vs, 1.5 * vs, 5.0 * vs):
        B = min(tau * tau, 12.0 * vs * vs)
        margin = max(8e-6 * vs * vs, 2.2 * (1.25e-6 * np.sqrt(B) * vs + 4.2e-6 * B))
        inside = D <= B
        assert np.all(2.0 * err[inside] <= margin)


def test_keys_order_like_floats_and_carry_the_position():
    """(bits & ~31) | position: unsigned order = float order for non-negative values; equal values order by position."""
    rng = np.random.Generator(np.random.PCG64(5))
    d = rng.uniform(0, 12, 4096).astype(np.float32)
    d[100:120] = d[100]                                                     # ties
    pos = np.arange(4096, dtype=np.uint32) % 20
    key = (d.view(np.uint32) & np.uint32(0xFFFFFFE0)) | pos
    order = np.argsort(key, kind="stable")
    trunc = (d.view(np.uint32) & np.uint32(0xFFFFFFE0)).view(np.float32)
    assert np.all(np.diff(trunc[order]) >= 0)
    tie = order[np.isin(order, np.arange(100, 120))]
    assert np.array_equal(pos[tie], np.sort(pos[tie]))                      # among equals the lower position comes first
    assert np.all((key & np.uint32(31)) == pos)


@pytest.mark.parametrize("vs", [1.0, 0.1, 0.37])
def test_face_bounds_never_exceed_the_true_distance_to_a_neighbour_voxel(vs):
    """Culling visits neighbour voxel c only if box_c <= current minimum + margin, with box_c built from the (rounded-down)
    squared distances to the own voxel's faces.  Sound iff box_c never exceeds the squared distance of ANY point of voxel c."""
    rng = np.random.Generator(np.random.PCG64(11))
    n = 300_000
    base = rng.integers(-2000, 2000, (n, 3)).astype(np.float64)
    q = (base + rng.uniform(0, 1, (n, 3))) * vs
    # queries hugging faces, edges and corners are the critical ones
    hug = rng.random((n, 3)) < 0.3
    q = np.where(hug, (base + rng.choice([1e-7, 1 - 1e-7, 1e-3, 1 - 1e-3], (n, 3))) * vs, q)
    shift = rng.integers(-1, 2, (n, 3))
    p = (np.floor(q / vs) + shift + rng.uniform(0, 1, (n, 3))) * vs
    same = np.all(np.floor(p / vs) == np.floor(q / vs) + shift, axis=1)
    q, p, shift = q[same], p[same], shift[same]
    fvs = np.float32(vs)
    B = min((0.6708 * vs) ** 2, 12.0 * vs * vs)
    margin = np.float32(max(8e-6 * vs * vs, 2.2 * (1.25e-6 * np.sqrt(B) * vs + 4.2e-6 * B)))
    l = f32(q - np.floor(q / vs) * vs)
    lo = (l * l).astype(np.float32) * np.float32(0.99999) - margin          # face[a][0]
    hi = ((fvs - l) * (fvs - l)).astype(np.float32) * np.float32(0.99999) - margin  # face[a][2]
    comp = np.where(shift < 0, lo, np.where(shift > 0, hi, np.float32(0.0))).astype(np.float32)
    box = (comp[:, 0] + comp[:, 1]).astype(np.float32) + comp[:, 2]
    D = np.sum((p - q) ** 2, axis=1)
    assert np.all(box.astype(np.float64) <= D)
