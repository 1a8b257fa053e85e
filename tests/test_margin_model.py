"""The pre-selection of the default pass kernel (kicp_kernels.hpp, k_pass_gather32) works on the 16-bit mirror
(kicp_common.hpp::MirrorPoint: offsets from the voxel corner in units of voxel_size / 65536) in fp32 and is only allowed to
decide what it can decide safely: two candidates are told apart there only if their keys differ by more than `margin`.
That is sound iff every key is within margin/2 of the true squared distance.  This test re-enacts the kernel's arithmetic
in numpy (quantised mirror coordinates, query offset in units seen from the neighbour's corner, fma-accumulated squares,
5 mantissa bits dropped for the integer tournament) on random queries / map points over the whole 27-voxel neighbourhood
and checks the documented error model (search_params() in kicp_kernels.hpp) - including voxel sizes that are not
representable in fp32, points hugging the voxel's upper faces (where the 16-bit range is clamped) and thresholds larger
than a voxel."""
import numpy as np
import pytest


def f32(x):
    return np.asarray(x, dtype=np.float64).astype(np.float32)


def fma32(a, b, c):
    """round_to_f32(a * b + c): the product of two floats is exact in float64; the sum's float64 rounding is far below
    the float32 rounding that follows."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def mirror_quant(offset, upm):
    """kicp_common.hpp::mirror_quant: round to the nearest unit, clamped into 16 bits"""
    return np.clip(np.floor(offset * upm + 0.5), 0.0, 65535.0)


def kernel_key_value(q, p, vs):
    """squared distance in METRES^2 as the kernel's tournament sees it (it works in units^2; converted back here), for query q
    and map point p (fp64 world coordinates)."""
    upm = 65536.0 / vs
    qv, pv = np.floor(q / vs), np.floor(p / vs)
    l = f32((q - qv * vs) * upm)                          # lx, ly, lz in units
    d = (pv - qv).astype(np.float32)                      # shift components in {-1, 0, 1}
    qrel = (l - d * np.float32(65536.0)).astype(np.float32)  # the query as seen from that voxel's corner
    c = mirror_quant(p - pv * vs, upm).astype(np.float32)  # mirror coordinate (kicp_host_map.hpp store32 / k_up_apply), exact in fp32
    dd = (c - qrel).astype(np.float32)
    d2 = fma32(dd[:, 2], dd[:, 2], fma32(dd[:, 1], dd[:, 1], (dd[:, 0] * dd[:, 0]).astype(np.float32)))
    bits = d2.view(np.uint32) & np.uint32(0xFFFFFFE0)     # 5 mantissa bits make room for the position
    return bits.view(np.float32).astype(np.float64) / (upm * upm)


def margin_of(tau, vs):
    """search_params(): the margin in metres^2"""
    B = min(tau * tau * (1.0 + 9.1e-13), 12.0 * vs * vs)
    return 2.2 * (5.55e-5 * np.sqrt(B) * vs + 4.2e-6 * B + 7.7e-10 * vs * vs), B


def sample(rng, n, vs, hug_fraction=0.2):
    base = rng.integers(-2000, 2000, (n, 3)).astype(np.float64)           # voxels up to 2000 voxel sizes from the origin
    q = (base + rng.uniform(0, 1, (n, 3))) * vs
    shift = rng.integers(-1, 2, (n, 3)).astype(np.float64)
    frac = rng.uniform(0, 1, (n, 3))
    hug = rng.random((n, 3)) < hug_fraction                                # points hugging the faces: the clamped top of the range
    frac = np.where(hug, rng.choice([0.0, 1e-9, 1 - 1e-9, 1 - 4e-6, 1 - 8e-6, 1 - 2e-5], (n, 3)), frac)
    p = (np.floor(q / vs) + shift + frac) * vs                              # anywhere in the 27-voxel neighbourhood
    same = np.all(np.floor(p / vs) == np.floor(q / vs) + shift, axis=1)     # (guard against rounding across a voxel border)
    return q[same], p[same]


@pytest.mark.parametrize("vs", [1.0, 0.5, 0.2, 0.1, 0.37, 2.5])
def test_mirror_keys_stay_within_the_documented_error_model(vs):
    rng = np.random.Generator(np.random.PCG64(int(vs * 1000)))
    q, p = sample(rng, 400_000, vs)
    D = np.sum((p - q) ** 2, axis=1)
    key = kernel_key_value(q, p, vs)
    err = np.abs(key - D)
    # per-candidate share of the kernel's margin: |delta| <= sqrt(3) 1.05 units, D off by 2 sqrt(D) |delta| + |delta|^2 + 4.2e-6 D
    model = 5.55e-5 * np.sqrt(D) * vs + 7.7e-10 * vs * vs + 4.2e-6 * D
    worst = float(np.max(err / np.maximum(model, 1e-300)))
    assert worst <= 1.0, "worst ratio %.3f" % worst
    assert worst > 0.3  # the model is not wildly pessimistic either
    # the kernel's margin for an acceptance bound B covers two such errors with 10 % to spare, for every D <= B
    for tau in (0.05 * vs, 0.3 * vs, 0.6708 * vs, 1.5 * vs, 5.0 * vs):
        margin, B = margin_of(tau, vs)
        inside = D <= B
        assert np.all(2.0 * err[inside] <= margin)


def test_mirror_coordinates_are_within_one_unit():
    """What the error model assumes about the quantisation itself (also asserted per point by kicp_map_check)."""
    rng = np.random.Generator(np.random.PCG64(3))
    for vs in (1.0, 0.1, 0.37):
        upm = 65536.0 / vs
        off = np.concatenate([rng.uniform(0, vs, 200_000), vs * (1 - rng.uniform(0, 3e-5, 50_000)), rng.uniform(0, 3e-5, 50_000) * vs])
        off = off[off < vs]
        qz = mirror_quant(off, upm)
        assert qz.min() >= 0 and qz.max() <= 65535
        assert np.max(np.abs(qz / upm - off)) * upm <= 1.0


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
    squared distances to the own voxel's faces, in mirror units.  Sound iff box_c never exceeds the squared distance of ANY
    point of voxel c (as the mirror represents it: the comparison partner is a mirror distance)."""
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
    upm = 65536.0 / vs
    margin_m, _ = margin_of(0.6708 * vs, vs)
    margin = np.float32(margin_m * upm * upm) * np.float32(1.00001)
    cell = np.float32(65536.0)
    l = f32((q - np.floor(q / vs) * vs) * upm)
    lo = (l * l).astype(np.float32) * np.float32(0.99999) - margin
    hi = ((cell - l) * (cell - l)).astype(np.float32) * np.float32(0.99999) - margin
    comp = np.where(shift < 0, lo, np.where(shift > 0, hi, np.float32(0.0))).astype(np.float32)
    box = (comp[:, 0] + comp[:, 1]).astype(np.float32) + comp[:, 2]
    D = np.sum((p - q) ** 2, axis=1) * upm * upm   # true squared distance, units^2
    assert np.all(box.astype(np.float64) <= D)
    key_u = kernel_key_value(q, p, vs) * upm * upm  # and what the kernel compares against
    assert np.all(box.astype(np.float64) <= key_u + float(margin))
