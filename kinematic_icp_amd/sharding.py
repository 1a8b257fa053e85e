"""Host-side pieces of the multi-GPU mode: how a scan is split across ranks and how the
per-rank sums travel through the all-reduce.

The device accumulates each of the seven sums exactly: every per-correspondence term is rounded once to a multiple
of 2^-40 and added as a 128-bit integer (kicp_kernels.hpp).  For the collective the 128-bit total T is split into
three signed 64-bit limbs, T = l0 + l1*2^40 + l2*2^80 (0 <= l0,l1 < 2^40), so that a plain int64 sum-all-reduce of
24 words (21 limbs + range flag + padding) is exact for any number of ranks and any reduction order.  These helpers
are the reference of that encoding; tests/test_sharding.py drives them through a real world_size-2 gloo all-reduce.
"""
import numpy as np

FIX_BITS = 40
LIMB_MASK = (1 << FIX_BITS) - 1
NUM_SUMS = 7
REDUCE_WORDS = 24


def shard_bounds(n, world, rank):
    """Contiguous, near-equal shards: rank r owns points [lo, hi)."""
    return (n * rank) // world, (n * (rank + 1)) // world


def quantize(x):
    """float -> fixed-point integer with 40 fractional bits, round-to-nearest-even (device: __double2ll_rn)."""
    return int(np.rint(np.float64(x) * np.float64(1 << FIX_BITS)))


def to_limbs(t):
    """python int (128-bit two's-complement range) -> [l0, l1, l2]."""
    return [t & LIMB_MASK, (t >> FIX_BITS) & LIMB_MASK, t >> (2 * FIX_BITS)]


def from_limbs(l):
    return int(l[0]) + (int(l[1]) << FIX_BITS) + (int(l[2]) << (2 * FIX_BITS))


def pack(totals):
    """seven fixed-point integer totals -> int64[24] all-reduce payload."""
    words = np.zeros(REDUCE_WORDS, dtype=np.int64)
    for i, t in enumerate(totals):
        words[3 * i:3 * i + 3] = to_limbs(int(t))
    return words


def unpack(words):
    """int64[24] (after the all-reduce) -> seven float64 sums."""
    return np.array([from_limbs(words[3 * i:3 * i + 3]) / float(1 << FIX_BITS) for i in range(NUM_SUMS)], dtype=np.float64)
