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


# ---- tagged rows of the hand-off (kicp_kernels.hpp finish_pass, mode 4; kicp_reg_launch.hip wait_rows) ----------------------
# Every word of a workgroup's / group's row carries the 16-bit tag of its pass in the low bits: word = value << 16 | tag.
# A reader knows a row has landed when all its words carry the current tag - no store acknowledgement, no fence.
TAG_BITS = 16
TAG_MASK = (1 << TAG_BITS) - 1
GROUP = 32  # workgroups per first-level group (kGroup)


def tag_row(words, tag):
    """int64[24] limb row -> uint64[24] tagged row (two's complement, like the device's `value << 16 | tag`)."""
    assert 1 <= tag <= TAG_MASK
    return np.array([((int(w) << TAG_BITS) | tag) & 0xFFFFFFFFFFFFFFFF for w in words], dtype=np.uint64)


def untag_row(row, tag):
    """uint64[24] -> (int64[24] values, landed?) : arithmetic shift, as the host and the group's reader do."""
    vals, ok = [], True
    for w in row:
        w = int(w)
        ok = ok and (w & TAG_MASK) == tag
        s = w - (1 << 64) if w >> 63 else w  # reinterpret as signed
        vals.append(s >> TAG_BITS)
    return np.array(vals, dtype=np.int64), ok


# ---- rows of the small-scan path (kicp_small.hpp small_publish / k_pass_wave; kicp_reg_launch.hip wait_rows_small) -----------------
# A workgroup's row is 16 tagged words = two 64-byte lines: per sum the low 48 bits and bits 48..95 of its 128-bit total (the
# upper half carries the sign), then the flag word (bit 0 range error, bit 1 "gave up"), then one spare word.
SMALL_ROW_WORDS = 16
HALF_BITS = 48
HALF_MASK = (1 << HALF_BITS) - 1


def small_row(totals, tag, flags=0):
    """seven fixed-point integer totals (|t| < 2^95) -> uint64[16] tagged row of the small-scan kernels."""
    assert 1 <= tag <= TAG_MASK
    words = []
    for t in totals:
        u = int(t) & ((1 << 128) - 1)  # two's complement
        words += [u & HALF_MASK, (u >> HALF_BITS) & HALF_MASK]
    words += [flags, 0]
    return np.array([((w << TAG_BITS) | tag) & 0xFFFFFFFFFFFFFFFF for w in words], dtype=np.uint64)


def add_small_rows(rows, tag):
    """what the host does with the rows of one pass: (int64[24] limb words of the totals, flags, all landed?)."""
    totals, flags, ok = [0] * NUM_SUMS, 0, True
    for row in rows:
        for w in row:
            ok = ok and (int(w) & TAG_MASK) == tag
        for i in range(NUM_SUMS):
            lo = int(row[2 * i]) >> TAG_BITS                    # logical shift: 48 unsigned bits
            hi = int(row[2 * i + 1])
            hi = (hi - (1 << 64) if hi >> 63 else hi) >> TAG_BITS  # arithmetic shift: 48 signed bits
            totals[i] += lo + (hi << HALF_BITS)
        flags |= int(row[2 * NUM_SUMS]) >> TAG_BITS
    return pack(totals), flags, ok


# ---- several SHARDED scans in flight: the lanes of the shared segment (kicp_reg_queues.hip run_batch_queues, `sharded`) ---------------
# A sharded batch call keeps `lanes` scans in flight on every rank.  Lane j registers scans j, j + lanes, j + 2 lanes, ... - the
# same deal on every rank, so lane j issues the same sequence of exchanges everywhere, whatever order the lanes' passes complete
# in on each rank.  Each lane owns an area [2 buffers][nranks] of slots (sequence word + 24 limb words) and counts its own
# hand-offs: hand-off s of a lane goes into buffer s & 1 with sequence s + 1; a rank may overwrite its slot of buffer s & 1 with
# hand-off s + 2 only after it has seen every rank's hand-off s + 1, which every rank publishes after reading hand-off s.
# This class is the reference of that protocol (tests/test_sharding.py runs it with eight processes); the library's loop is the
# same state machine around the GPU's rows.
SLOT_WORDS = 32  # 256-byte slots: sequence word, 24 limb words, padding


class SegmentLanes:
    def __init__(self, buffer, nranks, rank, lanes):
        self.slots = np.ndarray((lanes, 2, nranks, SLOT_WORDS), dtype=np.int64, buffer=buffer)
        self.nranks, self.rank, self.lanes = nranks, rank, lanes
        self.step = [0] * lanes

    @staticmethod
    def nbytes(nranks, lanes):
        return lanes * 2 * nranks * SLOT_WORDS * 8

    def publish(self, lane, words):
        """this rank's totals of the lane's next hand-off; returns the sequence value to collect"""
        step = self.step[lane]
        self.step[lane] += 1
        slot = self.slots[lane, step & 1, self.rank]
        slot[1:1 + REDUCE_WORDS] = words
        slot[0] = step + 1  # (last: a reader that sees the sequence word sees the words)
        return step + 1

    def collect(self, lane, value):
        """the sum over all ranks of hand-off `value` of the lane, or None while a rank's slot is missing (non-blocking)"""
        area = self.slots[lane, (value - 1) & 1]
        if not np.all(area[:, 0] == value):
            return None
        return area[:, 1:1 + REDUCE_WORDS].sum(axis=0)


    def collect_each(self, lane, value):
        """every rank's words of hand-off `value` of the lane, [nranks][REDUCE_WORDS], or None while a rank's slot is missing"""
        area = self.slots[lane, (value - 1) & 1]
        if not np.all(area[:, 0] == value):
            return None
        return area[:, 1:1 + REDUCE_WORDS].copy()


def part_bounds(part, parts, count):
    """the contiguous part `part` of a batch of `count` scans cut into `parts` (run_batch_resident_threads)"""
    return count * part // parts, count * (part + 1) // parts


def lane_scans(lane, lanes, count):
    """the scans lane `lane` of a sharded batch registers, in order (the static deal)"""
    return list(range(lane, count, lanes))
