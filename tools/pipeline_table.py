"""Bytes / time table of the pipeline's kernels: the rocprofv3 kernel-trace summary of a drive through the drop-in
(tools/gpu.sh pipetrace -> prof_summary.py) joined with what each kernel has to move per frame - its ALGORITHMIC bytes, from the
frame's point counts (the `chained pre-steps: a -> b -> c -> d points` line of a KICP_TRACE=1 run) - against the 8 TB/s HBM peak.
These kernels move a few MB each: none of them is anywhere near a bandwidth bound - they are dependent-latency bound (a load, an atomic
round trip, a dependent load; a kernel boundary costs ~1.5 us, a kernel's fill and drain ~4) - which is why round 6 cut their NUMBER.

    python tools/pipeline_table.py <pipeline_kernel_trace.txt> <n_in> <n0> <n1> <n2> [--point-step 16]
"""
import argparse
import re

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("n_in", type=int)
ap.add_argument("n0", type=int)
ap.add_argument("n1", type=int)
ap.add_argument("n2", type=int)
ap.add_argument("--point-step", type=int, default=16)
a = ap.parse_args()


def buckets(n):  # tsl::robin_map::reserve(n): ceil(n / 0.5) rounded up to a power of two
    b = 1
    while b < 2 * n:
        b <<= 1
    return b


A, B = buckets(a.n0), buckets(a.n1)
# bytes per frame and kernel: reads + writes the algorithm needs (atomics counted as 8 B each way)
BYTES = {
    "k_ingest": ("decode %d-byte records -> fp64 xyz + stamp" % a.point_step, a.n_in * (a.point_step + 32)),
    "k_frame_pre": ("xyz + stamp in, base-frame point + flag out, one claim per voxel run", a.n_in * (32 + 24 + 4) + a.n0 * 16),
    "k_frame_l1_replay": ("flags + staged -> buffer 0; table A keys + winners -> order", a.n_in * 4 + a.n0 * 48 + A * 8 + a.n1 * 24),
    "k_frame_l1_gather": ("table A keys + order -> buffer 1; reset; claims into table B", A * 8 + a.n1 * (4 + 24 + 24 + 20 + 16)),
    "k_frame_l2_replay": ("table B keys + winners -> order", B * 8 + a.n2 * 24),
    "k_frame_l2_gather": ("table B keys + order -> buffer 2 (HBM + host copy); reset", B * 8 + a.n2 * (4 + 24 + 48 + 20)),
    "k_push_frame": ("buffer 0 -> pinned host memory (PCIe: ~46 GB/s is the roof, not HBM)", a.n0 * 48),
    "k_up_claim": ("points of buffer 1 -> world frame, one claim per point", a.n1 * (24 + 24 + 16)),
    "k_up_scan": ("touched voxels' counters -> segment starts", a.n1 * 16),
    "k_up_scatter": ("points into their voxels' segments", a.n1 * (24 + 8)),
    "k_up_apply": ("per touched voxel: bucket in, accepted points out (fp64 pool + 16-bit mirror)", a.n1 * (24 + 20 * 24 // 4 + 32)),
    "k_up_remove": ("first point of every voxel against the origin", 0),
    "k_up_publish": ("the update's counters + sequence number -> host memory (40 B), per-update counters reset", 48),
}
rows = {}
for line in open(a.trace):
    m = re.match(r"\s*(?:void )?kicp::(?:host::)?(k_\w+)[<(].*?\s(\d+)\s+(\d+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
    if not m:
        continue
    name, grid, wg, calls, avg = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), float(m.group(5))
    r = rows.setdefault(name, [0, 0.0])
    r[0] += calls
    r[1] += calls * avg
print("# %s; per frame: %d points in, %d after the crop, %d after the 0.5-voxel downsample, %d after the 1.5-voxel downsample; tables of %d / %d buckets"
      % (a.trace, a.n_in, a.n0, a.n1, a.n2, A, B))
print("%-20s %8s %10s %12s %10s %9s  %s" % ("kernel", "calls", "avg us", "bytes/launch", "GB/s", "of 8 TB/s", "what it moves"))
for name, (what, nbytes) in BYTES.items():
    if name not in rows:
        continue
    calls, total = rows[name]
    avg = total / calls
    per_launch = nbytes / (4 if name == "k_ingest" else 1)  # (a message goes up in four pieces)
    gbs = per_launch / avg / 1e3
    print("%-20s %8d %10.2f %12d %10.1f %8.2f%%  %s" % (name, calls, avg, per_launch, gbs, 100.0 * gbs / 8000.0, what))
