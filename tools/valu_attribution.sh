# Where the generic pass kernel's VALU instructions go: SQ_INSTS_VALU / SQ_WAVES per dispatch of the four-waves
# build, k_pass_gather32<256, 1, 4, false, false>, on cfg2 - one rocprofv3 --pmc pass per ablation switch of the kernel (option "dbg",
# tools/gpu_dbg.py): 7 no query work (start of the lane + the hand-over) | 2 + the transform | 4 + the probe | 3 + the own voxel's
# visit | 5 + the face neighbours | 1 + edges and corners = the whole search | 13 + the exact phase | 0 + the terms (everything).
# usage: tools/valu_attribution.sh <outdir> [lib.so]     -> <outdir>/valu_attribution.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$1; LIB=$2; mkdir -p $O
[ -n "$LIB" ] && export KICP_AB_LIB=$LIB
for d in 7 2 4 3 5 1 13 0; do
  timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS -d $O/valu_$d -o pmc -- python tools/prof_target.py --workload cfg2 --calls 64 --option small=0 --option latency_kernel=0 --option lanes_per_query=1 --option dbg=$d > /dev/null 2> $O/valu_$d.err || echo "dbg $d failed"
done
python - $O <<'PY'
import sqlite3, sys, glob, numpy as np
O = sys.argv[1]
prev = None
out = open(O + "/valu_attribution.txt", "w")
def p(s):
    print(s); out.write(s + "\n")
p("dbg   VALU/dispatch   VALU/wave  (+ over the row above)   SALU/wave  LDS/wave   dispatches")
for d, what in ((7, "no query work: lane start without a query + hand-over"), (2, "+ transform, voxel, offsets"), (4, "+ probe"), (3, "+ own voxel's visit"),
                (5, "+ face neighbours"), (1, "+ edges, corners = whole search"), (13, "+ exact phase (no terms)"), (0, "+ terms = everything")):
    dbs = glob.glob("%s/valu_%d/**/*.db" % (O, d), recursive=True)
    if not dbs:
        p("%3d   (no database)" % d); continue
    c = sqlite3.connect(dbs[0])
    cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = c.execute("select counter_name, value from counters_collection where %s like '%%k_pass_gather32%%'" % name_col).fetchall()
    acc = {}
    for n, v in rows: acc.setdefault(n, []).append(float(v))
    m = {k: float(np.mean(v)) for k, v in acc.items()}
    waves = m.get("SQ_WAVES", float("nan"))
    per = m.get("SQ_INSTS_VALU", float("nan")) / waves
    p("%3d   %12.0f   %9.1f  %+9.1f   %9.1f  %8.1f   %d   %s" % (d, m.get("SQ_INSTS_VALU", float("nan")), per, per - (prev if prev is not None else per),
       m.get("SQ_INSTS_SALU", float("nan")) / waves, m.get("SQ_INSTS_LDS", float("nan")) / waves, len(acc.get("SQ_WAVES", [])), what))
    prev = per
PY
find $O -name "*.db" -delete; rm -rf $O/valu_*/
