# calibrate FETCH_SIZE for the 8-B-per-lane C-tile reads of the GEMM: same launches with beta = 1 and beta = 0 (scratch/gemm_beta.py);
# the difference is the C read as the counter sees it, the true bytes are tiles x 128 x 128 x 8
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmcal
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmcal -o r -- python3 $R/scratch/gemm_beta.py > /tmp/pmcal.log 2>&1
python3 - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob('/tmp/pmcal/**/*.db', recursive=True)[0])
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
print(cols)
key = [c for c in ("dispatch_id", "start", "id") if c in cols][0]
rows = db.execute(f"select kernel_name, value, duration from counters_collection where counter_name='FETCH_SIZE' order by {key}").fetchall()
g = [r for r in rows if 'gemm_nt_kernel' in r[0]]
print(len(g), "gemm launches")
# gemm_beta.py: for k in (256,512,1024): for beta in (1,0): 5 + 10 launches
i = 0
for k in (256, 512, 1024):
    vals = {}
    for beta in (1, 0):
        chunk = g[i:i + 17]; i += 17        # gemm_bench: one warm-up launch + reps, called with 5 and with 10 reps
        vals[beta] = sum(c[1] for c in chunk[7:]) / 10.0
    tiles = 121 * 120 // 2 + 121   # m=15488 -> 121 tile rows, n=15360 -> 120 tile cols, lower triangle incl. diagonal tiles
    true_kb = tiles * 128 * 128 * 8 / 1024.0
    print("k", k, "FETCH_SIZE beta1 %.0f KB  beta0 %.0f KB  difference %.0f KB  true C bytes %.0f KB  ratio %.3f" % (vals[1], vals[0], vals[1] - vals[0], true_kb, (vals[1] - vals[0]) / true_kb))
PY
