cd $GRAFT_REPO_ROOT
O=gpurun_out/c28; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CYCLES"; do
  n=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $GRAFT_REPO_ROOT/$O/pmc_$n -o r -- $GRAFT_REPO_ROOT/tools/_probe/r2p0 48 128 128 256 256 3 5 256 > $GRAFT_REPO_ROOT/$O/run_$n.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3, glob, json
out = {}
for d in glob.glob("gpurun_out/c28/pmc_*"):
    dbs = glob.glob(d + "/**/*results.db", recursive=True)
    if not dbs: continue
    c = sqlite3.connect(dbs[0])
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    try:
        rows = c.execute("select counter_name, sum(value), count(*) from counters_collection group by counter_name").fetchall()
    except Exception as e:
        rows = [("tables", str(tabs)[:400], 0)]
    kd = c.execute("select count(*), avg(duration), sum(duration) from kernels where name like '%igemm4%'").fetchall()
    out[d.split('/')[-1]] = {"counters": rows, "igemm4_launches_avg_ns_sum_ns": kd}
print(json.dumps(out, indent=1)[:3000])
json.dump(out, open("gpurun_out/c28/pmc_summary.json", "w"), indent=1)
PY
