# conv_split16 durations of the alternating dispatches of each (instantiation, grid): first / second conv of the ResBlock pairs
#   (on the GPU box: bash tools/conv_pair_roles.sh)
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/ppl
timeout 600 rocprofv3 --kernel-trace -d /tmp/ppl -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt > /dev/null 2>&1
python - <<PY
import sqlite3,glob,collections
db=glob.glob("/tmp/ppl/**/*.db",recursive=True)[0]
cur=sqlite3.connect(db).cursor()
rows=cur.execute("select name, grid_x/workgroup_x, duration, start from kernels where name like '%conv_split16_kernel%' order by start").fetchall()
seq=collections.defaultdict(list)
for n,g,d,s in rows: seq[(n[n.find('<'):n.find('>')+1],g)].append(d/1e3)
for k,v in seq.items():
    if len(v)>=8: print(k, len(v), "even %.1f odd %.1f"%(sum(v[0::2])/len(v[0::2]), sum(v[1::2])/len(v[1::2])))
PY
