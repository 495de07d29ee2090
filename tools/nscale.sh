#!/bin/bash
# usage: tools/nscale.sh op C H kh kw   -> kernel duration per batch size (grid size identifies the launch)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ns && rocprofv3 --kernel-trace -d /tmp/ns -o ns -- python $R/tools/nscale.py "$@" > /tmp/ns.log 2>&1
python - <<'PY'
import sqlite3, glob
db = glob.glob("/tmp/ns/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = list(c.execute("select name, end-start from kernels where name like '%slak%' order by start"))
name = rows[0][0][:90]
d = [r[1] / 1e3 for r in rows]
n = len(d) // 4
print(name)
for i, N in enumerate((32, 64, 128, 256)):
    seg = sorted(d[i * n:(i + 1) * n])
    print("   N=%3d: median %7.2f us  min %7.2f us" % (N, seg[len(seg) // 2], seg[0]))
PY
