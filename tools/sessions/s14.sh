#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python tools/probe_predict.py 60 2>/dev/null
cd /tmp
timeout 280 rocprofv3 --kernel-trace -d $O/prof_tmp -o trace -- python $R/tools/probe_predict.py 200 > $O/prof_tmp.log 2>&1
python $R/tools/rocpd_kernel_stats.py $O/prof_tmp/trace_results.db $O/s14_predict_kernels.csv
head -40 $O/s14_predict_kernels.csv
python - <<'PY'
import sqlite3,re
db=sqlite3.connect('/root/repo/gpurun_out/prof_tmp/trace_results.db')
rows=db.execute('select name,start,end from kernels order by start').fetchall()
# last predict: from last embedding_fwd_kernel to the end
idx=[i for i,r in enumerate(rows) if 'embedding_fwd' in r[0]]
a=idx[-2]; b=idx[-1]
seg=rows[a:b]
print('one predict: kernels',len(seg),'span us',(seg[-1][2]-seg[0][1])/1e3,'busy us',sum(e-s for _,s,e in seg)/1e3)
prev=None
for n,s,e in seg:
    n=re.sub(r'\(.*$','',n).replace('void ','')[:50]
    print(f'{(s-seg[0][1])/1e3:8.1f} {(e-s)/1e3:6.1f} gap {0 if prev is None else (s-prev)/1e3:5.1f} {n}')
    prev=e
PY
rm -rf $O/prof_tmp
