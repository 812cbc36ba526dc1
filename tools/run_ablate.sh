#!/bin/bash
# usage: run_ablate.sh <script> args...   -> prints per-launch igemm kernel durations in order
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/abl; rocprofv3 --kernel-trace --output-format csv -d /tmp/abl -- python $GRAFT_REPO_ROOT/tools/"$@" > /tmp/abl.log 2>&1 || tail -5 /tmp/abl.log
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/abl/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'igemm_kernel' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
print(' '.join(f"{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:.1f}" for r in rows))
PY
