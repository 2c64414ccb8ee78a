#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace -d $O/tr -o p --output-format csv -- python3 tools/pmc_run.py 2d 0 $2 strict > $O/tr.log 2>&1 || { tail -5 $O/tr.log; exit 1; }
python3 - <<PY
import csv, glob
f=glob.glob('$O/tr/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
steps=[]; cur=None
for r in rows:
    name=r['Kernel_Name']; d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    if 'k_bitonic_local<true' in name: cur={'init':d,'tail':0,'strided':0,'lean':0,'gen':0,'dens':0}; steps.append(cur)
    elif cur is None: continue
    elif 'k_bitonic_local<false' in name: cur['tail']+=d
    elif 'k_bitonic_strided' in name: cur['strided']+=d
    elif 'k_force_general' in name: cur['gen']+=d
    elif 'k_force<' in name: cur['lean']+=d
    elif 'k_density' in name: cur['dens']+=d
for i,s in enumerate(steps): print(i+1, {k:round(v) for k,v in s.items()})
PY
