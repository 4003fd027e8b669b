"""Per-launch durations and gaps of the last render batch in a rocprofv3 --kernel-trace CSV.
Usage: kernel_trace_step.py <trace output directory>"""
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# a batch starts with the extend launch that generates its primary rays (third template argument true)
idx=[i for i,r in enumerate(rows) if 'k_extend' in r['Kernel_Name'] and 'true>' in r['Kernel_Name']]
i0=idx[-1]
prev_end=None
tot={}
for r in rows[i0:]:
    st,en=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    nm=r['Kernel_Name']
    name='extend' if 'k_extend' in nm else 'shade' if 'k_shade' in nm else nm[:30]
    gap = (st-prev_end)/1e3 if prev_end else 0
    tot[name]=tot.get(name,0)+(en-st)/1e3
    tot['gap']=tot.get('gap',0)+max(gap,0)
    print(f"{name:32s} grid={r['Grid_Size_X']:>10s} dur={(en-st)/1e3:9.1f}us gap={gap:7.1f}us")
    prev_end=en
print(tot)
