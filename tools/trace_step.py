import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'gdn_graph_kernel' in r['Kernel_Name'] or 'gdn_graph_terms_kernel' in r['Kernel_Name']]
a,b=idx[-2],idx[-1]
t0=int(rows[a]['Start_Timestamp'])
print(b-a,'kernels per step; span', (int(rows[b]['Start_Timestamp'])-t0)/1e3,'us; busy', sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows[a:b])/1e3)
for r in rows[a:b]:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:8.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f}  {r['Kernel_Name'][:100]}")
