import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_prep' in r['Kernel_Name']]
a,b=idx[int(sys.argv[2]) if len(sys.argv)>2 else 2], idx[(int(sys.argv[2]) if len(sys.argv)>2 else 2)+1]
t0=int(rows[a]['Start_Timestamp'])
prev_end=t0
tot={}
for r in rows[a:b]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    name=r['Kernel_Name'].replace('void svsdf::','').replace('svsdf::','')[:22]
    tot[name]=tot.get(name,0)+(e-s)
    print(f"{name:22s} q{r['Queue_Id']:>2s} start+{(s-t0)/1e3:8.1f}us dur {(e-s)/1e3:7.1f}us gap {(s-prev_end)/1e3:6.1f} grid {r['Grid_Size_X']}")
    prev_end=max(prev_end,e)
print("span us", (prev_end-t0)/1e3)
for k,v in sorted(tot.items(), key=lambda kv:-kv[1]): print(f"  {k:22s} {v/1e3:9.1f} us")
