import csv,glob,re,sys
d=sys.argv[1]
rows=[]
for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True): rows+=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_tb3" in r["Kernel_Name"] and "false, 2>" in r["Kernel_Name"]]
i0=idx[-2]
# print from 12 kernels before the pair's box kernel to 12 after
t0=int(rows[i0-8]["Start_Timestamp"])
for r in rows[i0-8:i0+10]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    name=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void pf::","")
    print(f"{(s-t0)/1e3:9.1f} +{(e-s)/1e3:8.1f} us  q{r.get('Queue_Id','?'):>3}  {name[:90]}")
