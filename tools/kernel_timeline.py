import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# take the last 40 kernel records, print name, start offset, duration, gap to previous end
tail=rows[-int(sys.argv[2]):]
t0=int(tail[0]["Start_Timestamp"]); prev=None
for r in tail:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    name=r["Kernel_Name"][:60]
    print("%8.2f us  dur %6.2f  gap %6.2f  %s  grid %s" % ((s-t0)/1e3,(e-s)/1e3, (s-prev)/1e3 if prev else 0, name, r.get("Grid_Size","")))
    prev=e
