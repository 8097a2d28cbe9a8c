import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# print a window of ~3 iterations in the middle
mid = len(rows) // 2
names = lambda r: r["Kernel_Name"].split("(")[0].replace("void bnmf::", "").replace("bnmf::", "")[:22]
for r in rows[mid:mid + 18]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{names(r):24s} stream/queue {r.get('Queue_Id','?'):>3s} start {s/1e3:10.1f} us  dur {(e-s)/1e3:7.1f} us")
