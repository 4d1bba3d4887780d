import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"][:70], r["Calls"], "avg ms %.3f" % (float(r["AverageNs"])/1e6))
