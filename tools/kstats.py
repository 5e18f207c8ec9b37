import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "mlp_" in r["Name"] or "tail" in r["Name"]: print(sys.argv[1], r["Name"][:50], r["Calls"], round(float(r["AverageNs"])/1e3,1))
