import csv,glob,sys
f=[p for p in glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True)][0]
rows=list(csv.DictReader(open(f)))
for r in rows[:int(sys.argv[2]) if len(sys.argv)>2 else 40]:
    print("%-100s n=%6s avg=%10.1f us total=%8.1f ms %5s%%"%(r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
