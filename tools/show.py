import csv,glob,sys
tag=sys.argv[1]
f=sorted(glob.glob(f'gpurun_out/prof_{tag}/*/*kernel_stats.csv'))[-1]
tot=0
for r in list(csv.DictReader(open(f)))[:9]:
    if 'psm_' not in r['Name']: continue
    n=int(r['Calls']); avg=float(r['AverageNs'])/1e3
    per_solve = avg*n/651.0 if n>=651 else avg
    tot+=avg*n/651.0
    print(f"{r['Name'][:55]:55s} calls={n:6d} avg_us={avg:8.2f}")
print("sum per solve ~", round(tot,1),"us")
