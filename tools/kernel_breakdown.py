import csv,glob,sys
f=(glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')+glob.glob(sys.argv[1]+'/*kernel_stats.csv'))[0]
n=int(sys.argv[2])
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv)>3 else 24]:
    print("%-86s %6.1f/step %8.1f us/call %7.1f us/step %5.1f%%"%(r['Name'][:86],int(r['Calls'])/n,float(r['AverageNs'])/1e3,float(r['TotalDurationNs'])/n/1e3,float(r['Percentage'])))
print("total us/step", tot/n/1e3)
