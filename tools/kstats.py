"""print the top rows of a rocprofv3 --stats kernel_stats.csv found under a directory: name (shortened), calls, total, average (us)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for r in rows[:n]:
    nm = r["Name"].replace("ngpde::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    print(f'{nm:70s} calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"]) / 1e3:9.1f} us  total {float(r["TotalDurationNs"]) / 1e6:8.2f} ms  {float(r["Percentage"]):5.1f} %')
