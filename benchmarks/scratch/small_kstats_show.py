import csv, glob, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out/small/s*.csv")), key=lambda p: int(re.findall(r"s(\d+)", p)[-1])):
    print("==", os.path.basename(f))
    for r in list(csv.DictReader(open(f)))[:4]:
        print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e3:8.1f} us min {float(r['MinNs'])/1e3:8.1f}")
