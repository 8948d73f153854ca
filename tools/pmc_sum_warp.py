import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "warp_kernel" in r["Kernel_Name"]:
                agg[(r["Counter_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print(f"{k[0]:28s} grid={k[1]:>10s} avg/launch={sum(v)/len(v):16.1f} (n={len(v)})")
