import csv, glob, sys
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{sys.argv[1]}/pmc_{c}/*/*_counter_collection.csv")[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and "seg<" in r["Kernel_Name"]:
            print(c, r["Kernel_Name"][:40], f"{float(r['Counter_Value'])*1024/(512<<20):.3f} counted bytes per byte moved")
