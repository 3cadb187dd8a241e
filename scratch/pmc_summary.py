import sys, glob, csv, collections
d = sys.argv[1]
tot = collections.defaultdict(float)
for f in glob.glob(f"{d}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_render" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot): print(f"{k:28s} {tot[k]:.6g}")
g = tot
if "SQ_THREAD_CYCLES_VALU" in g and "SQ_ACTIVE_INST_VALU" in g:
    print("avg active lanes per VALU inst ~", g["SQ_THREAD_CYCLES_VALU"]/max(1,g["SQ_ACTIVE_INST_VALU"])/4 if g["SQ_ACTIVE_INST_VALU"] else 0)
if "SQ_INSTS_VALU" in g and "SQ_WAVES" in g: print("VALU insts per wave", g["SQ_INSTS_VALU"]/g["SQ_WAVES"])
