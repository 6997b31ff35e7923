"""Sum rocprofv3 --pmc counter_collection.csv rows per kernel (classification kernels only)."""
import csv, glob, sys, collections
keep = ("k_sketch_probe", "k_chain_small", "k_pair_pass", "k_local_cluster", "k_expand", "k_sort", "k_finalize", "k_giant", "k_cluster_dp", "k_chain_large",
        "k_ext_", "k_regs_align", "k_long_", "k_lext_", "k_k2_")
for d in sys.argv[1:]:
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(fs[0])):
        n = r["Kernel_Name"]
        if not any(k in n for k in keep): continue
        short = n.split("(")[0].replace("void ", "")[:28]
        acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (short, r["Dispatch_Id"])
        if key not in seen: seen.add(key); calls[short] += 1
    for k in acc:
        print(d.split("/")[-1], k.ljust(28), "calls", calls[k], " ".join(f"{c}={v:.4g}" for c, v in sorted(acc[k].items())))
