#!/usr/bin/env python3
"""profiles/traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only) of ONE bench step.

usage: make_traffic.py <fetch_dir> <write_dir> <records_per_launch> <out.json> [sr|ont|k2] [launches_in_the_profiled_step]
(one file per workload: profiles/traffic.json for the headline, traffic_ont.json, traffic_k2.json)
Counter values are KiB (x 1024 -> bytes), summed per bench stage.  Calibration for this path's access pattern, as
MI355X_MICROARCH.md (HBM section) asks for widths other than wide streams: `scripts/pmc_gather_calib.sh` runs the gather
micro-benchmark (a known number of random 16-B slot loads over the index table) under the same counter and finds exactly 64.0 B
per probe (profiles/r01_pmc_gather_calib.txt): one 64-B sector per 16-B gather, counted exactly.  The guide's x2 correction
applies only to the coalesced 16-B/lane streams (K1's 3 GB of bases: FETCH_SIZE shows half of them), so the figures are exact
for the gathers and a lower bound by at most that amount overall."""
import csv, glob, hashlib, json, os, sys, collections

KERNEL_SOURCES = ("sh_classify.hip", "sh_sketch.h", "sh_chain.h", "sh_align.h", "sh_long.h", "sh_rmq_tree.h", "sh_wave.h")
K2_SOURCES = ("sh_k2.hip",)


def source_hash(workload="sr"):
    """sha1 over the kernel sources the counters belong to: bench.py refuses a traffic.json measured on other code."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scrubby_amd", "csrc")
    h = hashlib.sha1()
    for f in (K2_SOURCES if workload == "k2" else KERNEL_SOURCES):
        h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()


STAGES = {      # bench.py's stage names -> kernel name prefixes; a kernel belongs to the FIRST stage one of whose prefixes it carries
    "extension stage (k_long_chains + k_regs_align_long)": ("k_lext_", "k_long_chains", "k_regs_align_long", "k_wait_started"),
    "k_sketch_probe": ("k_sketch_probe", "k_long_"),      # long reads: the segment-parallel front end stands where K1 does
    "k_chain_small": ("k_chain_small", "k_pair_pass"),
    "repeat path (k_local_cluster, k_expand, k_sort_top / k_sort_lds classes, k_giant_*, k_cluster_dp, k_finalize)": ("k_local_cluster", "k_expand", "k_lr_locus", "k_group_probe", "k_sort_", "k_giant", "k_cluster_dp", "k_finalize", "k_chain_large"),
    "extension stage (k_ext_* + k_regs_align)": ("k_ext_", "k_regs_align"),
    "k_k2_classify": ("k_k2_classify",),
}


def collect(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(float)
    kern = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        for st, pats in STAGES.items():
            if any(p in name for p in pats):
                acc[st] += float(r["Counter_Value"]) * 1024.0
                kern[name.split("(")[0].replace("void ", "")[:40]] += float(r["Counter_Value"]) * 1024.0
                break
    return acc, kern


def main():
    fd, wd, n_rec, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    workload = sys.argv[5] if len(sys.argv) > 5 else "sr"
    launches = int(sys.argv[6]) if len(sys.argv) > 6 else 1      # launches the profiled step made (long reads: records / --ont-chunk)
    fe, fk = collect(fd, "FETCH_SIZE")
    we, wk = collect(wd, "WRITE_SIZE")
    doc = {"comment": __doc__.split("\n\n")[1].replace("\n", " "), "workload": workload, "records_per_launch": n_rec,
           "source_sha1": source_hash(workload), "stages": {}, "kernels": {}}
    for st in STAGES:
        if fe[st] + we[st] > 0:
            doc["stages"][st] = {"fetch": fe[st], "write": we[st], "launches": launches, "hbm_bytes_per_launch": (fe[st] + we[st]) / launches}
    for k in sorted(set(fk) | set(wk)):
        doc["kernels"][k] = {"fetch": fk.get(k, 0.0), "write": wk.get(k, 0.0)}
    json.dump(doc, open(out, "w"), indent=1)
    for st, v in doc["stages"].items():
        print(f"{st[:40]:40s} fetch {v['fetch'] / 1e9:8.2f} GB  write {v['write'] / 1e9:8.2f} GB")


if __name__ == "__main__":
    main()
