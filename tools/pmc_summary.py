"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per (kernel, counter) over the profiled launches.
usage: python tools/pmc_summary.py <dir-or-csv> [...]  -> JSON on stdout {"kernel grid=<threads>": {counter: mean, "_launches": n}} (the grid size tells the shapes of one kernel apart)"""
import csv, glob, json, os, sys
acc = {}
for arg in sys.argv[1:]:
    files = [arg] if arg.endswith(".csv") else glob.glob(os.path.join(arg, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"] + " grid=" + r["Grid_Size"]
            if not any(t in k for t in ("conv3x3_tile", "gemm_dma", "gemm_wide", "gemm_p8", "gemm_rowpanel", "conv3x3_f32_mfma", "attn_kernel", "gn_apply", "gn_stats", "igemm_kernel")):
                continue
            d = acc.setdefault(k, {})
            v = d.setdefault(r["Counter_Name"], [0.0, 0])
            v[0] += float(r["Counter_Value"])
            v[1] += 1
out = {k: dict({c: v[0] / v[1] for c, v in d.items()}, _launches=max(v[1] for v in d.values())) for k, d in acc.items()}
print(json.dumps(out, indent=1))
