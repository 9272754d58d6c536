"""Average rocprofv3 --pmc counter CSVs per kernel (development aid): python tools/pmc_summary.py <dir with pass*/...counter_collection.csv>"""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r.get("Kernel_Name", "?")
            if "attn" not in k and len(sys.argv) < 3: continue
            acc[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(f"## {k}")
    print("| counter | mean per dispatch | n |\n|---|---|---|")
    for c, v in sorted(d.items()):
        v = v[1:] if len(v) > 2 else v          # drop the first (cold) dispatch
        print(f"| {c} | {sum(v)/len(v):,.0f} | {len(v)} |")
    g = {c: sum(v[1:] if len(v) > 2 else v) / len(v[1:] if len(v) > 2 else v) for c, v in d.items()}
    if "SQ_WAVE_CYCLES" in g and "SQ_BUSY_CYCLES" in g:
        print()
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY"):
            if c in g: print(f"- {c} / SQ_WAVE_CYCLES = {g[c]/g['SQ_WAVE_CYCLES']*100:.1f} %")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in g and "GRBM_GUI_ACTIVE" in g:
            # GRBM_GUI_ACTIVE is summed over 8 XCDs; 1024 SIMDs
            cyc = g["GRBM_GUI_ACTIVE"] / 8
            print(f"- MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x {cyc:,.0f} cycles) = {g['SQ_VALU_MFMA_BUSY_CYCLES']/1024/cyc*100:.1f} %")
            if "SQ_ACTIVE_INST_VALU" in g: print(f"- VALU busy = 4 x SQ_ACTIVE_INST_VALU / (1024 x cycles) = {4*g['SQ_ACTIVE_INST_VALU']/1024/cyc*100:.1f} %")
            if "SQ_VALU_MFMA_COEXEC_CYCLES" in g: print(f"- VALU+MFMA co-execution = {g['SQ_VALU_MFMA_COEXEC_CYCLES']/1024/cyc*100:.1f} % of cycles")
    print()
