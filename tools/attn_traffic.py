"""profiles/<tag>_attn_traffic.json from the PMC passes of tools/pmc_attn.sh: HBM bytes per launch of the attention forward =
2 x FETCH_SIZE x 1024 (gfx950 tallies wide reads at half, MI355X_MICROARCH.md 'HBM') + WRITE_SIZE x 1024, mean per dispatch without the first.
    python tools/attn_traffic.py <out.json> <key>=<gpurun_out/pmc_dir> ...      key: heads of a plain launch ("5", "15", "32"), "20" = the CFG form,
                                                                                 "15" = the optimisation pass's form, "15_plain" = 15 plain heads"""
import csv, glob, json, os, sys
out = sys.argv[1]
res = {"comment": "Attention forward at HEAD (round of the file name), N = M = 4096, D = 64, bf16: HBM bytes per launch = 2*FETCH_SIZE*1024 (gfx950 half-count "
                  "correction for wide reads) + WRITE_SIZE*1024; rocprofv3 --pmc, one counter per pass (tools/pmc_attn.sh -> tools/attn_one.py; tables: "
                  "profiles/pmc_r04_*.md).  '5' / '15_plain' / '32': plain head-major launches with pre-scaled queries; '20': the CFG pass's launch as an edit "
                  "issues it (4 token-major segments x 5 heads, fused warp + row list); '15': the optimisation pass's launch (3 segments x 5 heads, "
                  "row list, LSE, row sums over the rounded probabilities).",
       "algorithmic_bytes_per_launch": {"5": 10485760, "15": 31457280, "15_plain": 31457280, "20": 41943040, "32": 67108864},
       "bytes_per_launch": {}, "raw_KB": {}}
for arg in sys.argv[2:]:
    key, d = arg.split("=", 1)
    acc = {"FETCH_SIZE": [], "WRITE_SIZE": []}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_attn_fwd" in r.get("Kernel_Name", "") and r["Counter_Name"] in acc:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not acc["FETCH_SIZE"] or not acc["WRITE_SIZE"]:
        continue
    fe, wr = (v[1:] if len(v) > 2 else v for v in (acc["FETCH_SIZE"], acc["WRITE_SIZE"]))
    fe, wr = sum(fe) / len(fe), sum(wr) / len(wr)
    res["raw_KB"][key] = {"FETCH": round(fe), "WRITE": round(wr)}
    res["bytes_per_launch"][key] = int((2 * fe + wr) * 1024)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["bytes_per_launch"]))
