#!/usr/bin/env python3
"""Digest a tools/profile_bench.sh output directory into profiles/<tag>_*.{csv,md} (tracked files)."""
import csv, glob, collections, os, shutil, sys, json

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
ks = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
rows = list(csv.DictReader(open(ks[0]))) if ks else []
pmc = collections.OrderedDict()
meta = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "render_kernel<64, false" in r["Kernel_Name"]:
                pmc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                meta = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size")}
with open(os.path.join(dst, f"{tag}_summary.md"), "w") as o:
    o.write(f"# rocprofv3 summary `{tag}`\n\nCommand: `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline` (cfg3, scene S1), "
            "one `--kernel-trace --stats` pass and separate `--pmc` passes (tools/profile_bench.sh).\n\n## kernel-trace --stats\n\n")
    o.write("| kernel | calls | avg ms | min ms | max ms | % |\n|---|---|---|---|---|---|\n")
    for r in rows:
        if float(r["Percentage"]) < 0.01:
            continue
        o.write(f"| `{r['Name']}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} | {float(r['MinNs'])/1e6:.3f} | {float(r['MaxNs'])/1e6:.3f} | {r['Percentage']} |\n")
    o.write("\n## PMC, `render_kernel<64,false>` (mean over its dispatches)\n\n| counter | mean | n |\n|---|---|---|\n")
    m = {}
    for k, v in pmc.items():
        m[k] = sum(v) / len(v)
        o.write(f"| {k} | {m[k]:.6g} | {len(v)} |\n")
    o.write(f"\nDispatch: {json.dumps(meta)}\n\n## Derived\n\n")
    if "FETCH_SIZE" in m:
        o.write(f"- FETCH_SIZE = {m['FETCH_SIZE']/1e6:.2f} GB as reported (KB units); x2 per the gfx950 note for wide streams = {2*m['FETCH_SIZE']/1e6:.2f} GB "
                "(this kernel reads narrow gathers, so the true figure lies between the two)\n")
    if "WRITE_SIZE" in m:
        o.write(f"- WRITE_SIZE = {m['WRITE_SIZE']/1e6:.3f} GB\n")
    if "TCC_HIT_sum" in m:
        o.write(f"- L2 hit rate = {m['TCC_HIT_sum']/(m['TCC_HIT_sum']+m['TCC_MISS_sum']):.3f}\n")
    if "SQ_WAVE_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"]
        o.write(f"- per-wave time split: issuing {m['SQ_ACTIVE_INST_ANY']/wc:.2f}, waiting on memory/barrier {m['SQ_WAIT_ANY']/wc:.2f}, "
                f"waiting to issue {m['SQ_WAIT_INST_ANY']/wc:.2f}; VALU share of issue {m['SQ_ACTIVE_INST_VALU']/m['SQ_ACTIVE_INST_ANY']:.2f}\n")
    if "SQ_THREAD_CYCLES_VALU" in m and "SQ_INSTS_VALU" in m:
        o.write(f"- active lanes per VALU instruction = {m['SQ_THREAD_CYCLES_VALU']/m['SQ_INSTS_VALU']:.1f} of 64\n")
    if "GRBM_GUI_ACTIVE" in m and "SQ_ACTIVE_INST_ANY" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        o.write(f"- kernel cycles (GRBM_GUI_ACTIVE/8) = {cyc:.4g}; issue occupancy of the 1024 SIMDs = {m['SQ_ACTIVE_INST_ANY']*4/1024/cyc:.2f}\n")
# FETCH_SIZE calibration on a known 8-byte-per-lane stream (tools/calibrate_fetch.py: 4 GiB per dispatch)
cal = []
for f in glob.glob(os.path.join(src, "cal_fetch", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "probe_stream_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            cal.append(float(r["Counter_Value"]))
traffic = None
if cal and "FETCH_SIZE" in m:
    reported = sum(cal) / len(cal) * 1024.0
    factor = (4 << 30) / reported
    traffic = {"fetch_factor": round(factor, 4), "calibration": "4 GiB streamed with 8-byte-per-lane loads, FETCH_SIZE reported %.4g B" % reported,
               "hbm_read_bytes_per_launch": m["FETCH_SIZE"] * 1024.0 * factor,
               "hbm_write_bytes_per_launch": m.get("WRITE_SIZE", 0.0) * 1024.0, "tag": tag}
    traffic["hbm_bytes_per_launch"] = traffic["hbm_read_bytes_per_launch"] + traffic["hbm_write_bytes_per_launch"]
    json.dump(traffic, open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
    with open(os.path.join(dst, f"{tag}_summary.md"), "a") as o:
        o.write(f"- FETCH_SIZE calibration (8-byte-per-lane stream of 4 GiB): factor {factor:.3f} => HBM read "
                f"{traffic['hbm_read_bytes_per_launch']/1e9:.2f} GB + write {traffic['hbm_write_bytes_per_launch']/1e9:.2f} GB per launch\n")
print(open(os.path.join(dst, f"{tag}_summary.md")).read())
