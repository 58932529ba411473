#!/usr/bin/env python3
"""Digest a tools/profile_bench.sh output directory into tracked files: profiles/<tag>_kernel_stats.csv, profiles/<tag>_summary.md
and profiles/traffic_latest.json (what bench.py quotes as roofline.traffic / valu_issue_frac / hbm_utilisation -- accepted there
only while its `source_hash` equals the hash of the kernel sources bench.py runs).  usage: summarise_profile.py <dir> <tag> [--no-latest]
(--no-latest: a profile of another view, e.g. `profile_bench.sh <tag> --zoom 0.7`: traffic_latest.json stays the headline's)"""
import csv, glob, collections, os, shutil, sys, json

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
meta_in = json.load(open(os.path.join(src, "meta.json"))) if os.path.isfile(os.path.join(src, "meta.json")) else {}
path_seg = meta_in.get("path_seg", [2, 4])

KERNELS = (("render_kernel", "render_kernel<64, false"), ("path_kernel", "path_kernel<false"), ("resolve_paths_kernel", "resolve_paths_kernel<"))


def short(name):
    for s, pat in KERNELS:
        if pat in name:
            return s
    return None


ks = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
rows = []
if ks:
    shutil.copy(ks[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
    rows = list(csv.DictReader(open(ks[0])))
avg_ms, full_name = {}, {}
for r in rows:
    s = short(r["Name"])
    if s:
        if s in full_name and full_name[s] != r["Name"]:
            print(f"WARNING: two production instantiations of {s} in one profile: {full_name[s]} and {r['Name']}; profile one workload at a time")
        avg_ms[s] = float(r["AverageNs"]) / 1e6
        full_name[s] = r["Name"]           # bench.py accepts the counters only for exactly this instantiation
pmc = collections.defaultdict(lambda: collections.OrderedDict())
disp = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            s = short(r["Kernel_Name"])
            if s:
                pmc[s].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                disp[s] = {k: r.get(k) for k in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size")}
cal = []
for f in glob.glob(os.path.join(src, "cal_fetch", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "probe_stream_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            cal.append(float(r["Counter_Value"]))
factor = None
if cal:
    factor = (4 << 30) / (sum(cal) / len(cal) * 1024.0)

out = {"tag": tag, "source_hash": bench.source_hash(), "path_seg": list(path_seg), "command": meta_in.get("command"),
       "fetch_factor": None if factor is None else round(factor, 4),
       "calibration": None if factor is None else "4 GiB streamed with 8-byte-per-lane loads (tools/calibrate_fetch.py): FETCH_SIZE x factor = bytes",
       "kernels": {}}
with open(os.path.join(dst, f"{tag}_summary.md"), "w") as o:
    o.write(f"# rocprofv3 summary `{tag}`\n\nCommand: `{meta_in.get('command', 'python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary')}` "
            f"(cfg3, scene S1, path_seg_range {tuple(path_seg)}), kernel sources `{out['source_hash']}`; one `--kernel-trace --stats` pass and separate "
            "`--pmc` passes (tools/profile_bench.sh).  The profiled run is a different run (3 steps, another box) than the "
            "driver's bench: durations agree to a few per cent.\n\n## kernel-trace --stats\n\n")
    o.write("| kernel | calls | avg ms | min ms | max ms | % |\n|---|---|---|---|---|---|\n")
    for r in rows:
        if float(r["Percentage"]) < 0.01:
            continue
        o.write(f"| `{r['Name']}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} | {float(r['MinNs'])/1e6:.3f} | {float(r['MaxNs'])/1e6:.3f} | {r['Percentage']} |\n")
    for s, _ in KERNELS:
        if s not in pmc:
            continue
        m = {k: sum(v) / len(v) for k, v in pmc[s].items()}
        o.write(f"\n## PMC, `{s}` (production instantiation, mean over its dispatches)\n\n| counter | mean | n |\n|---|---|---|\n")
        for k, v in pmc[s].items():
            o.write(f"| {k} | {m[k]:.6g} | {len(v)} |\n")
        o.write(f"\nDispatch: {json.dumps(disp.get(s))}\n\nDerived:\n\n")
        k = {"avg_ms": round(avg_ms.get(s, 0.0), 4), "name": full_name.get(s)}
        if "FETCH_SIZE" in m and factor is not None:
            k["hbm_read_bytes"] = m["FETCH_SIZE"] * 1024.0 * factor
            k["hbm_write_bytes"] = m.get("WRITE_SIZE", 0.0) * 1024.0
            k["hbm_bytes"] = k["hbm_read_bytes"] + k["hbm_write_bytes"]
            o.write(f"- HBM traffic per launch: read {k['hbm_read_bytes']/1e9:.2f} GB (FETCH_SIZE x {factor:.3f}, calibrated) + write "
                    f"{k['hbm_write_bytes']/1e9:.2f} GB")
            if k["avg_ms"]:
                o.write(f" = {k['hbm_bytes']/1e9/(k['avg_ms']*1e-3)/1e3:.2f} TB/s = {k['hbm_bytes']/(k['avg_ms']*1e-3)/8e12:.3f} of the 8 TB/s peak")
            o.write("\n")
        if "TCC_HIT_sum" in m:
            k["l2_hit"] = round(m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 4)
            o.write(f"- L2 hit rate = {k['l2_hit']:.3f}\n")
        if "SQ_WAVE_CYCLES" in m:
            wc = m["SQ_WAVE_CYCLES"]
            o.write(f"- per-wave time split: issuing {m['SQ_ACTIVE_INST_ANY']/wc:.2f}, waiting on memory {m['SQ_WAIT_ANY']/wc:.2f}, "
                    f"waiting to issue {m['SQ_WAIT_INST_ANY']/wc:.2f}; VALU share of issue {m['SQ_ACTIVE_INST_VALU']/m['SQ_ACTIVE_INST_ANY']:.2f}\n")
        if "SQ_THREAD_CYCLES_VALU" in m and "SQ_ACTIVE_INST_VALU" in m:
            k["lanes_per_valu"] = round(m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"], 1)
            o.write(f"- active lanes per VALU instruction = {k['lanes_per_valu']} of 64\n")
        if "GRBM_GUI_ACTIVE" in m and "SQ_INSTS_VALU" in m:
            cyc = m["GRBM_GUI_ACTIVE"] / 8
            k["valu_insts"] = m["SQ_INSTS_VALU"]
            k["valu_issue_frac"] = round(m["SQ_INSTS_VALU"] * 2 / (1024 * cyc), 4)
            o.write(f"- kernel cycles (GRBM_GUI_ACTIVE/8) = {cyc:.4g}; VALU issue fraction = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x cycles) = {k['valu_issue_frac']:.3f}\n")
        out["kernels"][s] = k
if "--no-latest" not in sys.argv:
    json.dump(out, open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
print(open(os.path.join(dst, f"{tag}_summary.md")).read())
