#!/usr/bin/env python3
"""Copies the judged summaries of one profile set from gpurun_out/<tag>/ into profiles/ (tracked):
    python profiles/collect.py r01b r01
kernel_stats (first rows), the bench JSON line of each traced run, the PMC summaries of this
repo's kernels, and profiles/traffic.json (HBM bytes per k_pathtrace launch, read by bench.py)."""
import glob
import json
import os
import subprocess
import sys

tag, name = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
c3_out = None
for cfg in ("c2", "c3", "c4", "c4c", "c4pm", "c5"):
    ks = glob.glob(os.path.join(src, f"trace_{cfg}", "**", "*kernel_stats.csv"), recursive=True)
    if ks:
        ks.sort(key=os.path.getmtime)   # gpurun merges results into gpurun_out/: an earlier run of the same tag may still lie there
        lines = open(ks[-1]).read().splitlines()[:9]
        open(os.path.join(dst, f"{name}_{cfg}_kernel_stats.csv"), "w").write("\n".join(lines) + "\n")
    log = os.path.join(src, f"trace_{cfg}.log")
    if os.path.exists(log):
        js = [l for l in open(log, errors="replace") if l.startswith("{")]
        if js:
            open(os.path.join(dst, f"{name}_{cfg}_bench_under_trace.json"), "w").write(js[-1])
    pm = os.path.join(src, f"pmc_{cfg}")
    if os.path.isdir(pm):
        out = subprocess.run([sys.executable, os.path.join(dst, "summarize_pmc.py"), pm], capture_output=True, text=True, check=True).stdout
        open(os.path.join(dst, f"{name}_{cfg}_pmc.json"), "w").write(out)
        if cfg == "c3":
            c3_out = out
if c3_out:
    out = c3_out
    full = json.loads(out)
    d = next((v for k, v in full.items() if k.startswith("k_pathtrace<8, true, false")), {})
    if "hbm_write_bytes" in d:
        # k_pathtrace stores 12 B per pixel per chunk with dword stores (WRITE_SIZE is exact for those)
        # and reads only the 0.5 KB scene: FETCH_SIZE raw (no wide coalesced reads to correct for)
        bl = os.path.join(dst, f"{name}_c3_bench_under_trace.json")
        sha = (json.load(open(bl)).get("build") or {}).get("src_sha16") if os.path.exists(bl) else None
        c4 = {}
        p4 = os.path.join(dst, f"{name}_c4_pmc.json")
        if os.path.exists(p4):
            g = next((v for k, v in json.load(open(p4)).items() if k.startswith("void k_gather_knn") or k.startswith("k_gather_knn")), {})
            if g.get("TCC_REQ_sum"):
                c4 = {"c4_gather_l2_request_bytes_per_launch": 128.0 * g["TCC_REQ_sum"],
                      "c4_gather_hbm_bytes_per_launch": g.get("hbm_read_bytes_raw", 0.0) + g.get("hbm_write_bytes", 0.0)}
        p5 = os.path.join(dst, f"{name}_c5_pmc.json")
        if os.path.exists(p5):
            g = next((v for k, v in json.load(open(p5)).items() if k.startswith("k_pathtrace_bvh<7, 22, false, false, true")), {})
            if g.get("TCC_REQ_sum"):
                # the PMC run of C5 renders 16 spp (profiles/run_all.sh): lines the vector L1s ask the L2 for, and the launch's duration
                c4.update({"c5_l2_request_bytes_per_16spp_frame": 128.0 * g["TCC_REQ_sum"],
                           "c5_16spp_frame_us_under_pmc": g.get("mean_duration_us_under_pmc"),
                           "c5_l2_GBps": 128.0 * g["TCC_REQ_sum"] / (g["mean_duration_us_under_pmc"] * 1e-6) / 1e9,
                           "c5_hbm_bytes_per_16spp_frame": g.get("hbm_read_bytes_raw", 0.0) + g.get("hbm_write_bytes", 0.0)})
        json.dump({"tag": name, "src_sha16": sha, **c4,
                   "k_pathtrace_hbm_bytes_per_launch": d["hbm_write_bytes"] + d.get("hbm_read_bytes_raw", 0.0),
                   "read_bytes": d.get("hbm_read_bytes_raw"), "write_bytes": d["hbm_write_bytes"],
                   "k_pathtrace_valu_wave_instructions_per_launch": d.get("SQ_INSTS_VALU"),
                   "k_pathtrace_lane_utilisation": (d["SQ_THREAD_CYCLES_VALU"] / (64.0 * d["SQ_ACTIVE_INST_VALU"])
                                                    if d.get("SQ_ACTIVE_INST_VALU") else None),
                   "source": f"profiles/{name}_c3_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, KB units)",
                   "config": "1920x1080, 1000 spp, 1 GPU"}, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print("collected", sorted(os.listdir(dst)))
