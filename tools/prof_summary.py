#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (gpurun_out/prof_*) into the small files committed under profiles/.

  python tools/prof_summary.py --tag r01 --kt gpurun_out/prof_kt/r01_kernel_stats.csv \
      [--fetch gpurun_out/prof_fetch/r01_counter_collection.csv --write gpurun_out/prof_write/r01_counter_collection.csv] \
      --iters 73

Writes profiles/<tag>_kernel_stats.md (rocprofv3 --kernel-trace --stats summary, names shortened),
profiles/<tag>_traffic.json and profiles/traffic.json (per-launch HBM bytes of OUR kernels from the
FETCH_SIZE / WRITE_SIZE passes).  Unit and gfx950 correction per MI355X_MICROARCH.md section HBM:
counters are in KiB; FETCH_SIZE counts 64 B per 128-B request for wide coalesced streaming reads, so
the read side is reported both raw and x2 ("fetch_x2"); WRITE_SIZE is exact."""
import argparse
import csv
import json
import os
import re
import sys
from collections import defaultdict

csv.field_size_limit(sys.maxsize)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

OURS = {"k_preprocess_lean": "preprocess_lean", "k_preprocess_bin": "preprocess_bin", "k_preprocess(": "preprocess_fwd", "k_refine_init": "refine_init",
        "k_pose_load": "refine_init", "k_sh_color": "sh_color", "k_tile_count": "tile_count", "k_tile_scan": "tile_scan",
        "k_tile_emit": "tile_emit", "k_tracking_loss": "tracking_loss", "k_pose_step": "pose_step", "k_pose_init": "pose_step",
        "k_tau_finish": "pose_step", "k_preprocess_bwd": "preprocess_bwd", "k_render_fwd": "render_fwd", "k_render_bwd": "render_bwd",
        "k_ssim_fwd": "ssim_fwd", "k_ssim_bwd": "ssim_bwd", "k_pearson_sums": "pearson_sums", "k_train_loss_finish": "train_loss_finish",
        "k_densification_stats": "densification_stats", "k_backward_prologue": "backward_prologue",
        "k_gradmask_intensity": "gradmask_intensity", "k_gradmask_hist": "gradmask_hist", "k_gradmask_threshold": "gradmask_threshold",
        "k_gradmask_boxes": "gradmask_boxes", "k_gradmask_replica": "gradmask_replica"}


def short(name):
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([\w:]+)", n)
    base = m.group(1) if m else n[:60]
    if "rocprim" in n:
        mm = re.search(r"detail::(radix_sort_\w+|\w*scan\w*|\w+_kernel)", n)
        base = "rocprim::" + (mm.group(1) if mm else "kernel")
    if "at::native" in n:
        mm = re.search(r"at::native::(\w+)<[^>]*?(?:at::native::)?(?:binary_internal::)?(\w+Functor|\w+Ops|\w+_kernel)?", n)
        base = "at::" + (mm.group(1) + (":" + mm.group(2) if mm and mm.group(2) else "") if mm else base)
    return base[:80]


def ours(name):
    for k, v in OURS.items():
        if k in name and ("gsr::" in name or "rocprim" in name):
            return v
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--kt", required=True)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--sq", help="counter_collection.csv of an SQ pass: SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAVES")
    ap.add_argument("--iters", type=int, default=0, help="refinement iterations under the profiler (default: k_render_bwd launches)")
    ap.add_argument("--cmd", default="")
    ap.add_argument("--no-latest", action="store_true", help="do not overwrite profiles/traffic.json / issue.json (secondary passes, e.g. the plain loop)")
    a = ap.parse_args()
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    rows = list(csv.DictReader(open(a.kt)))
    if not a.iters:
        a.iters = sum(int(r["Calls"]) for r in rows if "k_render_bwd" in r["Name"])
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    agg = defaultdict(lambda: [0, 0.0])
    for r in rows:
        key = ours(r["Name"]) or short(r["Name"])
        agg[key][0] += int(r["Calls"])
        agg[key][1] += float(r["TotalDurationNs"])
    with open(os.path.join(ROOT, "profiles", f"{a.tag}_kernel_stats.md"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary ({a.tag})\n\ncommand: `{a.cmd}`\n\n")
        f.write(f"{a.iters} refinement iterations under the profiler; total GPU kernel time {tot/1e6:.2f} ms "
                f"= {tot/1e6/a.iters:.3f} ms/iteration\n\n(rows that are not this library's kernels -- `__amd_rocclr_*`, `at::*` -- are the scene set-up of the "
                f"command (host-to-device uploads in chunks, torch glue) and lie outside the timed loop; their 'us / iteration' is only total / iterations)\n\n"
                f"| kernel | calls | avg us | us / iteration | % |\n|---|---|---|---|---|\n")
        for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
            f.write(f"| {k} | {c} | {t/c/1e3:.2f} | {t/a.iters/1e3:.1f} | {100*t/tot:.1f} |\n")
    traffic = {}
    for which, path in (("fetch", a.fetch), ("write", a.write)):
        if not path:
            continue
        acc = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(path)):
            k = ours(r["Kernel_Name"])
            if k:
                acc[k][0] += 1
                acc[k][1] += float(r["Counter_Value"])
        # per LAUNCH of that kernel (the rocPRIM sort groups several kernels: per rocPRIM kernel launch)
        for k, (c, v) in acc.items():
            traffic.setdefault(k, {})[which + "_KiB_per_iter"] = v / max(c, 1)
            traffic[k]["launches_" + which] = c
    out = {}
    for k, d in traffic.items():
        fe, wr = d.get("fetch_KiB_per_iter", 0.0) * 1024, d.get("write_KiB_per_iter", 0.0) * 1024
        out[k] = {"per": "launch", "launches_profiled": d.get("launches_fetch", d.get("launches_write", 0)),
                  "fetch_raw_bytes": fe, "fetch_x2_bytes": 2 * fe, "write_bytes": wr, "hbm_bytes_corrected": 2 * fe + wr}
    if out:
        json.dump(out, open(os.path.join(ROOT, "profiles", f"{a.tag}_traffic.json"), "w"), indent=1)
        if not a.no_latest:
            json.dump({k: v["hbm_bytes_corrected"] for k, v in out.items()}, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    if a.sq:
        # Issue-slot utilisation per kernel: a CU issues at most one VALU instruction per cycle (4 SIMDs x one wave64 instruction
        # per 4 cycles) -> valu = SQ_INSTS_VALU / SQ_BUSY_CU_CYCLES; the fp32 MFMA occupies its SIMD's pipe for
        # SQ_VALU_MFMA_BUSY_CYCLES (SIMD cycles: / 4 for CU cycles) -> mfma = that / 4 / SQ_BUSY_CU_CYCLES.
        acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
        for r in csv.DictReader(open(a.sq)):
            k = ours(r["Kernel_Name"])
            if k:
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
        issue = {}
        with open(os.path.join(ROOT, "profiles", f"{a.tag}_issue_utilisation.md"), "w") as f:
            f.write(f"# Issue-slot counters of the hot kernels ({a.tag})\n\ncommand: `{a.cmd}` under `rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES "
                    "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS` (counters only); per-launch averages.\n\n"
                    "| kernel | launches | waves | VALU insts | MFMA insts | LDS insts | busy CU cycles | VALU issue | MFMA pipe | VALU insts / wave |\n|---|---|---|---|---|---|---|---|---|---|\n")
            for k, d in acc.items():
                g = lambda n: d.get(n, 0.0) / max(cnt[k].get(n, 1), 1)
                busy = max(g("SQ_BUSY_CU_CYCLES"), 1.0)
                valu, mfma = g("SQ_INSTS_VALU") / busy, g("SQ_VALU_MFMA_BUSY_CYCLES") / 4.0 / busy
                # (two different pipes: waves of one SIMD overlap their MFMA and VALU instructions, so the two fractions are NOT added up --
                # the binding one is the larger)
                issue[k] = {"valu_issue_frac": valu, "mfma_pipe_frac": mfma, "binding_pipe": "valu" if valu >= mfma else "mfma",
                            "valu_insts_per_launch": g("SQ_INSTS_VALU"), "waves_per_launch": g("SQ_WAVES"), "busy_cu_cycles_per_launch": busy,
                            "valu_insts_per_wave": g("SQ_INSTS_VALU") / max(g("SQ_WAVES"), 1.0),
                            "lds_insts_per_wave": g("SQ_INSTS_LDS") / max(g("SQ_WAVES"), 1.0), "mfma_insts_per_wave": g("SQ_INSTS_MFMA") / max(g("SQ_WAVES"), 1.0)}
                f.write(f"| {k} | {cnt[k].get('SQ_WAVES', 0)} | {g('SQ_WAVES'):.0f} | {g('SQ_INSTS_VALU') / 1e6:.2f} M | {g('SQ_INSTS_MFMA') / 1e6:.2f} M | "
                        f"{g('SQ_INSTS_LDS') / 1e6:.2f} M | {busy / 1e6:.1f} M | {valu:.2f} | {mfma:.2f} | {issue[k]['valu_insts_per_wave']:.0f} |\n")
        if not a.no_latest:
            json.dump(issue, open(os.path.join(ROOT, "profiles", "issue.json"), "w"), indent=1)
        print(json.dumps(issue, indent=1))
    print(open(os.path.join(ROOT, "profiles", f"{a.tag}_kernel_stats.md")).read())
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
