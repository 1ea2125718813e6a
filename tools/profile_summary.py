#!/usr/bin/env python3
"""Summarise one configuration of tools/profile_round.sh into <outdir>/<cfg>.summary.json + .txt:
per-kernel share of the step, launches per step, mean duration (rocprofv3 --kernel-trace), MFMA busy
fraction, VALU instructions per MFMA, HBM bytes per launch (FETCH_SIZE / WRITE_SIZE, KiB counters; FETCH_SIZE
is listed raw and x2; profiles/r03_fetch_calibration.txt: on this pool the counter is exactly half the bytes for every
coalesced read shape these kernels use, so 2 x FETCH_SIZE + WRITE_SIZE is the traffic)."""
import collections, csv, glob, json, os, sys

out, cfg = sys.argv[1], sys.argv[2]
whole = len(sys.argv) > 3 and sys.argv[3] == "whole"      # shares over the whole run (rollouts, inference probes) instead of one step
base = os.path.join(out, cfg)


def clean(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "")


def counters(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(base, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[clean(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


trace = glob.glob(os.path.join(base, "trace", "**", "*kernel_trace.csv"), recursive=True)
rows = list(csv.DictReader(open(trace[0]))) if trace else []
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [clean(r["Kernel_Name"]) for r in rows]
# one step = from a launch that occurs once per step to its next occurrence: the loss launch, or - round 5: the loss of the
# small-regime models rides in the decoder launches - the fused encoder-side launch that opens their step
marks = [i for i, n in enumerate(names) if "encoder_fwd_kernel" in n]
if len(marks) < 3:
    marks = [i for i, n in enumerate(names) if "rel_lp_fwd" in n]
summary = {"config": cfg, "kernels": []}
try:
    summary["bench"] = json.load(open(os.path.join(out, cfg + ".bench.json")))
except Exception:
    pass
if whole and rows:
    marks = [0, 0, len(rows)]
if len(marks) >= 3:
    a, b = marks[-3], marks[-2]                      # one steady-state step (loss launch to loss launch)
    step = rows[a:b]
    if whole:
        step = rows                                  # every launch of the run (warm-up included: same kernels)
    wall = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3
    agg = collections.OrderedDict()
    for r in step:
        n = clean(r["Kernel_Name"])
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        c, t = agg.get(n, (0, 0.0))
        agg[n] = (c + 1, t + d)
    busy = sum(t for _, t in agg.values())
    fe, wr, m1, m2 = counters("FETCH"), counters("WRITE"), counters("MFMA1"), counters("MFMA2")
    summary.update(launches_per_step=len(step), step_wall_us_under_profiler=round(wall, 1), kernel_time_us=round(busy, 1))
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        rec = {"kernel": n[:140], "launches": c, "total_us": round(t, 1), "mean_us": round(t / c, 2), "share": round(t / busy, 4)}
        if n in m1 and m1[n].get("GRBM_GUI_ACTIVE", 0) > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in m1[n]:
            rec["mfma_busy_frac"] = round(m1[n]["SQ_VALU_MFMA_BUSY_CYCLES"] / (m1[n]["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
        if n in m2 and m2[n].get("SQ_INSTS_MFMA", 0) > 0:
            rec["valu_per_mfma"] = round((m2[n]["SQ_INSTS_VALU"] - m2[n]["SQ_INSTS_MFMA"]) / m2[n]["SQ_INSTS_MFMA"], 2)
        if n in fe and n in wr:
            f, w = fe[n].get("FETCH_SIZE", 0.0), wr[n].get("WRITE_SIZE", 0.0)
            rec["hbm_kib_per_launch"] = {"fetch_raw": round(f, 1), "fetch_x2": round(2 * f, 1), "write": round(w, 1)}
            rec["hbm_GBps_fetch_x2_plus_write"] = round((2 * f + w) * 1024 / (t / c * 1e-6) / 1e9, 1)
        summary["kernels"].append(rec)
json.dump(summary, open(os.path.join(out, cfg + ".summary.json"), "w"), indent=1)
with open(os.path.join(out, cfg + ".summary.txt"), "w") as f:
    b = summary.get("bench", {})
    f.write(f"{cfg}: {b.get('config', {}).get('workload', '')}\n")
    if whole:
        f.write(f"whole run under the profiler ({b.get('ms_per_step')} ms/step reported): {summary.get('launches_per_step')} launches, "
                f"{summary.get('kernel_time_us')} us of kernels; shares over all launches (warm-up included)\n")
    else:
        f.write(f"bench (under the profiler): {b.get('ms_per_step')} ms/step, {b.get('value')} samples/s; one step = "
                f"{summary.get('launches_per_step')} launches, {summary.get('kernel_time_us')} us of kernels\n")
    f.write(f"{'share':>6} {'n':>3} {'mean us':>9} {'MFMA busy':>9} {'VALU/MFMA':>9} {'HBM GB/s':>9}  kernel\n")
    for r in summary["kernels"][:24]:
        f.write(f"{r['share']*100:5.1f}% {r['launches']:3d} {r['mean_us']:9.1f} {str(r.get('mfma_busy_frac', '-')):>9} "
                f"{str(r.get('valu_per_mfma', '-')):>9} {str(r.get('hbm_GBps_fetch_x2_plus_write', '-')):>9}  {r['kernel'][:100]}\n")
print(open(os.path.join(out, cfg + ".summary.txt")).read())
