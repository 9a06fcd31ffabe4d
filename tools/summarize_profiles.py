#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/collect_profiles.sh (gpurun_out/prof_<tag>/) into the committed summaries under
profiles/: the bench step's kernel statistics and, per micro-benchmarked kernel, one JSON with the kernel-trace average
duration and the PMC counters of the separate --pmc passes (FETCH_SIZE doubled per the gfx950 correction of
MI355X_MICROARCH.md for 16-byte-per-lane streaming reads; WRITE_SIZE exact)."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.environ.get("PROFILES_DST") or os.path.join(ROOT, "profiles")   # PROFILES_DST: scratch summaries of an experiment
os.makedirs(dst, exist_ok=True)


def rows(path):
    with open(path) as f:
        return list(csv.DictReader(f))


def counters(dirname, kernel_substr):
    out = {}
    for f in glob.glob(os.path.join(src, dirname, "*counter_collection.csv")):
        for r in rows(f):
            if kernel_substr in r["Kernel_Name"]:
                out.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}, {k: len(v) for k, v in out.items()}


def avg_duration_ns(dirname, kernel_substr):
    for f in glob.glob(os.path.join(src, dirname, "*kernel_stats.csv")):
        for r in rows(f):
            if kernel_substr in r["Name"]:
                return float(r["AverageNs"]), int(r["Calls"])
    return None, 0


bench_stats = glob.glob(os.path.join(src, "bench", "*kernel_stats.csv"))
if bench_stats:
    shutil.copy(bench_stats[0], os.path.join(dst, "%s_bench_kernel_stats.csv" % tag))
    for line in open(os.path.join(src, "bench.log")):
        if line.startswith("{"):
            open(os.path.join(dst, "%s_bench.json" % tag), "w").write(line)

# The step's dominant instantiation INSIDE the timed region + launches per step and hardware queue, from the bench trace itself
# (VERDICT r5 item 3): profiles/<tag>_bench_in_step.json, read back by bench.py (roofline.in_step, main_stream_launches_per_step).
bench_trace = glob.glob(os.path.join(src, "bench", "*kernel_trace.csv"))
if bench_trace:
    tr = []
    for r in rows(bench_trace[0]):
        tr.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id") or r.get("Stream_Id"),
                   int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)))
    tr.sort()
    marks = [i for i, r in enumerate(tr) if "gen_loss_kernel" in r[2]]
    # the timed region of the default bench.py run: skip the first 15 steps (warm-up) and whatever follows the last step mark
    lo, hi = (marks[15], marks[-1]) if len(marks) > 40 else (marks[0], marks[-1])
    inside = tr[lo:hi]
    nsteps = sum(1 for i in marks if lo <= i < hi)
    # (round 6: a backward pass' weight gradients go out as ONE grouped launch of 440-670 workgroups — two or three ResBlocks'
    #  768 x 41xx..46xx x B products with their riders; per_block_us = duration / ResBlocks in the launch, the figure
    #  comparable with the single-block launches of rounds 1-5 and with roofline.ms_per_launch)
    tn = [((e - b) / 1e3, g // 256) for b, e, k, q, g in inside if "gemm_x5_kernel<false, false" in k and g >= 192 * 256]
    big = [d for d, w in tn]
    per_block = [d / max(1, round(w / 215.0)) for d, w in tn]
    nt = [(e - b) / 1e3 for b, e, k, q, g in inside if "gemm_x5_kernel<true, true" in k and g >= 96 * 256]
    byq = {}
    for b, e, k, q, g in inside:
        byq[q] = byq.get(q, 0) + 1
    order = sorted(byq.items(), key=lambda kv: -kv[1])
    js = {"trace": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-sub --no-other-mode --repeats 3 (tools/collect_profiles.sh)",
          "steps_in_region": nsteps,
          "dominant_tn": {"what": "gemm_x5_kernel<false,false> (weight gradients, TN) launches of >= 192 workgroups inside the step: one "
                                  "grouped launch per backward pass (2-3 ResBlocks: 768 x 4101..4632 x B each, with their riders)",
                          "mean_us": round(sum(big) / len(big), 2), "min_us": round(min(big), 2), "max_us": round(max(big), 2),
                          "launches": len(big), "launches_per_step": round(len(big) / nsteps, 2),
                          "mean_workgroups": round(sum(w for d, w in tn) / len(tn), 1),
                          "per_block_mean_us": round(sum(per_block) / len(per_block), 2)},
          "forward_nt": {"what": "gemm_x5_kernel<true,true> (forward products, NT) launches of >= 96 workgroups inside the step "
                                 "(side streams run theirs with half the workgroups for twice as long)",
                         "mean_us": round(sum(nt) / len(nt), 2), "min_us": round(min(nt), 2), "max_us": round(max(nt), 2),
                         "launches": len(nt)},
          "launches_per_step_by_queue": {"main (cGAN chain)": round(order[0][1] / nsteps, 1),
                                         "side streams": [round(v / nsteps, 1) for _, v in order[1:]],
                                         "total": round(sum(v for _, v in order) / nsteps, 1)}}
    json.dump(js, open(os.path.join(dst, "%s_bench_in_step.json" % tag), "w"), indent=1)
    print("in_step", json.dumps(js["dominant_tn"]), json.dumps(js["launches_per_step_by_queue"]))

specs = [
    ("gemm_nt", "gemm_x5_kernel<true, true", "gemm_x5", {"M": 1024, "N": 768, "K": 4120, "splits": 4},
     "python3 tools/microbench.py gemm --k 4120 --splits 4 --iters 10", 2.0 * 1024 * 768 * 4120,
     4 * (1024 * 4120 + 768 * 4120 + 4 * 1024 * 768)),
    ("gemm_tn", "gemm_x5_kernel<false, false", "gemm_x5_tn", {"M": 768, "N": 4120, "K": 1024, "splits": 0},
     "python3 tools/microbench.py gemm --mode 2 --m 768 --n 4120 --k 1024 --splits 0 --iters 10", 2.0 * 768 * 4120 * 1024,
     4 * (1024 * 768 + 1024 * 4120 + 768 * 4120)),
    ("x4_nt", "gemm_x4_kernel<true, true", "gemm_x4", {"M": 1024, "N": 768, "K": 4120, "splits": 4},
     "DGG_X4=1 python3 tools/microbench.py gemm --k 4120 --splits 4 --iters 10", 2.0 * 1024 * 768 * 4120,
     4 * (1024 * 4120 + 768 * 4120 + 4 * 1024 * 768)),
    ("x4_tn", "gemm_x4_kernel<false, false", "gemm_x4_tn", {"M": 768, "N": 4120, "K": 1024, "splits": 0},
     "DGG_X4=1 python3 tools/microbench.py gemm --mode 2 --m 768 --n 4120 --k 1024 --splits 0 --iters 10", 2.0 * 768 * 4120 * 1024,
     4 * (1024 * 768 + 1024 * 4120 + 768 * 4120)),
    ("bps", "bps_mfma_kernel", "bps_encode", {"B": 1024, "N": 2048, "P": 4096},
     "python3 tools/microbench.py bps --iters 5", 1024 * 2048 * 4096 * 8.0, 1024 * (2048 * 12 + 4096 * 8) + 4096 * 12),
    ("bps_valu", "bps_batched_kernel", "bps_encode_valu", {"B": 1024, "N": 2048, "P": 4096},
     "DGG_BPS=valu python3 tools/microbench.py bps --iters 5", 1024 * 2048 * 4096 * 8.0, 1024 * (2048 * 12 + 4096 * 8) + 4096 * 12),
]
# FETCH_SIZE factor per kernel.  The guide's x2 is for 16-B-per-lane vector loads (the GEMMs' LDS-DMA; bps_mfma_kernel's
# staging and exact stage: raw 13.0 MB x 2 = 26 MB against 25.2 MB of clouds + 48 KB of basis).  Other widths have to be
# calibrated on a known byte count: bps_batched_kernel (the vector-ALU scan, DGG_BPS=valu) reads its clouds through the
# scalar cache, factor 1.
FETCH_FACTOR = {"bps_valu": (1.0, "scalar (s_load) reads: FETCH_SIZE taken as is (calibrated in round 2 on this kernel's "
                                   "known 25.2 MB of clouds); WRITE_SIZE exact")}
for name, ksub, outname, shape, cmd, flops, alg_bytes in specs:
    c = {}
    n = {}
    for d in glob.glob(os.path.join(src, name + "_pmc_*")):
        if os.path.isdir(d):
            cc, nn = counters(os.path.basename(d), ksub)
            c.update(cc)
            n.update(nn)
    dur, calls = avg_duration_ns(name + "_trace", ksub)
    if dur is None:
        continue
    ffac, fnote = FETCH_FACTOR.get(name, (2.0, "gfx950: FETCH_SIZE counts 128-B requests as 64 B for 16-B/lane loads "
                                                "-> doubled; WRITE_SIZE exact"))
    fetch = c.get("FETCH_SIZE", 0.0) * 1024 * ffac   # KB
    write = c.get("WRITE_SIZE", 0.0) * 1024
    out = {"round": int("".join(ch for ch in tag if ch.isdigit())[:2] or 0), "kernel": ksub, "shape": shape,
           "command": "rocprofv3 --pmc <counters> -- " + cmd + " (one pass per counter group; kernel-trace in its own run)",
           "avg_duration_ns_kernel_trace": dur, "launches_in_trace": calls,
           "FETCH_SIZE_KB_raw": c.get("FETCH_SIZE"), "WRITE_SIZE_KB_raw": c.get("WRITE_SIZE"),
           "correction": fnote,
           "fetch_bytes": fetch, "write_bytes": write, "traffic_bytes": fetch + write, "algorithmic_bytes": alg_bytes,
           "counters_avg_per_launch": c, "launches_per_counter": n}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        clk = c["GRBM_GUI_ACTIVE"] / 8.0 / (dur * 1e-9) / 1e9
        out["derived"] = {"shader_clock_ghz_from_GRBM_GUI_ACTIVE": round(clk, 3),
                          "mfma_pipe_busy_frac": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0) / (256 * 4), 4),
                          "tflops_algorithmic": round(flops / (dur * 1e-9) / 1e12, 2)}
        # The vector-issue port, measured: SQ_ACTIVE_INST_VALU counts quad-cycles in which a SIMD issues
        # vector-ALU instructions (MI355X_MICROARCH.md: SQ_ACTIVE_INST_* are in quad-cycles), a matrix instruction holds the
        # same port for 8 of its cycles; denominator = the launch's shader cycles x 1024 SIMDs.
        simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * (256 * 4)
        if "SQ_ACTIVE_INST_VALU" in c:
            va = 4.0 * c["SQ_ACTIVE_INST_VALU"] / simd_cycles
            mi = 8.0 * c.get("SQ_INSTS_MFMA", 0.0) / simd_cycles
            out["derived"].update({"valu_issue_active_frac": round(va, 4), "mfma_issue_frac": round(mi, 4),
                                   "vector_issue_port_busy_frac": round(va + mi, 4)})
        # GRBM_GUI_ACTIVE / 8 / duration reads HIGH on dispatches shorter than ~0.3 ms (MI355X_MICROARCH.md, DVFS give-back): 2.5 GHz
        # on a 60-us launch, 3.5 GHz on a 44-us one, where in-kernel s_memtime / s_memrealtime stamps say 1.85-2.05.
        # SQ_BUSY_CYCLES sums the busy cycles of the 32 shader engines of the chip: / 32 / duration reproduces the stamped
        # clock (round 5: 1.88 GHz on the 43.8-us weight-gradient launch, tools/exp/x5_stamps.py 1.87-1.95) — the
        # occupancies below are also given against THAT cycle count.
        if c.get("SQ_BUSY_CYCLES"):
            cyc = c["SQ_BUSY_CYCLES"] / 32.0
            d2 = {"shader_clock_ghz_from_SQ_BUSY_CYCLES": round(cyc / (dur * 1e-9) / 1e9, 3),
                  "mfma_pipe_busy_frac_of_SQ_BUSY_cycles": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4), 4)}
            if "SQ_ACTIVE_INST_VALU" in c:
                d2["vector_issue_port_busy_frac_of_SQ_BUSY_cycles"] = round(
                    (4.0 * c["SQ_ACTIVE_INST_VALU"] + 8.0 * c.get("SQ_INSTS_MFMA", 0.0)) / (cyc * 256 * 4), 4)
            out["derived"].update(d2)
        if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_ANY" in c:
            out["derived"].update({"wave_wait_any_frac": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4),
                                   "wave_wait_inst_any_frac": round(c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4)})
    json.dump(out, open(os.path.join(dst, "%s_%s_pmc.json" % (tag, outname)), "w"), indent=1)
    print(outname, json.dumps(out.get("derived")), "traffic MB %.1f alg MB %.1f dur us %.1f" % ((fetch + write) / 1e6, alg_bytes / 1e6, dur / 1e3))
