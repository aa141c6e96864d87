#!/usr/bin/env python3
"""Turns gpurun_out/<tag>/ (written by profiles/collect.sh on the GPU box) into the
committed summaries profiles/<tag>_summary.md and profiles/pmc_traffic.json.

HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB
per dispatch; on gfx950 FETCH_SIZE under-reports wide coalesced reads by up to 2x
and is uncalibrated for other widths, so both the raw and the doubled figure are
listed and bench.py reports the raw sum (lower bound)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def kernel_mode(kernel_name):
    """Arithmetic mode of a shading kernel from its namespace (libm_math / exact_math / fast_math)"""
    for mode in ("libm", "exact", "fast"):
        if mode + "_math" in kernel_name:
            return mode
    return None


def counters(directory):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for path in files:
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].split("(")[0]
            per_kernel[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
            meta[name] = {k: row.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size", "Grid_Size", "Workgroup_Size")}
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in per_kernel.items()}, meta


def kernel_stats(directory):
    out = []
    for path in glob.glob(os.path.join(directory, "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            out.append(row)
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    # the mode the collection ran in (profiles/collect.sh <tag> <mode>); entries of
    # profiles/pmc_traffic.json are keyed by configuration and mode and carry the hash of the kernel
    # sources they were measured with (bench.py refuses stale ones)
    run_mode = sys.argv[2] if len(sys.argv) > 2 else "libm"
    import bench
    csrc_hash = bench.kernel_source_hash()
    base = os.path.join(ROOT, "gpurun_out", tag)
    lines = ["# rocprofv3 summary %s" % tag, "",
             "Collected by `profiles/collect.sh %s` on an MI355X (gfx950), summarised by `profiles/summarize.py`." % tag, ""]
    traffic = {}
    traffic_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(traffic_path):
        traffic = json.load(open(traffic_path))
    # workloads of profiles/collect.sh: BASELINE configs, the north_star target shape, config 3 on the large scene; the
    # entries of pmc_traffic.json are keyed the way bench.py looks them up ("config3_libm", "config3_libm_large")
    workloads = [w for w in ("2", "3", "4", "target", "3_large") if glob.glob(os.path.join(base, "cfg%s_*" % w))]
    sizes = {"4": (3840, 2160)}
    for workload in workloads:
        cfg, _, scene = workload.partition("_")
        scene = scene or "bench"
        suffix = "" if scene == "bench" else "_" + scene
        lines += ["## BASELINE config %s%s" % (cfg, "" if scene == "bench" else " on the %s scene" % scene), ""]
        details = os.path.join(base, "cfg%s_bench_details.json" % workload)
        bench = os.path.join(base, "cfg%s_bench.json" % workload)
        d = None
        if os.path.exists(details):
            d = json.load(open(details))
        elif os.path.exists(bench):
            text = [l for l in open(bench).read().splitlines() if l.startswith("{")]
            d = json.loads(text[-1]) if text else None
        if d:
            parity = d.get("parity") or {}
            lines += ["bench.py (un-profiled): **%.1f %s**, %.4f ms/step, shade_pixels %.4f ms (HIP events), %.0f Mrays/s, parity: %s pixels of %s differ from the oracle" % (
                d["value"], d["unit"], d["ms_per_step"], d["roofline"]["kernel_ms"], d.get("Mrays_per_s", 0), parity.get("pixels_differing", "-"), parity.get("sample_pixels", "-")), ""]
        stats = kernel_stats(os.path.join(base, "cfg%s_trace" % workload))
        if stats:
            lines += ["`rocprofv3 --kernel-trace --stats` (top kernels):", "", "| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
            for row in stats[:8]:
                lines.append("| `%s` | %s | %.1f | %.1f | %.1f | %s |" % (row["Name"][:80], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3, row["Percentage"]))
            lines.append("")
        serial = kernel_stats(os.path.join(base, "cfg%s_serial" % workload))
        serial_us = {}
        if serial:
            lines += ["The same with `--frames-in-flight 1` (every kernel alone on the GPU):", "", "| kernel | calls | avg us | min us | max us |", "|---|---|---|---|---|"]
            for row in [r for r in serial if any(k in r["Name"] for k in ("shade_pixels", "trace_shadow_rays", "resolve_shadow", "light_shafts"))][:4]:
                lines.append("| `%s` | %s | %.1f | %.1f | %.1f |" % (row["Name"][:80], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
                serial_us[row["Name"].split("(")[0]] = float(row["AverageNs"]) / 1e3
            lines.append("")
        merged, meta = {}, {}
        for p in (1, 2, 3, 4, 5, 6, 7):
            c, m = counters(os.path.join(base, "cfg%s_pmc%d" % (workload, p)))
            for k, v in c.items():
                merged.setdefault(k, {}).update(v)
            meta.update(m)
        for kernel, c in merged.items():
            # rocprofv3 reports VGPR_Count in allocation units of two registers on gfx950 (84 = the 168 registers that
            # profiles/tools/kernel_resources.sh reads from the code object) and LDS_Block_Size without the dynamic
            # part (the polygon tables of shade_pixels, 11.3 KB per wave at V = 5, are dynamic LDS)
            labels = {"VGPR_Count": "VGPR_Count [units of 2 registers]", "LDS_Block_Size": "LDS_Block_Size [static part only, bytes]"}
            shown = []
            for key, value in meta.get(kernel, {}).items():
                if value is None:
                    continue
                shown.append("%s=%s" % (labels.get(key, key), value))
                if key == "VGPR_Count" and str(value).isdigit():
                    shown.append("registers=%d" % (2 * int(value)))
            lines += ["PMC, per dispatch of `%s` (%s):" % (kernel[:70], ", ".join(shown)), ""]
            lines += ["| counter | value |", "|---|---|"]
            for name in sorted(c):
                lines.append("| %s | %.5g |" % (name, c[name]))
            gui = c.get("GRBM_GUI_ACTIVE")
            if gui and "SQ_WAVE_CYCLES" in c:
                cycles = gui / 8.0  # summed over the 8 XCDs
                simds = 1024.0
                lines += ["", "Derived (SQ_* count quad-cycles; GRBM_GUI_ACTIVE is summed over 8 XCDs):", "",
                          "- kernel length ~ %.0f cycles; mean resident waves per SIMD = %.2f" % (cycles, 4 * c["SQ_WAVE_CYCLES"] / (simds * cycles)),
                          # (a "VALU busy" share of SIMD cycles used to be printed here from SQ_ACTIVE_INST_VALU; its unit on gfx950 is not
                          # the quad-cycle the formula assumed - it came out above 100 % -, so the issue floor below, from instruction
                          # counts and measured issue costs, is the only VALU figure that is reported)
                          "- lane utilisation of VALU instructions = %.1f %%" % (100 * c.get("SQ_THREAD_CYCLES_VALU", 0) / max(64 * c.get("SQ_ACTIVE_INST_VALU", 1), 1)),
                          "- wave time: %.1f %% issuing, %.1f %% waiting on memory (s_waitcnt), %.1f %% issue stalls" % (
                              100 * c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], 100 * c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], 100 * c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"])]
            if "SQ_INSTS_VALU" in c:
                # What a wave64 VALU instruction costs its SIMD was measured per class on this GPU
                # (profiles/tools/valu_rate.hip, profiles/r02d_valu_rate.txt): v_fma / v_mul / v_add / v_mov
                # and the plain integer ALU ops 2.5 clocks, conversions, compares, selects, min / max,
                # shifts, v_perm and the v_div_* helpers 4.3, v_rcp / v_rsq / v_sqrt 8.2.  The counters
                # separate ADD, MUL, FMA and the transcendental instructions; the rest is a mixture of
                # 2.5-clock (v_mov, v_and, v_add_u32) and 4.3-clock instructions, so the floor is a range.
                total = c["SQ_INSTS_VALU"]
                alone = serial_us.get(kernel)
                if "SQ_INSTS_VALU_FMA_F32" in c:
                    full = c["SQ_INSTS_VALU_ADD_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + c["SQ_INSTS_VALU_FMA_F32"]
                    trans = c.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
                    rest = max(total - full - trans, 0.0)
                    low = (2.5 * full + 8.2 * trans + 2.5 * rest) / 1024.0 / 2400.0
                    high = (2.5 * full + 8.2 * trans + 4.3 * rest) / 1024.0 / 2400.0
                    lines.append("- VALU instruction mix: %.3g wave instructions = %.0f %% add / mul / fma (2.5 clocks each), %.1f %% transcendental (8.2), %.0f %% other (2.5 - 4.3)" % (
                        total, 100 * full / total, 100 * trans / total, 100 * rest / total))
                    lines.append("- VALU issue floor at 2.4 GHz on 1024 SIMDs with the measured issue costs: %.1f - %.1f us%s" % (
                        low, high, (" = %.0f - %.0f %% of the %.1f us the kernel takes alone (un-profiled)" % (100 * low / alone, 100 * high / alone, alone)) if alone else ""))
                    floor_us = 0.5 * (low + high)
                else:
                    floor_us = 4 * c.get("SQ_ACTIVE_INST_VALU", total) / 1024.0 / 2400.0
                    lines.append("- VALU issue floor at 4 clocks per instruction: %.1f us" % floor_us)
                if "shade_pixels" in kernel or "trace_shadow_rays" in kernel or "resolve_shadow" in kernel or "light_shafts" in kernel:
                    valu_floor = traffic.setdefault("config%s_%s%s_valu_floor_us" % (cfg, kernel_mode(kernel) or run_mode, suffix), {})
                    valu_floor[kernel.split("::")[-1].split("<")[0]] = round(floor_us, 2)
                if "shade_pixels" in kernel and "SQ_INSTS_VALU_FMA_F32" in c:
                    flop = 64.0 * (c["SQ_INSTS_VALU_ADD_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + 2.0 * c["SQ_INSTS_VALU_FMA_F32"])
                    lines.append("- FP32 arithmetic: %.4g FLOP per dispatch (ADD + MUL + 2 FMA wave instructions x 64 lanes)%s" % (
                        flop, (" = %.1f TFLOP/s over the %.1f us alone = %.1f %% of the 157.3 TFLOP/s FP32 vector peak" % (flop / alone / 1e6, alone, 100 * flop / alone / 1e6 / 157.3)) if alone else ""))
                    traffic.setdefault("config%s_%s%s" % (cfg, kernel_mode(kernel) or run_mode, suffix), {})["fp32_flop_per_launch"] = flop
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                raw = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
                lines += ["- HBM traffic per dispatch: FETCH_SIZE %.1f MiB (x2 correction: %.1f MiB) + WRITE_SIZE %.1f MiB = %.1f MB raw" % (
                    c["FETCH_SIZE"] / 1024, 2 * c["FETCH_SIZE"] / 1024, c["WRITE_SIZE"] / 1024, raw / 1e6)]
                if "TCC_HIT_sum" in c:
                    lines.append("- L2 hit rate %.1f %%" % (100 * c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1)))
                if "shade_pixels" in kernel:
                    w, h = sizes.get(cfg, (1920, 1080))
                    entry = traffic.setdefault("config%s_%s%s" % (cfg, kernel_mode(kernel) or run_mode, suffix), {})
                    # (FETCH_SIZE with the guide's x2 correction for wide coalesced reads on gfx950)
                    entry.update({"width": w, "height": h, "scene": scene, "hbm_bytes_per_launch": int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024), "hbm_bytes_per_launch_uncorrected": int(raw),
                                  "source": "profiles/%s_summary.md" % tag, "csrc_hash": csrc_hash})
            lines.append("")
    open(os.path.join(ROOT, "profiles", "%s_summary.md" % tag), "w").write("\n".join(lines) + "\n")
    json.dump(traffic, open(traffic_path, "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
