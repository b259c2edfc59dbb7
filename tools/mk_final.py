"""Turns the raw measurement files of one build into the committed profiles/r<round>_final_* artefacts that bench.py reads:

    python tools/mk_final.py <round-tag> <gpurun_out/traffic_TAG.json> <gpurun_out/prof_TAG/r_kernel_trace.csv> <ms_per_step>

  profiles/<round-tag>_final_traffic.json   HBM-side bytes per step (separate FETCH_SIZE / WRITE_SIZE passes of tools/traffic.sh;
                                            FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; init fills subtracted)
  profiles/<round-tag>_gap_summary.txt      per step: sum of kernel durations, sum of the gaps between consecutive kernels, wall
"""
import csv
import json
import sys


def main():
    tag, traffic_json, trace_csv, ms = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
    t = json.load(open(traffic_json))
    fetch_kb = t["FETCH_SIZE"]["total_KB_per_step"]
    write_kb = t["WRITE_SIZE"]["total_KB_per_step"]
    fills = sum(v for k, v, _ in t["WRITE_SIZE"]["top"] if "fillBuffer" in k)
    out = {
        "command": "tools/traffic.sh: rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --steps 1 --warmup 0 --no-variants "
                   "(separate passes)",
        "units": "KB per (1D+1G) step at B=64,T=100 (the run includes model construction: its hipMemset fills are subtracted)",
        "caveat": "under rocprofv3 --pmc every dispatch is serialised and starts with cold L2s (tools/ubench/l2.hip section E under the "
                  "profiler: 100 % TCC misses on a pull that hits 100 % unprofiled), so these are the bytes a launch touches from beyond "
                  "its XCD's L2, an upper bound of the unprofiled HBM/MALL-side traffic",
        "ms_per_step": ms,
        "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb, "init_fill_KB": fills,
        "fetch_correction": "gfx950 FETCH_SIZE reports 1/2 of wide streaming reads (MI355X_MICROARCH.md HBM): doubled",
        "top_fetch": t["FETCH_SIZE"]["top"], "top_write": t["WRITE_SIZE"]["top"],
        "hbm_bytes_per_step": int((2 * fetch_kb + write_kb - fills) * 1024),
    }
    json.dump(out, open("profiles/%s_final_traffic.json" % tag, "w"), indent=1)

    rows = list(csv.DictReader(open(trace_csv)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # steps are delimited by the optimizer kernel of the generator (one k_apply_adam per step)
    ends = [i for i, r in enumerate(rows) if "k_apply_adam" in r["Kernel_Name"]]
    lines = ["# per (1D+1G) step, from the kernel trace of `tools/prof.sh` (rocprofv3 --kernel-trace): kernels, sum of kernel durations,",
             "# sum of positive gaps between consecutive kernels, wall (first start -> last end); ms",
             "# (the first line may be a warm-up step that still captures graph segments; the last line is bench.py's extra step with",
             "#  every launch bracketed by HIP events, issued eagerly: its gaps are the host, not the replayed graphs)"]
    for a, b in zip(ends[:-1], ends[1:]):
        seg = rows[a + 1:b + 1]
        dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e6
        gaps = sum(max(0, int(y["Start_Timestamp"]) - int(x["End_Timestamp"])) for x, y in zip(seg[:-1], seg[1:])) / 1e6
        wall = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
        lines.append("kernels %5d  sum_durations %.3f  sum_gaps %.3f  wall %.3f" % (len(seg), dur, gaps, wall))
    lines.append("# un-profiled wall of the same build: %.3f ms/step (bench.py)" % ms)
    open("profiles/%s_gap_summary.txt" % tag, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[-6:]))


if __name__ == "__main__":
    main()
