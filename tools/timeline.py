"""Timeline of one replayed step from a rocprofv3 kernel trace: python tools/timeline.py <r_kernel_trace.csv> [step index from the end]
-> per kernel: start offset from the step's first kernel (us), duration (us), queue, name; then the idle gaps of the device (no kernel
running on any queue) longer than 5 us.  Steps are delimited by k_apply_adam (one per step)."""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "k_apply_adam" in r["Kernel_Name"]]
    a, b = ends[-back - 1], ends[-back]
    seg = rows[a + 1:b + 1]
    t0 = int(seg[0]["Start_Timestamp"])
    busy_until = t0
    gaps = []
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("rsr::", "").replace("void ", "")
        name = name[:name.index("(")] if "(" in name else name
        print("%9.1f %8.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), name[:60]))
        if s - busy_until > 5000:
            gaps.append(((busy_until - t0) / 1e3, (s - busy_until) / 1e3))
        busy_until = max(busy_until, e)
    print("# step wall %.1f us, %d kernels; idle gaps > 5 us: %s" % ((busy_until - t0) / 1e3, len(seg),
                                                                   ", ".join("%.0f@%.0f" % (g, at) for at, g in gaps)))


if __name__ == "__main__":
    main()
