"""Timeline of one bench step from a rocprofv3 --kernel-trace --memory-copy-trace run (csv): kernels and copies of the
step that starts at the N-th matcher launch, with the gaps between consecutive GPU activities.

    python tools/step_timeline.py <trace dir> [step index] [matcher launches per step: 2 = the default two chunks]
"""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    step = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rows = list(csv.DictReader(open(glob.glob(d + "/*/*kernel_trace.csv")[0])))
    cf = glob.glob(d + "/*/*memory_copy_trace.csv")          # optional: a --kernel-trace only run has none
    cop = list(csv.DictReader(open(cf[0]))) if cf else []
    al = sorted(int(r["Start_Timestamp"]) for r in rows if "allpairs" in r["Kernel_Name"])
    lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    t_a, t_b = al[step * lanes], al[(step + 1) * lanes]
    ev = []
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if t_a - 100000 <= s < t_b - 100000:
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
            ev.append((s, e, name, "q" + str(r.get("Queue_Id", ""))))
    for r in cop:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if t_a - 100000 <= s < t_b - 100000:
            ev.append((s, e, "copy " + r["Direction"].replace("MEMORY_COPY_", ""), ""))
    ev.sort()
    busy_end = 0
    for s, e, name, q in ev:
        gap = (s - busy_end) / 1e3 if busy_end and s > busy_end else 0.0
        print(f"{(s - t_a) / 1e6:8.3f} ms  {(e - s) / 1e3:9.1f} us  {q:4s} {name[:48]:48s}" + (f"   <- GPU idle {gap:.0f} us" if gap > 20 else ""))
        busy_end = max(busy_end, e)
    print(f"step period {(t_b - t_a) / 1e6:.3f} ms")


if __name__ == "__main__":
    main()
