"""Reduce a rocprofv3 --kernel-trace CSV of tools/e2e_bench.py to ONE steady-state training step: the window between
the last two nsgp_project_kernel launches.  Prints GPU-busy fraction and the top kernels of that step.
Usage: python3 tools/step_window.py <kernel_trace.csv> [top_n]"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    top_n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "nsgp_project_kernel" in r["Kernel_Name"]]
    a, b = marks[-2], marks[-1]
    win = rows[a + 1:b + 1]
    t0, t1 = int(rows[a]["End_Timestamp"]), int(rows[b]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in win)
    agg = collections.defaultdict(lambda: [0, 0])
    for r in win:
        k = agg[r["Kernel_Name"][:110]]
        k[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        k[1] += 1
    print(f"step window {1e-6 * (t1 - t0):.2f} ms, {len(win)} kernel launches, kernels busy {1e-6 * busy:.2f} ms ({busy / (t1 - t0):.1%})")
    for name, (ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top_n]:
        print(f"{1e-6 * ns:9.3f} ms  {n:5d}x  {name}")


if __name__ == "__main__":
    main()
