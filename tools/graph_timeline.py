"""One step of a rocprofv3 --kernel-trace csv as a timeline: start (us from the end of the previous optimizer kernel),
duration, gap to the furthest end so far (negative: overlap), stream, kernel -- and per kernel how much it moved that
front (its share of the critical path under the profiler's serialisation).
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --workload cfg4 --steps 10 --warmup 3 ...
    python tools/graph_timeline.py DIR/*/*_kernel_trace.csv > profiles/rN_cfg4_graph_timeline.txt"""
import collections
import csv
import re
import sys


def name(r):
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", n)
    return (m.group(1) if m else n)[:64]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"], name(r)) for r in rows)
    idx = [i for i, e in enumerate(ev) if e[3].startswith("adam_step")]
    a, b = idx[-3], idx[-2]
    t0 = cur = ev[a][1]
    front = collections.defaultdict(float)
    count = collections.Counter()
    print(f"# one step of {sys.argv[1].split('/')[-1]}: {b - a} kernels, {(ev[b][1] - t0) / 1e3:.1f} us from optimizer to optimizer (profiled)")
    print(f"# {'start us':>9s} {'dur us':>8s} {'gap us':>8s} stream kernel")
    for s, e, st, n in ev[a + 1:b + 1]:
        print(f"{(s - t0) / 1e3:11.1f} {(e - s) / 1e3:8.1f} {(s - cur) / 1e3:8.1f} {st:>6s} {n}")
        front[n] += max(0, e - max(cur, s))
        count[n] += 1
        cur = max(cur, e)
    print("# time by which each kernel moved the front (us), launches")
    for k, v in sorted(front.items(), key=lambda x: -x[1]):
        print(f"# {v / 1e3:9.1f} {count[k]:4d} {k}")


if __name__ == "__main__":
    main()
