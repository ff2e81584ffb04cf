"""Compares two rocprofv3 kernel_stats.csv files of the same command: per kernel calls, average us (old -> new), total ms.
usage: python tools/kstat_diff.py old.csv new.csv [steps]"""
import csv
import re
import sys


def load(path):
    out = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            name = re.sub(r"\(.*", "", r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", ""))
            if "at::native" in name:
                name = "aten:" + name.split("<")[0][-30:]
            out[name] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3)
    return out


def main():
    a, b = load(sys.argv[1]), load(sys.argv[2])
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    names = sorted(set(a) | set(b), key=lambda n: -(b.get(n, (0, 0, 0))[1] + a.get(n, (0, 0, 0))[1]))
    ta = sum(v[1] for v in a.values()) / steps
    tb = sum(v[1] for v in b.values()) / steps
    print(f"total kernel ms per step: {ta:.3f} -> {tb:.3f}   launches per step: {sum(v[0] for v in a.values()) / steps:.0f} -> {sum(v[0] for v in b.values()) / steps:.0f}")
    for n in names:
        ca, ma, ua = a.get(n, (0, 0.0, 0.0))
        cb, mb, ub = b.get(n, (0, 0.0, 0.0))
        print(f"{ca / steps:6.1f} -> {cb / steps:6.1f} calls  {ua:7.2f} -> {ub:7.2f} us  {ma / steps:7.3f} -> {mb / steps:7.3f} ms  {n[:100]}")


if __name__ == "__main__":
    main()
