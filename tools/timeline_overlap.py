"""Reads a rocprofv3 --kernel-trace CSV and reports, for every collective kernel (RCCL / NCCL device kernels), how much of its
run time another kernel of the same process was running too -- the timeline evidence for "all-reduce overlapped with backward".
usage: python tools/timeline_overlap.py <kernel_trace.csv> [--last-fraction 0.5] [--out summary.txt]"""
import csv
import sys


def main():
    path = sys.argv[1]
    frac = float(sys.argv[sys.argv.index("--last-fraction") + 1]) if "--last-fraction" in sys.argv else 0.5
    out = open(sys.argv[sys.argv.index("--out") + 1], "w") if "--out" in sys.argv else sys.stdout
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            try:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                             r.get("Stream_Id", ""), r.get("Queue_Id", "")))
            except (KeyError, ValueError):
                continue
    rows.sort()
    if not rows:
        print("no kernel rows", file=out)
        return
    t0, t1 = rows[0][0], rows[-1][1]
    cut = t1 - (t1 - t0) * frac  # the tail of the run = the graph replays (the head holds warm-ups and the capture)
    rows = [r for r in rows if r[0] >= cut]
    is_coll = lambda n: any(k in n for k in ("nccl", "rccl", "Nccl", "Rccl", "oneRank", "OneRank"))  # noqa: E731
    coll = [r for r in rows if is_coll(r[2])]
    other = [r for r in rows if not is_coll(r[2])]
    print(f"{len(rows)} kernels in the last {frac:.0%} of the trace; {len(coll)} collective kernels; "
          f"names: {sorted({r[2][:60] for r in coll})}", file=out)
    print(f"queues used: collectives {sorted({r[4] for r in coll})}, compute {sorted({r[4] for r in other})}; "
          f"streams: collectives {sorted({r[3] for r in coll})}, compute {sorted({r[3] for r in other})}", file=out)
    # overlap of each collective kernel with the union of compute kernels
    import bisect
    starts = [r[0] for r in other]
    tot, tot_ov, stalls = 0, 0, []
    for s, e, name, st, q in coll:
        i = bisect.bisect_left(starts, s)
        j = max(0, i - 64)
        ov = 0
        segs = []
        for os_, oe, *_ in other[j:]:
            if os_ >= e:
                break
            a, b = max(s, os_), min(e, oe)
            if b > a:
                segs.append((a, b))
        segs.sort()
        cur_a = cur_b = None
        for a, b in segs:
            if cur_b is None or a > cur_b:
                if cur_b is not None:
                    ov += cur_b - cur_a
                cur_a, cur_b = a, b
            else:
                cur_b = max(cur_b, b)
        if cur_b is not None:
            ov += cur_b - cur_a
        tot += e - s
        tot_ov += ov
        stalls.append(((e - s) / 1e3, ov / max(1, e - s)))
    if coll:
        print(f"collective kernel time {tot / 1e6:.3f} ms, of which {tot_ov / 1e6:.3f} ms ({tot_ov / max(1, tot):.1%}) with a compute "
              f"kernel running concurrently", file=out)
        for us, f_ in stalls[:24]:
            print(f"   collective kernel {us:9.1f} us, {f_:6.1%} of it concurrent with compute", file=out)
    # how much wall time has NO kernel running between compute kernels while a collective runs (gaps the collective causes)
    busy = 0
    cur_a = cur_b = None
    for s, e, *_ in rows:
        if cur_b is None or s > cur_b:
            if cur_b is not None:
                busy += cur_b - cur_a
            cur_a, cur_b = s, e
        else:
            cur_b = max(cur_b, e)
    if cur_b is not None:
        busy += cur_b - cur_a
    span = rows[-1][1] - rows[0][0]
    print(f"window {span / 1e6:.3f} ms, some kernel running {busy / 1e6:.3f} ms ({busy / span:.1%})", file=out)


if __name__ == "__main__":
    main()
