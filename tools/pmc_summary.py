"""Reduces rocprofv3 --pmc counter_collection.csv files to per-kernel HBM traffic (profiles/pmc_traffic.json).

  --reduce <counter_collection.csv> <out.csv>   per kernel: dispatches, sum and mean of the counter
  --merge  <fetch.csv> <write.csv> <out.json>   per kernel: HBM bytes per launch

Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB;
FETCH_SIZE reports exactly half the bytes of wide (16 B/lane) coalesced streaming reads on gfx950, so the read side
is doubled for kernels whose global loads are 16-byte vector loads (every kernel of libadp_hip.so that streams
activations does that); the raw values are kept next to the corrected ones.  The calibration row is
v_noise_kernel: it reads 2 and writes 2 tensors of known size.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    depth, out = 0, []
    for ch in name:  # cut the argument list, keep template arguments
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()


def reduce(src, dst, trace=None):
    """Per kernel and counter: dispatches, sum, mean; with the kernel trace of the same run also the mean duration (ns) of
    the kernel's dispatches (joined on Dispatch_Id)."""
    dur = {}
    if trace:
        with open(trace) as f:
            for row in csv.DictReader(f):
                dur[row["Dispatch_Id"]] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    with open(src) as f:
        for row in csv.DictReader(f):
            a = acc[(short(row["Kernel_Name"]), row["Counter_Name"])]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
            a[2] += dur.get(row.get("Dispatch_Id"), 0.0)
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "sum", "mean", "mean_duration_ns"])
        for (k, c), (n, s, d) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, c, n, f"{s:.1f}", f"{s / n:.3f}", f"{d / n:.1f}"])


def merge(fetch, write, dst, mfma=None, grbm=None):
    def load(p, counter=None):
        with open(p) as f:
            return {r["kernel"]: (int(r["dispatches"]), float(r["mean"]), float(r.get("mean_duration_ns") or 0.0))
                    for r in csv.DictReader(f) if counter is None or r["counter"] == counter}
    fe, wr = load(fetch), load(write)
    busy = load(mfma, "SQ_VALU_MFMA_BUSY_CYCLES") if mfma else {}
    gui = load(grbm, "GRBM_GUI_ACTIVE") if grbm else {}
    out = {}
    for k in sorted(set(fe) | set(wr)):
        rd_kib = fe.get(k, (0, 0.0, 0.0))[1]
        wr_kib = wr.get(k, (0, 0.0, 0.0))[1]
        out[k] = {"dispatches_per_pass": fe.get(k, wr.get(k))[0], "fetch_size_kib_raw": round(rd_kib, 2),
                  "write_size_kib_raw": round(wr_kib, 2),
                  "hbm_bytes_per_launch": int((2.0 * rd_kib + wr_kib) * 1024)}
        if k in busy and busy[k][1] > 0 and busy[k][2] > 0:
            # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs (64 cycles per v_mfma_f32_32x32x2_f32); the duration
            # is the kernel's own dispatch time in the same counter pass.  GRBM_GUI_ACTIVE carries a large fixed
            # per-dispatch offset under the profiler (a 2.5 us copy reports ~37k cycles per XCD), so the busy fraction is
            # rated against the 2.4 GHz peak clock instead: = executed MFMA flops / the 157.3 TF matrix peak.
            out[k]["mfma_busy"] = round(busy[k][1] / NUM_SIMD / (busy[k][2] * PEAK_CLOCK_GHZ), 4)
            out[k]["mfma_busy_cycles_per_simd"] = round(busy[k][1] / NUM_SIMD, 1)
            out[k]["duration_ns_in_counter_pass"] = round(busy[k][2], 1)
            if k in gui and gui[k][2] > 0:
                out[k]["grbm_gui_active_per_xcd"] = round(gui[k][1] / NUM_XCD, 1)
    with open(dst, "w") as f:
        json.dump({"csrc_sha16": csrc_sha16(),
                   "note": "hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE doubled per the "
                           "gfx950 correction for 16-byte coalesced reads (MI355X_MICROARCH.md, HBM); reads served by "
                           "the 256 MiB Infinity Cache are included by these counters, not excluded.  mfma_busy = "
                           "SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (kernel duration x 2.4 GHz): the share of the matrix "
                           "pipes' peak issue capacity the kernel used (64 busy cycles per v_mfma_f32_32x32x2_f32), i.e. "
                           "EXECUTED MFMA flops / 157.3 TF -- for the Winograd variants executed flops are two thirds of "
                           "the direct-form flops bench.py rates them with; durations are those of the counter pass "
                           "(eager, profiled: 5-15 % longer than in the timed run)",
                   "kernels": out}, f, indent=1, sort_keys=True)


NUM_XCD, NUM_SIMD, PEAK_CLOCK_GHZ = 8, 1024, 2.4


def csrc_sha16():
    """Hash of the kernel sources the counters were collected from (bench.py compares it with the tree it runs in and marks
    stored counters of other code as stale)."""
    import hashlib
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audio_diffusion_pytorch_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(root)):
        if name.endswith((".hip", ".h")) and name != "probe.hip":  # (the calibration probes are not on the step's path)
            with open(os.path.join(root, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]

if __name__ == "__main__":
    if sys.argv[1] == "--reduce":
        reduce(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] else None)
    else:
        merge(*sys.argv[2:7])
