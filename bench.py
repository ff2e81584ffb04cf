"""Headline benchmark: denoising steps/s (UNetV0 fwd+bwd) at [B,2,2**18] on N MI355X (BASELINE.json `metric`).

A "step" = one `loss = model(audio); loss.backward()` of DiffusionModel(UNetV0 README config) on one batch of
synthetic randn waveforms already resident in HBM (VDiffusion noising + U-Net forward + MSE + full backward
incl. every weight gradient; with N > 1 also the RCCL gradient all-reduce).  Workload at N=1 = BASELINE.json
configs[1]: unconditional UNetV0 channels=[8,32,64,128,256,512,512,1024,1024], batch 4, fp32.  Weak scaling:
every rank runs batch 4.  Prints ONE JSON line on rank 0.

  python bench.py                                   # 1 GPU, defaults
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHANNELS = [8, 32, 64, 128, 256, 512, 512, 1024, 1024]
FACTORS = [1, 4, 4, 4, 2, 2, 2, 2, 2]
ITEMS = [1, 2, 2, 2, 2, 2, 2, 4, 4]
LENGTH = 2 ** 18
PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: FP32 matrix peak (v_mfma_f32_32x32x2_f32)
PEAK_HBM_GBPS = 8000.0         # HBM3E spec; 6290 GB/s measured copy ceiling


def build_model(dev):
    import audio_diffusion_pytorch_amd as adp
    torch.manual_seed(0)
    model = adp.DiffusionModel(net_t=adp.UNetV0, in_channels=2, channels=CHANNELS, factors=FACTORS, items=ITEMS)
    return model.to(dev)


def _cpu_worker(threads: int, batch: int, nsteps: int) -> None:
    """Child process of cpu_baseline: `nsteps` timed fwd+bwd steps (after one warm-up) of the oracle on `threads` host
    threads at the benchmarked batch; prints the per-step seconds as JSON."""
    from oracle import vdiffusion as ovd
    from oracle.a_unet_restatement import UNetV0Oracle
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    net = UNetV0Oracle(in_channels=2, channels=CHANNELS, factors=FACTORS, items=ITEMS)
    x = torch.randn(batch, 2, LENGTH)
    times = []
    for i in range(nsteps + 1):
        for p in net.parameters():
            p.grad = None
        t0 = time.perf_counter()
        loss = ovd.v_loss(net, x, torch.randn_like(x), torch.rand(batch))
        loss.backward()
        if i > 0:  # step 0 = warm-up (oneDNN primitive creation)
            times.append(time.perf_counter() - t0)
    print("CPU_WORKER " + json.dumps(times), flush=True)


def cpu_baseline(batch: int, sweep=(16, 32, 64, 128), final_steps: int = 5):
    """The reference CPU path (oracle restatement of the reference's a_unet composition + the live v-diffusion math) on
    this host's cores AT THE BENCHMARKED BATCH: one timed fwd+bwd step per thread count of `sweep` (each in its own process
    under a timeout -- with every hardware thread of a large host oneDNN's 8-channel depth-0 convs oversubscribe and a step
    takes minutes, which must not hang the bench), then `final_steps` steps with the best setting, median reported.  A
    baseline, never a target."""
    import subprocess
    host = os.cpu_count() or 1
    tried = {}

    def run(threads, nsteps, timeout):
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(threads), str(batch), str(nsteps)],
                                 capture_output=True, text=True, timeout=timeout).stdout
            for line in out.splitlines():
                if line.startswith("CPU_WORKER "):
                    return json.loads(line[len("CPU_WORKER "):])
        except (subprocess.TimeoutExpired, OSError, ValueError):
            pass
        return None

    for th in sorted({min(t, host) for t in sweep}):
        r = run(th, 1, 90)
        tried[th] = None if r is None else round(r[0], 3)
    ok = {t: v for t, v in tried.items() if v is not None}
    if not ok:
        return {"error": "no CPU-baseline trial finished inside its timeout", "host_cores": host, "host_cpu": _cpu_model()}
    best = min(ok, key=ok.get)
    times = run(best, final_steps, 60 + 3 * final_steps * ok[best]) or [ok[best]]
    med = sorted(times)[len(times) // 2]
    return {"value": round(1.0 / med, 4), "unit": "denoising steps/s", "cores": best, "host_cores": host,
            "host_cpu": _cpu_model(), "kind": "port",
            "thread_sweep_s_per_step": {str(k): v for k, v in tried.items()},
            "sample": f"median of {len(times)} fwd+bwd steps of the same UNetV0 at the BENCHMARKED batch [{batch},2,2**18] "
                      f"({med:.2f} s each, fp32, torch CPU, {best} threads = the best of the sweep "
                      f"{sorted(tried)} with one timed step each, one warm-up step per process)"}


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def winograd_fraction(kernel: str) -> float:
    """Share of the direct-form flops a kernel instantiation EXECUTES on the matrix cores: 1/2 for the Winograd F(4,3) kernels
    (conv_mm4, wgrad_mm's W4 flag, conv_tile32), 2/3 for F(2,3) (the WN flag of conv_mm / wgrad_mm), 1 otherwise.  The `roofline`
    objects rate kernels in DIRECT-FORM flops (the work the reference's conv does), so an F(4,3) kernel can read above 1.0 of the
    f32 MFMA peak; `frac_executed` = frac x this share is what the matrix pipes actually sustain."""
    head, _, rest = kernel.partition("<")
    args = [a.strip() for a in rest.rsplit(">", 1)[0].split(",")] if rest else []
    if head.endswith("conv_mm4_kernel") or head.endswith("conv_tile32_kernel") or head.endswith("conv_tilek_kernel"):
        return 0.5
    if head.endswith("wgrad_mm_kernel") and len(args) >= 7:
        return 0.5 if (len(args) >= 8 and args[7] == "true") else (2.0 / 3.0 if args[6] == "true" else 1.0)
    if head.endswith("conv_mm_kernel") and len(args) >= 9:
        return 2.0 / 3.0 if args[8] == "true" else 1.0
    return 1.0


def profiled_step(model, step):
    """Runs `step()` once with every C-ABI launch bracketed by HIP events on its launch stream; returns the records
    [(call, kernel, meta, ms), ...] (see audio_diffusion_pytorch_amd._C.profile_collect)."""
    from audio_diffusion_pytorch_amd import _C
    for p in model.parameters():
        p.grad = None
    _C.PROFILE = []
    try:
        step()
    finally:
        recs = _C.profile_collect()   # waits for the events
    return recs


def roofline_leg(model, x, top: int = 14):
    """Three instrumented eager steps (after two warm ones), averaged per step: while recording is on, libadp_hip.so brackets EVERY kernel launch with a pair of
    HIP events recorded on the stream the kernel is launched on (`adp_launch_trace` / `adp_launch_times`) and names
    the kernel instantiation (the spelling rocprofv3 prints); the host layer attaches the ALGORITHMIC flops / bytes of
    the call.  Reports
      roofline                 the single kernel with the largest total time, against the bound that limits it
                               (f32 MFMA peak for the implicit-GEMM convs, HBM for everything else);
      roofline_hbm_convblock   the depth-0/1 ConvBlock convs (north_star's HBM target) against the HBM peak;
      kernels                  the `top` kernels by total time.
    `traffic` (HBM bytes per launch from the rocprofv3 PMC passes) is looked up in profiles/pmc_traffic.json, which
    tools/pmc_summary.py writes from the FETCH_SIZE / WRITE_SIZE passes of this same command."""
    NREP = 3  # instrumented steps (after two warm ones: the clocks settle); everything below is per step
    for _ in range(2):
        profiled_step(model, lambda: model(x).backward())
    recs = []
    for _ in range(NREP):
        recs += profiled_step(model, lambda: model(x).backward())
    agg = {}
    for call, kern, meta, ms in recs:
        a = agg.setdefault(kern, {"launches": 0, "ms": 0.0, "flops": 0, "bytes": 0})
        a["launches"] += 1
        a["ms"] += ms
        a["flops"] += meta.get("flops", 0)
        a["bytes"] += meta.get("bytes", 0)
    for a in agg.values():  # per step (integer work counts stay integers)
        a["launches"] = max(1, round(a["launches"] / NREP))
        a["ms"] /= NREP
        a["flops"] //= NREP
        a["bytes"] //= NREP
    pmc, pmc_stale = {}, False
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            stored = json.load(f)
        pmc = stored.get("kernels", {})
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import pmc_summary  # the stored counters name the kernel sources they were collected from
        pmc_stale = stored.get("csrc_sha16") != pmc_summary.csrc_sha16()
    except (OSError, ValueError, ImportError):
        pass

    def pmc_of(name):
        """Counters of a kernel named by its launcher's template arguments: rocprofv3 prints trailing DEFAULTED template
        parameters too (`conv_tile32_kernel<false, 1, 16, false, true>` is `<..., true, false>` there)."""
        if name in pmc:
            return pmc[name]
        if name.endswith(">"):
            stem = name[:-1] + ", "
            cand = [k for k in pmc if k.startswith(stem)]
            if len(cand) == 1:
                return pmc[cand[0]]
            for k in cand:  # several instantiations share the stem: the one whose extra parameters are all defaults of the "off" kind
                if all(t.strip() in ("false", "0") for t in k[len(stem):-1].split(",")):
                    return pmc[k]
        return None

    def entry(name, a):
        sec = a["ms"] * 1e-3
        tf, gb = a["flops"] / sec / 1e12, a["bytes"] / sec / 1e9
        # bound: the roof this kernel would hit first at its algorithmic intensity
        mfma = a["flops"] > 0 and (a["flops"] / PEAK_F32_MFMA_TFLOPS / 1e12) > (a["bytes"] / PEAK_HBM_GBPS / 1e9)
        e = {"bound": "mfma" if mfma else "hbm", "kernel": name, "launches": a["launches"],
             "avg_us": round(a["ms"] / a["launches"] * 1e3, 2),
             "achieved": round(tf, 2) if mfma else round(gb, 1), "peak": PEAK_F32_MFMA_TFLOPS if mfma else PEAK_HBM_GBPS,
             "unit": "TFLOP/s" if mfma else "GB/s"}
        e["frac"] = round(e["achieved"] / e["peak"], 4)
        if mfma:
            # Winograd kernels execute 1/2 (F(4,3)) or 2/3 (F(2,3)) of the direct form's flops.  `achieved` / `frac` are what the
            # matrix pipes EXECUTE (round 6: a direct-form figure reads above 1.0 of the peak for an F(4,3) kernel and helps nobody);
            # the direct-form rate -- the work the reference's conv does per second -- stays next to it as `direct_form_*`.
            wf = winograd_fraction(name)
            e["executed_share_of_direct_form_flops"] = round(wf, 4)
            e["direct_form_achieved"] = e["achieved"]
            e["direct_form_frac"] = e["frac"]
            e["achieved"] = round(tf * wf, 2)
            e["frac"] = round(e["achieved"] / e["peak"], 4)
            e["frac_executed"] = e["frac"]
        t = pmc_of(name.split(" | ")[0])
        e["traffic"] = t.get("hbm_bytes_per_launch") if t else None
        e["traffic_source"] = ("stored rocprofv3 PMC pass of this command (profiles/pmc_traffic.json: separate FETCH_SIZE / "
                               "WRITE_SIZE runs, FETCH_SIZE doubled per the gfx950 correction)"
                               + ("; STALE: collected from kernel sources that differ from this tree's" if pmc_stale else "")) if t else None
        if t and "mfma_busy" in t:  # SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES-derived, same stored pass family
            e["mfma_busy"] = t["mfma_busy"]
        e["algorithmic_bytes_per_launch"] = int(a["bytes"] // a["launches"])
        return e

    detail_path = os.environ.get("ADP_BENCH_DETAIL")
    if detail_path:  # per (kernel, shape) table for kernel work; not part of the bench line
        det = {}
        for call, kern, meta, ms in recs:
            a = det.setdefault(kern + " :: " + meta.get("shape", ""), [0, 0.0, 0, 0])
            a[0] += 1
            a[1] += ms
            a[2] += meta.get("flops", 0)
            a[3] += meta.get("bytes", 0)
        with open(detail_path, "w") as f:
            for k, (n, ms, fl, by) in sorted(det.items(), key=lambda kv: -kv[1][1]):
                n, ms, fl, by = max(1, n // NREP), ms / NREP, fl / NREP, by / NREP  # per step
                f.write(f"{ms:8.3f} ms  n={n:3d}  avg {ms / n * 1e3:7.1f} us  {fl / ms / 1e9:6.1f} TF  {by / ms / 1e6:7.1f} GB/s  {k}\n")
    total_ms = sum(a["ms"] for a in agg.values())
    order = sorted(agg.items(), key=lambda kv: -kv[1]["ms"])
    rated = [(k, a) for k, a in order if (a["flops"] or a["bytes"])]
    rf = entry(*rated[0])
    rf["share_of_step"] = round(rated[0][1]["ms"] / total_ms, 4)
    # north_star's HBM target: the ConvBlock convs of depths 0-1 (GN+SiLU prologue, k=3, C = 8 / 32)
    hb = {"launches": 0, "ms": 0.0, "flops": 0, "bytes": 0}
    names = []
    for call, kern, meta, ms in recs:
        sh = meta.get("shape", "")
        if call == "adp_conv1d" and " pro1" in sh and " tr0" in sh and (" R8 " in sh or " R32 " in sh):
            hb["launches"] += 1
            hb["ms"] += ms
            hb["bytes"] += meta["bytes"]
            if kern not in names:
                names.append(kern)
    hbm = None
    if hb["launches"]:
        hb = {"launches": max(1, hb["launches"] // NREP), "ms": hb["ms"] / NREP, "flops": 0, "bytes": hb["bytes"] // NREP}
        hbm = entry(" | ".join(sorted(names)), hb)
        # HBM-side bytes per launch of each kernel in the group, from the stored PMC passes (the depth-0 and depth-1
        # launches move the same algorithmic bytes; NB the counters include Infinity-Cache hits)
        per = {n: pmc_of(n)["hbm_bytes_per_launch"] for n in sorted(names) if pmc_of(n) is not None}
        hbm["traffic"] = per or None
        hbm["traffic_source"] = "stored rocprofv3 PMC pass (profiles/pmc_traffic.json), per kernel instantiation" if per else None
        hbm["what"] = "forward ConvBlock convs (GroupNorm+SiLU prologue, k=3) of depths 0-1: A_in + A_out (+A_res) bytes"
        hbm["timing"] = ("HIP event pair around every launch of three instrumented eager steps (the pair itself adds 2-3 us to a "
                         "20 us kernel); replay_* = the same launches re-timed one by one from a hipGraph of 20 replays each")
        try:  # the same launches without the per-launch event pair
            rp = _replay_convblock(model, x)
            if rp:
                hbm.update(rp)
        except Exception as e:
            hbm["replay_error"] = f"{type(e).__name__}: {e}"
    table = {}
    for k, a in order[:top]:
        e = entry(k, a)
        table[k] = {"launches": e["launches"], "avg_us": e["avg_us"], "total_ms": round(a["ms"], 3),
                    "bound": e["bound"], "achieved": e["achieved"], "unit": e["unit"], "frac": e["frac"]}
        if "frac_executed" in e:
            table[k]["frac_executed"] = e["frac_executed"]
    return rf, hbm, table, round(total_ms, 3)


def _replay_convblock(model, x):
    """Re-times the depth-0/1 forward ConvBlock conv launches of one step: each launch (same tensors, inputs as warm in the
    Infinity Cache as behind their producer in the step) replayed 20 times back to back from its own hipGraph, HIP events
    around the replays.  Returns replay_avg_us / replay_achieved (GB/s) / replay_frac for the group."""
    from audio_diffusion_pytorch_amd import _C
    _C.REPLAY = []
    try:
        with torch.no_grad():
            model.net(x, torch.full((x.shape[0],), 0.5, device=x.device))
        torch.cuda.synchronize()
        recs = [r for r in _C.REPLAY if " pro1" in r[0] and " tr0" in r[0] and (" R8 " in r[0] or " R32 " in r[0])]
    finally:
        _C.REPLAY = None
    if not recs:
        return None
    tot_us, tot_bytes = 0.0, 0
    for label, nbytes, fn in recs:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                fn()
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        tot_us += a.elapsed_time(b) * 1e3 / 60
        tot_bytes += nbytes
    gbps = tot_bytes / (tot_us * 1e-6) / 1e9
    return {"replay_launches": len(recs), "replay_avg_us": round(tot_us / len(recs), 2), "replay_achieved": round(gbps, 1),
            "replay_frac": round(gbps / PEAK_HBM_GBPS, 4)}


def _our_pci_address():
    """PCI address ('0000:c5:00.0') of the GPU this process computes on, or None: a box's sysfs lists EVERY GPU of the host, also
    the ones this container cannot see -- reading the first card is how round 5 reported an idle neighbour's 116-161 MHz."""
    try:
        p = torch.cuda.get_device_properties(torch.cuda.current_device())
        return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except Exception:
        return None


def _read_clocks(pci=None):
    """sclk / mclk / power as the driver exposes them right now (sysfs first, rocm-smi as a fallback); best effort.  `pci`: the
    PCI address of the card to read (else every card is read and the one with the highest sclk is reported: under load that
    is the busy one)."""
    import glob
    import re
    cards = []
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        info = {}
        try:
            addr = os.path.basename(os.path.realpath(card))
            if pci is not None and addr.lower() != pci.lower():
                continue
            for key, fn in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk"), ("fclk_mhz", "pp_dpm_fclk")):
                path = os.path.join(card, fn)
                if os.path.exists(path):
                    levels = open(path).read().strip().splitlines()
                    act = [l for l in levels if l.rstrip().endswith("*")]
                    mhz = re.findall(r"(\d+)\s*[Mm][Hh]z", (act or levels[-1:])[0]) if levels else []
                    top = re.findall(r"(\d+)\s*[Mm][Hh]z", levels[-1]) if levels else []
                    if mhz:
                        info[key] = int(mhz[0])
                    if top:
                        info[key.replace("_mhz", "_max_mhz")] = int(top[0])
            for hw in glob.glob(os.path.join(card, "hwmon", "hwmon*")):
                for key, fn in (("power_cap_w", "power1_cap"), ("power_w", "power1_average"), ("power_w", "power1_input")):
                    path = os.path.join(hw, fn)
                    if os.path.exists(path) and key not in info:
                        info[key] = round(int(open(path).read().strip()) / 1e6, 1)
            busy = os.path.join(card, "gpu_busy_percent")
            if os.path.exists(busy):
                info["gpu_busy_percent"] = int(open(busy).read().strip())
            if info:
                info["source"] = card
                info["pci"] = addr
                info["matched_by"] = "pci address of the benchmarked device" if pci is not None else "highest sclk of all cards"
                cards.append(info)
        except (OSError, ValueError, IndexError):
            continue
    if cards:
        return max(cards, key=lambda c: (c.get("sclk_mhz", 0), c.get("power_w", 0)))
    info = {}
    if not info:
        try:
            import subprocess
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showmaxpower", "--showuse", "--json"],
                                 capture_output=True, text=True, timeout=15).stdout
            data = json.loads(out[out.index("{"):])
            best = None
            for name in sorted(data):
                card = {k: v for k, v in data[name].items() if any(t in k.lower() for t in ("sclk", "mclk", "fclk", "power", "use"))}
                sc = [float(x) for k, v in card.items() if "sclk" in k.lower() for x in re.findall(r"(\d+(?:\.\d+)?)", str(v))[:1]]
                card["source"] = f"rocm-smi {name}"
                if best is None or (sc and sc[0] > best[0]):
                    best = (sc[0] if sc else 0.0, card)
            info = best[1] if best else {"error": "rocm-smi listed no card"}
        except Exception as e:
            info = {"error": f"no clock source readable ({type(e).__name__})"}
    return info


def _clocks_under_load(busy, pci, idle):
    """Clocks WHILE the GPU is demonstrably busy: `busy()` queues ~1 s of asynchronous work, then the card is polled (every 50 ms,
    for as long as the work runs) and the reading with the highest sclk is kept; `polls` / `sclk_seen_mhz` say what was seen."""
    busy()
    done = torch.cuda.Event()
    done.record()
    best, seen = None, []
    t0 = time.perf_counter()
    while True:
        r = _read_clocks(pci)
        seen.append(r.get("sclk_mhz"))
        if best is None or (r.get("sclk_mhz") or 0) > (best.get("sclk_mhz") or 0):
            best = r
        if done.query() or time.perf_counter() - t0 > 5.0:
            break
        time.sleep(0.05)
    torch.cuda.synchronize()
    best = dict(best or {})
    best["polls"] = len(seen)
    best["sclk_seen_mhz"] = sorted({v for v in seen if v is not None})
    best["window_s"] = round(time.perf_counter() - t0, 3)
    if idle and best.get("sclk_mhz") is not None and best.get("sclk_mhz") == idle.get("sclk_mhz"):
        best["note"] = "no poll read a clock above the idle reading: this driver's sysfs does not follow the load on this box"
    return best


def _mixed_kernel_probe(dev, ev_ms):
    """us per launch of a hipGraph that cycles through eight different single-kernel library calls on tiny tensors."""
    from audio_diffusion_pytorch_amd import ops
    B, C, L, G = 1, 64, 256, 8
    x, y, z = (torch.randn(B, C, L, device=dev) for _ in range(3))
    st = ops.gn_stats(x, G)
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    ss = torch.randn(B, 2 * C, device=dev) * 0.1
    x8, w8 = torch.randn(1, 8, 4096, device=dev), torch.randn(8, 8, 3, device=dev) * 0.1
    o8 = torch.empty_like(x8)
    ab = torch.tensor([0.3, 0.9, 0.5, 0.8], device=dev)
    f2 = torch.randn(4, 512, device=dev)
    lst, mst = torch.empty(B, L, 2, device=dev), torch.empty(B, L, 2, device=dev)

    def round_():
        ops.add(x, y, out=z)
        ops.axpby(0.5, z, out=y)
        ops.act_fwd(f2, 2)
        ops.gn_act(x, st, G, gam, bet)
        ops.modulation_fwd(x, ss.view(-1), 2 * C, y=z, stats=mst)
        ops.ln_affine_fwd(z, gam, bet)
        ops.conv1d(x8, w8, None, pad=1, out=o8)
        ops.v_step(x, y, ab, out=z)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        round_()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(12):
            round_()
    return round(ev_ms(g.replay, 10) * 1e3 / (12 * 8), 3)


def calibration(dev, busy=None):
    """What THIS box does on three kernels whose ideal rates are known (csrc/probe.hip), measured in the benchmarked
    process before the timed window -- so that two bench lines of the same code on two boxes can be told apart from a
    regression: 16-byte copy GB/s at 256 MB (HBM), register-only exact-f32 MFMA TFLOP/s (matrix pipes x clock), the
    dependent-launch gap inside a hipGraph (100 empty kernels), and the clocks / power cap the driver reports idle and
    (`busy`: a callable that queues ~1 s of asynchronous work) under load."""
    from audio_diffusion_pytorch_amd import _C
    from audio_diffusion_pytorch_amd._C import ptr
    out = {}

    def ev_ms(fn, reps, warm_s=0.0):
        fn()
        torch.cuda.synchronize()
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < warm_s:  # (a probe that starts on an idle chip reads its clock ramp: MFMA 138-144 vs 154 TF)
            fn()
            torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    try:
        n = 64 << 20
        src, dst = torch.randn(n, device=dev), torch.empty(n, device=dev)
        ms = ev_ms(lambda: _C.call("adp_probe_copy", ptr(src), ptr(dst), n, _C.stream()), 10, warm_s=0.2)
        out["copy_256MB_gbps"] = round(8 * n / ms / 1e6, 1)
        del src, dst
        # Ceiling search (round 6): 1 GiB source + 1 GiB destination -- four times the Infinity Cache, so nothing of a pass is
        # served from it -- over the copy variants of csrc/probe.hip (persistent grids, 1-8 16-byte loads in flight per lane,
        # nontemporal access).  `copy_1GiB_best_gbps` is what "a plain copy reaches on this box"; the guide's 6290 GB/s
        # (MI355X_MICROARCH.md) is the reference the roofline_hbm_convblock object is read against.
        try:
            n1 = 256 << 20
            src, dst = torch.empty(n1, device=dev).normal_(), torch.empty(n1, device=dev)
            sweep = {}
            for v in range(8):
                ms = ev_ms(lambda: _C.call("adp_probe_copy_v", ptr(src), ptr(dst), n1, v, _C.stream()), 5)
                sweep[str(v)] = round(8 * n1 / ms / 1e6, 1)
            best = max(sweep, key=sweep.get)
            out["copy_1GiB_gbps_by_variant"] = sweep
            out["copy_1GiB_best_gbps"] = sweep[best]
            out["copy_1GiB_best_variant"] = {"0": "grid 4096 grid-stride, 1 load in flight", "1": "2048 x 4 loads", "2": "2048 x 4 loads, nt",
                                             "3": "2048 x 8 loads, nt", "4": "4096 x 2 loads, nt", "5": "1024 x 8 loads",
                                             "6": "1024 x 4 loads, nt", "7": "8192 x 1 load, nt"}[best]
            # one direction at a time: what the HBM delivers when it does not turn around between reads and writes
            ms = ev_ms(lambda: _C.call("adp_probe_copy_v", ptr(src), ptr(dst), n1, 8, _C.stream()), 5)
            out["read_1GiB_gbps"] = round(4 * n1 / ms / 1e6, 1)
            ms = ev_ms(lambda: _C.call("adp_probe_copy_v", ptr(src), ptr(dst), n1, 9, _C.stream()), 5)
            out["write_1GiB_gbps"] = round(4 * n1 / ms / 1e6, 1)
            del src, dst
        except Exception as e:
            out["copy_1GiB_error"] = f"{type(e).__name__}: {e}"
        buf = torch.empty(1024 * 256, device=dev)
        flops = [0]

        def mfma():
            flops[0] = _C.call_value("adp_probe_mfma", 8000, ptr(buf), buf.numel(), _C.stream())
        ms = ev_ms(mfma, 5, warm_s=0.3)
        out["mfma_f32_probe_tflops"] = round(flops[0] / ms / 1e9, 1)
        out["mfma_f32_probe_frac_of_peak"] = round(flops[0] / ms / 1e9 / PEAK_F32_MFMA_TFLOPS, 4)
        try:  # the same loop with 16 / 32 MFMAs per trip and 1 / 2 / 4 waves per SIMD (the guide measures 155 TF)
            sweep = {}
            for v in range(5):
                def mf(v=v):
                    flops[0] = _C.call_value("adp_probe_mfma_v", 8000, ptr(buf), buf.numel(), v, _C.stream())
                ms = ev_ms(mf, 5)
                sweep[str(v)] = round(flops[0] / ms / 1e9, 1)
            out["mfma_f32_probe_tflops_by_variant"] = sweep
            out["mfma_f32_probe_best_tflops"] = max(sweep.values())
        except Exception as e:
            out["mfma_f32_probe_sweep_error"] = f"{type(e).__name__}: {e}"
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            _C.call("adp_probe_launch", 1, _C.stream())
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(100):
                _C.call("adp_probe_launch", 1, _C.stream())
        out["graph_launch_gap_us"] = round(ev_ms(g.replay, 10) * 10.0, 3)  # ms per 100 launches -> us per launch
        del g
        # small dependent kernels (what two thirds of a step's launches are): 4 MB and 16 MB float4 copies, a -> b -> a ..., 50
        # per hipGraph: launch gap + first-load latency + drain per kernel, the working set resident in the Infinity Cache
        small = {}
        for mb in (4, 16):
            nn = mb << 18
            pa, pb = torch.randn(nn, device=dev), torch.empty(nn, device=dev)
            _C.call("adp_probe_copy", ptr(pa), ptr(pb), nn, _C.stream())
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(50):
                    s_, d_ = (pa, pb) if i % 2 == 0 else (pb, pa)
                    _C.call("adp_probe_copy", ptr(s_), ptr(d_), nn, _C.stream())
            small[f"{mb}MB"] = round(ev_ms(g.replay, 5) * 1e3 / 50, 3)
            del g, pa, pb
        out["small_copy_us_per_launch"] = small
        # DIFFERENT small kernels back to back (the step's ~470 small launches are ~40 different kernels, not one): eight
        # single-launch library calls on tiny tensors, 12 rounds per hipGraph.  Boxes whose same-kernel probes above agree to the
        # percent differ here (and in the step: 11.5 vs 12.6 ms with equal event-timed kernel durations) -- what a slow box adds
        # is per launch of a kernel that is not the previous one
        try:
            out["graph_mixed_kernels_us_per_launch"] = _mixed_kernel_probe(dev, ev_ms)
        except Exception as e:
            out["graph_mixed_kernels_us_per_launch"] = {"error": f"{type(e).__name__}: {e}"}
        # load-to-use latency: one lane chasing a random cycle of 64-byte lines over working sets that sit in the L2 (1 MB),
        # the Infinity Cache (64 MB) and HBM (2 GB)
        lat = {}
        for label, mbytes in (("1MB", 1), ("64MB", 64), ("2GB", 2048)):
            lines = mbytes << 14  # 64-byte lines
            perm = torch.randperm(lines, device=dev, dtype=torch.int32)
            chain = torch.zeros(lines * 16, dtype=torch.int32, device=dev)
            chain[perm.long() * 16] = torch.roll(perm, -1) * 16
            # (line 0 is on the cycle like every line: the walk starts there)
            o = torch.zeros(1, dtype=torch.int32, device=dev)
            steps = 40000  # (the first few thousand hops of a freshly written chain still hit the caches)
            fn = lambda: _C.call("adp_probe_chase", ptr(chain, torch.int32), steps, ptr(o, torch.int32), _C.stream())  # noqa: E731
            lat[label] = round(ev_ms(fn, 3) * 1e6 / steps, 1)
            del perm, chain
        out["load_latency_ns"] = lat
    except Exception as e:
        out["error"] = f"{type(e).__name__}: {e}"
    torch.cuda.synchronize()
    pci = _our_pci_address()
    if pci is not None and "error" in _read_clocks(pci):  # (no card with that address in this container's sysfs)
        pci = None
    out["clocks_idle"] = _read_clocks(pci)
    if busy is not None:
        try:
            out["clocks_under_load"] = _clocks_under_load(busy, pci, out["clocks_idle"])
        except Exception as e:
            out["clocks_under_load"] = {"error": f"{type(e).__name__}: {e}"}
    out["what"] = ("probes of csrc/probe.hip timed with HIP events in this process before the timed window: 256 MB float4 copy "
                   "(read + write bytes / time; 6290 GB/s is the best copy this pool has shown), register-only v_mfma_f32_32x32x2 "
                   f"loop against the {PEAK_F32_MFMA_TFLOPS} TF peak, hipGraph of 100 dependent empty kernels, hipGraphs of 50 dependent small "
                   "copies (per launch), hipGraph of 96 launches cycling through eight different small library kernels (per launch), "
                   "one lane chasing a random cycle of cache lines (ns per dependent load)")
    return out


def _dp1_worker() -> None:
    """Child process of the `dp1` leg: a ONE-rank RCCL group in a process of its own (so that nothing RCCL does can take the
    bench line down), the data-parallel wrapper with force_collectives=True -- bucketed AVG all-reduces from inside
    backward on RCCL's stream, exactly the code path of N > 1 -- against the same eager step without the wrapper."""
    import socket
    import torch.distributed as dist
    import audio_diffusion_pytorch_amd as adp
    from audio_diffusion_pytorch_amd import parallel
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    parallel.graph_safe_rccl_env()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    res = {}

    def leg(model, batch, kw, reps):
        x = torch.randn(batch, 2, LENGTH, device=dev)

        def zero():
            for p in model.parameters():
                p.grad = None

        def plain():
            zero()
            model(x, **kw).backward()
        t_plain = _time(plain, reps, warmup=3)
        dp = parallel.DataParallel(model, force_collectives=True)
        sent, orig = [], dp._send
        dp._send = lambda flat, a, b: (sent.append((a, b)), orig(flat, a, b))[1]

        def step():
            zero()
            dp(x, **kw).backward()
        step()
        dp._send = orig
        buckets = [round((b - a) * 4 / 2 ** 20, 1) for a, b in sent]
        t_dp = _time(step, reps, warmup=3)
        # the same two steps replayed from hipGraphs: the wrapped one holds its RCCL all-reduces (what bench.py times at N > 1),
        # and the bucket sequence alone, also replayed.  exposed_ms = wrapped - plain; against allreduce_alone_ms it says how
        # much of the collectives' time the backward hides (0 = all of it, 1 = none).  (The eager readings of this leg are host
        # launch time on a host-bound step; no overlap figure is derived from them any more.)
        graph_err = None
        try:
            t_dp_graph = _time(_graphed(step, zero, mode="thread_local"), 20)
        except Exception as e:
            t_dp_graph = None
            graph_err = f"{type(e).__name__}: {e}"
            torch.cuda.synchronize()
        t_comm_graph = None
        try:
            flat = torch.zeros(max(b for _, b in sent), dtype=torch.float32, device=dev)

            def comm_only():
                for a, b in sent:
                    dp._send(flat, a, b)
                dp._wait_all()
            t_comm_graph = _time(_graphed(comm_only, lambda: None, mode="thread_local"), 20)
            del flat
        except Exception as e:
            graph_err = (graph_err or "") + f" comm_only: {type(e).__name__}: {e}"
            torch.cuda.synchronize()
        dp.unet._grad_ready_hook = None
        t_plain2 = _time(plain, reps, warmup=2)  # (again after the wrapped steps: same clocks / allocator state)
        t_ref = min(t_plain, t_plain2)
        t_plain_graph = _time(_graphed(plain, zero, mode="thread_local"), 20)
        graphs = {"ms_per_step_without_wrapper": round(t_plain_graph * 1e3, 3)}
        if t_dp_graph is not None:
            graphs.update({"ms_per_step": round(t_dp_graph * 1e3, 3), "dp_over_plain": round(t_dp_graph / t_plain_graph, 4),
                           "exposed_ms": round((t_dp_graph - t_plain_graph) * 1e3, 3)})
        if t_comm_graph is not None:
            graphs["allreduce_alone_ms"] = round(t_comm_graph * 1e3, 3)
            if t_dp_graph is not None and t_comm_graph > 0:
                graphs["exposed_over_alone"] = round((t_dp_graph - t_plain_graph) / t_comm_graph, 4)
        if graph_err:
            graphs["error"] = graph_err
        return {"hipgraph_replay": graphs,
                "ms_per_step": round(t_dp * 1e3, 3), "ms_per_step_without_wrapper": round(t_ref * 1e3, 3),
                "dp_over_plain": round(t_dp / t_ref, 4), "buckets_mb": buckets, "launch": "eager (both; hipgraph_replay: replayed)",
                "ctx_bank_runs_under_hook": int(getattr(dp.unet, "_ctx_bank_hooked_backwards", 0))}

    try:
        res["headline"] = leg(build_model(dev), 4, {}, 10)
        res["headline"]["workload"] = "BASELINE configs[1]: batch 4, [4,2,2**18]"
    except Exception as e:
        res["headline"] = {"error": f"{type(e).__name__}: {e}"}
    try:
        torch.manual_seed(0)
        m4 = adp.DiffusionModel(net_t=adp.UNetV0, in_channels=2, channels=CHANNELS, factors=FACTORS, items=ITEMS,
                                cross_attentions=[0, 0, 0, 1, 1, 1, 1, 1, 1], embedding_features=768, attention_heads=8,
                                attention_features=64).to(dev)
        res["config4"] = leg(m4, 1, dict(embedding=torch.randn(1, 64, 768, device=dev)), 10)
        res["config4"]["workload"] = ("BASELINE configs[3] layout (cross attention at depths 3-8, embedding [1,64,768]) at the "
                                      "per-GPU batch of batch 8 over 8 GPUs: [1,2,2**18]")
    except Exception as e:
        res["config4"] = {"error": f"{type(e).__name__}: {e}"}
    torch.cuda.synchronize()
    print("DP1_WORKER " + json.dumps(res), flush=True)
    dist.destroy_process_group()


def dp1_leg(timeout: float = 240.0):
    """The multi-GPU code path on the one GPU a box has: see _dp1_worker.  backend nccl == RCCL."""
    import subprocess
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--dp1-worker"], capture_output=True, text=True,
                             timeout=timeout)
        for line in out.stdout.splitlines():
            if line.startswith("DP1_WORKER "):
                res = json.loads(line[len("DP1_WORKER "):])
                res["what"] = ("parallel.DataParallel(force_collectives=True) on a one-rank RCCL group in a child process: "
                               "the bucketed ReduceOp.AVG all-reduces issued from inside backward as at N > 1, against the same step "
                               "without the wrapper -- eager launches (mean of 10 steps each; host-bound, varies with the box's "
                               "host load) and, `hipgraph_replay`, both replayed from hipGraphs (the wrapped graph holds its "
                               "collectives; this is how bench.py runs at N > 1)")
                return res
        return {"error": "dp1 worker printed no result", "rc": out.returncode, "stderr_tail": out.stderr[-600:]}
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"}


def eager_api_leg(model, x, replay_ms, batch1_replay_ms):
    """The reference's README training loop VERBATIM (/root/reference/README.md:36-39: `loss = model(audio); loss.backward()`,
    plus the `p.grad = None` every optimizer's zero_grad() does) through the drop-in API: no capture code on the caller's side.
    VDiffusion.forward serves it from two hipGraphs (graphed.py); `launched_eagerly_ms` = the same loop with
    ADP_TRAIN_GRAPH=0 (~700 launches issued from Python per step) for comparison."""
    out = {}
    prev = os.environ.get("ADP_TRAIN_GRAPH")
    params = list(model.parameters())

    def loop_step(xx):
        for p in params:
            p.grad = None
        loss = model(xx)
        loss.backward()
    try:
        for name, xx, ref in (("batch4", x, replay_ms), ("batch1", x[:1].contiguous(), batch1_replay_ms)):
            os.environ["ADP_TRAIN_GRAPH"] = "1"
            dt = _time(lambda: loop_step(xx), 20, warmup=3, warm_s=LEG_WARM_S)
            os.environ["ADP_TRAIN_GRAPH"] = "0"
            de = _time(lambda: loop_step(xx), 5, warmup=2)
            e = {"ms_per_step": round(dt * 1e3, 3), "launched_eagerly_ms": round(de * 1e3, 3)}
            if ref:
                e["whole_step_replay_ms"] = round(ref, 3)
                e["over_whole_step_replay"] = round(dt * 1e3 / ref, 4)
            out[name] = e
        from audio_diffusion_pytorch_amd import graphed
        g = graphed.GRAPHS_OF.get(model.diffusion)
        out["captures"] = None if g is None else g.captures
        out["what"] = ("README loop verbatim (zero grads; loss = model(x); loss.backward()) timed from the host over 20 steps; "
                       "whole_step_replay_ms = the same step captured by bench.py as ONE hipGraph (the headline's launch mode)")
        if g is not None:
            g.cache.clear()  # (the entries own their activation pools)
    except Exception as e:
        out["error"] = f"{type(e).__name__}: {e}"
    finally:
        if prev is None:
            os.environ.pop("ADP_TRAIN_GRAPH", None)
        else:
            os.environ["ADP_TRAIN_GRAPH"] = prev
    torch.cuda.empty_cache()
    return out


LEG_WARM_S = 0.3  # untimed replay in front of every leg outside the headline (which has the calibration's 1 s)


def _graphed(step, zero, mode="global"):
    """Captures `step` in a hipGraph (after two eager warm-ups on a side stream); returns the replay callable.
    mode = "thread_local" for steps that hold RCCL collectives (see parallel.capture_step)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if mode == "thread_local":
        from audio_diffusion_pytorch_amd import parallel
        parallel.quiesce_watchdog()  # (the watchdog retires the warm-up collectives before the capture opens)
    zero()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode=mode):
        step()
    return graph.replay


def _time(fn, n, warmup=3, warm_s=0.0):
    """Seconds per call over n calls after `warmup` untimed ones; warm_s > 0 keeps calling for at least that long first (the
    legs outside the headline start on a GPU that idled through a model build: clocks come back within a few hundred ms)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t_w = time.perf_counter()
    while warm_s > 0.0 and time.perf_counter() - t_w < warm_s:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def _attn_summary(recs):
    """Attention kernels of one instrumented step: per kernel launches + average time; per C-ABI call (the backward
    is four kernels: delta, key/value pass, its reduce, query pass) the TFLOP/s of its algorithmic flops against the
    f32 MFMA peak."""
    kern, calls = {}, {}
    for call, k, meta, ms in recs:
        if not call.startswith("adp_attn"):
            continue
        a = kern.setdefault(k.split("(")[0], {"launches": 0, "ms": 0.0})
        a["launches"] += 1
        a["ms"] += ms
        c = calls.setdefault(call, {"calls": 0, "ms": 0.0, "flops": 0})
        c["ms"] += ms
        if meta.get("flops"):  # the tag rides on the first kernel of a call
            c["calls"] += 1
            c["flops"] += meta["flops"]
    res = {k: {"launches": a["launches"], "avg_us": round(a["ms"] / a["launches"] * 1e3, 2)} for k, a in kern.items()}
    for call, c in calls.items():
        if c["calls"] and c["ms"] > 0:
            tf = c["flops"] / (c["ms"] * 1e-3) / 1e12
            res[call] = {"calls": c["calls"], "avg_us_per_call": round(c["ms"] / c["calls"] * 1e3, 2), "tflops": round(tf, 2),
                         "frac_of_f32_mfma_peak": round(tf / PEAK_F32_MFMA_TFLOPS, 4)}
    return res


def extra_legs(model, x, dev):
    """Numbers SURVEY 8d asks for next to the headline, OUTSIDE its timed region (same random-init weights):
      batch1            config 1's shape on the GPU: fwd+bwd at [1,2,2**18], hipGraph-replayed;
      sampler           config 3: VSampler 50 steps on noise [1,2,2**18], hipGraph-captured step (sampler steps/s);
      readme_attention  config 1 + README attentions=[0,0,0,0,0,1,1,1,1] (8 heads x 64), fwd+bwd at batch 1, with
                        the attention kernels' TFLOP/s from an instrumented step;
      config4           text-conditional layout (cross attention at depths 3-8 over embedding [B,64,768]), batch 1."""
    import audio_diffusion_pytorch_amd as adp
    out = {}

    def zero(m):
        for p in m.parameters():
            p.grad = None

    x1 = x[:1].contiguous()
    try:
        def step1():
            zero(model)
            model(x1).backward()
        dt = _time(_graphed(step1, lambda: zero(model)), 20, warm_s=LEG_WARM_S)
        out["batch1"] = {"workload": "BASELINE configs[0] shape on the GPU: fwd+bwd, audio [1,2,2**18]",
                         "steps_per_s": round(1.0 / dt, 2), "ms_per_step": round(dt * 1e3, 3), "launch": "hipGraph replay"}
    except Exception as e:
        out["batch1"] = {"error": f"{type(e).__name__}: {e}"}
    try:
        noise = torch.randn(1, 2, LENGTH, device=dev)
        model.sample(noise, num_steps=2)  # captures the step
        dt = _time(lambda: model.sample(noise, num_steps=50), 2, warmup=1, warm_s=LEG_WARM_S)
        out["sampler"] = {"workload": "BASELINE configs[2]: VSampler.sample num_steps=50, noise [1,2,2**18], "
                                      "hipGraph-captured step", "sampler_steps_per_s": round(50.0 / dt, 2),
                          "ms_per_step": round(dt / 50 * 1e3, 3), "ms_per_50_step_sample": round(dt * 1e3, 2)}
    except Exception as e:
        out["sampler"] = {"error": f"{type(e).__name__}: {e}"}
    # kernel-family A/B on the headline step: the kernel-3 convs and weight gradients run in the Winograd domain (F(2,3) in conv_mm / wgrad_mm, F(4,3) in conv_tile32)
    # on the exact-f32 matrix cores by default (conv_mm / wgrad_mm WN variants: two thirds of the MFMAs, plain fp32);
    #   direct_form_convs   ADP_CONV_WINO=0: the same kernels in the direct form (the round-1 / round-2 arithmetic)
    for key, env, val, what in (("direct_form_convs", "ADP_CONV_WINO", "0", "in the direct form on the f32 matrix cores"),):
        prev = os.environ.get(env)
        try:
            os.environ[env] = val

            def stepb():
                zero(model)
                model(x).backward()
            dt = _time(_graphed(stepb, lambda: zero(model)), 20, warm_s=LEG_WARM_S)
            out[key] = {"workload": f"headline step ([{x.shape[0]},2,2**18] fwd+bwd) with {env}={val}: kernel-3 convs and "
                                    f"weight gradients {what}",
                        "steps_per_s": round(1.0 / dt, 2), "ms_per_step": round(dt * 1e3, 3)}
        except Exception as e:
            out[key] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            if prev is None:
                os.environ.pop(env, None)
            else:
                os.environ[env] = prev
    legs = {
        "readme_attention": (dict(attentions=[0, 0, 0, 0, 0, 1, 1, 1, 1], attention_heads=8, attention_features=64), False),
        "config4": (dict(cross_attentions=[0, 0, 0, 1, 1, 1, 1, 1, 1], embedding_features=768, attention_heads=8,
                         attention_features=64), True),
    }
    for name, (extra, use_emb) in legs.items():
        try:
            torch.manual_seed(0)
            m = adp.DiffusionModel(net_t=adp.UNetV0, in_channels=2, channels=CHANNELS, factors=FACTORS, items=ITEMS,
                                   **extra).to(dev)
            kw = dict(embedding=torch.randn(1, 64, 768, device=dev)) if use_emb else {}

            def stepa():
                zero(m)
                m(x1, **kw).backward()
            dt = _time(_graphed(stepa, lambda: zero(m)), 10, warm_s=LEG_WARM_S)
            recs = profiled_step(m, lambda: m(x1, **kw).backward())
            out[name] = {"workload": f"UNetV0 README channels + {extra}, fwd+bwd at [1,2,2**18]"
                                     + (", embedding [1,64,768]" if use_emb else ""),
                         "steps_per_s": round(1.0 / dt, 2), "ms_per_step": round(dt * 1e3, 3),
                         "attention_kernels": _attn_summary(recs)}
            del m
            torch.cuda.empty_cache()
        except Exception as e:
            out[name] = {"error": f"{type(e).__name__}: {e}"}
    return out


def _rendezvous_only() -> int:
    """Test hook (no GPU): what a rank does before any device work -- parse the contract's flags, join the process group from
    the launcher's environment (gloo without a GPU), agree on the world size, rank 0 prints one JSON line."""
    import torch.distributed as dist
    from audio_diffusion_pytorch_amd import parallel
    ap = argparse.ArgumentParser()
    for flag in ("--gpus", "--steps", "--warmup"):
        ap.add_argument(flag, type=int, default=1)
    args, _ = ap.parse_known_args()
    rank = parallel.init_process_group_from_env(backend="gloo")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    t = torch.tensor([float(rank + 1)])
    if world > 1:
        dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"rendezvous": "ok", "n_gpus": world, "rank_sum": t.item(), "steps": args.steps}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def _self_launch(n: int, argv=None, run=None) -> int:
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N ranks -- one process per GPU under
    torch.distributed.run on 127.0.0.1 with a free port -- with the same arguments, pass their output through (rank 0 prints the
    one JSON line) and return their exit code.  Under torchrun (WORLD_SIZE set) this is never reached."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return (run or subprocess.call)(cmd, env=env)


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--cpu-worker":
        return _cpu_worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    if len(sys.argv) >= 2 and sys.argv[1] == "--dp1-worker":
        return _dp1_worker()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4, help="per-GPU batch (BASELINE configs[1]: 4)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="STRONG scaling: total batch split over the ranks (BASELINE configs 4/5: 8 over 8 GPUs); "
                         "0 = weak scaling with --batch per GPU (the default the driver runs)")
    ap.add_argument("--graph", type=int, default=-1, help="1: replay the step from a hipGraph, 0: eager; "
                                                          "default: graph on 1 GPU, eager with RCCL")
    ap.add_argument("--dp-overlap", action="store_true",
                    help="N > 1: also report dp_overlap = {step, step without all-reduce, all-reduce alone, hidden_frac}")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the batch-1 / sampler / attention legs")
    ap.add_argument("--no-calibration", action="store_true", help="skip the calibration probes and the 1 s pre-warm")
    ap.add_argument("--no-dp1", action="store_true", help="skip the one-rank RCCL data-parallel leg")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return _self_launch(args.gpus)
    if os.environ.get("ADP_BENCH_RENDEZVOUS_ONLY") == "1":  # tests/test_bench_contract.py: the launch path without a GPU
        return _rendezvous_only()
    # Every leg launches its kernels itself (instrumented eager steps) or captures the whole step explicitly; the package's own
    # graph-replayed README loop (graphed.py) is measured by the `eager_api` leg only, which switches it back on for itself.
    os.environ.setdefault("ADP_TRAIN_GRAPH", "0")

    from audio_diffusion_pytorch_amd import parallel
    rank = parallel.init_process_group_from_env(graph_safe=True)  # (the step is captured with its collectives below)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py measures the MI355X path; no GPU visible"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist

    scaling = "weak"
    if args.global_batch:
        assert args.global_batch % world == 0, "--global-batch must be divisible by the number of GPUs"
        args.batch = args.global_batch // world
        scaling = "strong"
    model = build_model(dev)
    if world > 1:
        model = parallel.DataParallel(model)
    torch.manual_seed(1234 + rank)
    x = torch.randn(args.batch, 2, LENGTH).to(dev)  # synthetic waveforms, resident in HBM before timing
    # hipGraph replay also with RCCL (round 5: the collectives issued from inside backward are captured with the step once
    # ProcessGroupNCCL's watchdog is kept away from the capture -- parallel.graph_safe_rccl_env); --graph 0 = eager launches
    use_graph = args.graph if args.graph >= 0 else 1

    def zero():
        for p in model.parameters():
            p.grad = None

    def eager_step():
        zero()
        loss = model(x)
        loss.backward()
        return loss

    graph = None
    if use_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    eager_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if world > 1:
                parallel.quiesce_watchdog()  # (the watchdog retires the warm-up collectives before the capture opens)
            zero()
            graph = torch.cuda.CUDAGraph()
            # (with RCCL: thread_local -- the watchdog thread may still poll the warm-up collectives' events: parallel.capture_step)
            with torch.cuda.graph(graph, capture_error_mode="thread_local" if world > 1 else "global"):
                static_loss = model(x)
                static_loss.backward()
            # (drop the captured step's autograd graph: it would keep the parameters' AccumulateGrad nodes -- created on the
            #  capture stream -- alive, and every later eager backward in this process would run its 600 gradient accumulations
            #  on that stream behind an event pair each)
            static_loss = static_loss.detach()
        except Exception as e:  # capture is a launch-overhead optimisation only; the kernels are identical
            if rank == 0:
                print(f"[bench] hipGraph capture unavailable ({type(e).__name__}: {e}); timing eager launches",
                      file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
        if world > 1:  # every rank replays, or every rank launches eagerly (a graph holds its collectives)
            ok = torch.tensor([1.0 if graph is not None else 0.0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if ok.item() < 1.0:
                graph = None

    step = graph.replay if graph is not None else eager_step
    launch_mode = "hipGraph replay" if graph is not None else "eager"
    if world > 1 and graph is None and use_graph:
        # collectives could not be captured on this stack: capture the kernels, issue the bucket all-reduces behind every replay
        # (parallel.capture_step_deferred: no overlap with backward, but no per-kernel host launches either)
        try:
            _, step_deferred = parallel.capture_step_deferred(eager_step, model)
            ok = torch.tensor([1.0], device=dev)
        except Exception as e:
            step_deferred = None
            ok = torch.tensor([0.0], device=dev)
            if rank == 0:
                print(f"[bench] deferred-collective capture unavailable ({type(e).__name__}: {e})", file=sys.stderr)
            torch.cuda.synchronize()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() >= 1.0:
            step, launch_mode = step_deferred, "hipGraph replay of the kernels + bucket all-reduces issued behind it"
        else:
            model._deferred = None

    calib = None
    if world == 1 and not args.no_calibration:
        # >= 1 s of the step itself before anything is timed (clocks and power management settle under THIS load); the
        # calibration probes run first, and the clocks are read while the pre-warm replays are queued
        def busy():
            for _ in range(80):
                step()
        try:
            calib = calibration(dev, busy)
        except Exception as e:
            calib = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    windows = [dt / args.steps * 1e3]
    if world == 1 and not args.no_calibration:  # two more windows of the same K steps: is the first one representative?
        for _ in range(2):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            windows.append((time.perf_counter() - t1) / args.steps * 1e3)
    ms = dt / args.steps * 1e3
    value = world * args.steps / dt  # denoising steps (U-Net fwd+bwd evaluations on a batch) per second, whole job
    line = {
        "metric": "denoising steps/s (UNetV0 fwd+bwd) at [B,2,2**18]", "value": round(value, 3),
        "unit": "denoising steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[1]: unconditional DiffusionModel(UNetV0 channels="
                               "[8,32,64,128,256,512,512,1024,1024], factors=[1,4,4,4,2,2,2,2,2], "
                               "items=[1,2,2,2,2,2,2,4,4]) fwd+bwd, audio=randn(4,2,2**18) per GPU, random-init weights",
                   "per_gpu_batch": args.batch, "global_batch": args.batch * world, "length": LENGTH,
                   "samples_per_s": round(value * args.batch, 2), "launch": launch_mode,
                   "parallelism": f"dp{world}" if world > 1 else "single",
                   "collective_backend": (dist.get_backend() if world > 1 else None),
                   "collective_world_size": (dist.get_world_size() if world > 1 else 1),
                   "collectives_in_graph": bool(world > 1 and graph is not None),
                   "optimizer": "none (the metric is fwd+bwd; gradients for all 176M parameters are produced)"},
    }
    if len(windows) > 1:
        line["ms_per_step_windows"] = [round(w, 3) for w in windows]
        line["ms_per_step_best_of_3"] = round(min(windows), 3)
    if calib is not None:
        line["calibration"] = calib
    if world > 1 and args.dp_overlap:
        # outside the timed region, every rank in lockstep: how much of the gradient all-reduce hides under backward
        # (opt-in: three more untimed phases with collectives -- kept out of the default multi-GPU run)
        try:
            def dp_step():
                zero()
                model(x).backward()
            ov = model.measure_overlap(dp_step, lambda fn, n: _time(fn, n, warmup=1), reps=5)
            if rank == 0:
                line["dp_overlap"] = ov
        except Exception as e:
            if rank == 0:
                line["dp_overlap"] = {"error": f"{type(e).__name__}: {e}"}
    # rank 0 only, outside the timed region, at EVERY N (a scaling line carries its roofline and CPU baseline too): the other ranks
    # wait in the closing barrier.  The instrumented steps run on the unwrapped module with the data-parallel hook detached
    # (one rank issuing collectives alone would hang the job).
    inner = model.module if world > 1 else model
    if rank == 0 and not args.no_roofline:
        hook_owner = model.unet if world > 1 else None
        hook = getattr(hook_owner, "_grad_ready_hook", None)
        try:
            if hook_owner is not None:
                hook_owner._grad_ready_hook = None
            rf, hbm, extra, eager_ms = roofline_leg(inner, x)
            line["roofline"] = rf
            if hbm:
                copy = (calib or {}).get("copy_1GiB_best_gbps") or (calib or {}).get("copy_256MB_gbps")
                if copy:  # the same launches against what a plain 16-byte copy reaches on THIS box (calibration probe)
                    hbm["copy_ceiling_gbps"] = copy
                    hbm["frac_of_copy_ceiling"] = round(hbm["achieved"] / copy, 4)
                    if "replay_achieved" in hbm:
                        hbm["replay_frac_of_copy_ceiling"] = round(hbm["replay_achieved"] / copy, 4)
                line["roofline_hbm_convblock"] = hbm
            line["kernels"] = extra
            line["instrumented_kernel_ms_per_step"] = eager_ms
        except Exception as e:
            line["roofline"] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            if hook_owner is not None:
                hook_owner._grad_ready_hook = hook
    if rank == 0 and world == 1 and not args.no_extras:
        legs = extra_legs(model, x, dev)
        line.update(legs)
        line["eager_api"] = eager_api_leg(model, x, min(windows), legs.get("batch1", {}).get("ms_per_step"))
    if rank == 0 and world == 1 and not args.no_dp1:
        line["dp1"] = dp1_leg()
    if rank == 0 and not args.no_cpu_baseline:
        # (N > 1: one thread setting, three steps -- the other ranks are waiting)
        line["cpu_baseline"] = cpu_baseline(args.batch) if world == 1 else cpu_baseline(args.batch, sweep=(16,), final_steps=3)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
