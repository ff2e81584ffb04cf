"""bench.py's JSON assembly on synthetic launch records (no GPU): the `roofline` / `roofline_hbm_convblock` objects the
driver reads, per-step normalisation over the instrumented steps, and the attention summary of the extra legs."""
import json

import bench


def _fake_records():
    conv = ("adp_conv1d", "conv_mm_kernel<32, 3, 1, 1, true, 0, 32, 2>",
            {"flops": 2 * 4 * 1024 * 256 * 1024 * 3, "bytes": 17_000_000, "shape": "B4 R1024 M1024 N256 KT3 s1 up1 tr1 pro0"}, 0.050)
    shallow = ("adp_conv1d", "conv_tile32_kernel<false, 1, 16, true, false>",
               {"flops": 2 * 4 * 32 * 65536 * 32 * 3, "bytes": 100_000_000, "shape": "B4 R32 M32 N65536 KT3 s1 up1 tr0 pro1"}, 0.030)
    norm = ("adp_gn_silu_bwd_apply", "gn_bwd_apply_kernel", {"bytes": 50_000_000, "shape": "B4 C32 L65536"}, 0.012)
    return [conv] * 10 + [shallow] * 2 + [norm] * 4


def test_roofline_objects_from_instrumented_steps(monkeypatch):
    calls = []

    def fake_profiled_step(model, step):
        calls.append(1)
        return _fake_records()

    monkeypatch.setattr(bench, "profiled_step", fake_profiled_step)
    rf, hbm, table, total_ms = bench.roofline_leg(None, None)
    assert len(calls) == 5  # two warm steps + three measured ones
    # per-step normalisation: 10 conv launches per step, 0.05 ms each
    assert rf["kernel"].startswith("conv_mm_kernel") and rf["launches"] == 10 and abs(rf["avg_us"] - 50.0) < 1e-6
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == bench.PEAK_F32_MFMA_TFLOPS
    want_tf = 2 * 4 * 1024 * 256 * 1024 * 3 / 0.050e-3 / 1e12
    assert abs(rf["achieved"] - want_tf) < 0.01 and abs(rf["frac"] - want_tf / rf["peak"]) < 1e-3
    assert isinstance(rf["algorithmic_bytes_per_launch"], int) and rf["algorithmic_bytes_per_launch"] == 17_000_000
    assert "traffic" in rf and 0 < rf["share_of_step"] < 1
    assert hbm["bound"] == "hbm" and hbm["unit"] == "GB/s" and hbm["launches"] == 2
    assert abs(hbm["achieved"] - 100_000_000 / 0.030e-3 / 1e9) < 0.5 and hbm["peak"] == bench.PEAK_HBM_GBPS
    assert abs(total_ms - (10 * 0.050 + 2 * 0.030 + 4 * 0.012)) < 1e-6
    assert table["gn_bwd_apply_kernel"]["bound"] == "hbm" and table["gn_bwd_apply_kernel"]["launches"] == 4
    json.dumps({"roofline": rf, "roofline_hbm_convblock": hbm, "kernels": table})  # serialisable as emitted


def test_attention_summary_per_kernel_and_per_call():
    recs = [("adp_attn_fwd", "attn_fwd_kernel<true>(args)", {"flops": 4_000_000_000}, 0.020),
            ("adp_attn_fwd", "attn_fwd_combine_kernel(args)", {}, 0.008),
            ("adp_attn_bwd", "attn_sum_splits_kernel(args)", {"flops": 14_000_000_000}, 0.008),
            ("adp_attn_bwd", "attn_bwd_kv_kernel<true>(args)", {}, 0.040),
            ("adp_attn_bwd", "attn_bwd_q_kernel<true>(args)", {}, 0.022),
            ("adp_conv1d", "conv_mm_kernel<32>(args)", {"flops": 1}, 1.0)]
    s = bench._attn_summary(recs)
    assert "conv_mm_kernel<32>" not in s
    assert s["attn_fwd_kernel<true>"] == {"launches": 1, "avg_us": 20.0}
    assert s["adp_attn_fwd"]["calls"] == 1 and abs(s["adp_attn_fwd"]["avg_us_per_call"] - 28.0) < 1e-6
    assert abs(s["adp_attn_fwd"]["tflops"] - 4e9 / 0.028e-3 / 1e12) < 0.01
    # the backward's flops ride on its first kernel but are rated over the whole call (70 us), not over that kernel
    assert abs(s["adp_attn_bwd"]["tflops"] - 14e9 / 0.070e-3 / 1e12) < 0.01


def test_clock_reader_never_raises():
    """calibration's clock / power reader is best effort: a dict whatever the box exposes (no GPU here: an error entry or
    whatever sysfs holds), JSON-serialisable."""
    info = bench._read_clocks()
    assert isinstance(info, dict) and info
    json.dumps(info)


def test_winograd_fraction_by_kernel_instantiation():
    """The roofline objects rate kernels in direct-form flops; `frac_executed` needs the share each instantiation executes."""
    f = bench.winograd_fraction
    assert f("conv_mm4_kernel<false, 1, 64, 4>") == 0.5 and f("conv_tile32_kernel<false, 1, 16, true, false>") == 0.5
    assert f("wgrad_mm_kernel<64, 3, 1, 1, 0, 1, true, true>") == 0.5
    assert abs(f("wgrad_mm_kernel<64, 3, 1, 1, 0, 1, true, false>") - 2 / 3) < 1e-9
    assert abs(f("conv_mm_kernel<64, 3, 1, 1, true, 0, 32, 2, true, 1>") - 2 / 3) < 1e-9
    assert f("conv_mm_kernel<64, 1, 1, 1, true, 0, 32, 2, false, 1>") == 1.0 and f("gn_bwd_apply_vec_kernel<256>") == 1.0
    assert f("wgrad_mm_kernel<64, 2, 2, 1, 0, 1, false, false>") == 1.0


def test_bench_launches_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (how the driver calls --gpus 1) must start its two ranks itself; under a
    launcher it must not.  GPU-less: the ranks stop after the rendezvous (gloo) -- flags parsed, process group joined from the
    environment bench.py's own launcher set, one JSON line from rank 0."""
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["ADP_BENCH_RENDEZVOUS_ONLY"] = "1"
    out = subprocess.run([sys.executable, bench.__file__, "--gpus", "2", "--steps", "7", "--warmup", "1"], env=env,
                         capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert lines == [{"rendezvous": "ok", "n_gpus": 2, "rank_sum": 3.0, "steps": 7}], out.stdout
    # the command it builds is the contract's torchrun line with the caller's own flags
    seen = {}
    rc = bench._self_launch(4, argv=["--gpus", "4", "--steps", "3"], run=lambda cmd, env: seen.update(cmd=cmd, env=env) or 0)
    assert rc == 0 and seen["cmd"][1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=4" in seen["cmd"] and seen["cmd"][-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["cmd"][seen["cmd"].index("--master-addr") + 1] == "127.0.0.1"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
