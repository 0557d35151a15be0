#!/usr/bin/env python3
"""Mbp/s classified on the Jaeger predict hot path (window table -> encoder -> conv forward).

Workload (BASELINE.json configs[1]): jaeger_38341_1.4M_fragment stand-in (the in-tree
1500-bp "brain" architecture with seeded random weights, the real checkpoint is not
shippable), 1500-bp windows, 10 000 synthetic contigs with log-uniform lengths in
[1.5 kb, 200 kb] per GPU, PCG64(20260923 + rank).  One step = one pass over all of a
rank's contigs with the bases and the window table already resident in HBM; N > 1 ranks
each own such a contig set (weak scaling) and gather their logits to rank 0 over RCCL.

Prints ONE JSON line on rank 0 (see README / the driver contract).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

F32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md chip table: v_mfma_f32_32x32x2_f32
F16_MFMA_PEAK_TFLOPS = 2500.0     # dense f16/bf16 MFMA peak (same table); split-f16 issues 3 MFMAs/product


def synth_contigs(rng: np.random.Generator, n_contigs: int, lo: int = 1500, hi: int = 200_000):
    """Log-uniform contig lengths, iid uniform ACGT bases in one contiguous buffer."""
    lengths = np.exp(rng.uniform(np.log(lo), np.log(hi), n_contigs)).astype(np.int64)
    lengths = np.clip(lengths, lo, hi)
    total = int(lengths.sum())
    bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, total, dtype=np.uint8)]
    return lengths, bases


def cpu_baseline(cfg, weights, bases, offsets, table, fsize, target_s: float = 15.0):
    """Time the CPU oracle (fragment strings -> numpy encoder -> torch-CPU f32 forward) on a
    bounded sample of the same windows.  A reported baseline, not the optimisation target."""
    import torch
    from oracle import encoder as oenc
    from oracle import forward as ofwd
    from oracle import fragmenter as ofrag

    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # a container may see every host core but own a CPU-time quota of a few: threads beyond the quota only
    # preempt each other (cgroup v2 cpu.max "quota period", v1 cpu.cfs_quota_us / cpu.cfs_period_us)
    quota = avail
    try:
        q, per = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        if q != "max":
            quota = max(1, int(int(q) / int(per) + 0.5))
    except (OSError, ValueError):
        try:
            q = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
            per = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            if q > 0:
                quota = max(1, int(q / per + 0.5))
        except (OSError, ValueError):
            pass
    cores = max(1, min(avail, quota, 64))       # beyond ~64 threads the batch-96 GEMMs stop scaling
    torch.set_num_threads(cores)
    n_contigs_all = len(offsets) - 1
    win_per_contig = np.array([(offsets[i + 1] - offsets[i]) // fsize for i in range(min(n_contigs_all, 4000))])
    cum = np.cumsum(win_per_contig)

    def run(n_windows):
        n_c = int(np.searchsorted(cum, n_windows) + 1)
        recs = [(f"c{i}", bases[offsets[i]:offsets[i + 1]].tobytes().decode()) for i in range(n_c)]
        t0 = time.perf_counter()
        wins = []
        for frag in ofrag.fragment_strings(recs, fsize, fsize):
            wins.append(frag.split(",", 1)[0])
            if len(wins) >= n_windows:
                break
        ids = oenc.encode_windows(wins, fsize)
        n = 0
        for i in range(0, len(wins), 96):                       # reference default --batch 96
            n += ofwd.forward(cfg, weights, ids[i:i + 96])["prediction"].shape[0]
        return n, time.perf_counter() - t0

    run(8)                                                       # warm-up (thread pools, BLAS)
    n_w, dt = run(24)
    want = int(max(96, min(n_w / dt * target_s, 20000)))
    n_w, dt = run(want)
    return {"value": round(n_w * fsize / dt / 1e6, 5), "unit": "Mbp/s", "cores": cores, "kind": "port",
            "sample": f"first {n_w} windows x {fsize} bp of the workload; oracle = python fragmenter + "
                      f"numpy encoder + torch-CPU f32 forward, batch 96, {cores} threads "
                      f"({avail} cores visible, CPU quota {quota}), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--contigs", type=int, default=10_000)
    ap.add_argument("--fsize", type=int, default=1500)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--precision", choices=["f32", "f16x3"], default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true",
                    help="experiments only: no HIP events around the conv launches (roofline fields read 0)")
    ap.add_argument("--timed-dbg", type=int, default=None,
                    help="experiments only: set the conv kernel's JG_DBG ablation mask after the warm-up steps "
                         "(the timed steps then read real activations; their results are wrong)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import yaml

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev_t = torch.device("cuda", local_rank)

    from jaeger_amd.engine import JaegerHipEngine, frame_length
    from jaeger_amd.fragment import build_window_table
    from jaeger_amd import dist as jdist
    from jaeger_amd.plan import build_plan
    from jaeger_amd.weights import random_weights

    cfg = yaml.safe_load((ROOT / "tests" / "golden" / "brain_project.yaml").read_text())["model"]
    weights = random_weights(build_plan(cfg), seed=38341)
    import warnings
    warnings.simplefilter("ignore")
    eng = JaegerHipEngine(model_cfg=cfg, weights=weights, device_id=local_rank, chunk=args.chunk,
                          precision=args.precision)
    mode = eng.model.precision

    fsize = args.fsize
    l_pad = frame_length(fsize)
    rng = np.random.Generator(np.random.PCG64(20260923 + rank))
    lengths, bases = synth_contigs(rng, args.contigs)
    offsets = np.zeros(lengths.size + 1, np.int64)
    np.cumsum(lengths, out=offsets[1:])
    table = build_window_table(lengths, fsize, fsize)
    n_win = len(table)
    win_start = (offsets[table.contig] + table.start).astype(np.int64)
    win_len = table.length.astype(np.int32)
    bp_per_step = int(win_len.sum())

    # inputs resident in HBM before the timed region
    d_bases = torch.from_numpy(bases).to(dev_t)
    d_start = torch.from_numpy(win_start).to(dev_t)
    d_len = torch.from_numpy(win_len).to(dev_t)
    n_cls = eng.model.widths["prediction"]
    d_pred = torch.zeros((n_win, n_cls), dtype=torch.float32, device=dev_t)
    d_rel = torch.zeros((n_win, max(eng.model.widths["reliability"], 1)), dtype=torch.float32, device=dev_t)
    d_counts = torch.zeros((n_win, 4), dtype=torch.int32, device=dev_t)
    torch.cuda.synchronize()

    def step():
        eng.model.predict_windows_raw(
            d_bases.data_ptr(), bases.size, d_start.data_ptr(), d_len.data_ptr(), n_win, fsize, eng.lut,
            eng.encode_flags, l_pad,
            {"prediction": d_pred.data_ptr(),
             "reliability": d_rel.data_ptr() if eng.model.widths["reliability"] else 0},
            counts_ptr=d_counts.data_ptr(), chunk=args.chunk)
        eng.device.sync()
        if world > 1:                       # the final gather of per-window logits to rank 0
            return jdist.gather_rows(d_pred, dst=0)
        return None

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    if args.timed_dbg is not None:
        os.environ["JG_DBG"] = str(args.timed_dbg)
    eng.device.profile_enable(not args.no_profile)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gathered = step()
    fence()
    dt = time.perf_counter() - t0
    prof = eng.device.profile_read()
    eng.device.profile_enable(False)

    t = torch.tensor([dt, float(bp_per_step), float(n_win)], dtype=torch.float64, device=dev_t)
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_max, bp_total, win_total = float(tmax[0]), float(tsum[1]), float(tsum[2])
    else:
        dt_max, bp_total, win_total = dt, float(bp_per_step), float(n_win)

    if rank == 0:
        steps = max(args.steps, 1)
        value = bp_total * steps / dt_max / 1e6
        conv_s = prof["conv_ms"] / 1e3
        ach = prof["conv_flops"] / conv_s / 1e12 if conv_s > 0 else 0.0
        # roofline: algorithmic (f32-equivalent) conv FLOP/s against the matrix-core peak the kernel
        # can reach: exact-f32 MFMA, or the f16 MFMA peak / 3 for the split-f16 scheme
        peak = F32_MFMA_PEAK_TFLOPS if mode == "f32" else F16_MFMA_PEAK_TFLOPS / 3.0
        # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (it cannot be
        # read from inside the process); the committed summary applies to the default configuration only
        traffic = None
        try:
            pmc = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())["conv_f16x3_kernel"]
            if mode == pmc["precision"] and fsize == pmc["fsize"] and args.chunk in (0, pmc["chunk"]):
                traffic = pmc["traffic_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        line = {
            "metric": "Mbp/s classified (1500bp frags)", "value": round(value, 3), "unit": "Mbp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if mode == "f32" else "f16x3 (split-f16, f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": f"jaeger_38341_1.4M_fragment stand-in (nn_config_1500bp_nmd_merge_6_class_brain "
                                   f"architecture, seeded random weights), {fsize}bp windows stride {fsize}, "
                                   f"{args.contigs} synthetic contigs/GPU log-uniform 1.5-200 kb",
                       "windows_per_gpu": int(win_total / world), "bp_per_gpu": int(bp_total / world),
                       "parallelism": f"contig-sharded x{world}, final RCCL gather"},
            "roofline": {"bound": "mfma", "achieved": round(ach, 3), "peak": round(peak, 1),
                         "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                         "kernel": "conv_f32_kernel" if mode == "f32" else "conv_f16x3_kernel", "launches": int(prof["conv_launches"]),
                         "avg_launch_ms": round(prof["conv_ms"] / max(prof["conv_launches"], 1), 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, weights, bases, offsets, table, fsize)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
