#!/usr/bin/env python3
"""Mbp/s classified on the Jaeger predict hot path (window table -> encoder -> conv forward).

Workloads (``--config``), all synthetic (PCG64-seeded uniform ACGT), one contig set per GPU (weak scaling):

* ``default``     BASELINE.json configs[1]: jaeger_38341_1.4M_fragment stand-in (the in-tree 1500-bp "brain"
                  architecture with seeded random weights - the real checkpoint is not shippable), 1500-bp
                  windows, 10 000 contigs with log-uniform lengths in [1.5 kb, 200 kb], PCG64(20260923 + rank).
* ``frag1m``      the per-GPU shard of configs[2]: 125 000 fragments of exactly 1 500 bp (1 M over 8 GPUs),
                  PCG64(20260924 + rank), same model.
* ``baseline500`` configs[3]: nn_config_500bp_baseline architecture, 1 M fragments of exactly 500 bp per GPU.

One step = one pass over all of a rank's contigs with the bases and the window table already resident in HBM;
ranks gather their logits to rank 0 over RCCL inside the timed region.  ``--gpus N`` without a torchrun
environment starts N ranks itself (fresh child processes, created before this process touches the GPU).

Prints ONE JSON line on rank 0 (see README / the driver contract).
"""

from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

F32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md chip table: v_mfma_f32_32x32x2_f32
F16_MFMA_PEAK_TFLOPS = 2500.0     # dense f16/bf16 MFMA peak (same table); split-f16 issues 3 MFMAs/product
HBM_PEAK_GBS = 8000.0

CONFIGS = {
    "default": dict(model="brain", fsize=1500, contigs=10_000, seed=20260923, exact=None,
                    label="jaeger_38341_1.4M_fragment stand-in (nn_config_1500bp_nmd_merge_6_class_brain architecture, "
                          "seeded random weights)"),
    "frag1m": dict(model="brain", fsize=1500, contigs=125_000, seed=20260924, exact=1500,
                   label="jaeger_38341_1.4M_fragment stand-in (brain architecture, seeded random weights), per-GPU "
                         "shard of the 1 M x 1500 bp fragment set"),
    "baseline500": dict(model="baseline500", fsize=500, contigs=1_000_000, seed=20260925, exact=500,
                        label="nn_config_500bp_baseline architecture (seeded random weights)"),
    # not a BASELINE.json config: the reference's pyramid ResNet (train_config/nn_config_baseline.yaml: widths 32 - 256,
    # stride-2 blocks, dilations 1 - 8) at its own 2 000-bp crop, on the default workload's contig mixture
    "pyramid": dict(model="pyramid", fsize=2000, contigs=10_000, seed=20260926, exact=None, gain=0.85,
                    label="nn_config_baseline architecture (pyramid ResNet 32/64/128/256 channels; seeded random weights, "
                          "block kernels x0.85)"),
}


def synth_contigs(rng: np.random.Generator, n_contigs: int, lo: int = 1500, hi: int = 200_000, exact: int | None = None):
    """Log-uniform contig lengths (or ``exact``-length fragments), iid uniform ACGT bases in one buffer."""
    if exact is not None:
        lengths = np.full(n_contigs, exact, np.int64)
    else:
        lengths = np.exp(rng.uniform(np.log(lo), np.log(hi), n_contigs)).astype(np.int64)
        lengths = np.clip(lengths, lo, hi)
    total = int(lengths.sum())
    bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, total, dtype=np.uint8)]
    return lengths, bases


def cpu_quota():
    """(cores visible, CPU quota): a container may see every host core but own a CPU-time quota of a few; threads
    beyond the quota only preempt each other (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
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
    return avail, quota


def cpu_baseline(cfg, weights, bases, offsets, fsize, flops_per_window: float, target_s: float = 11.0):
    """Time the CPU oracle (python fragment strings -> numpy encoder -> torch-CPU f32 forward on the fused oneDNN
    kernels, ``oracle.forward.FAST``) on a bounded sample of the same windows, twice: 4 threads (the reference's
    default ``--workers 4``, cli.py:260) and every core of the CPU quota.  A reported baseline, not the target."""
    import torch
    from oracle import encoder as oenc
    from oracle import forward as ofwd
    from oracle import fragmenter as ofrag

    avail, quota = cpu_quota()
    all_cores = max(1, min(avail, quota, 64))
    n_contigs_all = len(offsets) - 1
    win_per_contig = np.array([(offsets[i + 1] - offsets[i]) // fsize for i in range(min(n_contigs_all, 20000))])
    cum = np.cumsum(win_per_contig)
    ofwd.FAST = True

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
        with torch.no_grad():
            for i in range(0, len(wins), 96):                   # reference default --batch 96
                n += ofwd.forward(cfg, weights, ids[i:i + 96])["prediction"].shape[0]
        return n, time.perf_counter() - t0

    legs = []
    try:
        for threads in sorted({min(4, all_cores), all_cores}):
            torch.set_num_threads(threads)
            run(8)                                               # warm-up (thread pools, oneDNN primitives)
            n_w, dt = run(96)
            want = int(max(96, min(n_w / dt * target_s, 20000)))
            if threads == all_cores:                         # at least 2 000 windows on the full-quota leg
                want = max(want, 2000)
            if want > 96 * 1.5:
                n_w, dt = run(want)
            legs.append({"threads": threads, "value": round(n_w * fsize / dt / 1e6, 5), "windows": n_w,
                         "seconds": round(dt, 1), "gflops": round(n_w * flops_per_window / dt / 1e9, 1)})
    finally:
        ofwd.FAST = False
    best = max(legs, key=lambda leg: leg["value"])
    return {"value": best["value"], "unit": "Mbp/s", "cores": best["threads"], "kind": "port",
            "gflops": best["gflops"], "legs": legs,
            "sample": f"first {best['windows']} windows x {fsize} bp of the workload; reference-equivalent CPU restatement "
                      f"(python fragmenter + numpy encoder + torch-CPU f32 forward on oneDNN channels-last convs, fused "
                      f"GELU, batch 96), not TensorFlow; legs at {' and '.join(str(l['threads']) for l in legs)} threads "
                      f"({avail} cores visible, CPU quota {quota})"}


def visible_gpus() -> int:
    """GPUs this process may use, counted WITHOUT touching the HIP runtime (a process that has initialised the GPU must
    not start other programs on this pool): the *_VISIBLE_DEVICES lists if set, else the KFD topology in sysfs (a node
    with simd_count > 0 is a GPU), else /dev/dri render nodes."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    nodes = Path("/sys/class/kfd/kfd/topology/nodes")
    n = 0
    if nodes.is_dir():
        for node in nodes.iterdir():
            try:
                props = dict(line.split()[:2] for line in (node / "properties").read_text().splitlines() if line.strip())
                if int(props.get("simd_count", "0")) > 0:
                    n += 1
            except (OSError, ValueError):
                continue
        return n
    dri = Path("/dev/dri")
    return len(list(dri.glob("renderD*"))) if dri.is_dir() else 0


def kernel_hash(mode: str = "f16x3") -> str:
    """Fingerprint of the sources of the kernels a line's dominant-kernel figures come from: the PMC summaries under
    profiles/ record it, and a bench line only quotes counters that were collected on the build it is timing.  Split-f16
    lines: the conv device headers, the instantiation file and the fused small-window kernel; exact-f32 lines add
    ``jg_kernels.hip`` (``conv_f32_kernel``) - their hash is a different one on purpose."""
    import hashlib
    h = hashlib.sha256()
    src = ROOT / "jaeger_amd" / "csrc"
    names = ["jg_common.h", "jg_conv_dev.h", "jg_conv_f16.hip", "jg_conv_f16_impl.h", "jg_small.h", "jg_small.hip"]
    if mode == "f32":
        names.append("jg_kernels.hip")
    for name in names:
        h.update((src / name).read_bytes())
    return h.hexdigest()[:16]


def write_fasta_contigs(path, lengths, bases):
    """Contigs of any length, 80 bases per line, headers ``>contig_<i> len=<n>``."""
    with open(path, "wb") as fh:
        off = 0
        for i, n in enumerate(lengths.tolist()):
            fh.write(b">contig_%d len=%d\n" % (i, n))
            seq = bases[off:off + n].tobytes()
            off += n
            fh.write(b"\n".join(seq[j:j + 80] for j in range(0, n, 80)) + b"\n")


def write_fasta_records(path, bases2d):
    """``n`` records of one length (the fragment workloads as a file), one sequence line each, fixed-width headers
    ``>r0000123`` - the whole file image is built as one numpy array (a million records in well under a second)."""
    n, length = bases2d.shape
    width = max(7, len(str(max(n - 1, 0))))
    img = np.empty((n, 1 + 1 + width + 1 + length + 1), np.uint8)
    img[:, 0] = ord(">")
    img[:, 1] = ord("r")
    idx = np.arange(n, dtype=np.int64)
    for d in range(width):
        img[:, 2 + d] = (idx // 10 ** (width - 1 - d)) % 10 + ord("0")
    img[:, 2 + width] = ord("\n")
    img[:, 3 + width:3 + width + length] = bases2d
    img[:, -1] = ord("\n")
    with open(path, "wb") as fh:
        fh.write(memoryview(img.reshape(-1)))


def e2e_leg(cfg, weights, wl, fsize, records: int = 0, seed: int | None = None):
    """FASTA on tmpfs -> ``jaeger_amd.predict.run_core`` in-process (defaults: DUST on, host pipeline on) -> TSV:
    the end-to-end figure SURVEY 8(d) asks for next to the resident-input metric (commands/predict.py:488-860).
    ``records == 0``: the file is the 10 000-contig mixture of BASELINE configs[1] (PCG64(20260923), log-uniform
    1.5 - 200 kb) for every model family - a realistic assembly.  ``records > 0``: the fragment workload itself as a file
    (``records`` sequences of exactly ``fsize`` bases, PCG64(``seed``)) - per-record host work (names, window table,
    aggregation, repeat scan, one TSV row per record) is what that run prices.  The FASTA and the model directory are
    written before the clock starts; the stage split is run_core's own (``LAST_RUN``).  Two runs in this process: the
    FIRST pays the file's first page-cache touch and whatever run_core imports lazily (the HIP runtime and the library are
    already loaded by the bench itself - a CLI user's cold start is in DESIGN 6, not here), the second is warm; both are
    reported by name and ``mbps`` is the better of the two (the protocol of round 4's lines)."""
    import shutil
    import tempfile

    import yaml

    from jaeger_amd import predict as P
    from jaeger_amd.weights import save_npz
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    tmp = Path(tempfile.mkdtemp(prefix="jaeger_bench_e2e_", dir=base))
    try:
        fa = tmp / "bench.fasta"
        if records:
            rng = np.random.Generator(np.random.PCG64(wl["seed"] if seed is None else seed))
            bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, records * fsize, dtype=np.uint8)]
            write_fasta_records(fa, bases.reshape(records, fsize))
        else:
            rng = np.random.Generator(np.random.PCG64(CONFIGS["default"]["seed"]))
            lengths, bases = synth_contigs(rng, CONFIGS["default"]["contigs"])
            write_fasta_contigs(fa, lengths, bases)
        mdir = tmp / "model_root" / "model"
        mdir.mkdir(parents=True)
        name = "bench_model"
        shutil.copyfile(ROOT / "tests" / "golden" / f"{wl['model']}_project.yaml", mdir / f"{name}_project.yaml")
        (mdir / f"{name}_classes.yaml").write_text(yaml.safe_dump({"classes": cfg["class_label_map"]}))
        save_npz(mdir / f"{name}.weights.npz", weights)
        out = tmp / "out"
        runs = []
        for _ in range(2):
            P.wait_for_release()             # the run before has given its device memory and pinned staging back
            t0 = time.perf_counter()
            n_rows = P.run_core(input=str(fa), output=str(out), model_path=str(tmp / "model_root"), fsize=fsize, stride=fsize,
                                overwrite=True, dustmask=True, verbose=0, batch=96, rc=0.1, pc=3)
            runs.append((time.perf_counter() - t0, {k: v for k, v in P.LAST_RUN.items() if k not in ("timeline", "t_start_epoch")}))
        P.wait_for_release()
        dt, stages = min(runs, key=lambda r: r[0])
        tsv = next(out.rglob("bench.tsv"), None)
        what = (f"{records} records of exactly {fsize} bp as a FASTA (tmpfs)" if records else
                "10 000-contig synthetic FASTA (tmpfs)")
        return {"mbps": round(bases.size / dt / 1e6, 2), "seconds": round(dt, 3), "bp": int(bases.size),
                "first_run_mbps": round(bases.size / runs[0][0] / 1e6, 2),
                "second_run_mbps": round(bases.size / runs[1][0] / 1e6, 2),
                "seconds_each_run": [round(r[0], 3) for r in runs], "tsv_rows": int(n_rows or 0),
                "tsv_bytes": tsv.stat().st_size if tsv else 0, "stages": stages,
                "what": what + " -> parallel ingest -> one fused call (DUST on the GPU, window table, encode + forward, "
                        "host buffers over PCIe through pinned staging) with the terminal-repeat scan and the per-contig "
                        "aggregation beside it -> TSV; in-process run_core with its defaults; mbps = the better of two "
                        "runs in this process (first_run_mbps / second_run_mbps name them); writing the FASTA / model "
                        "directory is outside the clock"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_fields(name: str, mode: str, fsize: int, chunk: int, avg_launch_ms: float):
    """HBM traffic and matrix-core utilisation of the dominant kernel from the committed rocprofv3 --pmc summaries
    (profiles/pmc_traffic.json, profiles/mfma_util.json) - attached only when they were collected on THIS kernel build."""
    out = {"traffic": None, "hbm_gbs": None, "other_device": None, "pmc_kernel_hash": None, "pmc_stale": None}
    try:
        tr = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())
        mu = json.loads((ROOT / "profiles" / "mfma_util.json").read_text())
    except (OSError, ValueError):
        return out
    here = kernel_hash(mode)
    key = {"default": "conv_f16x3_kernel", "baseline500": "small_net_kernel"}.get(name)
    if key is None or mode != "f16x3" or chunk != 0:
        return out
    t, m = tr.get(key, {}), mu.get(key, {})
    if t.get("fsize", fsize) != fsize:
        return out
    out["pmc_kernel_hash"] = tr.get("kernel_hash")
    out["pmc_stale"] = tr.get("kernel_hash") != here or mu.get("kernel_hash") != here
    if out["pmc_stale"]:
        return out
    out["traffic"] = t.get("traffic_bytes_per_launch")
    if out["traffic"] and avg_launch_ms > 0:
        out["hbm_gbs"] = round(out["traffic"] / (avg_launch_ms * 1e-3) / 1e9, 1)
    # matrix-core busy share and effective clock are properties of the DEVICE the counters were collected on (and of a
    # profiled run): they are quoted as such, next to the launch duration they belong to - this run's device is
    # calibrated by `box` instead
    out["other_device"] = {"mfma_busy_frac": m.get("mfma_busy_frac"), "eff_clock_ghz": m.get("eff_clock_ghz"),
                           "avg_launch_ms_there": m.get("avg_launch_ms_sq_pass"),
                           "note": "collected under rocprofv3 --pmc on another device (profiles/mfma_util.json), not on "
                                   "the device this line was timed on"}
    return out


def spawn_ranks(n: int, oversubscribe: bool = False, timeout_s: float = 1800.0) -> int:
    """``--gpus N`` outside torchrun: N fresh child ranks over RCCL.  The parent never initialises HIP (GPUs are counted
    from sysfs / the environment), children are plain ``subprocess`` launches in a process group of their own; a rank
    that fails has its stderr surfaced, and a rank that outlives ``timeout_s`` takes the whole group down (exit 124)."""
    import signal
    import socket
    import tempfile
    have = visible_gpus()
    if have < n and not (oversubscribe and have >= 1):
        print(f"bench.py: --gpus {n} but only {have} GPU(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs, errs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r % have), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        err = tempfile.TemporaryFile()
        errs.append(err)
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=err,
                                      start_new_session=True))
    deadline = time.monotonic() + timeout_s
    timed_out, failed_at = False, None
    out_chunks: list[bytes] = []
    import threading
    reader = threading.Thread(target=lambda: out_chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()                                        # rank 0's one JSON line (drained so that the pipe never fills)
    while any(p.poll() is None for p in procs):
        now = time.monotonic()
        if failed_at is None and any(p.poll() not in (None, 0) for p in procs):
            failed_at = now                               # a rank died: the others sit in a collective - a short grace, then end them
        if now > deadline:
            timed_out = True
            break
        if failed_at is not None and now > failed_at + 15.0:
            break
        time.sleep(0.2)
    for p in procs:
        if p.poll() is None:
            try:
                os.killpg(p.pid, signal.SIGKILL)          # the exact process group this function created
            except ProcessLookupError:
                pass
            p.wait()
    reader.join(timeout=10.0)
    out = b"".join(out_chunks)
    rcs = [p.returncode for p in procs]
    for r, (rc, err) in enumerate(zip(rcs, errs)):
        err.seek(0)
        text = err.read().decode(errors="replace")
        err.close()
        if rc != 0 and text.strip():
            sys.stderr.write(f"---- bench.py rank {r} (exit {rc}) stderr ----\n{text[-4000:]}\n")
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    if timed_out:
        print(f"bench.py: ranks still running after {timeout_s:.0f} s were killed", file=sys.stderr)
        return 124
    return max(abs(rc) for rc in rcs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="default")
    ap.add_argument("--contigs", type=int, default=None, help="contigs / fragments per GPU (default: the config's)")
    ap.add_argument("--fsize", type=int, default=None)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--precision", choices=["f32", "f16x3"], default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact-f32", action="store_true", help="skip the short exact-f32 side measurement")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end FASTA -> TSV leg")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the other BASELINE configs (`also`: frag1m and baseline500 at 2 timed steps each)")
    ap.add_argument("--no-box", action="store_true", help="skip the box calibration (bare MFMA loop behind the timed steps)")
    ap.add_argument("--box-seconds", type=float, default=0.5)
    ap.add_argument("--no-profile", action="store_true",
                    help="experiments only: no HIP events around the conv launches (roofline fields read 0)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="tests only: let the N ranks share the visible GPUs (rank r on GPU r %% visible) and exchange "
                         "over gloo instead of RCCL, so that the N-rank launch can be exercised on a 1-GPU box")
    ap.add_argument("--collective", choices=["auto", "nccl"], default="auto",
                    help="'nccl': take the N-rank exchange path over RCCL even with ONE rank (world 1: init_process_group on "
                         "the device, the padded gather of the logits on device tensors inside the timed region, the "
                         "all_gather of the per-rank statistics, barrier, destroy) - first contact for the collective code "
                         "on a one-GPU box; a failure prints the RCCL / HIP error and exits non-zero, there is no gloo fallback")
    ap.add_argument("--rank-timeout", type=float, default=1800.0,
                    help="--gpus N outside torchrun: seconds after which the child ranks are killed (exit 124)")
    ap.add_argument("--rank-seed", type=int, default=None,
                    help="tests only: generate the synthetic contigs of rank R (seed = config seed + R) in a "
                         "single-rank run")
    ap.add_argument("--rank-contigs", default=None,
                    help="tests only: comma-separated contigs / fragments per rank (unequal and empty shards), e.g. "
                         "'200,0,150,40'; overrides --contigs")
    ap.add_argument("--dump-gather", default=None,
                    help="tests only: rank 0 writes the gathered (or, with one rank, its own) logits of the last step to "
                         "this .npy file")
    ap.add_argument("--conv-pc", type=int, choices=[0, 1, 2], default=0,
                    help="experiments only (libjaeger_hip_exp.so): 128-channel five-tap convs on the two-workgroup kernel (0, the "
                         "default and the only one in the shipped library), on the producer / consumer kernel (1) or on "
                         "the two-workgroup kernel with the pipelined main loop (2); same results bit for bit")
    ap.add_argument("--timed-dbg", type=int, default=None,
                    help="experiments only (libjaeger_hip_exp.so): set the conv kernel's JG_DBG ablation mask after "
                         "the warm-up steps (the timed steps then read real activations; their results are wrong)")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, args.oversubscribe, args.rank_timeout))
    if world_env is not None and int(world_env) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE={world_env}", file=sys.stderr)
        sys.exit(2)

    # The contract is ONE JSON line on stdout.  RCCL prints a version banner through C stdio at communicator set-up, and
    # with stdout redirected that buffer only empties at exit - behind the line (seen in round 6's first world-1 RCCL run).
    # So this process's fd 1 goes to /dev/null for its whole life, and rank 0 writes the line to the real stdout itself.
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(os.open(os.devnull, os.O_WRONLY), 1)

    import torch
    import torch.distributed as dist
    import yaml

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("JAEGER_BENCH_FAIL_RANK") == str(rank) and world > 1:      # tests: a rank that dies before the rendezvous
        raise RuntimeError(f"JAEGER_BENCH_FAIL_RANK={rank}: this rank was told to fail")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    # the ranks of a node share its cores: every rank generates its own contigs and runs its own host threads
    lws = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    torch.set_num_threads(max(1, min(cpu_quota()) // lws))
    coll = world > 1 or args.collective == "nccl"           # the exchange path: every N > 1 run, and world 1 on request
    if coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:                  # (world 1 outside torchrun)
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if args.oversubscribe and world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            except Exception as e:                           # noqa: BLE001 - surfaced, never replaced by another backend
                print(f"bench.py: RCCL initialisation failed on rank {rank} (HSA_ENABLE_IPC_MODE_LEGACY="
                      f"{os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}): {type(e).__name__}: {e}", file=sys.stderr)
                raise
    dev_t = torch.device("cuda", local_rank)
    coll_dev = torch.device("cpu") if (world > 1 and args.oversubscribe) else dev_t

    ctx = dict(torch=torch, dist=dist, rank=rank, world=world, local_rank=local_rank, dev_t=dev_t, coll_dev=coll_dev,
               coll=coll)
    line = measure(args, args.config, args.steps, args.warmup, ctx, headline=True)
    if rank == 0:
        data = (json.dumps(line) + "\n").encode()
        while data:
            data = data[os.write(real_stdout, data):]
    if coll:
        dist.barrier()
        dist.destroy_process_group()


def measure(args, name: str, steps: int, warmup: int, ctx: dict, headline: bool):
    """One workload: ``warmup`` untimed steps, ``steps`` timed steps between fences, the line's fields on rank 0 (None on
    the other ranks).  ``headline``: the run's own config - it carries the exact-f32 side measurement, the end-to-end
    leg, the box calibration (straight behind the timed steps, before anything else touches the GPU), the other BASELINE
    configs as ``also`` and the CPU baseline."""
    import warnings

    import yaml

    from jaeger_amd import _lib
    from jaeger_amd import dist as jdist
    from jaeger_amd.engine import JaegerHipEngine, frame_length
    from jaeger_amd.fragment import build_window_table
    from jaeger_amd.plan import build_plan
    from jaeger_amd.weights import random_weights

    torch, dist = ctx["torch"], ctx["dist"]
    rank, world, local_rank, dev_t, coll_dev = ctx["rank"], ctx["world"], ctx["local_rank"], ctx["dev_t"], ctx["coll_dev"]
    coll = ctx["coll"]
    wl = CONFIGS[name]
    cfg = yaml.safe_load((ROOT / "tests" / "golden" / f"{wl['model']}_project.yaml").read_text())["model"]
    weights = random_weights(build_plan(cfg), seed=38341)
    if wl.get("gain"):                  # He-uniform stand-in kernels blow a 36-conv residual pyramid's logits up to +-900
        for key in weights:
            if key.startswith("rep/") and key.endswith("/kernel"):
                weights[key] = weights[key] * np.float32(wl["gain"])
    warnings.simplefilter("ignore")
    eng = JaegerHipEngine(model_cfg=cfg, weights=weights, device_id=local_rank, chunk=args.chunk,
                          precision=args.precision)
    mode = eng.model.precision
    if args.conv_pc:
        eng.device.set_conv_pc(args.conv_pc)
    if args.timed_dbg is not None and "_exp" not in _lib.lib_path().name:
        print("bench.py: --timed-dbg needs the experiment build (make -C jaeger_amd/csrc exp; "
              "JAEGER_HIP_LIB=jaeger_amd/libjaeger_hip_exp.so)", file=sys.stderr)
        sys.exit(2)

    fsize = (args.fsize if headline else None) or wl["fsize"]
    n_contigs = (args.contigs if headline else None) or wl["contigs"]
    if args.rank_contigs and headline:
        per_rank = [int(x) for x in args.rank_contigs.split(",")]
        n_contigs = per_rank[(rank if args.rank_seed is None else args.rank_seed) % len(per_rank)]
    l_pad = eng.model.row_length(fsize)
    rng = np.random.Generator(np.random.PCG64(wl["seed"] + (rank if args.rank_seed is None else args.rank_seed)))
    lengths, bases = synth_contigs(rng, n_contigs, exact=(fsize if wl["exact"] else None))
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
    n_rel = eng.model.widths["reliability"]
    d_pred = torch.zeros((n_win, n_cls), dtype=torch.float32, device=dev_t)
    d_rel = torch.zeros((n_win, max(n_rel, 1)), dtype=torch.float32, device=dev_t)
    d_counts = torch.zeros((n_win, 4), dtype=torch.int32, device=dev_t)
    torch.cuda.synchronize()

    split = {"compute_s": 0.0, "gather_s": 0.0}    # per rank, timed steps only: where a step's time goes

    def step(n=n_win):
        ta = time.perf_counter()
        if n > 0:                           # (an empty shard - tests - still takes part in the gather)
            eng.model.predict_windows_raw(
                d_bases.data_ptr(), bases.size, d_start.data_ptr(), d_len.data_ptr(), n, fsize, eng.lut,
                eng.encode_flags, l_pad,
                {"prediction": d_pred.data_ptr(), "reliability": d_rel.data_ptr() if n_rel else 0},
                counts_ptr=d_counts.data_ptr(), chunk=args.chunk)
        eng.device.sync()
        tb = time.perf_counter()
        got = None
        if coll:                            # the final gather of per-window logits to rank 0
            got = jdist.gather_rows(d_pred.to(coll_dev), dst=0)
        split["compute_s"] += tb - ta
        split["gather_s"] += time.perf_counter() - tb
        return got

    def fence():
        if coll:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    fence()
    if args.timed_dbg is not None:
        os.environ["JG_DBG"] = str(args.timed_dbg)
    eng.device.profile_enable(not args.no_profile)
    split["compute_s"] = split["gather_s"] = 0.0
    t0 = time.perf_counter()
    gathered = None
    for _ in range(steps):
        gathered = step()
    fence()
    dt = time.perf_counter() - t0
    # the box calibration runs straight behind the timed steps, on the chip as the steps left it
    box = None
    if headline and world == 1 and not args.no_box and args.timed_dbg is None:
        box = eng.device.box_calibrate(args.box_seconds)
    prof = eng.device.profile_read()
    eng.device.profile_enable(False)
    if args.timed_dbg is not None:
        os.environ["JG_DBG"] = "0"

    t = torch.tensor([dt, float(bp_per_step), float(n_win), split["compute_s"], split["gather_s"]], dtype=torch.float64,
                     device=coll_dev)
    if coll:
        rows = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(rows, t)
        per_rank = torch.stack(rows).cpu().numpy()
    else:
        per_rank = t.cpu().numpy()[None, :]
    dt_max, bp_total, win_total = float(per_rank[:, 0].max()), float(per_rank[:, 1].sum()), float(per_rank[:, 2].sum())

    if rank == 0 and args.dump_gather and headline:
        parts = [g.cpu().numpy() for g in gathered] if gathered is not None else [d_pred.cpu().numpy()]
        np.save(args.dump_gather, np.concatenate(parts, axis=0))
    line = None
    if rank == 0:
        n_steps = max(steps, 1)
        value = bp_total * n_steps / dt_max / 1e6
        # dominant kernel = the matrix-core convolutions of the arithmetic in use (the first layer's table-lookup
        # kernel is reported beside it, not folded in: it runs no MFMA)
        dom = prof["mfma_f16x3"] if mode == "f16x3" else prof["mfma_f32"]
        dom_name = "conv_f32_kernel" if mode == "f32" else "conv_f16x3_kernel (matrix-core convs only)"
        mfma_share = 1.0
        if prof["fused_small"]["launches"] and prof["fused_small"]["ms"] > dom["ms"]:
            # the 32-channel family: ONE fused kernel from ids to pooled sums.  Its first conv is table lookups (no
            # matrix-core work), so the MFMA roofline is quoted on the k = 3 convs' share of the algorithmic FLOPs
            from jaeger_amd.plan import conv_flops_per_position
            rows = conv_flops_per_position(build_plan(cfg))
            fl = [2.0 * k * ci * co for _, k, ci, co, _, _ in rows]
            mfma_share = sum(fl[1:]) / sum(fl)
            dom, dom_name = prof["fused_small"], "small_net_kernel (fused ids -> pooled sums; MFMA share = its k=3 convs)"
        dom_s = dom["ms"] / 1e3
        ach = dom["flops"] * mfma_share / dom_s / 1e12 if dom_s > 0 else 0.0
        # roofline: algorithmic (f32-equivalent) conv FLOP/s against the matrix-core peak the kernel can reach:
        # exact-f32 MFMA, or the f16 MFMA peak / 3 for the split-f16 scheme
        peak = F32_MFMA_PEAK_TFLOPS if mode == "f32" else F16_MFMA_PEAK_TFLOPS / 3.0
        # HBM traffic and matrix-core utilisation of the dominant kernel come from separate rocprofv3 --pmc passes
        # (counters cannot be read from inside the process): scripts/r5_profiles.sh -> profiles/*.json, keyed by
        # the kernel-source hash so that a stale summary is never quoted
        pmc = pmc_fields(name if (args.fsize is None or not headline) else "", mode, fsize, args.chunk,
                         dom["ms"] / max(dom["launches"], 1))
        all_s = prof["conv_ms"] / 1e3
        line = {
            "metric": f"Mbp/s classified ({fsize}bp frags)", "value": round(value, 3), "unit": "Mbp/s",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(dt_max / n_steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if mode == "f32" else "f16x3 (split-f16, f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": f"{wl['label']}, {fsize}bp windows stride {fsize}, {n_contigs} synthetic "
                                   + (f"fragments/GPU of exactly {fsize} bp" if wl["exact"] else
                                      "contigs/GPU log-uniform 1.5-200 kb"),
                       "name": name,
                       "windows_per_gpu": int(win_total / world), "bp_per_gpu": int(bp_total / world),
                       "windows_per_gpu_min": int(per_rank[:, 2].min()), "windows_per_gpu_max": int(per_rank[:, 2].max()),
                       "parallelism": f"contig-sharded x{world}, final {'gloo (test mode)' if args.oversubscribe and world > 1 else 'RCCL'} gather"
                                      + (" (executed)" if coll else " (one rank: no exchange)"),
                       "collective_backend": (dist.get_backend() if coll else None),
                       "outputs": "prediction + reliability + G/C/A/T counts per window stay in HBM (the logits are "
                                  "gathered); embedding / nmd vectors (InferModel.predict also returns them, 2.6 kB "
                                  "per window) are computed but not copied out in the timed region"},
            "roofline": {"bound": "mfma", "achieved": round(ach, 3), "peak": round(peak, 1),
                         "unit": "TFLOP/s", "frac": round(ach / peak, 4) if peak else 0.0, "traffic": pmc["traffic"],
                         "traffic_what": ("HBM bytes per full-size launch from rocprofv3 --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE, "
                                          "profiles/pmc_traffic.json) of THIS kernel build collected on ANOTHER device - a "
                                          "byte count, the same on every device; hbm_gbs divides it by this run's launch time"
                                          if pmc["traffic"] else None),
                         "hbm_gbs": pmc["hbm_gbs"], "kernel_hash": kernel_hash(mode),
                         "pmc_kernel_hash": pmc["pmc_kernel_hash"], "pmc_stale": pmc["pmc_stale"],
                         "pmc_other_device": pmc["other_device"],
                         "kernel": dom_name,
                         "launches": int(dom["launches"]),
                         "avg_launch_ms": round(dom["ms"] / max(dom["launches"], 1), 4),
                         "hip_events_in_timed_region": not args.no_profile,
                         "all_convs_incl_table_kernel": {
                             "achieved": round(prof["conv_flops"] / all_s / 1e12, 3) if all_s > 0 else 0.0,
                             "frac": round(prof["conv_flops"] / all_s / 1e12 / peak, 4) if all_s > 0 else 0.0,
                             "launches": int(prof["conv_launches"])},
                         "table_kernel": {"launches": int(prof["table"]["launches"]),
                                          "avg_launch_ms": round(prof["table"]["ms"] / max(prof["table"]["launches"], 1), 4)}},
        }
        # per-rank split of the timed steps: a weak-scaling efficiency below target is imbalance (compute spread between
        # ranks) or communication (the gather's share on the slowest rank) - readable from this line alone
        slow = int(per_rank[:, 0].argmax())
        line["per_rank"] = {
            "slowest_rank": slow, "ms_per_step_slowest": round(float(per_rank[slow, 0]) / n_steps * 1e3, 3),
            "ms_per_step_fastest": round(float(per_rank[:, 0].min()) / n_steps * 1e3, 3),
            "compute_ms_per_step": {"min": round(float(per_rank[:, 3].min()) / n_steps * 1e3, 3),
                                    "max": round(float(per_rank[:, 3].max()) / n_steps * 1e3, 3)},
            "gather_ms_per_step": {"min": round(float(per_rank[:, 4].min()) / n_steps * 1e3, 3),
                                   "max": round(float(per_rank[:, 4].max()) / n_steps * 1e3, 3),
                                   "what": "final gather of the logits incl. waiting for the slowest rank; 0 with one GPU"},
            "windows": [int(v) for v in per_rank[:, 2]]}
        if prof["fused_small"]["launches"]:
            fs = prof["fused_small"]
            line["roofline"]["fused_small_kernel"] = {
                "launches": int(fs["launches"]), "avg_launch_ms": round(fs["ms"] / fs["launches"], 4),
                "achieved": round(fs["flops"] / (fs["ms"] / 1e3) / 1e12, 3)}
        if box is not None:
            # the headline relative to what THIS device's matrix cores deliver under load: comparable across boxes / rounds
            box["value_per_box_tflop"] = round(value / box["mfma_loop_tflops"], 5) if box["mfma_loop_tflops"] else None
            box["dominant_kernel_issued_f16_tflops_over_box"] = (
                round(3.0 * ach / box["mfma_loop_tflops"], 4) if (mode == "f16x3" and box["mfma_loop_tflops"]) else None)
            line["box"] = box
        if world == 1 and headline and not args.no_exact_f32 and mode == "f16x3" and args.timed_dbg is None:
            # the exact-f32 MFMA arithmetic on a bounded sample of the same windows (about 2 s)
            n_s = int(min(n_win, max(256, 40e6 // fsize)))
            eng.model.set_precision("f32")
            step(n_s)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step(n_s)
            torch.cuda.synchronize()
            line["exact_f32_mbps"] = round(float(win_len[:n_s].sum()) / (time.perf_counter() - t1) / 1e6, 2)
            eng.model.set_precision("f16x3")
    flops_per_window = eng.model.flops_per_window(l_pad)
    eng.close()
    del d_bases, d_start, d_len, d_pred, d_rel, d_counts
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and args.timed_dbg is None:
        if not args.no_e2e and (headline or name != "pyramid"):      # (pyramid rides along as a kernel figure only)
            try:
                if headline or wl["model"] != CONFIGS[args.config]["model"]:
                    line["e2e"] = e2e_leg(cfg, weights, wl, fsize)
                else:
                    line["e2e"] = "same model and file as the headline's e2e"
                if not headline and wl["exact"]:
                    # the fragment workload itself as a file: a record per window
                    line["e2e_records"] = e2e_leg(cfg, weights, wl, fsize, records=n_contigs)
            except SystemExit as e:               # run_core exits on its own errors: report, do not lose the bench line
                line["e2e"] = {"error": f"run_core exited with {e.code}"}
        if headline and not args.no_also:
            # the other BASELINE configs in the same driver-run line (2 timed steps each): configs[2]'s per-GPU shard
            # and configs[3] - and the reference's pyramid ResNet (not a BASELINE config: the width- / stride-general
            # kernels' figure); a failure there must not lose the headline
            line["also"] = {}
            for other in ("default", "frag1m", "baseline500", "pyramid"):
                if other == name:
                    continue
                try:
                    sub = measure(args, other, 2, 1, ctx, headline=False)
                    keep = {k: sub[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "dtype") if k in sub}
                    keep["workload"] = sub["config"]["workload"]
                    keep["roofline"] = {k: sub["roofline"][k] for k in
                                        ("kernel", "achieved", "peak", "frac", "avg_launch_ms", "launches", "traffic",
                                         "pmc_stale")}
                    for k in ("e2e", "e2e_records"):
                        if k in sub:
                            keep[k] = sub[k]
                    line["also"][other] = keep
                except (Exception, SystemExit) as e:   # noqa: BLE001
                    line["also"][other] = {"error": f"{type(e).__name__}: {e}"}
        if headline:
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(cfg, weights, bases, offsets, fsize, flops_per_window)
            else:
                line["cpu_baseline"] = None
    elif rank == 0 and headline:
        line["cpu_baseline"] = None
    return line


if __name__ == "__main__":
    main()
