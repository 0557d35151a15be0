import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def _cpu_quota() -> int:
    """Cores this container may actually use (cgroup CPU quota; a GPU box shows 256 cores and grants 16)."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, per = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    # the CPU oracle (torch) otherwise starts one thread per VISIBLE core and the threads preempt each other
    try:
        import torch
        torch.set_num_threads(max(1, min(_cpu_quota(), 32)))
    except ImportError:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_model_cfg(name: str) -> dict:
    import yaml
    return yaml.safe_load((GOLDEN / f"{name}_project.yaml").read_text())["model"]


def make_model_dir(root: Path, name: str = "brain", model_name: str = "jaeger_38341_1.4M_fragment",
                   seed: int = 38341) -> Path:
    """A model directory as ``AvailableModels`` expects it (utils/misc.py:346-392), with seeded
    stand-in weights in the engine's canonical npz form."""
    import shutil

    import yaml

    from jaeger_amd.plan import build_plan
    from jaeger_amd.weights import random_weights, save_npz
    d = Path(root) / "model"
    d.mkdir(parents=True, exist_ok=True)
    shutil.copyfile(GOLDEN / f"{name}_project.yaml", d / f"{model_name}_project.yaml")
    cfg = load_model_cfg(name)
    (d / f"{model_name}_classes.yaml").write_text(yaml.safe_dump({"classes": cfg["class_label_map"]}))
    save_npz(d / f"{model_name}.weights.npz", random_weights(build_plan(cfg), seed))
    return Path(root)


def oracle_term_repeats(records, fsize: int):
    """Terminal-repeat table (contig_id, terminal_repeats, repeat_length) from the CPU oracle for (name, seq)
    records; what ``scan_for_terminal_repeats`` feeds into the summaries (utils/termini.py:88-189)."""
    import numpy as np
    import pandas as pd

    from oracle import termini as ot
    rows = []
    for name, seq in records:
        if len(seq) < fsize:
            continue
        kind, length = ot.scan_record(seq)
        rows.append({"contig_id": name.strip().replace(",", "___"), "terminal_repeats": kind,
                     "repeat_length": np.nan if length is None else float(length)})
    return pd.DataFrame(rows, columns=["contig_id", "terminal_repeats", "repeat_length"])
