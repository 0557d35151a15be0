"""The torchrun path of ``run_core`` on CPU: world_size 2 over gloo with a stand-in engine (deterministic logits
computed from the window bases on the host - the GPU engine itself is covered by the -m gpu tests).  Contig
sharding (LPT), the object gather, the restoration of the reference's emission order (FASTA order, long pass before
short pass) and the TSV must equal a single-process run byte for byte.  Under torchrun rank 0 indexes the FASTA,
every rank reads / masks / classifies only its own contigs (asserted: no rank holds the whole base buffer), one
gather of f32 rows + one of the int32 repeat table; duplicate record names survive (order is by record index)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import ROOT, make_model_dir


class _StandInModel:
    precision = "stand-in"

    def __init__(self, n_classes):
        self.n_classes = n_classes

    def host_outputs(self, n_win, want=("prediction", "reliability"), counts=True):
        out = {"prediction": np.zeros((n_win, self.n_classes), np.float32), "reliability": np.zeros((n_win, 1), np.float32)}
        out = {k: v for k, v in out.items() if k in want}
        if counts:
            out["counts"] = np.zeros((n_win, 4), np.int32)
        return out


class _StandInDevice:
    """JG_STAT_WINDOWS_DONE of the stand-in: rows below the mark are final while predict_windows is still running."""
    done = 0

    def windows_done(self):
        return self.done


class _StandInEngine:
    """Same surface as JaegerHipEngine for run_core: class_map, string_processor_config, model.host_outputs,
    predict_windows(out=...), device.windows_done."""

    def __init__(self, path_dict=None, **kw):
        import yaml
        classes = yaml.safe_load(Path(path_dict["classes"]).read_text())["classes"]
        self.class_map = {"num_classes": len(classes), "class": [c["class"] for c in classes],
                          "index": [c["label"] for c in classes]}
        self.string_processor_config = {"crop_size_codons": None, "crop_size_nt": None}
        self.model = _StandInModel(len(classes))
        self.device = _StandInDevice()

    dust_masked_total = 0

    def predict_windows(self, bases, win_start, win_len, fsize, l_pad=None, pre_cased=False,
                        want=("prediction", "reliability"), dust_records=None, out=None):
        import time
        if dust_records is not None:
            # what jg_engine_set_dust does on the device: DUST on a COPY of the bases by their record table, then the
            # encoder respects the case (the host scan gives the same masks bit for bit: tests/test_gpu_dust.py)
            from jaeger_amd import fragment as frag
            off = np.asarray(dust_records, np.int64)
            assert off[0] >= 0 and off[-1] <= len(bases) and (np.diff(off) >= 0).all()
            tmp = frag.FastaBatch([""] * (len(off) - 1), np.array(bases, np.uint8, copy=True), off)
            self.dust_masked_total += frag.dust_mask(tmp, threads=1)
            bases, pre_cased = tmp.bases, True
        n = len(win_start)
        res = out if out is not None else self.model.host_outputs(n, want)
        pred, rel, counts = res["prediction"], res["reliability"], res["counts"]
        self.device.done = 0
        for i, (s, ln) in enumerate(zip(np.asarray(win_start).tolist(), np.asarray(win_len).tolist())):
            w = np.asarray(bases[s:s + ln])
            up = w & 0xDF if not pre_cased else w
            c = np.array([(up == ord(ch)).sum() for ch in "GCAT"], np.int32)          # meta_5..8 order: G C A T
            counts[i] = c
            h = np.cumsum(w.astype(np.int64) * (np.arange(ln) % 7 + 1))[-1] if ln else 0
            pred[i] = [((h >> (3 * k)) % 97) / 9.0 - 5.0 for k in range(pred.shape[1])]
            rel[i, 0] = ((h >> 5) % 31) / 5.0 - 3.0
            if out is not None and i % 3 == 2:           # progress in steps of three windows, slow enough to be polled
                self.device.done = i + 1
                time.sleep(0.01)
        self.device.done = n
        return res

    def close(self):
        pass


def _write_fasta(path, seed=5):
    rng = np.random.Generator(np.random.PCG64(seed))
    lens = [5200, 700, 1499, 3100, 650, 1203, 400, 1500, 999, 2999, 12000, 1800, 4700, 800, 6100]
    with open(path, "w") as fh:
        for i, n in enumerate(lens):
            seq = "".join(rng.choice(list("ACGT"), n))
            if i == 3:
                seq = seq[:500] + "AT" * 60 + seq[620:]          # a low-complexity stretch for DUST
            fh.write(f">{'ctg_dup' if i in (9, 11) else f'ctg_{i}'} len={n}\n")     # two records share a name
            for j in range(0, n, 60):
                fh.write(seq[j:j + 60] + "\n")
    return lens


def _run(out_dir, fasta, model_root, min_len, no_pipeline=True, dust_host=False):
    import pandas as pd

    import jaeger_amd.engine as E
    import jaeger_amd.termini as T
    from jaeger_amd.predict import run_core
    E.JaegerHipEngine = _StandInEngine
    E.HipDevice = lambda *a, **k: type("SideStream", (), {"close": lambda self: None})()
    T.scan_for_terminal_repeats = lambda device, fa, fsize: pd.DataFrame(
        {"contig_id": [n.strip().replace(",", "___") for n, ln in zip(fa.names, fa.lengths.tolist()) if ln >= fsize],
         "terminal_repeats": None, "repeat_length": np.nan})
    T.terminal_repeat_table = lambda device, fa, fsize, report_min=0: np.where(
        (fa.lengths >= fsize)[:, None], np.zeros((len(fa), 10), np.int32), np.int32(-1)).astype(np.int32)
    return run_core(input=str(fasta), output=str(out_dir), model_path=str(model_root), fsize=1500, stride=1500,
                    min_len=min_len, batch=2, dustmask=True, rc=0.1, pc=1, overwrite=True, verbose=1,
                    no_pipeline=no_pipeline, prophage=True, lc=4000, dust_host=dust_host)


def _worker(rank, world, port, tmp, min_len):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _run(Path(tmp) / "sharded", Path(tmp) / "in.fasta", Path(tmp) / "m", min_len)
    import json

    import jaeger_amd.predict as P
    (Path(tmp) / f"shard_stats_{rank}.json").write_text(json.dumps(P.SHARD_STATS))
    dist.destroy_process_group()


@pytest.mark.parametrize("min_len", [None, 600])
def test_sharded_run_core_equals_single_process(tmp_path, min_len, monkeypatch):
    import torch.multiprocessing as mp
    _write_fasta(tmp_path / "in.fasta")
    make_model_dir(tmp_path / "m")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    import jaeger_amd.engine as E
    import jaeger_amd.termini as T
    keep = (E.JaegerHipEngine, E.HipDevice, T.scan_for_terminal_repeats, T.terminal_repeat_table)
    try:
        n_single = _run(tmp_path / "single", tmp_path / "in.fasta", tmp_path / "m", min_len)
        # the single-GPU host pipeline (worker thread owns the engine; the calling thread aggregates finished contigs in
        # batches beside the forward - here every two windows or more)
        import jaeger_amd.predict as P
        real_agg = P._Aggregator
        batches = []

        class SmallBatches(real_agg):
            def __init__(self, *a, **k):
                k["min_batch"] = 2
                super().__init__(*a, **k)
                batches.append(self)

        monkeypatch.setattr(P, "_Aggregator", SmallBatches)
        n_piped = _run(tmp_path / "piped", tmp_path / "in.fasta", tmp_path / "m", min_len, no_pipeline=False)
        monkeypatch.setattr(P, "_Aggregator", real_agg)
        assert len(batches) == 1 and len(batches[0].parts) >= 3        # aggregation really ran in several batches
        # --dust-host: the host pass over the FASTA image instead of DUST inside the fused calls (record tables of the
        # whole-buffer and of the compacted short-contig batches): same TSV
        n_host = _run(tmp_path / "hostdust", tmp_path / "in.fasta", tmp_path / "m", min_len, dust_host=True)
    finally:
        E.JaegerHipEngine, E.HipDevice, T.scan_for_terminal_repeats, T.terminal_repeat_table = keep
    assert n_piped == n_single == n_host
    assert (tmp_path / "hostdust" / "38341_1.4M" / "in.tsv").read_text() == \
        (tmp_path / "single" / "38341_1.4M" / "in.tsv").read_text()
    assert (tmp_path / "piped" / "38341_1.4M" / "in.tsv").read_text() == \
        (tmp_path / "single" / "38341_1.4M" / "in.tsv").read_text()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path), min_len), nprocs=2, join=True)
    single = (tmp_path / "single" / "38341_1.4M" / "in.tsv").read_text()
    sharded = (tmp_path / "sharded" / "38341_1.4M" / "in.tsv").read_text()
    assert n_single > 0 and single == sharded
    # -p: the segmentation-input frames of the contigs >= --lc, identical from one process and from two ranks
    seg = [np.load(tmp_path / d / "38341_1.4M" / "in_prophages" / "in_segmentation_inputs.npz", allow_pickle=True)
           for d in ("single", "sharded")]
    assert list(seg[0]["contigs"]) == list(seg[1]["contigs"]) == ["ctg_0", "ctg_10", "ctg_12", "ctg_14"]
    for a, b in zip(seg[0]["tracks"], seg[1]["tracks"]):
        np.testing.assert_array_equal(a, b)
    ids = [ln.split("\t")[0] for ln in single.splitlines()[1:]]
    long_first = [f"ctg_{i}" for i in (0, 3, 7, 10, 12, 14)]
    assert ids.count("ctg_dup") >= 2                                # both records that share a name are reported
    ids = [i for i in ids if i != "ctg_dup"]
    # per-rank ingest: each rank parsed only its own contigs, together they cover the used ones exactly once
    import json
    stats = [json.loads((tmp_path / f"shard_stats_{r}.json").read_text()) for r in range(2)]
    total = stats[0]["total_bases"]
    assert all(0 < st["local_bases"] < total for st in stats), stats
    assert sum(st["local_bases"] for st in stats) <= total and stats[0]["total_records"] == 15
    assert ids[:len(long_first)] == long_first                      # FASTA order, long pass first
    if min_len is not None:
        assert set(ids[len(long_first):]) <= {"ctg_1", "ctg_2", "ctg_4", "ctg_5", "ctg_8", "ctg_13"}
