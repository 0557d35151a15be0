"""``jaeger predict`` orchestration on the MI355X engine.

Mirrors ``commands/predict.py:488-861`` (``run_core``) for the conv model family: model
discovery, FASTA validation, output layout ``<output>/<model_id>/<stem>.tsv``, two-pass
short-contig mode, auxiliary npz writers - with the tf.data pipeline
(``_build_prediction_dataset`` :186-245) and ``InferModel.predict`` replaced by one window
table + ``JaegerHipEngine.predict_windows`` call per pass.  Under ``torchrun`` (WORLD_SIZE > 1)
contigs are sharded over the ranks and gathered on rank 0 (RCCL).
"""

from __future__ import annotations

import json
import logging
import os
import sys
import threading
import time
import traceback
from collections import defaultdict
from pathlib import Path
from typing import Any

import numpy as np

from . import fragment as frag

logger = logging.getLogger("Jaeger")


# ---- model registry (utils/misc.py:334-396) ------------------------------------------------
class AvailableModels:
    """Scan ``model/`` directories for ``<name>_graph/``, ``<name>_classes.yaml``,
    ``<name>_project.yaml`` and ``<name>.weights.h5`` / ``<name>.npz`` entries."""

    def __init__(self, path):
        self.paths = [Path(path)] if isinstance(path, (str, Path)) else [Path(p) for p in path]
        self.info = self._scan()

    def _scan(self):
        models: dict[str, dict] = defaultdict(dict)
        for path in self.paths:
            dirs = [p for p in path.rglob("model") if p.is_dir()]
            if path.name == "model" and path.is_dir():
                dirs.append(path)
            for d in dirs:
                for e in d.iterdir():
                    if e.is_dir() and e.name.endswith("_graph"):
                        models[e.name.removesuffix("_graph")]["graph"] = e
                    elif e.is_file():
                        name = e.name
                        if "_classes.yaml" in name:
                            key = "classes"
                        elif "_project.yaml" in name:
                            key = "project"
                        elif e.suffix == ".h5" and ".weights" in e.stem:
                            key = "weights"
                        elif e.suffix == ".npz" and ".weights" in e.stem:
                            key = "weights_npz"           # canonical weights for the MI355X engine
                        else:
                            continue
                        base = (name.replace("_classes.yaml", "").replace("_project.yaml", "")
                                .replace(".weights.h5", "").replace(".weights.npz", ""))
                        models[base][key] = e
        return models


def get_model_id(model: str) -> str:
    """utils/misc.py:395-396: ``jaeger_38341_1.4M_fragment`` -> ``38341_1.4M``."""
    return model.split("_", 1)[1].rsplit("_", 1)[0]


def validate_fasta_entries(path, min_len: int) -> int:
    """utils/fs.py:99-115: record count; raises when no record reaches ``min_len``.  ``path`` may be
    an already loaded :class:`~jaeger_amd.fragment.FastaBatch` (one ingest pass instead of three)."""
    fa = path if isinstance(path, frag.FastaBatch) else frag.load_fasta(path)
    num, ok = len(fa), int((fa.lengths >= min_len).sum())
    logger.info(f"{ok}/{num} entries in {'input' if isinstance(path, frag.FastaBatch) else path}")
    if ok == 0:
        raise Exception(f"all records in {'input' if isinstance(path, frag.FastaBatch) else path} are < {min_len}bp")
    return num


def get_logger(out_dir: Path, log_file: Path, level: int = 1) -> logging.Logger:
    """utils/logging.py:30-75: console + DEBUG file handler."""
    lg = logging.getLogger("Jaeger")
    lg.setLevel(logging.DEBUG)
    for h in list(lg.handlers):
        lg.removeHandler(h)
    ch = logging.StreamHandler(sys.stderr)
    ch.setLevel(logging.DEBUG if level and level > 1 else logging.INFO)
    ch.setFormatter(logging.Formatter("%(asctime)s %(levelname)-8s [jaeger] %(message)s", "%Y-%m-%d %H:%M:%S"))
    lg.addHandler(ch)
    fh = logging.FileHandler(out_dir / f"{time.strftime('%m%d%Y_%H%M%S')}_{log_file}")
    fh.setLevel(logging.DEBUG)
    fh.setFormatter(ch.formatter)
    lg.addHandler(fh)
    return lg


def _crop_length_warning(trained_codons, trained_nt, fsize: int) -> str | None:
    """commands/predict.py:36-63."""
    if trained_codons is not None:
        runtime = (int(fsize) - 5) // 3
        if runtime == trained_codons:
            return None
        hint = f" ({trained_nt} nt)" if trained_nt is not None else ""
        return (f"runtime --fsize {fsize} maps to {runtime} codon frames, but the model was trained on "
                f"{trained_codons} codons{hint}; prefer --fsize "
                f"{trained_nt if trained_nt is not None else 'used at training'} for this model.")
    if trained_nt is not None and int(fsize) != int(trained_nt):
        return (f"runtime --fsize {fsize} differs from the model's trained fragment length ({trained_nt} nt).")
    return None


def _concat_predictions(a: dict, b: dict) -> dict:
    """commands/predict.py:248-258."""
    if not a:
        return b
    if not b:
        return a
    return {k: np.concatenate([a[k], b[k]], axis=0) for k in a}


# ---- one prediction pass --------------------------------------------------------------------
def predict_batch(engine, fa: "frag.FastaBatch", fsize: int, stride: int | None, min_len: int | None = None,
                  max_len: int | None = None, dynamic_stride: bool = False,
                  dynamic_stride_threshold: float = 10.0, batch: int = 96, padded: bool = False,
                  subset=None, pre_cased: bool = False,
                  want=("prediction", "reliability", "embedding", "nmd"), meta: bool = True,
                  dust_device: bool = False) -> dict[str, np.ndarray]:
    """Window table + GPU encode/forward for the records of ``fa`` (optionally only those listed in
    ``subset``, kept in that order); returns the dict ``InferModel.predict`` would (model outputs +
    ``meta_0..9``).  ``padded`` reproduces ``padded_batch`` of the short-contig pass: windows run in
    groups of ``batch`` padded to the longest frame of the group (commands/predict.py:236-245).  ``dust_device``: the
    bases are DUST soft-masked on the GPU after their upload (the host buffer stays as read) instead of ``pre_cased``
    by :func:`jaeger_amd.fragment.dust_mask`."""
    idx = np.arange(len(fa)) if subset is None else np.asarray(subset, np.int64)
    lengths = fa.lengths[idx]
    names = [fa.names[i] for i in idx.tolist()] if subset is not None else fa.names
    table = frag.build_window_table(lengths, fsize, stride, dynamic_stride, dynamic_stride_threshold,
                                    min_len, max_len)
    if len(table) == 0:
        return {}
    starts = fa.offsets[idx][table.contig] + table.start
    if not padded:
        out = engine.predict_windows(fa.bases, starts, table.length, fsize, pre_cased=pre_cased, want=want,
                                     dust_records=fa.offsets if dust_device else None)
    else:
        # the short-contig pass: one whole-contig window per record.  The selected records are compacted into a
        # buffer of their own once, so that a batch of 96 windows uploads the few hundred kB it covers and not the
        # whole FASTA image (a 2 Gbp assembly with 1 M short contigs would otherwise move 2 GB per batch)
        # (compacted batch by batch, run by run: records are in FASTA order, so adjacent short records merge into one
        # slice copy - no per-base index array over every short contig of the assembly)
        wl = table.length.astype(np.int64)
        cstart = np.cumsum(wl) - wl
        off3 = (-2, -1, 0)[fsize % 3]
        parts = []
        for i in range(0, len(table), batch):
            sl = slice(i, i + batch)
            lmax = int(max(0, -(-(int(table.length[sl].max()) - 5 + off3) // 3)))
            if getattr(engine.model, "strands", 1) > 1:                       # nucleotide rows hold bases
                lmax = int(table.length[sl].max())
            elif getattr(engine.model, "wide_ids", False):                    # dicodon rows: codon pairs six bases apart
                lmax = int(max(0, -(-(int(table.length[sl].max()) - 8 + off3) // 6)))
            c0, c1 = int(cstart[i]), int(cstart[sl][-1] + wl[sl][-1])
            s_b, l_b = starts[sl], wl[sl]
            cut = np.nonzero(s_b[1:] != s_b[:-1] + l_b[:-1])[0] + 1          # where a new run of adjacent records starts
            run_a = s_b[np.concatenate(([0], cut))]
            run_b = (s_b + l_b)[np.concatenate((cut - 1, [len(s_b) - 1]))]
            compact = np.concatenate([fa.bases[x:y] for x, y in zip(run_a.tolist(), run_b.tolist())]) \
                if len(run_a) > 1 else fa.bases[int(run_a[0]):int(run_b[0])]
            # (every window of this pass is a whole record: the compact buffer's record table is its window table)
            recs = np.append(cstart[sl] - c0, c1 - c0) if dust_device else None
            parts.append(engine.predict_windows(compact, cstart[sl] - c0, table.length[sl], fsize,
                                                l_pad=max(lmax, 1), pre_cased=pre_cased, want=want, dust_records=recs))
        out = {k: np.concatenate([p[k] for p in parts], axis=0) for k in parts[0]}
    if not meta:                       # sharded runs: rank 0 rebuilds the metadata from its own window table
        return out
    counts = out.pop("counts")
    out.update(frag.window_metadata(table, names, counts))
    return out


def predict_records(engine, names: list[str], seqs: list[bytes], fsize: int, stride: int | None, **kw):
    """:func:`predict_batch` for records given as Python lists."""
    bases, offsets = frag.concat_records(seqs)
    return predict_batch(engine, frag.FastaBatch(list(names), bases, offsets), fsize, stride, **kw)


def _shard_records(lengths: np.ndarray, fsize: int, stride: int | None, world: int):
    """Whole contigs to ranks, balanced by window count (LPT); returns one index list per rank."""
    from .dist import lpt_partition
    step = fsize if stride is None else stride
    w = np.where(lengths >= fsize, np.maximum(1, (lengths - fsize) // step + 1), 1)
    return lpt_partition(w, world)


class _LazyTableWriter:
    """``postprocess.TableWriter`` created at its first batch: pandas is then imported beside the forward, not in front of it."""

    def __init__(self, class_map: dict, table_path, phage_path, reliability_cutoff, phage_score):
        self._args = (class_map.get("class"), class_map.get("index"), table_path, phage_path)
        self._kw = dict(reliability_cutoff=reliability_cutoff, phage_score=phage_score)
        self.w = None

    def append(self, data: dict) -> None:
        if self.w is None:
            from .postprocess import TableWriter
            self.w = TableWriter(*self._args, **self._kw)
        self.w.append(data)

    def close(self) -> int:
        return self.w.close() if self.w is not None else 0

    def abort(self) -> None:
        if self.w is not None:
            self.w.abort()


class _Aggregator:
    """Per-contig aggregation of the long pass BESIDE the forward: ``advance(done)`` is called with the engine's progress
    mark (output rows of windows [0, done) are final, ``HipDevice.windows_done``) and runs ``pred_to_dict`` on the contigs
    whose windows are all below it, batch by batch; the reference aggregates after ``InferModel.predict`` has returned
    (commands/predict.py:846).  Per-contig statistics do not depend on which other contigs share a batch, so the merged
    result equals one ``pred_to_dict`` call over everything (tests/test_postprocess.py)."""

    def __init__(self, table: "frag.WindowTable", names: list[str], out: dict, pred_kw: dict, min_batch: int, workers: int = 1):
        self.table, self.out, self.kw = table, out, pred_kw
        # workers > 1 (round 6): batches are aggregated on a small pool - ``parts`` then holds futures, in batch order - so
        # that on files of very many short contigs the aggregation keeps pace with the forward (numpy and the native
        # reductions run without the interpreter lock); consumers take the batches in order whatever finished first
        self.pool = None
        if workers > 1:
            from concurrent.futures import ThreadPoolExecutor
            self.pool = ThreadPoolExecutor(max_workers=int(workers), thread_name_prefix="jaeger-agg")
        # names straight from the native parser (fragment.Names) that io.py:109's normalisation leaves as they are stay BYTES
        # from the FASTA image to the table rows; anything else goes through the per-record strings as before
        self._names = names
        self.names_bytes = names if isinstance(names, frag.Names) and names.plain() else None
        self._hdr = None
        n_rec = len(names)
        per_rec = np.bincount(table.contig, minlength=n_rec).astype(np.int64) if len(table) else np.zeros(n_rec, np.int64)
        self.ends = np.cumsum(per_rec[per_rec > 0])              # window index one past each contig's last window
        self.w_done = 0                                           # windows aggregated so far (always a contig boundary)
        self.min_batch = max(1, int(min_batch))
        self.parts: list = []
        self.busy_s = 0.0
        self.flushed = 0                                          # parts already handed to the table writer
        self._unique = None

    @property
    def hdr(self) -> np.ndarray:
        """The records' normalised names as an object array of Python strings (made when something needs them)."""
        if self._hdr is None:
            self._hdr = frag.normalise_headers(self._names)
        return self._hdr

    def advance(self, done: int, final: bool = False) -> None:
        if len(self.ends) == 0:
            return
        k = int(np.searchsorted(self.ends, done, side="right"))
        w1 = int(self.ends[k - 1]) if k else 0
        if w1 <= self.w_done or (not final and w1 - self.w_done < self.min_batch):
            return
        w0, self.w_done = self.w_done, w1
        if self.pool is not None:
            self.parts.append(self.pool.submit(self._batch, w0, w1))
        else:
            part = self._batch(w0, w1)
            if part is not None:
                self.parts.append(part)

    def _batch(self, w0: int, w1: int):
        """(data, data_full) of the contigs whose windows are [w0, w1), or None."""
        t0 = time.time()
        rec = self.table.contig[w0:w1]
        first = np.ones(len(rec), dtype=bool)
        first[1:] = rec[1:] != rec[:-1]
        part = self._aggregate(self.slice(w0, w1, with_names=self.names_bytes is None), np.asarray(rec[first], dtype=np.int64))
        self.busy_s += time.time() - t0
        return part

    def _part(self, i: int, wait: bool = True):
        """Batch ``i`` as (data, data_full) (None: an empty one); with ``wait`` False, the string "pending" while a worker
        still has it."""
        item = self.parts[i]
        if hasattr(item, "result"):
            if not wait and not item.done():
                return "pending"
            item = self.parts[i] = item.result()          # (a worker's exception surfaces here, in the consumer)
        return item

    def close(self) -> None:
        if self.pool is not None:
            self.pool.shutdown(wait=True)
            self.pool = None

    def names_unique(self) -> bool:
        """No record name twice (then a join by record number IS the reference's merge on the name).  One hash pass over
        the names, made once - the polling loop calls it beside the forward."""
        if self._unique is None:
            if self.names_bytes is not None:
                self._unique = self.names_bytes.is_unique()          # (jg_names_unique: no Python strings)
            else:
                import pandas as pd
                self._unique = bool(pd.Index(self.hdr).is_unique)
        return self._unique

    def slice(self, w0: int, w1: int, with_names: bool = True) -> dict:
        """Engine outputs + window metadata of windows [w0, w1) in ``InferModel.predict``'s dict form (``with_names`` False:
        without the per-window header array ``meta_0`` - the batch's contig names travel as bytes)."""
        t = self.table
        sub = frag.WindowTable(*(getattr(t, f)[w0:w1] for f in ("contig", "start", "length", "is_last", "ordinal", "seqlen")))
        y = {k: v[w0:w1] for k, v in self.out.items() if k != "counts"}
        y.update(frag.window_metadata(sub, self.hdr if with_names else None, self.out["counts"][w0:w1], normalised=True))
        return y

    def add(self, y_pred: dict, records: np.ndarray | None = None) -> None:
        part = self._aggregate(y_pred, records)
        if part is not None:
            self.parts.append(part)

    def _aggregate(self, y_pred: dict, records: np.ndarray | None = None):
        from .postprocess import SpanColumn, _Summaries, pred_to_dict, window_letters
        if y_pred and len(y_pred["meta_2"]):
            kw = self.kw
            if y_pred.get("meta_0") is None:                       # names as bytes: one span per contig of the batch
                kw = dict(kw, headers=SpanColumn(*self.names_bytes.spans(records), facts=self.names_bytes.facts))
            data, full = pred_to_dict(y_pred, **kw)
            # the run-length strings of the batch's contigs belong beside the forward too: the library's text as it comes
            runs, letters = data["frag_pred"], window_letters(self.kw["class_map"])
            blob = runs.summaries_blob(letters)
            data["frag_pred"] = _Summaries(blob=blob, n=len(runs)) if blob is not None else _Summaries(runs.summaries(letters))
            if records is not None and len(records) == len(data["headers"]):
                data["record_index"] = records        # FASTA record of every contig of the batch: the repeat table joins by it
            return data, full
        return None

    def flush(self, writer, term_repeats, wait: bool = True) -> None:
        """Hand the batches aggregated so far to the table writer, in order (needs the repeat table: the left merge of
        collect.py:527-532 is part of a row).  ``wait`` False: stop at the first batch a worker has not finished."""
        t0 = time.time()
        while self.flushed < len(self.parts):
            part = self._part(self.flushed, wait)
            if part == "pending":
                break
            if part is None:
                self.flushed += 1
                continue
            data = part[0]
            data["repeats"] = term_repeats
            row_of = getattr(term_repeats, "row_of_record", None)
            if row_of is None and term_repeats is not None and hasattr(term_repeats, "attrs"):
                row_of = term_repeats.attrs.get("_row_of_record")
            if row_of is not None and "record_index" in data and len(row_of) == len(self._names):
                data["repeat_rows"] = row_of[data["record_index"]]
                data["names_unique"] = self.names_unique()
            writer.append(data)
            self.flushed += 1
        self.busy_s += time.time() - t0

    def result_full(self):
        from .postprocess import merge_data
        parts = [self._part(i) for i in range(len(self.parts))]
        return merge_data([p[1] for p in parts if p is not None])


# ---- torchrun: contig-sharded prediction, one gather of f32 rows ---------------------------------------------
SHARD_STATS: dict = {}        # filled by the sharded path (tests read it): local / total bases of this rank
LAST_RUN: dict = {}           # stage split of the last single-GPU run_core of this process (bench.py's e2e leg reads it)


_PENDING_RELEASE: list = []      # threads still releasing a finished run's engine / host buffers


def wait_for_release(timeout: float | None = 60.0) -> None:
    """Join the background release of earlier ``run_core`` calls (device memory and pinned staging are back when this
    returns).  ``run_core`` calls it on entry, so two runs never overlap one's teardown with the other's forward."""
    while _PENDING_RELEASE:
        _PENDING_RELEASE.pop().join(timeout)


def _close_process_group():
    """Destroy the process group if ``_predict_sharded`` created it (a caller's own group is left alone)."""
    if SHARD_STATS.pop("own_process_group", False):
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


def _coll_device(local_rank: int):
    import torch
    import torch.distributed as dist
    return torch.device("cuda", local_rank) if dist.get_backend() == "nccl" else torch.device("cpu")


def _bcast(arr: np.ndarray | None, dtype, n: int, dev):
    """Broadcast a 1-D array from rank 0 (``arr`` is None elsewhere)."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(arr, dtype)).to(dev) if arr is not None else \
        torch.zeros(n, dtype=torch.from_numpy(np.zeros(0, dtype)).dtype, device=dev)
    dist.broadcast(t, src=0)
    return t.cpu().numpy()


def _place_rows(out: np.ndarray, part: np.ndarray, items: np.ndarray, rows_per_item: np.ndarray, first: np.ndarray):
    """``part`` holds the rows of ``items`` back to back (``rows_per_item[i]`` each); write them to their global
    positions ``first[i]...`` (vectorised :func:`jaeger_amd.dist.restore_order`)."""
    n = rows_per_item[items]
    total = int(n.sum())
    if total == 0:
        return
    local_first = np.cumsum(n) - n
    dest = np.repeat(first[items] - local_first, n) + np.arange(total, dtype=np.int64)
    out[dest] = part[:total]


def _predict_sharded(make_engine, input_path, fsize, stride, user_min_len, min_len, dust, common, want, lg,
                     rank, world, local_rank, log_setup):
    """Rank 0 indexes the FASTA and broadcasts (record byte offsets, lengths); every rank reads, soft-masks, scans
    and classifies only the contigs LPT deals it; one padded gather of f32 rows (logits | reliability | G C A T counts
    [| embedding | nmd]) and one of the int32 repeat table go to rank 0, which rebuilds the window metadata from
    its own window table and restores FASTA order (long pass before short pass)."""
    import torch
    import torch.distributed as dist

    from . import dist as jdist
    from .termini import repeats_frame, terminal_repeat_table
    if not dist.is_initialized():
        SHARD_STATS["own_process_group"] = True          # run_core destroys what it created
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
        # RCCL ("nccl") when GPUs are there; JAEGER_DIST_BACKEND=gloo lets several ranks share one GPU (tests)
        backend = os.environ.get("JAEGER_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ and world == 1:      # one rank outside torchrun (JAEGER_SHARDED=1)
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        try:
            if backend == "nccl":
                dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
        except Exception as e:              # no fallback to another backend: the error is the result
            lg.error(f"could not initialise the {backend} process group on rank {rank}/{world} "
                     f"(HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}): {type(e).__name__}: {e}")
            raise
    dev = _coll_device(local_rank)

    def all_ok(ok: bool) -> bool:
        t = torch.tensor([0 if ok else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t.item()) == 0

    t_ingest = time.time()
    ix, head = None, np.zeros(2, np.int64)
    if rank == 0:
        try:
            ix = frag.index_fasta(str(input_path))
            ok = int((ix.lengths >= min_len).sum())
            lg.info(f"{ok}/{len(ix)} entries in {input_path}")
            if ok == 0:
                raise Exception(f"all records in {input_path} are < {min_len}bp")
            head[:] = (0, len(ix))
        except Exception as e:
            lg.error(e)
            head[:] = (1, 0)
    head = _bcast(head if rank == 0 else None, np.int64, 2, dev)
    if head[0] != 0:
        sys.exit(1)
    n_rec = int(head[1])
    rec_off = _bcast(ix.rec_off if rank == 0 else None, np.int64, n_rec + 1, dev)
    lengths = _bcast(ix.lengths if rank == 0 else None, np.int64, n_rec, dev)
    groups = [np.sort(g) for g in _shard_records(lengths, fsize, stride, world)]
    used = lengths >= min_len                                   # contigs that yield a window in either pass
    mine = groups[rank][used[groups[rank]]]
    ok, fa, engine, local, rep_local = True, None, None, None, None
    t_predict = 0.0
    two_pass = user_min_len is not None and user_min_len < fsize
    try:
        fa = frag.load_fasta_records(str(input_path), rec_off, mine)
        if not np.array_equal(fa.lengths, lengths[mine]):
            raise RuntimeError(f"{input_path}: record lengths changed between the index pass and this rank's read")
        t_ingest = time.time() - t_ingest
        SHARD_STATS.update(rank=rank, local_bases=int(fa.bases.size), total_bases=int(lengths.sum()),
                           local_records=len(fa), total_records=n_rec)
        if dust:
            t0 = time.time()
            n_masked = frag.dust_mask(fa)
            lg.info(f"DUST (window 64, threshold 20): {n_masked} of {fa.bases.size} bases of this rank's "
                    f"{len(fa)} contigs soft-masked in {time.time() - t0:.2f} s")
        engine = make_engine()
        log_setup(engine)
        t_predict = time.time()
        kw = dict(common, want=want, meta=False)
        if two_pass:
            lg.info(f"Two-pass prediction: long contigs (>= {fsize} bp) then short contigs "
                    f"({user_min_len}-{fsize - 1} bp)")
            parts = [predict_batch(engine, fa, fsize, stride, min_len=fsize, max_len=None, **kw),
                     predict_batch(engine, fa, fsize, stride, min_len=user_min_len, max_len=fsize - 1, padded=True, **kw)]
        else:
            parts = [predict_batch(engine, fa, fsize, stride, min_len=min_len, max_len=None, **kw)]
        cols = [k for k in ("prediction", "reliability", "counts", "embedding", "nmd") if any(k in p for p in parts)]
        mats = [np.concatenate([np.asarray(p[k], np.float32).reshape(len(p[k]), -1) for k in cols], axis=1)
                for p in parts if p]
        widths = {k: next(np.asarray(p[k]).reshape(len(p[k]), -1).shape[1] for p in parts if p) for k in cols}
        local = np.concatenate(mats, axis=0) if mats else None
        t_predict = time.time() - t_predict
        rep_local = terminal_repeat_table(engine.device, fa, fsize)
    except BaseException as e:               # every rank must reach the collectives below, or the others hang
        lg.debug(traceback.format_exc())
        lg.error(f"an error {e} occured during inference on MI355X #{local_rank} (rank {rank})!")
        ok = False
    if not all_ok(ok):
        if engine is not None:
            engine.close()
        sys.exit(1)
    # column layout is the same on every rank (same model); ranks without windows send zero rows
    wtab = torch.zeros(8, dtype=torch.int64, device=dev)
    if local is not None:
        for j, k in enumerate(("prediction", "reliability", "counts", "embedding", "nmd")):
            wtab[j] = widths.get(k, 0)
    dist.all_reduce(wtab, op=dist.ReduceOp.MAX)
    wlist = [int(v) for v in wtab.tolist()[:5]]
    n_col = sum(wlist)
    if local is None:
        local = np.zeros((0, n_col), np.float32)
    got = jdist.gather_rows(torch.from_numpy(np.ascontiguousarray(local)).to(dev), dst=0)
    rep = jdist.gather_rows(torch.from_numpy(np.ascontiguousarray(rep_local.reshape(-1, 10))).to(dev), dst=0)
    class_map = engine.class_map
    engine.close()
    if rank != 0:
        dist.barrier()
        return None
    # ---- rank 0: metadata from its own window table, rows back in the reference's emission order -------------
    passes = [dict(min_len=fsize, max_len=None), dict(min_len=user_min_len, max_len=fsize - 1)] if two_pass else \
        [dict(min_len=min_len, max_len=None)]
    tables = [frag.build_window_table(lengths, fsize, stride, common["dynamic_stride"],
                                      common["dynamic_stride_threshold"], **pk) for pk in passes]
    per_rec = [np.bincount(t.contig, minlength=n_rec).astype(np.int64) for t in tables]
    y_pred: dict = {}
    consumed = [0] * world
    for t, n_w in zip(tables, per_rec):
        rows = np.zeros((len(t), n_col), np.float32)
        first = np.cumsum(n_w) - n_w
        for q in range(world):
            items = groups[q][used[groups[q]]]
            take = int(n_w[items].sum())
            part = got[q].cpu().numpy()[consumed[q]:consumed[q] + take]
            _place_rows(rows, part, items, n_w, first)
            consumed[q] += take
        out, c0 = {}, 0
        for k, w in zip(("prediction", "reliability", "counts", "embedding", "nmd"), wlist):
            if w:
                out[k] = rows[:, c0:c0 + w]
            c0 += w
        counts = np.rint(out.pop("counts")).astype(np.int32)
        out.update(frag.window_metadata(t, ix.names, counts))
        y_pred = _concat_predictions(y_pred, out) if len(t) else y_pred
    res = np.full((n_rec, 10), -1, np.int32)
    for q in range(world):
        items = groups[q][used[groups[q]]]
        res[items] = rep[q].cpu().numpy()
    term_repeats = repeats_frame(res, ix.names, lengths)
    lg.info(f"terminal repeats: {int(term_repeats['terminal_repeats'].notna().sum())} of {len(term_repeats)} contigs")
    dist.barrier()
    return dict(y_pred=y_pred, term_repeats=term_repeats, class_map=class_map, num=n_rec, t_ingest=t_ingest,
                t_predict=t_predict)


# ---- run_core ---------------------------------------------------------------------------------
def run_core(**kwargs) -> int:
    """Equivalent of ``commands/predict.py:run_core``; returns the number of table rows written."""
    import yaml

    from .engine import JaegerHipEngine

    wait_for_release()
    t_start = time.time()
    LAST_RUN.clear()
    timeline: list = []                     # (event, seconds since run_core was entered): where the wall time of a short run goes
    LAST_RUN["timeline"] = timeline
    LAST_RUN["t_start_epoch"] = t_start

    def mark(name):
        timeline.append((name, round(time.time() - t_start, 4)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(kwargs.get("physicalid", 0))))

    model_path = kwargs.get("model_path")
    if model_path:
        info = AvailableModels(path=model_path).info
        if not info:
            print(f"No model found in {model_path}", file=sys.stderr)
            sys.exit(1)
        cands = {n: m for n, m in info.items() if m.get("project") is not None and m.get("classes") is not None}
        if not cands:
            print(f"No classification model found in {model_path}. Expected *_classes.yaml, *_project.yaml "
                  "and *.weights.h5 / *.weights.npz files.", file=sys.stderr)
            sys.exit(1)
        non_emb = {n: m for n, m in cands.items() if not n.endswith("_embedding")}
        cands = non_emb or cands
        model_name = next(iter(cands))
        model_info = cands[model_name]
    else:
        cfg_path = kwargs.get("config")
        if not cfg_path:
            print("jaeger_amd: pass --model_path <dir with model/> or --config <json with model_paths>",
                  file=sys.stderr)
            sys.exit(1)
        paths = json.loads(Path(cfg_path).read_text()).get("model_paths", [])
        info = AvailableModels(path=paths).info
        model_name = kwargs.get("model")
        if model_name not in info:
            print(f"model {model_name!r} not found under {paths}", file=sys.stderr)
            sys.exit(1)
        model_info = info[model_name]
    model_id = get_model_id(model_name)

    input_path = Path(kwargs.get("input"))
    file_base = input_path.stem
    out_dir = Path(kwargs.get("output")) / model_id
    out_dir.mkdir(parents=True, exist_ok=True)
    lg = get_logger(out_dir, Path(f"{file_base}_jaeger.log"), kwargs.get("verbose", 1))
    fsize, stride = kwargs.get("fsize", 2000), kwargs.get("stride", 1500)
    user_min_len = kwargs.get("min_len")
    min_len = user_min_len or fsize
    fa, num, t_ingest = None, 0, 0.0
    table_path, phage_path = out_dir / f"{file_base}.tsv", out_dir / f"{file_base}_phages.tsv"
    if table_path.exists() and not kwargs.get("overwrite"):
        lg.error("output file exists. enable --overwrite option to overwrite the output file.")
        sys.exit(1)
    for flag in ("refine", "quantized", "onnx", "int8", "cpu"):
        if kwargs.get(flag):
            lg.error(f"--{flag} is not available on the MI355X predict path (jaeger_amd has no CPU / "
                     "alternative-backend fallback; refinement post-processing is out of scope)")
            sys.exit(1)
    # experimental CRF (Viterbi) window decoding (commands/predict.py:288-307)
    crf_kw = {}
    if kwargs.get("crf"):
        lg.warning("CRF window decoding is experimental; results may change between releases")
        crf_kw = {"crf_switch_cost": kwargs.get("crf_switch_cost", 2.0), "crf_prior": kwargs.get("crf_prior", "biological")}
        matrix_path = kwargs.get("crf_transition_matrix")
        if matrix_path:
            crf_kw["crf_transition_matrix"] = json.loads(Path(matrix_path).read_text())
    precision = "f32" if kwargs.get("exact_f32") else None

    def make_engine():
        t_eng = time.time()
        mark("engine_setup_begin")
        try:
            return _make_engine()
        finally:
            LAST_RUN["engine_create_s"] = round(time.time() - t_eng, 3)
            mark("engine_ready")

    def _make_engine():
        # the engine resolves the weights itself (weights.load_weights): the graph's variable bundle - what the reference
        # executes - first, then the .weights.h5, then the canonical .npz
        eng = JaegerHipEngine(model_info, device_id=local_rank, chunk=kwargs.get("chunk", 0),
                              precision=precision, trust_project=True if kwargs.get("trust_project") else None)
        if kwargs.get("stream_bytes"):              # span budget of the host -> HBM ingest (default 32 MiB)
            eng.device.set_stream_bytes(int(kwargs["stream_bytes"]))
        if kwargs.get("dust_stream") is not None:   # A/B: 1 = DUST on the copy stream (default), 0 = on the compute stream
            from . import _lib as L
            L.check(eng.device.lib.jg_engine_set_option(eng.device.handle, L.JG_OPT_DUST_ON_COPY_STREAM, int(kwargs["dust_stream"])))
        return eng

    def ingest():
        """(under torchrun rank 0 indexes the file and every rank reads its own contigs instead)"""
        nonlocal fa, num, t_ingest
        try:
            t_ingest = time.time()
            mark("ingest_begin")
            fa = frag.load_fasta(str(input_path))
            t_ingest = time.time() - t_ingest
            mark("ingest_done")
            num = validate_fasta_entries(fa, min_len=min_len)
        except Exception as e:
            lg.error(e)
            sys.exit(1)

    def engine_failed(e, tb):
        lg.debug(tb)
        lg.error(f"could not set up the model on GPU {local_rank}: {e}")
        sys.exit(1)

    def log_setup(engine):
        sp = engine.string_processor_config
        lg.info(f"input file: {input_path.name}")
        lg.info(f"outpath: {out_dir.resolve()}")
        lg.info(f"fragment size: {fsize}  stride: {stride}  batch: {kwargs.get('batch', 96)}")
        lg.info(f"model: {model_id}  arithmetic: {engine.model.precision}  device: MI355X #{local_rank} "
                f"(rank {rank}/{world})")
        try:                                   # kernel placement (engines of other kinds have none)
            pl = engine.model.placement()
            if pl["small_fused"]:
                lg.info("kernels: fused small-window kernel (ids -> pooled sums in one launch)")
            elif engine.model.precision == "f16x3":
                lg.info(f"kernels: {pl['convs_f16x3']} of {pl['convs']} convolutions on the split-f16 kernels"
                        + (f", {pl['layout_conversions']} layout conversions" if pl["layout_conversions"] else ""))
                if pl["convs_f16x3"] < pl["convs"]:
                    slow = [ln for ln in engine.model.describe().splitlines() if "exact-f32" in ln]
                    lg.warning("convolutions left on the exact-f32 kernel (about 4x slower):\n  " + "\n  ".join(slow))
        except AttributeError:
            pass
        msg = _crop_length_warning(sp.get("crop_size_codons"), sp.get("crop_size_nt"), fsize)
        if msg:
            lg.warning(msg)

    def scan_repeats(device):
        t_term = time.time()
        mark("repeat_scan_begin")
        from .termini import REPORT_MIN_COLUMNS, RepeatColumns, terminal_repeat_table
        # (alignments of at most 12 columns never reach the table: the scan may skip them - same repeat columns)
        table = terminal_repeat_table(device, fa, fsize, report_min=REPORT_MIN_COLUMNS)
        mark("repeat_table_done")
        rep = RepeatColumns(table, fa.names, fa.lengths)       # (the DataFrame form only where something asks for it)
        LAST_RUN["terminal_repeats_s"] = round(time.time() - t_term, 3)
        mark("repeat_scan_done")
        lg.info(f"terminal repeats: {rep.n_found} of {len(rep)} contigs in {time.time() - t_term:.2f} s")
        return rep

    # DUST soft-masking (on by default like the reference): on the GPU, on the uploaded copy of the bases inside the
    # fused call; --dust-host runs the host scan over the FASTA image instead (same masks bit for bit)
    dust_any = bool(kwargs.get("dustmask", True))
    dust = dust_any and bool(kwargs.get("dust_host", False))          # the host pass
    dust_dev = dust_any and not dust
    two_pass = user_min_len is not None and user_min_len < fsize
    want = ("prediction", "reliability") + (("embedding",) if kwargs.get("save_embedding") else ()) \
        + (("nmd",) if kwargs.get("save_nmd") else ())
    common = dict(dynamic_stride=kwargs.get("dynamic_stride", False),
                  dynamic_stride_threshold=kwargs.get("dynamic_stride_threshold", 10.0),
                  batch=kwargs.get("batch", 96), pre_cased=dust, want=want, dust_device=dust_dev)
    term_repeats = None
    class_map = None
    t_predict = time.time()
    # JAEGER_SHARDED=1: take the sharded path whatever the world size - with ONE rank under torchrun every collective of the
    # 8-GPU run (init over RCCL, broadcasts, the error all_reduce, both padded gathers, barrier) executes on the one GPU
    sharded = world > 1 or os.environ.get("JAEGER_SHARDED") == "1"
    if sharded:
        got = _predict_sharded(make_engine, input_path, fsize, stride, user_min_len, min_len, dust, common, want, lg,
                               rank, world, local_rank, log_setup)
        if got is None:                 # ranks other than 0 are done after the gather
            _close_process_group()
            return 0
        y_pred, term_repeats, class_map = got["y_pred"], got["term_repeats"], got["class_map"]
        num, t_ingest, t_predict = got["num"], got["t_ingest"], got["t_predict"]
        engine = None
    else:
        # ---- one GPU.  A worker thread owns the engine: it sets the model up while the FASTA is read (every core, native),
        # then runs ONE fused call over all windows of the long pass (DUST on its uploaded spans, encoder, forward; the
        # library streams the spans through pinned staging and publishes its progress).  Beside it the calling thread scans
        # for terminal repeats on a stream of its own and aggregates the contigs whose windows are final, so that what is
        # left behind the forward is the last batch and the TSV.  --no-pipeline runs the same steps one after the other.
        from concurrent.futures import ThreadPoolExecutor
        piped = not kwargs.get("no_pipeline")
        pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="jaeger-gpu") if piped else None
        t_setup = time.time()
        f_engine = pool.submit(make_engine) if piped else None
        ingest()
        scan: dict = {}
        th = None
        if piped:
            # the terminal-repeat scan starts as soon as the bases are there - a thread and a stream of its own - so that most
            # of it runs while the model is still being set up: beside the forward its workgroups only get the CUs in the gaps
            # between the network's launches (the conv and small-window kernels take a CU's whole LDS), and the rows of
            # finished batches cannot go to the table before the repeat columns exist

            def scan_side():
                try:
                    from .engine import HipDevice
                    side = HipDevice(local_rank)
                    # the scan's short kernels go in front of the network's launches (stream priority): the repeat table is
                    # there long before the forward ends, and finished batches' rows are written beside it, not behind it
                    if not kwargs.get("scan_low_priority") and hasattr(side, "set_stream_priority"):   # (duck-typed devices)
                        side.set_stream_priority(True)
                    try:
                        scan["frame"] = scan_repeats(side)
                    finally:
                        side.close()
                except BaseException as e:
                    scan["error"] = e

            th = threading.Thread(target=scan_side, name="jaeger-termini", daemon=True)
            th.start()
        if dust:
            t_dust = time.time()
            n_masked = frag.dust_mask(fa)
            lg.info(f"DUST (window 64, threshold 20): {n_masked} of {fa.bases.size} bases soft-masked in "
                    f"{time.time() - t_dust:.2f} s")
        if two_pass:
            lg.info(f"Two-pass prediction: long contigs (>= {fsize} bp) then short contigs "
                    f"({user_min_len}-{fsize - 1} bp)")
        table = frag.build_window_table(fa.lengths, fsize, stride, common["dynamic_stride"],
                                        common["dynamic_stride_threshold"], fsize if two_pass else min_len, None)
        n_long = len(table)
        starts = fa.offsets[table.contig] + table.start
        mark("window_table_done")
        LAST_RUN["ingest_and_table_s"] = round(time.time() - t_setup, 3)
        try:
            engine = f_engine.result() if piped else make_engine()
        except Exception as e:
            engine_failed(e, traceback.format_exc())
        t_setup = time.time() - t_setup
        class_map = engine.class_map
        out = engine.model.host_outputs(n_long, want)

        writer = _LazyTableWriter(class_map, table_path, phage_path, kwargs.get("rc", 0.5), kwargs.get("pc", 1))
        agg = _Aggregator(table, fa.names, out, dict(class_map=class_map, fsize=fsize, term_repeats=None,
                                                     want_full=bool(kwargs.get("window_scores") or kwargs.get("prophage")),
                                                     **crf_kw), min_batch=n_long // 16, workers=2 if piped else 1)

        def classify():
            t0 = time.time()
            mark("fused_call_begin")
            if n_long:
                engine.predict_windows(fa.bases, starts, table.length, fsize, pre_cased=dust, want=want,
                                       dust_records=fa.offsets if dust_dev else None, out=out)
            mark("fused_call_done")
            return time.time() - t0

        if piped and th is not None and kwargs.get("scan_first"):
            th.join()                       # A/B: the repeat scan alone on the GPU, the forward behind it
        flusher, flush_stop, flush_err = None, None, {}
        try:
            f_pred = pool.submit(classify) if piped else None
            log_setup(engine)
            if piped:
                # rows of finished batches go to the table beside the forward, on a thread of their own (the calling thread
                # aggregates; formatting a batch's rows is native code and numpy: both run without the interpreter lock)
                flush_stop = threading.Event()

                def flush_side():
                    try:
                        while True:
                            stopping = flush_stop.is_set()
                            if "frame" in scan:     # (the pass after the stop signal waits for the pool's last batches)
                                agg.flush(writer, scan["frame"], wait=stopping)
                            if stopping:
                                break
                            time.sleep(0.002)
                    except BaseException as e:          # noqa: BLE001 - handed to the calling thread
                        flush_err["error"] = e

                flusher = threading.Thread(target=flush_side, name="jaeger-rows", daemon=True)
                flusher.start()
                while not f_pred.done():
                    agg.names_unique()
                    agg.advance(engine.device.windows_done())
                    time.sleep(0.004)
                t_forward = f_pred.result()
                agg.advance(n_long, final=True)       # the last contigs: needs no repeat column, runs while the scan finishes
                mark("aggregated")
                th.join()
                flush_stop.set()
                flusher.join()
                flusher = None
                mark("rows_beside_forward_done")
                if "error" in flush_err:
                    raise flush_err["error"]
                if "error" in scan:
                    raise scan["error"]
                term_repeats = scan["frame"]
            else:
                term_repeats = scan_repeats(engine.device)
                t_forward = classify()
            agg.advance(n_long, final=True)
            if two_pass:
                y_short = predict_batch(engine, fa, fsize, stride, min_len=user_min_len, max_len=fsize - 1,
                                        padded=True, **common)
                agg.add(y_short)
        except Exception as e:
            writer.abort()                  # rows of finished batches were appended beside the forward: no truncated table stays
            lg.debug(traceback.format_exc())
            lg.error(f"an error {e} occured during inference on MI355X #{local_rank}!")
            sys.exit(1)
        finally:
            if flusher is not None:             # (an error path: the row thread must not outlive the writer it uses)
                flush_stop.set()
                flusher.join()
            agg.close()                         # (every batch handed to the pool has been aggregated)
            if pool is not None:
                pool.shutdown(wait=True)
        t_predict = time.time() - t_predict
        n_windows = n_long + (len(y_short.get("meta_2", ())) if two_pass else 0)
        n_bp = float(np.minimum(table.seqlen, fsize).sum()) + \
            (float(np.minimum(np.asarray(y_short["meta_4"], np.int64), fsize).sum()) if two_pass and y_short else 0.0)
        y_pred = None
        if kwargs.get("save_embedding") or kwargs.get("save_nmd"):        # the per-window vectors with their headers
            y_pred = _concat_predictions(agg.slice(0, n_long) if n_long else {}, y_short if two_pass else {})
        lg.info(f"GPU worker: model set-up {t_setup:.2f} s (beside the FASTA ingest), {n_long} windows classified in "
                f"{t_forward:.2f} s; {len(agg.parts)} aggregation batches, {agg.busy_s:.2f} s beside the forward")
        LAST_RUN.update(model_setup_and_ingest_s=round(t_setup, 3), ingest_s=round(t_ingest, 3), forward_s=round(t_forward, 3),
                        aggregation_batches=len(agg.parts), aggregation_beside_forward_s=round(agg.busy_s, 3),
                        windows=n_windows, pipelined=bool(piped))
    if class_map is None:
        class_map = engine.class_map
    if dust_dev and engine is not None:
        lg.info(f"DUST (window 64, threshold 20) on the GPU: {engine.dust_masked_total} bases soft-masked inside the "
                f"fused calls")
    t_post = time.time()

    from .postprocess import header_strings, pred_to_dict, write_output
    if sharded:
        data, data_full = pred_to_dict(y_pred, class_map=class_map, fsize=fsize, term_repeats=term_repeats,
                                       want_full=bool(kwargs.get("window_scores") or kwargs.get("prophage")), **crf_kw)
        n_windows = len(y_pred["meta_2"])
        n_bp = float(np.minimum(np.asarray(y_pred["meta_4"], np.int64), fsize).sum())
        LAST_RUN["merge_s"] = round(time.time() - t_post, 3)
        n_written = write_output(data, labels=class_map.get("class"), indices=class_map.get("index"),
                                 output_table_path=table_path, output_phage_table_path=phage_path,
                                 reliability_cutoff=kwargs.get("rc", 0.5), phage_score=kwargs.get("pc", 1))
    else:
        try:
            mark("last_flush_begin")
            agg.flush(writer, term_repeats)            # what is left: the last batch (and the short-contig pass)
            n_written = writer.close()
            mark("tables_closed")
        except BaseException:
            writer.abort()
            raise
        data_full = agg.result_full()
        LAST_RUN["merge_s"] = 0.0
    LAST_RUN["tsv_s"] = round(time.time() - t_post - LAST_RUN["merge_s"], 3)
    lg.info(f"processed {n_written}/{num} sequences")
    if kwargs.get("window_scores"):
        np.savez(out_dir / f"{file_base}_window_scores.npz", headers=header_strings(data_full["headers"]),
                 lengths=data_full["lengths"], predictions=np.array(data_full["predictions"], dtype=object),
                 gc_skews=np.array(data_full["gc_skews"], dtype=object),
                 gcs=np.array(data_full["gcs"], dtype=object))
    if kwargs.get("prophage"):
        # --- prophage segmentation inputs (commands/predict.py:353-442): the per-contig frames logits_to_df_v2 builds
        # from the window logits.  The change-point segmentation, boundary refinement and plots that consume them
        # (ruptures / kneed / pycirclize) are not part of this path; the frames are written for them.
        try:
            from .prophage_inputs import logits_to_df_v2
            frames = logits_to_df_v2(class_map=class_map, cmdline_kwargs=kwargs, headers=header_strings(data_full["headers"]),
                                     predictions=data_full["predictions"], lengths=data_full["lengths"],
                                     gc_skews=data_full["gc_skews"], gcs=data_full["gcs"])
            if frames:
                pro_dir = out_dir / f"{file_base}_prophages"
                pro_dir.mkdir(parents=True, exist_ok=True)
                keys = list(frames)
                np.savez(pro_dir / f"{file_base}_segmentation_inputs.npz",
                         contigs=np.array(keys, dtype=object), hosts=np.array([frames[k][1] for k in keys], dtype=object),
                         lengths=np.array([frames[k][2] for k in keys], np.int64),
                         columns=np.array(list(frames[keys[0]][0].columns), dtype=object),
                         tracks=np.array([frames[k][0].to_numpy(np.float64) for k in keys], dtype=object))
                lg.info(f"prophage segmentation inputs of {len(keys)} contigs (>= {kwargs.get('lc', 500000)} bp) written "
                        f"to {pro_dir}; the segmentation / plotting step itself is not part of the MI355X path")
            else:
                lg.info("no prophage regions found")
        except Exception as e:
            lg.error(f"an error {e} occurred during the prophage prediction step")
            lg.debug(traceback.format_exc())
    if kwargs.get("getsequences"):
        # --- phage sequences as FASTA (commands/predict.py:444-456) ---
        from .postprocess import write_fasta_from_results
        out_fasta = out_dir / f"{file_base}_phages_jaeger.fasta"
        lg.info(f"generating fasta file {out_fasta}")
        n_seq = write_fasta_from_results(input_path, phage_path, out_fasta)
        lg.info(f"{n_seq} phage sequences written")
    if y_pred is not None:
        headers = np.asarray(y_pred.get("meta_0", np.array([], dtype=object))).astype(str)
        if kwargs.get("save_embedding") and "embedding" in y_pred:
            np.savez(out_dir / f"{file_base}_embedding.npz", embedding=y_pred["embedding"], headers=headers)
        if kwargs.get("save_nmd") and "nmd" in y_pred:
            np.savez(out_dir / f"{file_base}_nmd.npz", embedding=y_pred["nmd"], headers=headers)   # legacy key name
    mark("end")
    t_all = time.time() - t_start
    LAST_RUN.update(behind_forward_s=round(time.time() - t_post, 3), wall_s=round(t_all, 3))
    lg.info(f"wall time(s) : {t_all:.2f}  ({n_windows} windows; FASTA ingest {t_ingest:.2f} s, "
            f"encode+forward {t_predict:.2f} s = {n_bp / 1e6 / max(t_predict, 1e-9):.1f} Mbp/s, aggregation+TSV behind the forward "
            f"{time.time() - t_post:.2f} s; end to end {n_bp / 1e6 / max(t_all, 1e-9):.1f} Mbp/s)")
    if engine is not None:
        # the engine's buffers (workspace, pinned staging) are released beside the caller: freeing them takes tens of
        # milliseconds - a tenth of a short run - and nothing the caller gets depends on it
        # ... nor on the large host buffers of the run (bases, window table, per-window outputs: unmapping half a gigabyte
        # takes another 20 ms): their last references are dropped on that thread too
        garbage = [engine, locals().get("fa"), locals().get("table"), locals().get("out"), locals().get("agg"),
                   locals().get("starts"), locals().get("y_pred")]

        def release(objs):
            try:
                objs[0].close()
            except Exception as e:          # noqa: BLE001 - nobody joins this thread for its result: say it in the run log
                lg.warning(f"releasing the engine failed: {type(e).__name__}: {e}")
            finally:
                objs.clear()

        th_close = threading.Thread(target=release, args=(garbage,), name="jaeger-engine-close")
        _PENDING_RELEASE.append(th_close)   # the next run_core (and bench.py before it times anything) joins it
        th_close.start()
        del garbage
    if sharded:
        _close_process_group()
    return n_written
