"""``jaeger predict -m default`` (the legacy workflow, ``commands/predict_legacy.py:34-357``) on the
MI355X engine: same data layout (``config.json`` + ``models/default/{WRes_1024.h5, LR_ood_4_class_default.pkl,
batch_means.npy, batch_std.npy}``), same output files ``<output>/default/<stem>_jaeger.tsv`` and
``<stem>_phages_jaeger.tsv``.  The data directory is the reference's ``jaeger/data`` (``--legacy-data``,
``$JAEGER_DATA``, or the installed ``jaeger`` package when importable).
"""

from __future__ import annotations

import json
import os
import sys
import time
import traceback
from pathlib import Path

import numpy as np

from . import fragment as frag
from .postprocess_legacy import pred_to_dict_legacy, write_output_legacy
from .predict import get_logger, validate_fasta_entries


def find_legacy_data(explicit=None) -> Path:
    for cand in (explicit, os.environ.get("JAEGER_DATA")):
        if cand:
            return Path(cand)
    try:
        from importlib.resources import files
        return Path(str(files("jaeger.data")))
    except Exception as e:                                   # the reference package is not installed
        raise FileNotFoundError("legacy model data not found: pass --legacy-data <dir with config.json and "
                                "models/default/> or set JAEGER_DATA") from e


def load_ood_params(model_path: Path, config: dict):
    """predict_legacy.py:82-109."""
    if not config.get("ood"):
        return None
    p = model_path / config["ood"]
    if p.suffix == ".h5":
        from .hdf5_lite import read_datasets
        d = read_datasets(p)
        return {"type": "params", "coeff": d["/coeff"], "intercept": d["/intercept"], "batch_mean": d["/mean_batch"]}
    if p.suffix == ".pkl":
        import joblib
        return {"type": "sklearn", "model": joblib.load(p), "batch_mean": np.load(model_path / "batch_means.npy"),
                "batch_std": np.load(model_path / "batch_std.npy")}
    raise ValueError(f"unsupported reliability model file {p.name}")


def predict_batch_legacy(engine, fa, fsize, stride, min_len, dynamic_stride=False,
                         dynamic_stride_threshold=10.0, pre_cased=False) -> dict:
    table = frag.build_window_table(fa.lengths, fsize, stride, dynamic_stride, dynamic_stride_threshold, min_len, None)
    if len(table) == 0:
        return {}
    out = engine.predict_windows(fa.bases, fa.offsets[table.contig] + table.start, table.length, fsize,
                                 pre_cased=pre_cased)
    meta = frag.window_metadata(table, fa.names, out.pop("counts"))
    return {"y_hat": {"output": out["output"], "embedding": out["embedding"]},
            "meta": [meta[f"meta_{i}"] for i in range(10)]}


def run_core(**kwargs) -> int:
    from .legacy import LegacyHipEngine

    t_start = time.time()
    model = kwargs.get("model") or "default"
    data_path = find_legacy_data(kwargs.get("legacy_data"))
    model_path = data_path / "models" / model
    config = json.loads((data_path / "config.json").read_text()).get(model)
    config["model"] = model
    input_path = Path(kwargs.get("input"))
    file_base = input_path.stem
    out_dir = Path(kwargs.get("output")) / model
    out_dir.mkdir(parents=True, exist_ok=True)
    lg = get_logger(out_dir, Path(f"{file_base}_jaeger.log"), kwargs.get("verbose", 1))
    fsize = kwargs.get("fsize", 2000)
    try:
        min_len = kwargs.get("min_len") or fsize
        if min_len < fsize:
            lg.warning(f"--min-len < --fsize is not supported in legacy prediction mode; using --min-len={fsize}.")
            min_len = fsize
        fa = frag.load_fasta(str(input_path))
        num = validate_fasta_entries(fa, min_len=min_len)
    except Exception as e:
        lg.error(e)
        sys.exit(1)
    table_path = out_dir / f"{file_base}_jaeger.tsv"
    phage_path = out_dir / f"{file_base}_phages_jaeger.tsv"
    if table_path.exists() and not kwargs.get("overwrite"):
        lg.error("output file exists. enable --overwrite option to overwrite the output file.")
        sys.exit(1)
    weights_path = model_path / config["weights"]
    if not weights_path.exists():
        lg.error("could not find model weights. please check the data dir")
        sys.exit(1)
    for flag in ("prophage", "cpu", "getsequences"):
        if kwargs.get(flag):
            lg.error(f"--{flag} is not available on the MI355X predict path")
            sys.exit(1)
    dusted = False
    if kwargs.get("dustmask", True):
        n_masked = frag.dust_mask(fa)
        dusted = True
        lg.info(f"DUST (window 64, threshold 20): {n_masked} of {fa.bases.size} bases soft-masked")
    ood_params = load_ood_params(model_path, config)
    try:
        engine = LegacyHipEngine(weights_path, device_id=kwargs.get("physicalid", 0), chunk=kwargs.get("chunk", 0),
                                 precision="f32" if kwargs.get("exact_f32") else None)
    except Exception as e:
        lg.debug(traceback.format_exc())
        lg.error(f"could not set up the legacy model on GPU {kwargs.get('physicalid', 0)}: {e}")
        sys.exit(1)
    from .termini import scan_for_terminal_repeats
    term_repeats = scan_for_terminal_repeats(engine.device, fa, fsize)
    lg.info(f"input file: {input_path.name}  fragment size: {fsize}  stride: {kwargs.get('stride')}  "
            f"model: {model}  arithmetic: {engine.model.precision}")
    try:
        y_pred = predict_batch_legacy(engine, fa, fsize, kwargs.get("stride", 1500), min_len,
                                      kwargs.get("dynamic_stride", False),
                                      kwargs.get("dynamic_stride_threshold", 10.0), pre_cased=dusted)
    except Exception as e:
        lg.debug(traceback.format_exc())
        lg.error(f"an error {e} occured during inference!")
        sys.exit(1)
    engine.close()
    key = "all_labels" if kwargs.get("getalllabels") else "default_labels"
    config["labels"] = [v for _, v in config[key].items()]
    data, data_full = pred_to_dict_legacy(config, y_pred, model=model, fsize=fsize, ood_params=ood_params,
                                          term_repeats=term_repeats)
    n = write_output_legacy(config, data, output_table_path=table_path, output_phage_table_path=phage_path,
                            reliability_cutoff=kwargs.get("rc", 0.5), phage_score=kwargs.get("pc", 3))
    lg.info(f"processed {data.get('headers').shape[0]}/{num} sequences")
    if kwargs.get("window_scores"):
        np.savez(out_dir / f"{file_base}_{config['suffix']}_window_scores.npz", headers=data_full["headers"],
                 lengths=data_full["lengths"], predictions=np.array(data_full["predictions"], dtype=object),
                 gc_skews=np.array(data_full["gc_skews"], dtype=object),
                 gcs=np.array(data_full["gcs"], dtype=object))
    lg.info(f"wall time(s) : {time.time() - t_start:.2f}")
    return n
