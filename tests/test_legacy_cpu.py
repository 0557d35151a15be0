"""Legacy ``default`` model (BASELINE config #1): host-side pieces that need no GPU."""
import hashlib
import json

import numpy as np
import pandas as pd
import pytest
from click.testing import CliRunner
from conftest import GOLDEN

LEGACY = GOLDEN / "legacy_data"
H5 = LEGACY / "models" / "default" / "WRes_1024.h5"


def test_hdf5_reader_matches_h5dump_view():
    """Every dataset of the reference's weight file: shape and bytes as h5dump printed them
    (digests written by make_golden.py after an element-wise comparison with h5dump)."""
    from jaeger_amd.hdf5_lite import read_datasets
    want = json.loads((GOLDEN / "legacy_h5_datasets.json").read_text())
    got = read_datasets(H5)
    assert sorted(got) == sorted(want) and len(got) == 79
    for key, meta in want.items():
        assert list(got[key].shape) == meta["shape"], key
        assert hashlib.sha256(got[key].tobytes()).hexdigest() == meta["sha256"], key
    assert sum(v.size for v in got.values()) == 947036


def test_hdf5_reader_rejects_garbage(tmp_path):
    from jaeger_amd.hdf5_lite import H5File, H5Unsupported
    with pytest.raises(H5Unsupported):
        H5File(b"not an hdf5 file at all")


def test_v1_trimer_table():
    from jaeger_amd import maps
    tables = json.loads((GOLDEN / "maps.json").read_text())
    ref = dict(zip(tables["V1_TRIMERS"], tables["V1_TRIMER_INT"]))
    assert [ref[c] for c in maps.CODONS] == maps.V1_TRIMER_INT
    assert min(maps.V1_TRIMER_INT) == 1 and max(maps.V1_TRIMER_INT) == 21


def test_legacy_weights_and_program():
    from jaeger_amd import _lib as L
    from jaeger_amd import legacy
    w = legacy.load_legacy_h5(H5)
    assert set(w) == set(legacy.weight_shapes())
    assert w["block1_0/kernel"].shape == (9, 4, 128)          # stored as `conv1d` in the file
    prog = legacy.compile_legacy(w)
    kinds = [op.kind for op in prog.ops]
    assert kinds.count(L.OP_CONV) == 12 and kinds.count(L.OP_MAXPOOL1D) == 2 and kinds.count(L.OP_FRAMESUM) == 1
    assert kinds[-3:] == [L.OP_DENSE] * 3 and prog.vocab == 22
    convs = [op for op in prog.ops if op.kind == L.OP_CONV]
    assert [op.dilation for op in convs] == [1, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7]
    assert [op.n_stages for op in convs] == [3, 3] + [3, 4] * 5   # extra GELU after each block's second conv


def test_oracle_legacy_forward_runs_on_real_weights():
    from jaeger_amd import legacy
    from oracle import legacy as ol
    w = legacy.load_legacy_h5(H5)
    ids = np.random.Generator(np.random.PCG64(0)).integers(0, 22, (3, 6, 665))
    out = ol.forward(w, ids)
    assert out["output"].shape == (3, 4) and out["embedding"].shape == (3, 128)
    assert np.isfinite(out["output"]).all() and (out["embedding"] >= -0.2).all()   # GELU range


def test_postprocess_legacy_tsv_identical_to_reference(tmp_path):
    import joblib

    from jaeger_amd.postprocess_legacy import pred_to_dict_legacy, write_output_legacy
    z = np.load(GOLDEN / "postprocess_legacy_input.npz", allow_pickle=True)
    y = {"y_hat": {"output": z["output"], "embedding": z["embedding"]}, "meta": [z[f"meta_{i}"] for i in range(10)]}
    conf = json.loads((LEGACY / "config.json").read_text())["default"]
    conf["model"] = "default"
    conf["labels"] = [v for _, v in conf["default_labels"].items()]
    ood = {"type": "sklearn", "model": joblib.load(LEGACY / "models/default/LR_ood_4_class_default.pkl"),
           "batch_mean": np.load(LEGACY / "models/default/batch_means.npy"),
           "batch_std": np.load(LEGACY / "models/default/batch_std.npy")}
    rep = pd.read_csv(GOLDEN / "postprocess_legacy_repeats.csv")
    data, _ = pred_to_dict_legacy(conf, y, model="default", fsize=2000, ood_params=ood, term_repeats=rep)
    write_output_legacy(conf, data, output_table_path=tmp_path / "a.tsv", output_phage_table_path=tmp_path / "b.tsv",
                        reliability_cutoff=0.1, phage_score=3)
    assert (tmp_path / "a.tsv").read_bytes() == (GOLDEN / "postprocess_legacy.tsv").read_bytes()
    assert (tmp_path / "b.tsv").read_bytes() == (GOLDEN / "postprocess_legacy_phages.tsv").read_bytes()


def test_cli_default_model_needs_data_dir(tmp_path, monkeypatch):
    from jaeger_amd.cli import main
    monkeypatch.delenv("JAEGER_DATA", raising=False)
    r = CliRunner().invoke(main, ["predict", "-i", str(GOLDEN / "test_contigs.fasta"), "-o", str(tmp_path / "o")])
    assert r.exit_code != 0 and isinstance(r.exception, FileNotFoundError)
    out = tmp_path / "o" / "default"
    out.mkdir(parents=True)
    (out / "test_contigs_jaeger.tsv").write_text("x")
    r = CliRunner().invoke(main, ["predict", "-i", str(GOLDEN / "test_contigs.fasta"), "-o", str(tmp_path / "o"),
                                  "--legacy-data", str(LEGACY)])
    assert r.exit_code == 1 and (out / "test_contigs_jaeger.tsv").read_text() == "x"
