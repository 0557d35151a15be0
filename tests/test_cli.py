"""CLI / run_core plumbing that does not need a GPU (cli.py:122-370, commands/predict.py:488-620)."""
import numpy as np
import pytest
from click.testing import CliRunner
from conftest import GOLDEN, make_model_dir

from jaeger_amd import predict as P
from jaeger_amd.cli import main


def test_model_id_and_registry(tmp_path):
    root = make_model_dir(tmp_path)
    info = P.AvailableModels(root).info
    assert list(info) == ["jaeger_38341_1.4M_fragment"]
    entry = info["jaeger_38341_1.4M_fragment"]
    assert {"project", "classes", "weights_npz"} <= set(entry)
    assert P.get_model_id("jaeger_38341_1.4M_fragment") == "38341_1.4M"


def test_validate_fasta_entries():
    assert P.validate_fasta_entries(str(GOLDEN / "test_contigs.fasta"), 2000) == 9
    with pytest.raises(Exception, match="< 2000bp"):
        P.validate_fasta_entries(str(GOLDEN / "test_short.fasta"), 2000)
    with pytest.raises(Exception):
        P.validate_fasta_entries(str(GOLDEN / "test_empty.fasta"), 1)


def test_crop_length_warning():
    assert P._crop_length_warning(498, 1500, 1500) is None
    msg = P._crop_length_warning(498, 1500, 2000)
    assert "665 codon frames" in msg and "498 codons (1500 nt)" in msg
    assert P._crop_length_warning(None, 1500, 2000) is not None
    assert P._crop_length_warning(None, None, 2000) is None


def test_concat_predictions():
    a = {"prediction": np.ones((2, 3)), "meta_0": np.array(["a", "b"])}
    assert P._concat_predictions({}, a) is a and P._concat_predictions(a, {}) is a
    c = P._concat_predictions(a, a)
    assert c["prediction"].shape == (4, 3) and list(c["meta_0"]) == ["a", "b", "a", "b"]


def test_cli_defaults_and_required():
    r = CliRunner().invoke(main, ["predict"])
    assert r.exit_code != 0 and "--input" in r.output
    r = CliRunner().invoke(main, ["predict", "--help"])
    for flag in ("--fsize", "--stride", "--dynamic-stride", "--min-len", "--model_path", "--rc", "--pc",
                 "--window-scores", "--save-embedding", "--save-nmd", "--overwrite", "--no-dustmask", "--crf",
                 "--crf-switch-cost", "--crf-prior", "--crf-transition-matrix"):
        assert flag in r.output


@pytest.mark.parametrize("flag", ["--cpu", "--onnx", "--refine", "--quantized"])
def test_cli_rejects_out_of_scope_flags(tmp_path, flag):
    root = make_model_dir(tmp_path / "m")
    r = CliRunner().invoke(main, ["predict", "-i", str(GOLDEN / "test_contigs.fasta"), "-o", str(tmp_path / "out"),
                                  "--model_path", str(root), "--fsize", "1500", flag])
    assert r.exit_code == 1
    logs = list((tmp_path / "out" / "38341_1.4M").glob("*_jaeger.log"))
    assert logs and "not available on the MI355X predict path" in logs[0].read_text()


def test_cli_refuses_existing_output(tmp_path):
    root = make_model_dir(tmp_path / "m")
    out = tmp_path / "out" / "38341_1.4M"
    out.mkdir(parents=True)
    (out / "test_contigs.tsv").write_text("x")
    r = CliRunner().invoke(main, ["predict", "-i", str(GOLDEN / "test_contigs.fasta"), "-o", str(tmp_path / "out"),
                                  "--model_path", str(root), "--fsize", "1500"])
    assert r.exit_code == 1 and (out / "test_contigs.tsv").read_text() == "x"
