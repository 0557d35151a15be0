"""Nucleotide two-strand model (train_config/nn_config_500bp_dvf.yaml) on the GPU against oracle/strands.py: the encoder bit
for bit, logits and embedding within 1e-4, the id-tensor entry point, ragged / N-holding windows, merge methods, the
streamed host pipeline, and the CLI end to end."""
import copy

import numpy as np
import pytest
from conftest import load_model_cfg, make_model_dir

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _dna(rng, n, n_frac=0.0, lower_frac=0.0):
    seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)].copy()
    if n_frac:
        seq[rng.random(n) < n_frac] = ord("N")
    if lower_frac:
        low = rng.random(n) < lower_frac
        seq[low] |= 0x20
    return seq


def test_nucleotide_encoder_bit_exact():
    from jaeger_amd import _lib as L
    from jaeger_amd.engine import HipDevice
    from oracle import encoder as oenc
    from oracle import strands as ost
    rng = np.random.default_rng(5)
    fsize = 500
    lens = np.array([500, 500, 123, 10, 1, 499, 500, 37], np.int32)
    seq = _dna(rng, int(lens.sum()) + 700, n_frac=0.02, lower_frac=0.3)
    seq[40:43] = np.frombuffer(b"RY-", np.uint8)
    starts = (np.cumsum(lens) - lens).astype(np.int64)
    starts[6] += 100                                     # a window that is not at a record start
    dev = HipDevice(0)
    windows = [seq[s:s + n].tobytes() for s, n in zip(starts, lens)]
    for flags in (0, 1, 2, 3):                           # case flags change the COUNTS, never the nucleotide ids
        ids, counts = dev.encode(seq, starts, lens, fsize, np.zeros(65, np.uint8), flags=flags | L.JG_ENC_NUCLEOTIDE)
        assert ids.shape == (len(lens), 2, fsize)
        np.testing.assert_array_equal(ids, ost.encode_nucleotide(windows, fsize, pad_to=fsize))
        cased = [w if flags & 1 else w.upper() for w in windows]
        np.testing.assert_array_equal(counts, np.array([oenc.window_counts(w) for w in cased], np.int32))
    # a crop shorter than the windows: the reverse strand is the reverse complement of the CROPPED bases
    ids, _ = dev.encode(seq, starts[:2], lens[:2], 200, np.zeros(65, np.uint8), flags=L.JG_ENC_NUCLEOTIDE)
    np.testing.assert_array_equal(ids, ost.encode_nucleotide(windows[:2], 200))
    dev.close()


def _case(cfg, n_win, fsize, seed, short=False, n_frac=0.01, chunk=0, precision=None, lds=False):
    from jaeger_amd.engine import JaegerHipEngine
    from oracle import encoder as oenc
    from oracle import strands as ost
    weights = ost.random_weights(cfg, seed=38341)
    rng = np.random.Generator(np.random.PCG64(seed))
    seq = _dna(rng, fsize * n_win, n_frac=n_frac, lower_frac=0.1)
    starts = (np.arange(n_win) * fsize).astype(np.int64)
    lens = np.full(n_win, fsize, np.int32)
    if short:
        lens[1::3] = rng.integers(fsize // 2, fsize, lens[1::3].size)
    with pytest.warns(UserWarning, match="embedding.type is not 'nucleotide'"):
        eng = JaegerHipEngine(model_cfg=cfg, weights=weights, device_id=0, chunk=chunk, precision=precision)
    assert eng.string_processor_config["input_type"] == "nucleotide" and eng.model.strands == 2
    eng.device.set_table_net_lds(lds)        # the strand branch on the LDS-table kernel instead of the matrix cores
    got = eng.predict_windows(seq, starts, lens, fsize)
    windows = [seq[s:s + n].tobytes() for s, n in zip(starts, lens)]
    ids = ost.encode_nucleotide(windows, fsize, pad_to=fsize)
    ref = ost.forward(cfg, weights, ids)
    got2 = eng.model.forward(ids, chunk=chunk)             # the id-tensor entry point: bit for bit the fused one
    mode = eng.model.precision
    eng.close()
    assert "reliability" not in got and "nmd" not in got
    for k, r in ref.items():
        assert got[k].shape == r.shape, k
        err = float(np.abs(got[k] - r).max())
        print("dvf500", fsize, mode, k, f"{err:.2e}")
        assert err <= TOL, (k, err)
        np.testing.assert_array_equal(got[k], got2[k])
    np.testing.assert_array_equal(got["counts"], np.array([oenc.window_counts(w.upper()) for w in windows], np.int32))
    return got


@pytest.mark.parametrize("lds", [False, True])
def test_forward_dvf500_vs_oracle(lds):
    _case(load_model_cfg("dvf500"), 96, 500, seed=1, lds=lds)


@pytest.mark.parametrize("lds", [False, True])
def test_forward_dvf500_ragged_windows_and_chunks(lds):
    """Windows shorter than the crop are zero padded to the row length (what padded_batch does to a batch whose longest
    window is fsize); three passes of 40 windows."""
    _case(load_model_cfg("dvf500"), 100, 500, seed=2, short=True, n_frac=0.05, chunk=40, lds=lds)


def test_forward_dvf500_exact_f32_and_other_fsize():
    _case(load_model_cfg("dvf500"), 32, 300, seed=3, precision="f32")


def test_forward_dvf500_long_rows_take_the_lds_form():
    """Rows of 1 200 bases do not fit the matrix-core kernel's prefetched id image (1 024 bytes): the LDS-table kernel runs."""
    _case(load_model_cfg("dvf500"), 12, 1200, seed=10, short=True)


@pytest.mark.parametrize("method", ["sum", "max", "concat"])
def test_forward_dvf500_merge_methods(method):
    cfg = copy.deepcopy(load_model_cfg("dvf500"))
    cfg["classifier"]["branch"]["hidden_layers"][-1]["config"]["method"] = method
    _case(cfg, 24, 500, seed=4)


@pytest.mark.parametrize("lds", [False, True])
def test_dvf500_average_pool_and_gelu_branch(lds):
    cfg = copy.deepcopy(load_model_cfg("dvf500"))
    cfg["representation_learner"]["branch"] = {"hidden_layers": [
        {"name": "conv1d", "config": {"filters": 64, "kernel_size": 7, "activation": "gelu"}}], "pooling": "average1d"}
    cfg["classifier"]["branch"]["hidden_layers"] = [
        {"name": "dense", "config": {"units": 16}}, {"name": "tanh"}, {"name": "dense", "config": {"units": 3}},
        {"name": "merge", "config": {"method": "average"}}]
    _case(cfg, 40, 500, seed=6, lds=lds)


@pytest.mark.parametrize("lds", [False, True])
def test_dvf500_same_padding_dilated_branch(lds):
    """'same' padding and a dilated kernel: taps outside the strand add nothing (table-net kernel's pad_left path)."""
    cfg = copy.deepcopy(load_model_cfg("dvf500"))
    cfg["representation_learner"]["branch"] = {"hidden_layers": [
        {"name": "conv1d", "config": {"filters": 100, "kernel_size": 6, "padding": "same", "dilation_rate": 3}},
        {"name": "sigmoid"}], "pooling": "max1d"}
    cfg["classifier"]["branch"]["hidden_layers"] = [
        {"name": "dense", "config": {"units": 3}}, {"name": "merge", "config": {"method": "average"}}]
    _case(cfg, 40, 400, seed=7, short=True, lds=lds)


def test_dvf_wide_conv_runs_layer_by_layer():
    """800 filters x 10 taps do not fit the table-net kernel's LDS image: the same program runs conv -> pool on the
    exact-f32 kernels (the path every strand model took before that kernel existed)."""
    from jaeger_amd.engine import JaegerHipEngine
    from oracle import strands as ost
    cfg = copy.deepcopy(load_model_cfg("dvf500"))
    cfg["representation_learner"]["branch"]["hidden_layers"][0]["config"]["filters"] = 800
    for layer in cfg["classifier"]["branch"]["hidden_layers"]:
        if layer["name"] == "dense" and layer["config"]["units"] == 500:
            layer["config"]["units"] = 64
    _case(cfg, 16, 500, seed=8)
    with pytest.warns(UserWarning):
        eng = JaegerHipEngine(model_cfg=cfg, weights=ost.random_weights(cfg), device_id=0)
    assert "table-net" not in eng.model.describe()
    eng.close()
    with pytest.warns(UserWarning):
        eng = JaegerHipEngine(model_cfg=load_model_cfg("dvf500"), weights=ost.random_weights(load_model_cfg("dvf500")), device_id=0)
    assert "table-net" in eng.model.describe()
    eng.close()


def test_dvf500_streamed_pipeline_equals_one_call(tmp_path):
    """Host buffers above the span budget go through the two-deep pipeline: same rows as the resident call."""
    from jaeger_amd.engine import JaegerHipEngine
    from oracle import strands as ost
    cfg = load_model_cfg("dvf500")
    weights = ost.random_weights(cfg)
    rng = np.random.default_rng(9)
    n_win, fsize = 3000, 500
    seq = _dna(rng, n_win * fsize, n_frac=0.01)
    starts = (np.arange(n_win) * fsize).astype(np.int64)
    lens = np.full(n_win, fsize, np.int32)
    with pytest.warns(UserWarning):
        eng = JaegerHipEngine(model_cfg=cfg, weights=weights, device_id=0)
    eng.device.set_stream_bytes(1 << 30)
    whole = eng.predict_windows(seq, starts, lens, fsize)
    assert eng.device.stream_stats()["groups"] == 0
    eng.device.set_stream_bytes(200_000)
    streamed = eng.predict_windows(seq, starts, lens, fsize)
    groups = eng.device.stream_stats()["groups"]
    eng.close()
    assert groups >= 4
    for k in whole:
        np.testing.assert_array_equal(whole[k], streamed[k])


@pytest.mark.parametrize("dustmask", [False, True])
def test_cli_dvf500_end_to_end(tmp_path, dustmask):
    """`jaeger predict` with the two-strand model: TSV of the GPU run against the oracle pipeline's per-window logits.
    With DUST on, soft-masked bases leave the G / C / A / T counts but not the nucleotide ids (both cases are keys of the
    lookup, encode.py:36-41): same calls."""
    import pandas as pd

    from jaeger_amd import predict as P
    from jaeger_amd.weights import load_npz
    from oracle import strands as ost
    rng = np.random.default_rng(12)
    lengths = [2600, 500, 1234, 800, 5000]
    fa = tmp_path / "in.fasta"
    seqs = [_dna(rng, n, n_frac=0.003).tobytes() for n in lengths]
    seqs[0] = seqs[0][:700] + b"A" * 300 + b"AC" * 100 + seqs[0][1200:]          # low complexity: DUST masks it
    fa.write_bytes(b"".join(b">c%d some text\n%s\n" % (i, s) for i, s in enumerate(seqs)))
    mdir = make_model_dir(tmp_path / "m", name="dvf500", model_name="jaeger_500bp_dvf")
    P.run_core(input=str(fa), output=str(tmp_path / "out"), model_path=str(mdir), fsize=500, stride=500, overwrite=True,
               dustmask=dustmask, verbose=0, batch=96, rc=0.1, pc=3)
    tsv = next((tmp_path / "out").rglob("in.tsv"))
    df = pd.read_csv(tsv, sep="\t")
    assert df["contig_id"].tolist() == [f"c{i}" for i in range(5)]
    assert (df["reliability_score"] == "unavailable").all()
    cfg = load_model_cfg("dvf500")
    weights = load_npz(next(mdir.rglob("*.weights.npz")))
    labels = [c["class"] for c in cfg["class_label_map"]]
    for i, s in enumerate(seqs):
        wins = [s[p:p + 500] for p in range(0, len(s) - 499, 500)]
        logits = ost.forward(cfg, weights, ost.encode_nucleotide(wins, 500, pad_to=500))["prediction"]
        calls = logits.argmax(1)
        for c, name in enumerate(labels):
            assert int(df.loc[i, f"#_{name}_windows"]) == int((calls == c).sum())
    gc0 = float(df.loc[0, "G+C"])
    if dustmask:
        assert gc0 != pytest.approx(test_cli_dvf500_end_to_end.gc_plain, abs=1e-4)      # masked bases are not counted
    else:
        test_cli_dvf500_end_to_end.gc_plain = gc0


def test_translated_unmasked_conv_then_pool_is_not_taken_by_the_table_net():
    """A six-frame model whose first conv is unmasked and feeds the pool directly has the op pattern of a strand branch;
    its pool runs over frames x positions, so it must stay on the layer-by-layer kernels."""
    from jaeger_amd.engine import JaegerHipEngine, frame_length
    from oracle import encoder as oenc
    from oracle import forward as ofwd
    cfg = copy.deepcopy(load_model_cfg("baseline500"))
    cfg["use_masking"] = False
    cfg["representation_learner"] = {"hidden_layers": [
        {"name": "masked_conv1d", "config": {"filters": 32, "kernel_size": 7, "activation": "gelu", "use_masking": False}}],
        "pooling": "max"}
    cfg["classifier"]["input_shape"] = 32
    weights = ofwd.random_weights(cfg, seed=5)
    rng = np.random.default_rng(13)
    fsize, n_win = 500, 40
    seq = _dna(rng, fsize * n_win, n_frac=0.01)
    starts = (np.arange(n_win) * fsize).astype(np.int64)
    lens = np.full(n_win, fsize, np.int32)
    eng = JaegerHipEngine(model_cfg=cfg, weights=weights, device_id=0)
    assert "table-net" not in eng.model.describe()
    got = eng.predict_windows(seq, starts, lens, fsize)
    eng.close()
    ids = oenc.encode_windows([seq[s:s + fsize].tobytes() for s in starts], fsize, pad_to=frame_length(fsize))
    ref = ofwd.forward(cfg, weights, ids)
    assert float(np.abs(got["prediction"] - ref["prediction"]).max()) <= TOL
