"""The float forward pinned to a reference-held artefact: the reference's bundled SavedModel
(``src/jaeger/data/models/test/jaeger_fragment_graph``, committed as a data fixture) executed WITHOUT TensorFlow by
``oracle/graphdef.py``; golden logits in ``tests/golden/legacy_savedmodel_logits.npz`` (generator:
``tests/golden/make_golden_savedmodel.py``)."""
import numpy as np
import pytest

from conftest import GOLDEN

GRAPH = GOLDEN / "legacy_data" / "models" / "test" / "jaeger_fragment_graph"
H5 = GOLDEN / "legacy_data" / "models" / "default" / "WRes_1024.h5"
TOL = 1e-4


@pytest.fixture(scope="module")
def golden():
    return dict(np.load(GOLDEN / "legacy_savedmodel_logits.npz"))


def test_bundle_reader_matches_the_h5_weights():
    """The SavedModel's variable bundle (TF-free SSTable reader) holds the same 947 036 values as WRes_1024.h5."""
    from jaeger_amd import legacy
    from jaeger_amd import savedmodel_lite as S
    bundle = S.read_bundle(GRAPH / "variables")
    floats = {k: v for k, v in bundle.items() if v.dtype == np.float32 and k.endswith("VARIABLE_VALUE")}
    assert sum(v.size for v in floats.values()) == 947_036
    h5 = legacy.load_legacy_h5(H5)
    # Keras names differ; match the two sets by value (every tensor of one side appears bit for bit in the other)
    sigs = {(v.shape, v.tobytes()) for v in floats.values()}
    for name, arr in h5.items():
        assert (arr.shape, np.asarray(arr, np.float32).tobytes()) in sigs, name


def test_census_of_the_bundled_graph():
    """SURVEY Appendix D census, reproduced by the committed reader."""
    from jaeger_amd import savedmodel_lite as S
    c = S.census(GRAPH)
    assert c["n_nodes"] == 6324 and c["n_captured_variables"] == 79 and c["n_parameters"] == 947_036
    assert c["ops"]["Conv2D"] == 72 and c["ops"]["Erfc"] == 104 and c["ops"]["MaxPool"] == 12
    assert c["ops"]["SpaceToBatchND"] == 66 and c["ops"]["MatMul"] == 3 and c["ops"]["Max"] == 1
    assert c["gelu_form"] == "erf"
    assert list(c["batchnorm_eps"].values()) == [72] and abs(list(c["batchnorm_eps"])[0] - 1e-3) < 1e-9
    assert c["inputs"] == ["inputs", "inputs_1", "inputs_2", "inputs_3", "inputs_4", "inputs_5"]


def test_graph_interpreter_reproduces_golden(golden):
    """Re-executing the committed graph on a few windows gives the committed logits (the fixture is what it says)."""
    from oracle import graphdef
    ids = golden["ids"][[0, 67, 134]]
    feeds = {("inputs" if f == 0 else f"inputs_{f}"): ids[:, f, :].astype(np.float32) for f in range(6)}
    out = graphdef.run_saved_model(GRAPH, feeds)
    np.testing.assert_allclose(out["identity_1"], golden["output"][[0, 67, 134]], atol=2e-6, rtol=0)
    np.testing.assert_allclose(out["identity"], golden["embedding"][[0, 67, 134]], atol=2e-6, rtol=0)


def test_legacy_oracle_pinned_to_the_reference_graph(golden):
    """oracle/legacy.py (the restatement every legacy GPU test is checked against) vs the reference's own graph +
    weights: conv SAME-pad split, dilations, bias, exact-erf GELU, batch norm eps 1e-3, MaxPool, frame sum, global max,
    dense stack - within 1e-4 of the graph's exact (f64) value on all 135 windows."""
    from jaeger_amd import legacy
    from oracle import legacy as ol
    w = legacy.load_legacy_h5(H5)
    ref = ol.forward(w, golden["ids"])
    e_out = float(np.abs(ref["output"] - golden["output_f64"]).max())
    e_emb = float(np.abs(ref["embedding"] - golden["embedding_f64"]).max())
    print(f"oracle/legacy.py vs SavedModel (f64): logits {e_out:.2e} (range {np.abs(golden['output_f64']).max():.1f}), "
          f"embedding {e_emb:.2e}")
    assert e_out < TOL and e_emb < TOL
    # and the graph's own f32 evaluation is as far from its f64 value as the oracle is: the gate is rounding-level
    assert float(np.abs(golden["output"] - golden["output_f64"]).max()) < TOL


def test_modern_conv_semantics_match_tf_ops():
    """oracle/forward.py's conv (SAME / VALID, stride, dilation) against the TF op chain the SavedModel interpreter
    implements (ExpandDims -> [SpaceToBatchND] -> Conv2D -> [BatchToSpaceND]), the way tf.nn.conv1d lowers."""
    import torch
    from oracle import forward as ofwd
    from oracle import graphdef as G

    class N:                       # a stand-in NodeDef with the attributes _conv2d reads
        def __init__(self, padding, strides):
            self.p, self.s = padding, strides
        def attr_s(self, k, d=None):
            return {"padding": self.p, "data_format": "NHWC"}.get(k, d)
        def attr_ints(self, k):
            return {"strides": [1, 1, self.s, 1], "dilations": [1, 1, 1, 1]}.get(k, [])

    rng = np.random.default_rng(0)
    for L, k, s, d, pad in [(50, 5, 1, 3, "SAME"), (51, 7, 1, 1, "SAME"), (40, 3, 2, 1, "SAME"), (41, 3, 2, 1, "SAME"),
                            (33, 1, 2, 1, "SAME"), (50, 5, 1, 2, "VALID"), (30, 9, 1, 1, "VALID")]:
        x = rng.standard_normal((3, L, 6)).astype(np.float32)
        w = rng.standard_normal((k, 6, 4)).astype(np.float32)
        got = ofwd.conv1d_nwc(torch.from_numpy(x), torch.from_numpy(w), s, pad, d).numpy()
        x4, w4 = x[:, None, :, :], w[None, :, :, :]
        if d > 1:                   # tf.nn.convolution lowers dilation to SpaceToBatchND / VALID conv / BatchToSpaceND
            assert s == 1
            span = (k - 1) * d
            pl, pr = (span // 2, span - span // 2) if pad == "SAME" else (0, 0)
            Lp = L + pl + pr
            extra = (-Lp) % d
            xb = G._space_to_batch(x4[:, 0], [d], [[pl, pr + extra]])
            yb = G._conv2d(xb[:, None], w4, N("VALID", 1))[:, 0]
            ref = G._batch_to_space(yb, [d], [[0, extra]])
        else:
            ref = G._conv2d(x4, w4, N(pad, s))[:, 0]
        assert got.shape == ref.shape, (L, k, s, d, pad, got.shape, ref.shape)
        np.testing.assert_allclose(got, ref, atol=1e-5, rtol=0)


def test_fast_cpu_mode_agrees_with_the_checker():
    """bench.py's cpu_baseline runs oracle.forward with FAST = True (oneDNN convs, fused GELU): same results."""
    from conftest import load_model_cfg
    from oracle import forward as ofwd
    cfg = load_model_cfg("brain")
    w = ofwd.random_weights(cfg, seed=38341)
    ids = np.random.default_rng(1).integers(0, 65, (4, 6, 120)).astype(np.uint8)
    ids[:, :, 100:] = 0
    ref = ofwd.forward(cfg, w, ids)
    ofwd.FAST = True
    try:
        fast = ofwd.forward(cfg, w, ids)
    finally:
        ofwd.FAST = False
    for k in ref:
        assert float(np.abs(ref[k] - fast[k]).max()) < 1e-4, k


def test_verify_model_accepts_the_matching_plan_and_flags_others():
    from conftest import load_model_cfg
    from jaeger_amd.plan import build_plan
    from jaeger_amd.verify import verify_model
    assert verify_model(GRAPH, legacy=True) == []
    findings = verify_model(GRAPH, build_plan(load_model_cfg("brain")))
    text = " ".join(findings)
    assert "variable shapes differ" in text and "GELU form" in text and "epsilons differ" in text
    assert "no mask comparison ops" in text


def test_legacy_bundle_loader_equals_the_h5_loader():
    """Weights straight out of ``<name>_graph/variables`` (Keras-3 ``_operations/<n>/<attribute>`` keys, mapped by
    object-graph order and variable names) are the ``WRes_1024.h5`` values tensor for tensor, bit for bit."""
    from jaeger_amd import legacy
    wb = legacy.load_legacy_bundle(GRAPH)
    wh = legacy.load_legacy_h5(GRAPH.parents[2] / "models" / "default" / "WRes_1024.h5"
                               if (GRAPH.parents[2] / "models" / "default" / "WRes_1024.h5").exists()
                               else next(GRAPH.parents[2].rglob("WRes_1024.h5")))
    assert list(wb) == list(wh) and len(wb) == 79
    for k in wb:
        np.testing.assert_array_equal(wb[k], wh[k])


@pytest.mark.parametrize("name", ["baseline500", "brain", "pyramid", "zeus"])
def test_hand_built_keras3_bundle_round_trips_through_the_loader(tmp_path, name):
    """A variable bundle written with the repo's own SSTable writer under the Keras-3 key scheme
    (``_operations/<n>[/convK|bnK]/_kernel|bias|gamma|...``) loads back to the canonical weights by order and names -
    including through ``load_weights`` / the engine's path_dict route, which prefers the graph's bundle to a weights file."""
    from conftest import load_model_cfg
    from jaeger_amd import savedmodel_lite as S
    from jaeger_amd.plan import build_plan
    from jaeger_amd.weights import (bundle_checkpoint_keys, load_savedmodel_bundle, load_weights, random_weights,
                                    save_npz)
    plan = build_plan(load_model_cfg(name))
    w = random_weights(plan, seed=11)
    keys = bundle_checkpoint_keys(plan)
    assert set(keys) == set(w)
    graph = tmp_path / f"{name}_graph"
    tensors = {keys[k]: v for k, v in w.items()}
    tensors["_CHECKPOINTABLE_OBJECT_GRAPH_SIZE"] = np.array([len(tensors)], np.int64)      # a non-float entry is skipped
    S.write_bundle(graph / "variables", tensors)
    assert S.crc32c(b"123456789") == 0xE3069283
    back = load_savedmodel_bundle(graph, plan)
    assert set(back) == set(w)
    for k in w:
        np.testing.assert_array_equal(back[k], w[k])
    # the model-entry route: the bundle wins over a (here: different) weights file
    other = tmp_path / "other.weights.npz"
    save_npz(other, random_weights(plan, seed=12))
    got = load_weights({"graph": graph, "weights_npz": other}, plan)
    for k in w:
        np.testing.assert_array_equal(got[k], w[k])


def test_bundle_loader_refuses_instead_of_guessing(tmp_path):
    """A bundle whose layers do not line up with the plan - a missing layer, swapped variable names, a wrong shape - is
    an error that names both sides, never a best-effort assignment."""
    from conftest import load_model_cfg
    from jaeger_amd import savedmodel_lite as S
    from jaeger_amd.plan import build_plan
    from jaeger_amd.weights import bundle_checkpoint_keys, load_savedmodel_bundle, random_weights
    plan = build_plan(load_model_cfg("baseline500"))
    w = random_weights(plan, seed=11)
    keys = bundle_checkpoint_keys(plan)
    base = {keys[k]: v for k, v in w.items()}

    def write(tensors, sub):
        S.write_bundle(tmp_path / sub / "variables", tensors)
        return tmp_path / sub

    first_bias = next(k for k in sorted(base) if k.endswith("/bias/.ATTRIBUTES/VARIABLE_VALUE"))
    missing = {k: v for k, v in base.items() if k != first_bias}
    with pytest.raises(ValueError, match="holds|weighted layers"):
        load_savedmodel_bundle(write(missing, "a"), plan)
    renamed = {k.replace("/gamma/", "/scale/"): v for k, v in base.items()}
    with pytest.raises(ValueError, match="holds"):
        load_savedmodel_bundle(write(renamed, "b"), plan)
    kern = next(k for k in sorted(base) if k.endswith("/_kernel/.ATTRIBUTES/VARIABLE_VALUE") and base[k].ndim == 3)
    reshaped = dict(base)
    reshaped[kern] = np.zeros(base[kern].shape[:-1] + (base[kern].shape[-1] + 1,), np.float32)
    with pytest.raises(ValueError, match="shape"):
        load_savedmodel_bundle(write(reshaped, "c"), plan)


def test_unmappable_bundle_falls_back_to_the_weights_file_loudly(tmp_path):
    """a bundle under another key scheme (here ``layers/...`` object paths and a renamed variable) must not
    turn a model directory that loaded from its weights file into a hard error: ``load_weights`` falls back to the file
    with a warning that names the failure; without a file the error stands; ``trust_project`` skips the bundle."""
    from conftest import load_model_cfg
    from jaeger_amd import savedmodel_lite as S
    from jaeger_amd.plan import build_plan
    from jaeger_amd.weights import bundle_checkpoint_keys, load_weights, random_weights, save_npz
    plan = build_plan(load_model_cfg("baseline500"))
    w_bundle, w_file = random_weights(plan, seed=11), random_weights(plan, seed=12)
    keys = bundle_checkpoint_keys(plan)
    odd = {keys[k].replace("_operations/", "layers/functional/layers/").replace("/gamma/", "/scale/"): v
           for k, v in w_bundle.items()}
    S.write_bundle(tmp_path / "m_graph" / "variables", odd)
    npz = tmp_path / "m.weights.npz"
    save_npz(npz, w_file)
    with pytest.warns(RuntimeWarning, match="could not be mapped onto the layer plan"):
        got = load_weights({"graph": tmp_path / "m_graph", "weights_npz": npz}, plan)
    for k in w_file:
        np.testing.assert_array_equal(got[k], w_file[k])
    with pytest.raises(ValueError):
        load_weights({"graph": tmp_path / "m_graph"}, plan)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        got = load_weights({"graph": tmp_path / "m_graph", "weights_npz": npz}, plan, trust_project=True)
    for k in w_file:
        np.testing.assert_array_equal(got[k], w_file[k])


def test_understood_bundle_that_disagrees_with_the_plan_stays_fatal(tmp_path, caplog):
    """the file fallback is for an unknown key SCHEME only.  A bundle under the known scheme whose variables
    disagree with the plan (a wrong shape, a missing layer) must refuse even when a weights file sits beside it - the
    SavedModel is what the reference executes, a file that differs from it would predict something else - unless
    ``trust_project`` says so; and the scheme fallback's warning also goes to the run log."""
    import logging

    from conftest import load_model_cfg
    from jaeger_amd import savedmodel_lite as S
    from jaeger_amd.plan import build_plan
    from jaeger_amd.weights import BundleSchemeError, bundle_checkpoint_keys, load_weights, random_weights, save_npz
    plan = build_plan(load_model_cfg("baseline500"))
    w = random_weights(plan, seed=11)
    keys = bundle_checkpoint_keys(plan)
    base = {keys[k]: v for k, v in w.items()}
    kern = next(k for k in sorted(base) if k.endswith("/_kernel/.ATTRIBUTES/VARIABLE_VALUE") and base[k].ndim == 3)
    reshaped = dict(base)
    reshaped[kern] = np.zeros(base[kern].shape[:-1] + (base[kern].shape[-1] + 1,), np.float32)
    S.write_bundle(tmp_path / "a_graph" / "variables", reshaped)
    npz = tmp_path / "m.weights.npz"
    save_npz(npz, w)
    with pytest.raises(ValueError, match="shape") as ei:
        load_weights({"graph": tmp_path / "a_graph", "weights_npz": npz}, plan)
    assert not isinstance(ei.value, BundleSchemeError)
    first_bias = next(k for k in sorted(base) if k.endswith("/bias/.ATTRIBUTES/VARIABLE_VALUE"))
    S.write_bundle(tmp_path / "b_graph" / "variables", {k: v for k, v in base.items() if k != first_bias})
    with pytest.raises(ValueError, match="holds|weighted layers"):
        load_weights({"graph": tmp_path / "b_graph", "weights_npz": npz}, plan)
    got = load_weights({"graph": tmp_path / "a_graph", "weights_npz": npz}, plan, trust_project=True)
    assert all(np.array_equal(got[k], w[k]) for k in w)
    # unknown scheme: falls back, and the run log hears about it
    odd = {k.replace("_operations/", "model/layer_with_weights-"): v for k, v in base.items()}
    S.write_bundle(tmp_path / "c_graph" / "variables", odd)
    with caplog.at_level(logging.WARNING, logger="Jaeger"), pytest.warns(RuntimeWarning):
        got = load_weights({"graph": tmp_path / "c_graph", "weights_npz": npz}, plan)
    assert any("could not be mapped onto the layer plan" in r.getMessage() for r in caplog.records)
    assert all(np.array_equal(got[k], w[k]) for k in w)
    # the reference's own file (.weights.h5) outranks the derived .npz when both are present
    from jaeger_amd.weights import load_keras3_h5  # noqa: F401
    from conftest import GOLDEN
    w500 = random_weights(plan, seed=500)
    got = load_weights({"weights": GOLDEN / "baseline500_keras3.weights.h5", "weights_npz": npz}, plan)
    assert all(np.array_equal(got[k], w500[k]) for k in w500)


def _merge_cfg(mode):
    from test_plan_program import _nmd_merge_builder_config
    cfg = _nmd_merge_builder_config()
    cfg["reliability_model"] = {"merge": {"mode": mode, "target_dim": 8}, "input_shape": 8,
                                "hidden_layers": [{"name": "dense", "config": {"units": 1, "activation": None}}]}
    return cfg


@pytest.mark.parametrize("mode", ["sum", "max", "weighted"])
def test_nmd_merge_variables_load_from_a_bundle(tmp_path, mode):
    """a model whose NMD vectors are merged by projections (``reliability_model.merge.mode`` sum / mean /
    max / weighted, nmd.py:93-155) can be loaded from the reference's containers, not only from random weights: the bundle
    holds ``layer_weights`` on the NMDMerge operation and the bias-free kernels under its ``projections/<i>``."""
    from jaeger_amd import savedmodel_lite as S
    from jaeger_amd.plan import build_plan
    from jaeger_amd.weights import _layer_order, bundle_checkpoint_keys, load_savedmodel_bundle, load_weights, random_weights
    plan = build_plan(_merge_cfg(mode))
    w = random_weights(plan, seed=5)
    assert "rep/nmd_merge/proj_1/kernel" in w and ("rep/nmd_merge/layer_weights" in w) == (mode == "weighted")
    order = [p for p, _ in _layer_order(plan)]
    assert order.index("rep/nmd_merge/proj_0") > order.index("rep/3") and order.index("rep/nmd_merge/proj_1") < order.index("classifier/0")
    keys = bundle_checkpoint_keys(plan)
    assert set(keys) == set(w)
    merge_keys = sorted(v for k, v in keys.items() if k.startswith("rep/nmd_merge"))
    op = merge_keys[0].split("/")[1]
    assert all(k.split("/")[1] == op for k in merge_keys)                          # ONE operation of the graph
    assert f"_operations/{op}/projections/1/_kernel/.ATTRIBUTES/VARIABLE_VALUE" in merge_keys
    if mode == "weighted":
        assert f"_operations/{op}/layer_weights/.ATTRIBUTES/VARIABLE_VALUE" in merge_keys
    graph = tmp_path / "m_graph"
    S.write_bundle(graph / "variables", {keys[k]: v for k, v in w.items()})
    back = load_savedmodel_bundle(graph, plan)
    assert set(back) == set(w)
    for k in w:
        np.testing.assert_array_equal(back[k], w[k])
    got = load_weights({"graph": graph}, plan)                                      # no BundleSchemeError, no fallback
    np.testing.assert_array_equal(got["rep/nmd_merge/proj_0/kernel"], w["rep/nmd_merge/proj_0/kernel"])


@pytest.mark.parametrize("mode", ["mean", "weighted"])
def test_nmd_merge_variables_load_from_a_weights_h5(monkeypatch, mode):
    """The same from a Keras-3 ``.weights.h5`` as ``save_weights`` lays it out (layers numbered per container; the merge
    layer's own variable under ``layers/<merge>/vars/0``, its projections as the container ``projections`` holding
    ``dense``, ``dense_1``): the datasets as ``hdf5_lite.read_datasets`` returns them (no HDF5 writer in the image - the
    container itself is covered by the committed fixtures), a classifier ``dense`` of the SAME shape as a projection
    included: per-container names keep path order, so nothing is swapped."""
    from jaeger_amd import hdf5_lite
    from jaeger_amd.plan import build_plan
    from jaeger_amd.weights import load_keras3_h5, random_weights
    cfg = _merge_cfg(mode)
    cfg["classifier_out_dim"] = 8                  # classifier Dense(8 -> 8, bias) next to proj_1 (8 -> 8, no bias)
    cfg["class_label_map"] = [{"class": f"c{i}", "label": i} for i in range(8)]
    cfg["classifier"]["hidden_layers"] = [{"name": "dense", "config": {"units": 8, "activation": None, "use_bias": False}},
                                          {"name": "dense", "config": {"units": 8, "activation": None, "use_bias": False}}]
    plan = build_plan(cfg)
    w = random_weights(plan, seed=9)
    assert w["rep/nmd_merge/proj_1/kernel"].shape == w["classifier/0/kernel"].shape == w["classifier/1/kernel"].shape == (8, 8)
    d = {"/layers/functional/layers/embedding/vars/0": w["embedding/embeddings"],
         "/layers/functional/layers/masked_conv1d/vars/0": w["rep/0/kernel"], "/layers/functional/layers/masked_conv1d/vars/1": w["rep/0/bias"],
         "/layers/functional/layers/nmd_layer/vars/0": w["rep/1/moving_mean"],
         "/layers/functional/layers/masked_conv1d_1/vars/0": w["rep/3/kernel"], "/layers/functional/layers/masked_conv1d_1/vars/1": w["rep/3/bias"],
         "/layers/functional/layers/nmd_layer_1/vars/0": w["rep/4/moving_mean"],
         "/layers/functional/layers/rep_nmd_merge/projections/dense/vars/0": w["rep/nmd_merge/proj_0/kernel"],
         "/layers/functional/layers/rep_nmd_merge/projections/dense_1/vars/0": w["rep/nmd_merge/proj_1/kernel"],
         "/layers/functional_1/layers/dense/vars/0": w["classifier/0/kernel"],
         "/layers/functional_1/layers/dense_1/vars/0": w["classifier/1/kernel"],
         "/layers/functional_2/layers/dense/vars/0": w["reliability/0/kernel"], "/layers/functional_2/layers/dense/vars/1": w["reliability/0/bias"]}
    if mode == "weighted":
        d["/layers/functional/layers/rep_nmd_merge/vars/0"] = w["rep/nmd_merge/layer_weights"]
    monkeypatch.setattr(hdf5_lite, "read_datasets", lambda path: dict(d))
    got = load_keras3_h5("model.weights.h5", plan)
    assert set(got) == set(w)
    for k in w:
        np.testing.assert_array_equal(got[k], w[k], err_msg=k)
    # globally numbered leaf names (every name once) still go name-major: the layout of the committed nested fixture
    d2 = {k.replace("functional_1/layers/dense_1", "functional_1/layers/dense_3").replace("functional_1/layers/dense/", "functional_1/layers/dense_2/")
           .replace("functional_2/layers/dense/", "functional_2/layers/dense_4/"): v for k, v in d.items()}
    monkeypatch.setattr(hdf5_lite, "read_datasets", lambda path: dict(d2))
    got = load_keras3_h5("model.weights.h5", plan)
    for k in w:
        np.testing.assert_array_equal(got[k], w[k], err_msg=k)
