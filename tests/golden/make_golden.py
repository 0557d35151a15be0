#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference; TensorFlow, pyfastx and
pydustmasker are absent there, so the two C-extension packages are replaced by tiny
in-process stubs - a 20-line FASTA iterator and an identity masker).  Nothing of the
reference travels: the outputs are data (inputs + expected outputs).

    python tests/golden/make_golden.py

Produces
  maps.json               codon / id tables                    (seqops/maps.py)
  crop.json               nt <-> codon-frame contract          (seqops/crop.py)
  fragments_<cfg>.json    fragment_generator fields per window (seqops/io.py) for the bundled
                          test FASTA at three (fsize, stride, dynamic) settings; window
                          sequences are kept as sha1 digests
  encoder_ids.json        6-frame codon ids of selected windows from the reference's second
                          statement of the encoder, dataops/convert.py:_process_batch_numba
  postprocess_*.tsv/json  pred_to_dict + write_output on seeded synthetic logits
                          (postprocess/collect.py)
  *.fasta                 the reference's bundled test FASTA files (test data)
  *_project.yaml          `model:` sections of the reference's train_config/*.yaml (configuration data): brain, zeus,
                          baseline500, nmdmerge500 = nn_config_500bp_nmd_merge.yaml and pyramid =
                          nn_config_baseline.yaml (its `training:` section is not valid YAML - a stray quote - so
                          the text is cut in front of it), written by
                          yaml.safe_dump({"model": cfg["model"]}, sort_keys=False) - see write_project_yamls()
"""
import hashlib
import json
import shutil
import sys
import types
from pathlib import Path

import numpy as np
import pandas as pd

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")
sys.path.insert(0, str(REF / "src"))


def write_project_yamls(reference_root, here):
    """The `model:` sections the tests compile (configuration data, not code)."""
    import yaml
    names = {"brain": "nn_config_1500bp_nmd_merge_6_class_brain.yaml", "zeus": "nn_config_1500bp_nmd_merge_6_class_zeus.yaml",
             "baseline500": "nn_config_500bp_baseline.yaml", "nmdmerge500": "nn_config_500bp_nmd_merge.yaml",
             "pyramid": "nn_config_baseline.yaml", "dvf500": "nn_config_500bp_dvf.yaml"}
    for name, fn in names.items():
        text = (reference_root / "train_config" / fn).read_text()
        cut = text.find("\ntraining:")
        cfg = yaml.safe_load(text[:cut] if cut > 0 else text)
        (here / f"{name}_project.yaml").write_text(yaml.safe_dump({"model": cfg["model"]}, sort_keys=False))


def _stub_modules():
    pyfastx = types.ModuleType("pyfastx")

    class Fasta:
        def __init__(self, path, build_index=False, **kw):
            self.path = path

        def __iter__(self):
            name, chunks = None, []
            with open(self.path) as fh:
                for line in fh:
                    line = line.rstrip("\n")
                    if line.startswith(">"):
                        if name is not None:
                            yield name, "".join(chunks)
                        name, chunks = line[1:].split()[0], []
                    elif name is not None:
                        chunks.append(line.strip())
            if name is not None:
                yield name, "".join(chunks)

    pyfastx.Fasta = Fasta
    sys.modules["pyfastx"] = pyfastx
    dust = types.ModuleType("pydustmasker")

    class DustMasker:
        def __init__(self, seq, **kw):
            self.seq = seq

        def mask(self):
            return self.seq

    dust.DustMasker = DustMasker
    sys.modules["pydustmasker"] = dust


def main():
    _stub_modules()
    write_project_yamls(REF, HERE)
    from jaeger.dataops import convert
    from jaeger.postprocess import collect
    from jaeger.seqops import crop, io, maps

    # ---- tables ------------------------------------------------------------------------
    (HERE / "maps.json").write_text(json.dumps({
        "CODONS": maps.CODONS, "CODON_ID": maps.CODON_ID, "AA_ID": maps.AA_ID,
        "MURPHY10_ID": maps.MURPHY10_ID, "PC5_ID": maps.PC5_ID,
        # the 4 096 codon pairs of `codon: DICODON` (maps.py:544-546): digest + ends instead of the whole list
        "DICODONS_SHA256": hashlib.sha256(",".join(maps.DICODONS).encode()).hexdigest(),
        "DICODONS_HEAD": maps.DICODONS[:5], "DICODONS_TAIL": maps.DICODONS[-3:], "DICODONS_LEN": len(maps.DICODONS),
        "DICODON_ID_IS_IDENTITY": maps.DICODON_ID == list(range(len(maps.DICODONS)))}, indent=0))
    (HERE / "crop.json").write_text(json.dumps({
        "tf_frame_length": {str(n): crop.tf_frame_length(n) for n in (2, 3, 5, 6, 7, 8, 9, 500, 1500, 1501,
                                                                        1502, 1505, 2000, 2048)},
        "nucleotides_to_codons": {str(n): crop.nucleotides_to_codons(n) for n in (500, 1500, 1505, 2000)},
        "codons_to_nucleotides": {str(n): crop.codons_to_nucleotides(n) for n in (165, 498, 500, 665)},
    }, indent=0))

    # ---- bundled test data ---------------------------------------------------------------
    for name in ("test_contigs.fasta", "test_short.fasta", "test_empty.fasta"):
        shutil.copyfile(REF / "src/jaeger/data/test" / name, HERE / name)
    fasta = str(HERE / "test_contigs.fasta")

    # ---- fragmenter ----------------------------------------------------------------------
    settings = {"1500_1500": dict(fragsize=1500, stride=1500),
                "2000_1500": dict(fragsize=2000, stride=1500),
                "2000_2000_dyn": dict(fragsize=2000, stride=2000, dynamic_stride=True,
                                      dynamic_stride_threshold=10.0),
                "4000_4000_min1000": dict(fragsize=40000, stride=40000, min_len=9000)}
    windows_for_encoder = {}
    for tag, kw in settings.items():
        rows = []
        for frag in io.fragment_generator(fasta, num=9, no_progress=True, dustmask=False, **kw):
            f = frag.split(",")
            rows.append({"sha1": hashlib.sha1(f[0].encode()).hexdigest(), "len": len(f[0]), "fields": f[1:]})
            if tag in ("1500_1500", "2000_1500") and len(windows_for_encoder.setdefault(tag, [])) < 40:
                windows_for_encoder[tag].append(f[0])
        (HERE / f"fragments_{tag}.json").write_text(json.dumps(rows, indent=0))

    # ---- encoder (numba statement; first tf_frame_length columns, see seqops/crop.py:25-34)
    lut, ascii_lut, comp = convert._build_numba_lookups()
    enc = {}
    for tag, wins in windows_for_encoder.items():
        fsize = int(tag.split("_")[0])
        wins = list(wins)
        # edge cases: N runs, lower case (upper-cased by the converter's caller in predict: keep
        # upper here), a non-ACGT letter
        w = list(wins[0]); w[10:17] = "NNNNNNN"; w[100] = "R"; wins.append("".join(w))
        arr = np.zeros((len(wins), fsize), np.uint8)
        for i, s in enumerate(wins):
            arr[i] = np.frombuffer(s.encode(), np.uint8)
        lens = np.full(len(wins), fsize, np.int64)
        ids = convert._process_batch_numba(arr, lens, fsize, fsize // 3 - 1, lut, comp, ascii_lut)
        n = crop.tf_frame_length(fsize)
        ids = np.asarray(ids)[:, :, :n].astype(np.uint8)
        enc[tag] = {"fsize": fsize, "n_frames": n,
                    "windows_sha1": [hashlib.sha1(s.encode()).hexdigest() for s in wins],
                    "edge_window": wins[-1],
                    "ids_sha256": hashlib.sha256(ids.tobytes()).hexdigest(),
                    "first_window_ids": ids[0].tolist(), "edge_window_ids": ids[-1].tolist()}
    (HERE / "encoder_ids.json").write_text(json.dumps(enc))

    # ---- postprocess -------------------------------------------------------------------------
    rng = np.random.Generator(np.random.PCG64(1234))
    n_win = [1, 3, 7, 2, 12]
    n = sum(n_win)
    classes = ["bacteria", "phage", "eukarya", "archaea", "plasmid", "virus"]
    y = {
        "prediction": (rng.normal(0, 3, (n, 6))).astype(np.float32),
        "reliability": rng.normal(0, 2, (n, 1)).astype(np.float32),
        "meta_0": np.array(sum([[f"contig_{i}___x"] * k for i, k in enumerate(n_win)], [])),
        "meta_2": np.array(sum([[0] * (k - 1) + [1] for k in n_win], [])),
        "meta_4": np.array(sum([[1500 * k + 17] * k for k in n_win], [])),
        "meta_5": rng.integers(300, 400, n), "meta_6": rng.integers(300, 400, n),
        "meta_7": rng.integers(300, 400, n), "meta_8": rng.integers(300, 400, n),
        "meta_9": np.round(rng.normal(0, 0.1, n), 2),
    }
    y["prediction"][0:1, 1] += 8.0            # a confident single-window phage contig
    y["meta_5"][-12:] = 10                     # a contig that is mostly N -> dropped by N% < 0.3
    y["meta_6"][-12:] = 10
    y["meta_7"][-12:] = 10
    y["meta_8"][-12:] = 10
    repeats = pd.DataFrame({"contig_id": [f"contig_{i}___x" for i in range(len(n_win))],
                            "terminal_repeats": [None, "DTR", None, None, "ITR"],
                            "repeat_length": [np.nan, 55, np.nan, np.nan, 31]})
    np.savez(HERE / "postprocess_input.npz", **{k: v for k, v in y.items()})
    repeats.to_csv(HERE / "postprocess_repeats.csv", index=False)
    for tag, yy in (("rel", y), ("norel", {k: v for k, v in y.items() if k != "reliability"})):
        data, data_full = collect.pred_to_dict(yy, class_map={"num_classes": 6}, fsize=1500, term_repeats=repeats)
        out = HERE / f"postprocess_{tag}.tsv"
        out_ph = HERE / f"postprocess_{tag}_phages.tsv"
        for p in (out, out_ph):
            p.unlink(missing_ok=True)
        nrows = collect.write_output(data, labels=classes, indices=list(range(6)), output_table_path=out,
                                     output_phage_table_path=out_ph, reliability_cutoff=0.1, phage_score=3)
        (HERE / f"postprocess_{tag}.json").write_text(json.dumps({
            "n_written": int(nrows), "consensus": np.asarray(data["consensus"]).tolist(),
            "entropy": np.asarray(data["entropy"], np.float64).tolist(),
            "energy": np.asarray(data["energy"], np.float64).tolist(),
            "pred_sum": np.asarray(data["pred_sum"], np.float64).tolist(),
            "ood": None if data["ood"] is None else np.asarray(data["ood"], np.float64).tolist()}))
    # ---- legacy `default` model (BASELINE config #1) -------------------------------------------
    # data files the reference ships (weights, reliability model, batch statistics) + its config entry
    import importlib.util
    legacy = HERE / "legacy_data"
    (legacy / "models" / "default").mkdir(parents=True, exist_ok=True)
    for name in ("WRes_1024.h5", "LR_ood_4_class_default.pkl", "batch_means.npy", "batch_std.npy"):
        shutil.copyfile(REF / "src/jaeger/data/models/default" / name, legacy / "models" / "default" / name)
    cfg_all = json.loads((REF / "src/jaeger/data/config.json").read_text())
    (legacy / "config.json").write_text(json.dumps({"default": cfg_all["default"]}, indent=1))
    spec = importlib.util.spec_from_file_location("v1maps", REF / "src/jaeger/preprocess/v1/maps.py")
    v1maps = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(v1maps)
    tables = json.loads((HERE / "maps.json").read_text())
    tables["V1_TRIMERS"], tables["V1_TRIMER_INT"] = list(v1maps.TRIMERS), list(v1maps.TRIMER_INT)
    (HERE / "maps.json").write_text(json.dumps(tables, indent=0))
    # h5dump cross-check of the HDF5 reader's view of the weight file: per-dataset shape + sha256
    sys.path.insert(0, str(HERE.parents[1]))
    from jaeger_amd.hdf5_lite import read_datasets
    import re
    import subprocess
    dsets = read_datasets(REF / "src/jaeger/data/models/default/WRes_1024.h5")
    summary = {}
    for key, arr in sorted(dsets.items()):
        out = subprocess.run(["/opt/conda/bin/h5dump", "-d", key, "-y", "-w", "0", "-m", "%.9g",
                              str(REF / "src/jaeger/data/models/default/WRes_1024.h5")],
                             capture_output=True, text=True, check=True).stdout
        body = out[out.index("DATA {") + 6:]
        vals = np.array([float(x) for x in re.findall(r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?", body)], np.float32)
        assert vals.size == arr.size and np.array_equal(vals, arr.ravel()), key
        summary[key] = {"shape": list(arr.shape), "sha256": hashlib.sha256(arr.tobytes()).hexdigest()}
    (HERE / "legacy_h5_datasets.json").write_text(json.dumps(summary, indent=0))
    # legacy postprocess on seeded synthetic outputs
    import joblib
    rng = np.random.Generator(np.random.PCG64(4321))
    n_win = [2, 5, 1, 9]
    n = sum(n_win)
    yl = {"y_hat": {"output": rng.normal(0, 3, (n, 4)).astype(np.float32),
                    "embedding": np.abs(rng.normal(0, 1, (n, 128))).astype(np.float32)},
          "meta": [np.array(sum([[f"c{i}___y"] * k for i, k in enumerate(n_win)], [])),
                   np.arange(n), np.array(sum([[0] * (k - 1) + [1] for k in n_win], [])), np.arange(n),
                   np.array(sum([[2000 * k + 3] * k for k in n_win], [])),
                   rng.integers(400, 500, n), rng.integers(400, 500, n), rng.integers(400, 500, n),
                   rng.integers(400, 500, n), np.round(rng.normal(0, 0.1, n), 2)]}
    yl["y_hat"]["output"][2:7, 1] += 9.0
    np.savez(HERE / "postprocess_legacy_input.npz", output=yl["y_hat"]["output"], embedding=yl["y_hat"]["embedding"],
             **{f"meta_{i}": m for i, m in enumerate(yl["meta"])})
    conf = dict(cfg_all["default"])
    conf["model"] = "default"
    conf["labels"] = [v for _, v in conf["default_labels"].items()]
    ood_params = {"type": "sklearn", "model": joblib.load(legacy / "models/default/LR_ood_4_class_default.pkl"),
                  "batch_mean": np.load(legacy / "models/default/batch_means.npy"),
                  "batch_std": np.load(legacy / "models/default/batch_std.npy")}
    rep = pd.DataFrame({"contig_id": [f"c{i}___y" for i in range(len(n_win))],
                        "terminal_repeats": [None, "ITR", None, None], "repeat_length": [np.nan, 77, np.nan, np.nan]})
    data, _ = collect.pred_to_dict_legacy(conf, yl, model="default", fsize=2000, ood_params=ood_params,
                                          term_repeats=rep)
    collect.write_output_legacy(conf, data, output_table_path=HERE / "postprocess_legacy.tsv",
                                output_phage_table_path=HERE / "postprocess_legacy_phages.tsv",
                                reliability_cutoff=0.1, phage_score=3)
    rep.to_csv(HERE / "postprocess_legacy_repeats.csv", index=False)
    # ---- a Keras-3 style .weights.h5 (layers/<name>/vars/<i>) for the 500-bp baseline plan, written
    # with HDF5's own h5import tool from seeded random weights: exercises the h5py-free reader on a
    # file produced by the HDF5 library with default settings
    import tempfile
    import yaml
    from jaeger_amd.plan import build_plan
    from jaeger_amd.weights import _layer_order, random_weights
    plan = build_plan(yaml.safe_load((HERE / "baseline500_project.yaml").read_text())["model"])
    w500 = random_weights(plan, seed=500)
    keras_names = {"embedding": "embedding", "rep/0": "masked_conv1d", "rep/1": "masked_batch_norm",
                   "rep/4": "masked_batch_norm_1", "classifier/1": "dense"}
    out_h5 = HERE / "baseline500_keras3.weights.h5"
    out_h5.unlink(missing_ok=True)
    with tempfile.TemporaryDirectory() as td:
        for prefix, leaves in _layer_order(plan):
            if prefix in keras_names:
                group = f"layers/{keras_names[prefix]}"
            else:                                              # rep/3/block0/conv1 -> residual_block/conv1
                _, _, blk, sub = prefix.split("/")
                n = int(blk.removeprefix("block"))
                group = f"layers/residual_block_stack/layers/residual_block{'' if n == 0 else '_' + str(n)}/{sub}"
            for i, leaf in enumerate(leaves):
                arr = w500[f"{prefix}/{leaf}"]
                txt, cfg = Path(td) / "t.txt", Path(td) / "t.cfg"
                np.savetxt(txt, arr.reshape(-1, 1), fmt="%.9g")
                cfg.write_text(f"PATH {group}/vars/{i}\nINPUT-CLASS TEXTFP\nRANK {arr.ndim}\n"
                               f"DIMENSION-SIZES {' '.join(map(str, arr.shape))}\nOUTPUT-CLASS FP\nOUTPUT-SIZE 32\n")
                subprocess.run(["/opt/conda/bin/h5import", str(txt), "-c", str(cfg), "-o", str(out_h5)], check=True,
                               capture_output=True)         # one call per dataset: appends to the file
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
