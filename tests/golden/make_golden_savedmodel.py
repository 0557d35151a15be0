#!/usr/bin/env python3
"""Golden logits of the reference's bundled SavedModel, executed WITHOUT TensorFlow.

Source artefact (data, copied verbatim into ``tests/golden/legacy_data/models/test/jaeger_fragment_graph``):
``/root/reference/src/jaeger/data/models/test/jaeger_fragment_graph`` = ``saved_model.pb`` + variable bundle of the
legacy ``default`` tower (947 036 parameters, the same values as ``WRes_1024.h5``).  Its serving function
(``__inference_serving_default_*``, 6 324 nodes, 29 op kinds) is what ``InferModel`` / ``JaegerModel`` execute
(``nnlib/inference.py:307-325``); ``oracle/graphdef.py`` interprets it op by op in numpy/torch-CPU.

Inputs: the 135 windows of the reference's ``test_contigs.fasta`` at fsize 2000 / stride 1500, as v1 amino-acid ids
(``preprocess/v1/convert.py:56-125`` semantics, fed as the float tensors the signature declares: ``inputs``,
``inputs_1`` ... ``inputs_5`` = frames f1,f2,f3,r1,r2,r3).

Writes ``tests/golden/legacy_savedmodel_logits.npz``: ``ids`` (135,6,665) u8, ``output`` / ``embedding`` evaluated in
f32, ``output_f64`` / ``embedding_f64`` evaluated in f64 (the graph's exact value, to which TensorFlow's f32 kernels and
every f32 implementation here are rounding-level approximations).

Run from the repo root: ``python tests/golden/make_golden_savedmodel.py [graph_dir]``.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
GOLDEN = ROOT / "tests" / "golden"


def main():
    from jaeger_amd.fragment import read_fasta
    from jaeger_amd.maps import V1_TRIMER_INT
    from oracle import encoder as oenc
    from oracle import fragmenter as ofr
    from oracle import graphdef

    ref_dir = Path("/root/reference/src/jaeger/data/models/test/jaeger_fragment_graph")
    graph_dir = Path(sys.argv[1]) if len(sys.argv) > 1 else (ref_dir if ref_dir.exists() else
                                                               GOLDEN / "legacy_data/models/test/jaeger_fragment_graph")
    records = [(n, s.decode()) for n, s in read_fasta(str(GOLDEN / "test_contigs.fasta"))]
    rows = [r.split(",") for r in ofr.fragment_strings(records, 2000, 1500)]
    ids = oenc.encode_windows([r[0] for r in rows], 2000, codon_id=[v - 1 for v in V1_TRIMER_INT], masking=True)
    assert ids.shape == (135, 6, 665)
    out = {"ids": ids.astype(np.uint8)}
    for tag, dt in (("", np.float32), ("_f64", np.float64)):
        parts = []
        for i in range(0, len(ids), 45):
            feeds = {("inputs" if f == 0 else f"inputs_{f}"): ids[i:i + 45, f, :].astype(np.float32) for f in range(6)}
            parts.append(graphdef.run_saved_model(graph_dir, feeds, float_dtype=dt))
        # signature outputs: identity = (W, 128) embedding, identity_1 = (W, 4) class logits
        out["embedding" + tag] = np.concatenate([p["identity"] for p in parts]).astype(dt)
        out["output" + tag] = np.concatenate([p["identity_1"] for p in parts]).astype(dt)
    print("f32 vs f64 evaluation of the graph: logits", float(np.abs(out["output"] - out["output_f64"]).max()),
          "embedding", float(np.abs(out["embedding"] - out["embedding_f64"]).max()))
    np.savez_compressed(GOLDEN / "legacy_savedmodel_logits.npz", **out)
    print("wrote", GOLDEN / "legacy_savedmodel_logits.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
