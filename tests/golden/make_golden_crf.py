#!/usr/bin/env python3
"""Golden vectors for `--crf` window decoding, produced by IMPORTING the reference
(postprocess/helpers.py: viterbi_decode, build_transition_costs; postprocess/collect.py:
pred_to_dict + write_output with crf_switch_cost).  Build container only; outputs are data.

    python tests/golden/make_golden_crf.py

Produces
  crf_cases.json          seeded f32 logits (several T, C, lambda, priors, a user matrix) -> the reference's
                          cost matrix and decoded path; plus the binary-head [0, z] stacking
  postprocess_crf.tsv     pred_to_dict(crf_switch_cost=2.0, biological prior) + write_output on
                          postprocess_input.npz (the same seeded windows as postprocess_rel.tsv)
"""
import json
import sys
from pathlib import Path

import numpy as np
import pandas as pd

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_golden  # noqa: E402  (adds /root/reference/src to sys.path, provides the pyfastx / pydustmasker stubs)

make_golden._stub_modules()
from jaeger.postprocess import collect  # noqa: E402
from jaeger.postprocess.helpers import build_transition_costs, viterbi_decode  # noqa: E402

SIX = ["bacteria", "phage", "eukarya", "archaea", "plasmid", "virus"]


def main():
    rng = np.random.Generator(np.random.PCG64(77))
    cases = []
    settings = [
        (1, SIX, 2.0, "biological", None), (2, SIX, 2.0, "biological", None), (40, SIX, 2.0, "biological", None),
        (133, SIX, 0.5, "biological", None), (57, SIX, 6.0, "uniform", None), (25, SIX, 0.0, "uniform", None),
        (64, ["bacteria", "phage", "eukarya", "archaea"], 3.0, "biological", None),
        (31, ["Bacteria", "Phage", "Other"], 1.5, "biological", None),
        (48, SIX, 2.0, "biological", {"bacteria": {"phage": 0.1, "nonsense": 9}, "Virus": {"eukarya": 4.0}, "x": 3}),
    ]
    for t_len, names, lam, prior, user in settings:
        z = (rng.normal(0, 2.0, (t_len, len(names))) + 3.0 * np.eye(len(names))[rng.integers(0, len(names), 1)[0]]
             ).astype(np.float32)
        # plant a block of another class so that smoothing has something to decide
        if t_len > 8:
            a = t_len // 3
            z[a:a + max(1, t_len // 10), 1] += 4.0
        costs = build_transition_costs(names, switch_cost=lam, prior=prior, user_matrix=user)
        path = viterbi_decode(z, lam, costs)
        cases.append({"kind": "softmax", "names": names, "switch_cost": lam, "prior": prior, "user_matrix": user,
                      "logits": z.tolist(), "costs": costs.tolist(), "path": np.asarray(path).tolist()})
    for t_len, lam in ((1, 2.0), (9, 1.0), (60, 2.5)):
        z = rng.normal(0, 2.0, (t_len, 1)).astype(np.float32)
        path = viterbi_decode(np.concatenate([np.zeros_like(z), z], axis=-1), lam)      # collect.py:365-372
        cases.append({"kind": "binary", "switch_cost": lam, "logits": z.tolist(), "path": np.asarray(path).tolist()})
    (HERE / "crf_cases.json").write_text(json.dumps(cases))

    y = dict(np.load(HERE / "postprocess_input.npz", allow_pickle=True))
    repeats = pd.read_csv(HERE / "postprocess_repeats.csv")
    cm = {"num_classes": 6, "class": SIX, "index": list(range(6))}
    data, _ = collect.pred_to_dict(y, class_map=cm, fsize=1500, term_repeats=repeats, crf_switch_cost=2.0,
                                   crf_prior="biological")
    out, out_ph = HERE / "postprocess_crf.tsv", HERE / "postprocess_crf_phages.tsv"
    for p in (out, out_ph):
        p.unlink(missing_ok=True)
    collect.write_output(data, labels=SIX, indices=list(range(6)), output_table_path=out,
                         output_phage_table_path=out_ph, reliability_cutoff=0.1, phage_score=3)
    print("crf golden vectors written to", HERE)


if __name__ == "__main__":
    main()
