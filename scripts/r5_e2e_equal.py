"""Round 5: the end-to-end tables of the pipeline as it ships against the same run with every shortcut of the round switched
off - the repeat scan scoring every alignment (JG_OPT_TERMINI_REPORT_MIN 0) and the table rows through the DataFrame path -
on the 10 000-contig FASTA and on 200 000 records of 500 - 800 bp (overlapping ends), 500-bp model, DUST on.  Prints the
sizes and whether <base>.tsv / <base>_phages.tsv are byte-identical.
usage: python scripts/r5_e2e_equal.py"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bench  # noqa: E402
from conftest import make_model_dir  # noqa: E402
from jaeger_amd import postprocess as PP  # noqa: E402
from jaeger_amd import predict as P  # noqa: E402
from jaeger_amd import termini as T  # noqa: E402
tmp = Path("/dev/shm/jaeger_r5_equal"); tmp.mkdir(exist_ok=True)
mdir = make_model_dir(tmp / "m", name="baseline500", model_name="jaeger_500bp_baseline")
rng = np.random.Generator(np.random.PCG64(20260923))
cases = {}
lengths, bases = bench.synth_contigs(rng, 10000)
bench.write_fasta_contigs(tmp / "10k.fasta", lengths, bases)
cases["10k"] = tmp / "10k.fasta"
n = 200_000
lens = rng.integers(500, 800, n)
b2 = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(lens.sum()), dtype=np.uint8)]
b2[rng.integers(0, b2.size, 2000)] = ord("N")
bench.write_fasta_contigs(tmp / "short.fasta", lens, b2)
cases["200k short records"] = tmp / "short.fasta"
init = PP.TableWriter.__init__
for name, fa in cases.items():
    outs = []
    for slow in (False, True):
        T.REPORT_MIN_COLUMNS = 0 if slow else 13
        if slow:
            PP.TableWriter.__init__ = lambda self, *a, **k: init(self, *a, **{**k, "columns_path": False})
        else:
            PP.TableWriter.__init__ = init
        out = tmp / f"out_{int(slow)}"
        P.run_core(input=str(fa), output=str(out), model_path=str(mdir), fsize=500, stride=500, overwrite=True, dustmask=True,
                   verbose=0, batch=96, rc=0.1, pc=3)
        base = fa.stem
        tsv = next(out.rglob(f"{base}*.tsv"))
        files = sorted(p for p in out.rglob("*.tsv"))
        outs.append({p.name: p.read_bytes() for p in files})
    same = outs[0].keys() == outs[1].keys() and all(outs[0][k] == outs[1][k] for k in outs[0])
    print(f"{name}: {[(k, len(v)) for k, v in outs[0].items()]}  byte-identical: {same}", flush=True)
    assert same
