"""``python -m jaeger_amd predict ...`` - the reference's ``jaeger predict`` flag surface
(cli.py:122-370) in front of :func:`jaeger_amd.predict.run_core`.

Flags that select machinery outside the MI355X hot path (``--cpu``, ``--onnx``, ``--quantized``,
``--int8``, ``--refine``) are accepted so existing
command lines parse, and rejected at run time with an explicit message - there is no silent
fallback.  Extra flags: ``--exact-f32`` (disable the split-f16 conv path), ``--chunk``.
"""

from __future__ import annotations

import click


@click.group()
def main():
    """Jaeger predict path on AMD MI355X."""


@main.command(context_settings=dict(ignore_unknown_options=True), help="Run jaeger inference pipeline on MI355X.")
@click.option("-i", "--input", type=click.Path(exists=True), required=True, help="Path to input file")
@click.option("-o", "--output", type=str, required=True, help="Path to output directory")
@click.option("--fsize", type=int, default=2000, help="Length of the sliding window (value must be 2^n).")
@click.option("--stride", type=int, default=1500, help="Stride of the sliding window.")
@click.option("--dynamic-stride", is_flag=True, help="adaptive overlap per contig")
@click.option("--dynamic-stride-threshold", type=float, default=10.0)
@click.option("--crf", is_flag=True, help="(experimental) decode per-window calls jointly with a linear-chain CRF "
                                          "(Viterbi) instead of independent argmax")
@click.option("--crf-switch-cost", type=float, default=2.0, help="global CRF transition cost lambda (log-prob units)")
@click.option("--crf-prior", type=click.Choice(["biological", "uniform"]), default="biological")
@click.option("--crf-transition-matrix", type=click.Path(exists=True), default=None,
              help='JSON class-name-keyed cost matrix, e.g. {"bacteria": {"phage": 0.5}}; overrides --crf-prior')
@click.option("--dustmask/--no-dustmask", default=True, help="soft-mask low-complexity regions (symmetric DUST)")
@click.option("--dust-host", is_flag=True, default=False,
              help="run DUST as a host pass over the FASTA image instead of on the GPU inside the fused call (same masks)")
@click.option("--min-len", "min_len", type=int, default=None, help="Minimum contig length to process")
@click.option("-m", "--model", type=str, default="default")
@click.option("--model_path", type=click.Path(exists=True), default=None,
              help="directory containing model/ with *_project.yaml, *_classes.yaml and weights")
@click.option("--config", type=click.Path(exists=True), default=None)
@click.option("-p", "--prophage", is_flag=True, help="write the prophage-segmentation input frames of contigs >= --lc "
                                                     "(the segmentation / plots themselves are not built here)")
@click.option("-s", "--sensitivity", type=float, default=1.5)
@click.option("--lc", type=int, default=500000)
@click.option("--plot-type", type=click.Choice(["circular", "linear"]), default="circular")
@click.option("--rc", type=float, default=0.1, help="Minimum reliability score for the phage table")
@click.option("--pc", type=int, default=3, help="Minimum phage score for the phage table")
@click.option("--batch", type=int, default=96, help="batch of the short-contig padded pass")
@click.option("--workers", type=int, default=4, help="accepted, unused (no host input pipeline)")
@click.option("--window-scores", is_flag=True)
@click.option("--getsequences", is_flag=True, help="write the sequences of the phage table to <stem>_phages_jaeger.fasta")
@click.option("--cpu", is_flag=True, help="[rejected: no CPU fallback]")
@click.option("--physicalid", type=int, default=0, help="GPU ordinal")
@click.option("--mem", type=int, default=4, help="accepted, unused")
@click.option("--getalllabels", is_flag=True)
@click.option("-v", "--verbose", count=True, default=1)
@click.option("-f", "--overwrite", is_flag=True)
@click.option("--quantized", is_flag=True, help="[unsupported here]")
@click.option("--precision", type=click.Choice(["fp32", "fp16", "bf16"]), default="fp32",
              help="accepted; the engine always returns f32-accurate logits")
@click.option("--xla", is_flag=True, help="accepted, no-op")
@click.option("--onnx", is_flag=True, help="[unsupported here]")
@click.option("--int8", is_flag=True, help="[unsupported here]")
@click.option("--save-embedding", is_flag=True)
@click.option("--save-nmd", is_flag=True)
@click.option("--refine", is_flag=True, help="[unsupported here]")
@click.option("--exact-f32", is_flag=True, help="run every convolution on the exact-f32 MFMA kernels")
@click.option("--chunk", type=int, default=0, help="windows per device pass (0 = library default)")
@click.option("--no-pipeline", is_flag=True, help="run model set-up, repeat scan, forward and aggregation one after the "
                                                   "other (single-GPU runs overlap them by default)")
@click.option("--trust-project", is_flag=True, help="take the layer plan from <name>_project.yaml and the weights from the "
                                                    "weights file even when a <name>_graph/ SavedModel is there (skips its "
                                                    "census check and its variable bundle)")
@click.option("--stream-bytes", type=int, default=None, help="span budget of the host -> HBM base ingest (default 32 MiB)")
@click.option("--legacy-data", type=click.Path(exists=True), default=None,
              help="reference data directory (config.json, models/default/) for -m default")
def predict(**kwargs):
    # cli.py:375-410: --model_path wins; the `default` model goes through the legacy workflow
    if not kwargs.get("model_path") and (kwargs.get("model") or "default") == "default" and not kwargs.get("config"):
        click.echo(click.style("Warning: model 'default' uses the legacy prediction workflow and is deprecated.",
                               fg="yellow"), err=True)
        from .predict_legacy import run_core
    else:
        from .predict import run_core
    run_core(**kwargs)


@main.command("verify-model", help="Compare a SavedModel directory (saved_model.pb + variables) with the layer plan "
                                   "the MI355X engine compiles - no TensorFlow needed.")
@click.argument("graph_dir", type=click.Path(exists=True))
@click.option("--project", type=click.Path(exists=True), default=None, help="the model's *_project.yaml")
@click.option("--legacy", is_flag=True, help="check against the legacy `default` tower instead of a project.yaml")
def verify_model_cmd(graph_dir, project, legacy):
    import sys

    import yaml

    from .plan import build_plan
    from .verify import report, verify_model
    if not legacy and project is None:
        raise click.UsageError("pass --project <name>_project.yaml or --legacy")
    plan = None if legacy else build_plan(yaml.safe_load(open(project).read())["model"])
    click.echo(report(graph_dir, plan, legacy))
    sys.exit(1 if verify_model(graph_dir, plan, legacy) else 0)


if __name__ == "__main__":
    main()
