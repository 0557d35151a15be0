"""Host-side pieces of bench.py that need no GPU: counting GPUs without touching HIP, the kernel-source fingerprint and
the rule that PMC-derived roofline fields are only quoted for the kernel build they were collected on."""
import json
import sys

import pytest
from conftest import ROOT

sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def test_visible_gpus_reads_the_environment_first(monkeypatch):
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0,1,2")
    assert bench.visible_gpus() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "4")           # HIP's list wins over ROCR's, as in the runtime
    assert bench.visible_gpus() == 1
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpus() == 0


def test_visible_gpus_does_not_import_torch(monkeypatch):
    """The parent of ``--gpus N`` must not initialise the GPU runtime: counting goes through sysfs / the environment."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import bench; bench.visible_gpus(); "
            "print('torch' in sys.modules)" % str(ROOT))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "False", out.stderr[-500:]


def test_kernel_hash_is_stable_and_source_sensitive(tmp_path, monkeypatch):
    h1, h2 = bench.kernel_hash(), bench.kernel_hash()
    assert h1 == h2 and len(h1) == 16 and int(h1, 16) >= 0
    # a copy of the tree's kernel sources with one byte changed hashes differently
    src = ROOT / "jaeger_amd" / "csrc"
    fake = tmp_path / "jaeger_amd" / "csrc"
    fake.mkdir(parents=True)
    for name in ("jg_common.h", "jg_conv_dev.h", "jg_conv_f16.hip", "jg_conv_f16_impl.h", "jg_small.h", "jg_small.hip"):
        (fake / name).write_bytes((src / name).read_bytes())
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    assert bench.kernel_hash() == h1
    (fake / "jg_small.hip").write_bytes((src / "jg_small.hip").read_bytes() + b"\n")
    assert bench.kernel_hash() != h1
    # exact-f32 lines are keyed on conv_f32_kernel's source as well
    (fake / "jg_kernels.hip").write_bytes((src / "jg_kernels.hip").read_bytes())
    assert bench.kernel_hash("f32") != bench.kernel_hash("f16x3")


def test_write_fasta_records_is_what_the_reader_parses(tmp_path):
    import numpy as np
    rng = np.random.default_rng(3)
    b2 = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (1203, 37))]
    bench.write_fasta_records(tmp_path / "r.fa", b2)
    lines = (tmp_path / "r.fa").read_bytes().split(b"\n")
    assert lines[-1] == b"" and len(lines) == 2 * 1203 + 1
    assert lines[0] == b">r0000000" and lines[2 * 1202] == b">r0001202"
    assert lines[1] == b2[0].tobytes() and lines[2 * 1202 + 1] == b2[1202].tobytes()


@pytest.mark.parametrize("same_build", [True, False])
def test_pmc_fields_only_for_the_build_they_were_collected_on(tmp_path, monkeypatch, same_build):
    src = ROOT / "jaeger_amd" / "csrc"
    (tmp_path / "jaeger_amd" / "csrc").mkdir(parents=True)
    for name in ("jg_common.h", "jg_conv_dev.h", "jg_conv_f16.hip", "jg_conv_f16_impl.h", "jg_small.h", "jg_small.hip",
                 "jg_kernels.hip"):
        (tmp_path / "jaeger_amd" / "csrc" / name).write_bytes((src / name).read_bytes())
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    here = bench.kernel_hash()
    stamp = here if same_build else "0" * 16
    (tmp_path / "profiles").mkdir()
    (tmp_path / "profiles" / "pmc_traffic.json").write_text(json.dumps(
        {"kernel_hash": stamp, "conv_f16x3_kernel": {"fsize": 1500, "traffic_bytes_per_launch": 7_000_000_000}}))
    (tmp_path / "profiles" / "mfma_util.json").write_text(json.dumps(
        {"kernel_hash": stamp, "conv_f16x3_kernel": {"mfma_busy_frac": 0.61, "eff_clock_ghz": 1.7}}))
    got = bench.pmc_fields("default", "f16x3", 1500, 0, 2.8)
    if same_build:
        assert got["pmc_stale"] is False and got["traffic"] == 7_000_000_000
        # busy share and clock belong to the device they were counted on: quoted under a label that says so
        od = got["other_device"]
        assert od["mfma_busy_frac"] == 0.61 and od["eff_clock_ghz"] == 1.7 and "another device" in od["note"]
        assert got["hbm_gbs"] == pytest.approx(2500.0) and "mfma_busy_frac" not in got
    else:
        assert got["pmc_stale"] is True and got["traffic"] is None and got["other_device"] is None and got["hbm_gbs"] is None
    # other precisions / window sizes / chunkings never get the default configuration's counters
    for args in (("default", "f32", 1500, 0, 2.8), ("default", "f16x3", 2000, 0, 2.8), ("default", "f16x3", 1500, 512, 2.8),
                 ("pyramid", "f16x3", 2000, 0, 1.0)):
        assert bench.pmc_fields(*args)["traffic"] is None
