import os

# single-GPU command-line runs do not need PyTorch (see _lib.load); under torchrun they do
if int(os.environ.get("WORLD_SIZE", "1")) <= 1 and os.environ.get("JAEGER_SHARDED") != "1":
    os.environ.setdefault("JAEGER_HIP_TORCH", "0")

from .cli import main  # noqa: E402

main()
