"""Drop-in entry point: `python train.py --config YAML [--config-override K V ...] --num-gpus-per-machine N ...` — the reference's
command line (reference train.py:38-58,299-313), running the MI355X-native train step of clip_lite_amd."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from clip_lite_amd.train_loop import cli  # noqa: E402

if __name__ == "__main__":
    cli()
