"""`python train_stage1.py --uid U` (3_style_translator/train_stage1.py)."""
from ._train_stage import run

if __name__ == "__main__":
    run(1)
