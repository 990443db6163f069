"""`python train_stage2.py --uid U` (3_style_translator/train_stage2.py)."""
from ._train_stage import run

if __name__ == "__main__":
    run(2)
