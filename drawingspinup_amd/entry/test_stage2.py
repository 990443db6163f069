"""`python test_stage2.py --uid U` (3_style_translator/test_stage2.py)."""
from ._test_stage import run

if __name__ == "__main__":
    run(2)
