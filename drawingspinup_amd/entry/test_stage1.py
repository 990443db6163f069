"""`python test_stage1.py --uid U` (3_style_translator/test_stage1.py)."""
from ._test_stage import run

if __name__ == "__main__":
    run(1)
