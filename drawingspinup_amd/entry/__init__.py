"""Entry points with the reference's command lines and on-disk layout:
    python -m drawingspinup_amd.entry.mv          --uid U [--all]      (2_charactor_reconstructor/mv.py)
    python -m drawingspinup_amd.entry.recon       --uid U [--all]      (2_charactor_reconstructor/recon.py)
    python -m drawingspinup_amd.entry.test_stage1 --uid U              (3_style_translator/test_stage1.py)
    python -m drawingspinup_amd.entry.test_stage2 --uid U              (3_style_translator/test_stage2.py)
Data root: <root>/<uid>/{char,mv,mesh}/... exactly as README.md:60-78 of the reference."""
