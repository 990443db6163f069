"""diffusers stand-in (TEST INFRASTRUCTURE ONLY; see oracle/stubs/README.md): the symbols the
reference's mvdiffusion package imports, restated from the published definitions of diffusers
0.19.3 (reference requirements.txt:9) in plain torch so they run on the CPU in float64."""
__version__ = "0.19.3"
