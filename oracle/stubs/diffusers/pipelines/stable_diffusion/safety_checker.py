class StableDiffusionSafetyChecker:      # type annotation only; mv.py loads the pipeline without one
    pass
