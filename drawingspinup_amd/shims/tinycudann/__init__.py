"""`import tinycudann as tcnn` for instant_nsr/models/network_utils.py:5 and models/utils.py:9."""
from drawingspinup_amd.nsr.encoding import Encoding  # noqa: F401


def free_temporary_memory():      # instant_nsr/models/utils.py:110
    return None


class Network:                    # only reached when mlp otype != VanillaMLP (not the shipped config)
    def __init__(self, *a, **k):
        raise NotImplementedError("tinycudann.Network: the reference configs use VanillaMLP")
